import torch, time
dev=torch.device('cuda:0')
def bench(fn, iters=200):
    for _ in range(10): fn()
    torch.cuda.synchronize()
    e0=torch.cuda.Event(enable_timing=True); e1=torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1)/iters
for nbytes in (156237824, 616300544, 2_000_000_000):
    x=torch.empty(nbytes, dtype=torch.uint8, device=dev)
    y=torch.empty(nbytes, dtype=torch.uint8, device=dev)
    xf=x[:nbytes//4*4].view(torch.float32)
    t=bench(lambda: xf.zero_())
    print("fill  %4d MB: %.1f us  %.2f TB/s" % (nbytes/1e6, t*1e3, nbytes/t/1e9))
    t=bench(lambda: xf.fill_(1.5))
    print("fill1 %4d MB: %.1f us  %.2f TB/s" % (nbytes/1e6, t*1e3, nbytes/t/1e9))
    t=bench(lambda: y.copy_(x))
    print("copy  %4d MB: %.1f us  %.2f TB/s (read+write)" % (nbytes/1e6, t*1e3, 2*nbytes/t/1e9))
