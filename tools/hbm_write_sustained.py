"""Sustained HBM write rate of this box over several seconds (what bounds a kernel that only writes, like the rollout's
outputs): python tools/hbm_write_sustained.py [seconds].  Prints TB/s of torch's fill kernel on a 4 GiB buffer per ~0.1 s window."""
import sys
import time

import torch

secs = float(sys.argv[1]) if len(sys.argv) > 1 else 8.0
dev = torch.device("cuda:0")
buf = torch.empty(1 << 30, dtype=torch.float32, device=dev)  # 4 GiB
nbytes = buf.numel() * 4
for _ in range(3):
    buf.fill_(1.0)
torch.cuda.synchronize()
per = 128  # fills per window
rates = []
t_start = time.perf_counter()
while time.perf_counter() - t_start < secs:
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(per):
        buf.fill_(1.0)
    e1.record()
    e1.synchronize()
    rates.append((time.perf_counter() - t_start, per * nbytes / (e0.elapsed_time(e1) * 1e-3) / 1e12))
print("fill_ of 4 GiB, %d fills per window: TB/s over time" % per)
print(" ".join("%.1fs:%.2f" % r for r in rates))
# the same bytes as 1 KiB chunks at a 1 MiB stride per "step" would be the rollout's pattern; a strided variant:
rows = buf.view(4096, 1 << 18)  # 4096 rows of 1 MiB
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(64):
    rows[:, : 1 << 17].fill_(2.0)  # half of every row: 2 GiB per pass, strided
e1.record(); e1.synchronize()
print("strided half-rows (512 KiB of every 1 MiB): %.2f TB/s" % (64 * (nbytes // 2) / (e0.elapsed_time(e1) * 1e-3) / 1e12))
