"""What sets a process's level on the headline launch?  One process = one line per placement of the output
arrays: python tools/process_level.py [tag]   (run it several times, and under different HSA_* / GPU_* settings, on ONE box)
  base      rewards / flags as torch hands them out (2 MiB-aligned segments)
  +4K/+64K/+1M  both arrays carved from one big buffer at that offset from a 2 MiB boundary
  fresh     the arrays freed and allocated again (empty_cache in between): another physical placement, same process
Each figure: median of 12 launches of 4 000 steps x 65 536 cramped_room envs, tiled flags (the driver's headline launch)."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from overcooked_ai_amd.vec_env import VecOvercookedEnv

tag = sys.argv[1] if len(sys.argv) > 1 else "run"
n, T = 65536, 4000
dev = torch.device("cuda:0")
env = VecOvercookedEnv("cramped_room", n, horizon=400, device=dev, auto_reset=True, seed=0)


def level(rew, fl):
    for _ in range(3):
        env.rollout_random(T, rew, fl, flags_tiled8=True)
    torch.cuda.synchronize()
    evs = [torch.cuda.Event(enable_timing=True) for _ in range(13)]
    for i in range(12):
        evs[i].record()
        env.rollout_random(T, rew, fl, flags_tiled8=True)
    evs[12].record()
    torch.cuda.synchronize()
    ms = sorted(a.elapsed_time(b) for a, b in zip(evs[:-1], evs[1:]))
    return n * T / (ms[len(ms) // 2] * 1e-3) / 1e9, n * T / (ms[0] * 1e-3) / 1e9, n * T / (ms[-1] * 1e-3) / 1e9


out = []
rew = torch.zeros((T, n, 4), dtype=torch.float32, device=dev)
fl = torch.zeros((T // 8, n, 8), dtype=torch.uint8, device=dev)
out.append(("base", level(rew, fl)))
del rew, fl
torch.cuda.empty_cache()
big = torch.zeros((T * n * 17 + (8 << 20),), dtype=torch.uint8, device=dev)
base = (-big.data_ptr()) % (2 << 20)
for name, off in (("+4K", 4096), ("+64K", 65536), ("+1M", 1 << 20), ("+0", 0)):
    r = big[base + off: base + off + T * n * 16].view(torch.float32).view(T, n, 4)
    f = big[base + off + T * n * 16: base + off + T * n * 17].view(T // 8, n, 8)
    out.append((name, level(r, f)))
del big, r, f
torch.cuda.empty_cache()
for k in range(2):
    pad = torch.empty(((37 + 64 * k) << 20,), dtype=torch.uint8, device=dev)  # shifts where the next segments land
    rew = torch.zeros((T, n, 4), dtype=torch.float32, device=dev)
    fl = torch.zeros((T // 8, n, 8), dtype=torch.uint8, device=dev)
    out.append(("fresh%d" % k, level(rew, fl)))
    del rew, fl, pad
    torch.cuda.empty_cache()
print("%-22s " % tag + "  ".join("%s %.1f (%.1f..%.1f)" % (k, v[0], v[2], v[1]) for k, v in out))
