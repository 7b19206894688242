"""Is a slow 'level' transient?  Per-launch rate of the first 80 headline launches after the output arrays are allocated, three times
in one process (separate torch allocations each time).  python tools/process_level3.py"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from overcooked_ai_amd.vec_env import VecOvercookedEnv

n, T = 65536, 4000
dev = torch.device("cuda:0")
env = VecOvercookedEnv("cramped_room", n, horizon=400, device=dev, auto_reset=True, seed=0)
for rnd in range(3):
    rew = torch.zeros((T, n, 4), dtype=torch.float32, device=dev)
    fl = torch.zeros((T // 8, n, 8), dtype=torch.uint8, device=dev)
    torch.cuda.synchronize()
    N = 80
    evs = [torch.cuda.Event(enable_timing=True) for _ in range(N + 1)]
    for i in range(N):
        evs[i].record()
        env.rollout_random(T, rew, fl, flags_tiled8=True)
    evs[N].record()
    torch.cuda.synchronize()
    g = [n * T / (a.elapsed_time(b) * 1e-3) / 1e9 for a, b in zip(evs[:-1], evs[1:])]
    print("allocation %d (rewards at %#x): launches 1-80, G env-steps/s: %s" % (rnd, rew.data_ptr(), " ".join("%.0f" % x for x in g)))
    del rew, fl
    torch.cuda.empty_cache()
    pad = torch.empty(((53 + 64 * rnd) << 20,), dtype=torch.uint8, device=dev)
