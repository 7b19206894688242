"""Fixed cost of an oc_rollout_random launch: time launches of 8, 16, 32, 64 steps: python tools/time_rollout_fixed.py [layout] [n_envs]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from overcooked_ai_amd.vec_env import VecOvercookedEnv

layout = sys.argv[1] if len(sys.argv) > 1 else "cramped_room"
n = int(sys.argv[2]) if len(sys.argv) > 2 else 65536
dev = torch.device("cuda:0")
env = VecOvercookedEnv(layout, n, horizon=400, device=dev, auto_reset=True, seed=0)
rew = torch.zeros((400, n, 4), dtype=torch.float32, device=dev)
fl = torch.zeros((400, n), dtype=torch.uint8, device=dev)
res = []
for T in (8, 16, 32, 64, 400):
    for _ in range(5):
        env.rollout_random(T, rew[:T], fl[:T])
    torch.cuda.synchronize()
    evs = [torch.cuda.Event(enable_timing=True) for _ in range(61)]
    for i in range(60):
        evs[i].record()
        env.rollout_random(T, rew[:T], fl[:T])
    evs[60].record()
    torch.cuda.synchronize()
    ms = sorted(a.elapsed_time(b) for a, b in zip(evs[:-1], evs[1:]))
    res.append((T, ms[len(ms) // 2] * 1e3))
    print("%s n=%d T=%d: %.1f us per launch" % (layout, n, T, res[-1][1]))
(t1, a), (t2, b) = res[0], res[3]
slope = (b - a) / (t2 - t1)
print("per step %.3f us, fixed per launch %.1f us (events around back-to-back launches: includes the dispatch gap)" % (slope, a - slope * t1))
