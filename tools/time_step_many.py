"""Time oc_step_many (K caller-supplied-action steps per launch): python tools/time_step_many.py [layout] [n_envs] [K]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from overcooked_ai_amd.vec_env import VecOvercookedEnv

layout = sys.argv[1] if len(sys.argv) > 1 else "cramped_room"
n = int(sys.argv[2]) if len(sys.argv) > 2 else 65536
K = int(sys.argv[3]) if len(sys.argv) > 3 else 400
dev = torch.device("cuda:0")
env = VecOvercookedEnv(layout, n, horizon=400, device=dev, auto_reset=True, seed=0)
acts = torch.randint(0, 6, (K, n, 2), dtype=torch.uint8, device=dev)
rew = torch.zeros((K, n, 4), dtype=torch.float32, device=dev)
fl = torch.zeros((K, n), dtype=torch.uint8, device=dev)
for _ in range(3):
    env.step_many(acts, rew, fl)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
R = 10
for _ in range(R):
    env.step_many(acts, rew, fl)
e1.record()
torch.cuda.synchronize()
us = e0.elapsed_time(e1) / (R * K) * 1e3
print("%s n=%d K=%d: %.3f us per batched step -> %.1f G env-steps/s" % (layout, n, K, us, n / us / 1e3))
