"""Time one-step launches (oc_step through VecOvercookedEnv.step): python tools/time_step.py [layout] [n_envs]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from overcooked_ai_amd.vec_env import VecOvercookedEnv

layout = sys.argv[1] if len(sys.argv) > 1 else "cramped_room"
n = int(sys.argv[2]) if len(sys.argv) > 2 else 65536
dev = torch.device("cuda:0")
env = VecOvercookedEnv(layout, n, horizon=400, device=dev, auto_reset=True, seed=0)
acts = torch.randint(0, 6, (64, n, 2), dtype=torch.uint8, device=dev)
for i in range(50):
    env.step(acts[i % 64])
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for i in range(500):
    env.step(acts[i % 64])
e1.record()
torch.cuda.synchronize()
print("%s n=%d lib=%s: %.2f us per oc_step call (back to back)" % (layout, n, os.environ.get("OC_AMD_LIB", "default"), e0.elapsed_time(e1) / 500 * 1e3))
