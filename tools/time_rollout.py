"""Time oc_rollout_random launches (400 fused steps) for a layout / batch size:  python tools/time_rollout.py [layout] [n_envs] [v3]"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from overcooked_ai_amd.vec_env import VecOvercookedEnv

layout = sys.argv[1] if len(sys.argv) > 1 else "cramped_room"
n = int(sys.argv[2]) if len(sys.argv) > 2 else 65536
dev = torch.device("cuda:0")
env = VecOvercookedEnv(layout, n, horizon=400, device=dev, auto_reset=True, seed=0, track_events=os.environ.get("TRACK_EVENTS") == "1")
T = 400
rew = torch.zeros((T, n, 4), dtype=torch.float32, device=dev)
fl = torch.zeros((T, n), dtype=torch.uint8, device=dev)
if "noout" in sys.argv[3:]:
    rew = fl = None
for _ in range(5):
    env.rollout_random(T, rew, fl)
torch.cuda.synchronize()
evs = [torch.cuda.Event(enable_timing=True) for _ in range(41)]
for i in range(40):
    evs[i].record()
    env.rollout_random(T, rew, fl)
evs[40].record()
torch.cuda.synchronize()
ms = sorted(a.elapsed_time(b) for a, b in zip(evs[:-1], evs[1:]))
med = ms[len(ms) // 2]
print("%s n=%d %s: launch median %.1f us, min %.1f us -> %.3f us/step, %.1f G env-steps/s" % (
    layout, n, "v4", med * 1e3, ms[0] * 1e3, med * 1e3 / T, n * T / (med * 1e-3) / 1e9))
