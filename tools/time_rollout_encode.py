"""Time K transitions + the observation of every step (oc_rollout_encode) against the one-step calls:
   python tools/time_rollout_encode.py [layout] [n_envs] [K] [f32] [actions]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from overcooked_ai_amd.vec_env import VecOvercookedEnv

layout = sys.argv[1] if len(sys.argv) > 1 else "asymmetric_advantages"
n = int(sys.argv[2]) if len(sys.argv) > 2 else 65536
K = int(sys.argv[3]) if len(sys.argv) > 3 else 40
dt = torch.float32 if "f32" in sys.argv[4:] else torch.uint8
use_actions = "actions" in sys.argv[4:]
dev = torch.device("cuda:0")
env = VecOvercookedEnv(layout, n, horizon=400, device=dev, auto_reset=True, seed=0)
per = (n, 2, env.width, env.height, 26)
rew = torch.zeros((K, n, 4), dtype=torch.float32, device=dev)
fl = torch.zeros((K, n), dtype=torch.uint8, device=dev)
acts = torch.randint(0, 6, (K, n, 2), dtype=torch.uint8, device=dev) if use_actions else None
for name, shape in (("trajectory buffer [K][n]", (K,) + per), ("single buffer [n]", per)):
    obs = torch.empty(shape, dtype=dt, device=dev)
    for _ in range(3):
        env.rollout_encode(K, obs, rew, fl, actions=acts, dtype=dt)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    R = 10
    for _ in range(R):
        env.rollout_encode(K, obs, rew, fl, actions=acts, dtype=dt)
    e1.record()
    torch.cuda.synchronize()
    us = e0.elapsed_time(e1) / (R * K) * 1e3
    b = n * 2 * env.width * env.height * 26 * obs.element_size()
    print("%s n=%d K=%d %s: %.1f us per step+observation -> %.2f G env-steps/s, %.2f TB/s of observations"
          % (layout, n, K, name, us, n / us / 1e3, b / us / 1e6))
    del obs
