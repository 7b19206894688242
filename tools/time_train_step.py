"""Time one step of the batched training environment (VecOvercookedMultiAgent): python tools/time_train_step.py [layout] [n_envs]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from overcooked_ai_amd.multi_agent import VecOvercookedMultiAgent

layout = sys.argv[1] if len(sys.argv) > 1 else "cramped_room"
n = int(sys.argv[2]) if len(sys.argv) > 2 else 65536
dev = torch.device("cuda:0")
for use_phi in (True, False):
    for dt in (torch.uint8, torch.float32):
        env = VecOvercookedMultiAgent(layout, n, horizon=400, reward_shaping_factor=1.0, device=dev, use_phi=use_phi, obs_dtype=dt)
        acts = torch.randint(0, 6, (64, n, 2), dtype=torch.uint8, device=dev)
        for i in range(20):
            env.step(acts[i % 64])
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for i in range(300):
            env.step(acts[i % 64])
        e1.record()
        torch.cuda.synchronize()
        us = e0.elapsed_time(e1) / 300 * 1e3
        print("%s n=%d use_phi=%s obs=%s: %.1f us per step" % (layout, n, use_phi, str(dt).split(".")[-1], us))
