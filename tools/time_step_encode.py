"""Time one training-shaped step = transition + lossless observation: oc_step_encode (fused) vs oc_step + oc_encode_lossless.
   python tools/time_step_encode.py [layout] [n_envs]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from overcooked_ai_amd.vec_env import VecOvercookedEnv

layout = sys.argv[1] if len(sys.argv) > 1 else "asymmetric_advantages"
n = int(sys.argv[2]) if len(sys.argv) > 2 else 65536
dev = torch.device("cuda:0")
env = VecOvercookedEnv(layout, n, horizon=400, device=dev, auto_reset=True, seed=0)
acts = torch.randint(0, 6, (64, n, 2), dtype=torch.uint8, device=dev)
for dt in (torch.uint8, torch.float32):
    obs = torch.empty((n, 2, env.width, env.height, 26), dtype=dt, device=dev)
    for name in ("fused", "two kernels"):
        def step(i):
            if name == "fused":
                env.step_encode(acts[i % 64], dt, out=obs)
            else:
                env.step(acts[i % 64])
                env.encode_lossless(dt, out=obs)
        for i in range(20):
            step(i)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for i in range(300):
            step(i)
        e1.record()
        torch.cuda.synchronize()
        us = e0.elapsed_time(e1) / 300 * 1e3
        print("%s n=%d %s %-11s %.1f us per step+observation -> %.2f G env-steps/s" % (layout, n, str(dt).split(".")[-1], name, us, n / us / 1e3))
