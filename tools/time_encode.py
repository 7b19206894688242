"""Time oc_encode_lossless: python tools/time_encode.py [layout] [n_envs]   (OC_ENC_LDS=<bytes> varies the LDS image budget in a library built with `python -m overcooked_ai_amd.build --force -DOC_AMD_TUNING`)"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from overcooked_ai_amd.vec_env import VecOvercookedEnv

layout = sys.argv[1] if len(sys.argv) > 1 else "cramped_room"
n = int(sys.argv[2]) if len(sys.argv) > 2 else 65536
dev = torch.device("cuda:0")
env = VecOvercookedEnv(layout, n, horizon=400, device=dev, auto_reset=True, seed=1)
env.rollout_random(150)
for dt, el in ((torch.uint8, 1), (torch.float32, 4)):
    obs = torch.empty((n, 2, env.width, env.height, 26), dtype=dt, device=dev)
    for _ in range(10):
        env.encode_lossless(dt, out=obs)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(200):
        env.encode_lossless(dt, out=obs)
    e1.record()
    torch.cuda.synchronize()
    us = e0.elapsed_time(e1) / 200 * 1e3
    b = n * 2 * env.width * env.height * 26 * el
    print("%s n=%d %s OC_ENC_LDS=%s: %.1f us, %.2f TB/s" % (layout, n, str(dt).split(".")[-1], os.environ.get("OC_ENC_LDS", "default"), us, b / us / 1e6))
