"""Tiny driver for rocprofv3 counter passes over the fused rollout kernel (keeps the rocpd database small):
    MODE=<lane_pair|predicate_interact> LAYOUT=<name> ENVS=<n> STEPS=<fused steps> python tools/prof_rollout.py
    CONFIG=4 | CONFIG=5: the 5-layout mix / the 4 096 generated terrains of BASELINE configs[3] / [4] instead of one layout"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from overcooked_ai_amd.vec_env import VecOvercookedEnv  # noqa: E402

dev = torch.device("cuda:0")
n = int(os.environ.get("ENVS", "65536"))
cfg = os.environ.get("CONFIG", "")
if cfg in ("4", "5"):
    import argparse

    import bench

    wl = bench.make_workload(argparse.Namespace(config=int(cfg), envs=n, layout="cramped_room"), 0)
    env = VecOvercookedEnv(wl["table"], n, horizon=400, device=dev, auto_reset=True, seed=0, layout_id=wl["lid"])
else:
    env = VecOvercookedEnv(os.environ.get("LAYOUT", "cramped_room"), n, horizon=400, device=dev, auto_reset=True, seed=0)
mode = os.environ.get("MODE", "")
if mode:
    setattr(env, mode, True)
T = int(os.environ.get("STEPS", "100"))
rew = torch.zeros((T, n, 4), dtype=torch.float32, device=dev)
fl = torch.zeros((T, n), dtype=torch.uint8, device=dev)
tiled8 = os.environ.get("TILED8", "") == "1"  # OC_OPT_FLAGS_TILED8: what bench.py's headline launches use
for _ in range(12):
    if tiled8:
        env.rollout_random(T, rew, fl.view(T // 8, n, 8), flags_tiled8=True)
    else:
        env.rollout_random(T, rew, fl)
torch.cuda.synchronize()
