"""Time oc_featurize (k_featurize): python tools/time_featurize.py [layout ...]  (65 536 envs, mid-episode states)"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from overcooked_ai_amd.vec_env import VecOvercookedEnv

dev = torch.device("cuda:0")
n = 65536
for layout in (sys.argv[1:] or ["asymmetric_advantages", "cramped_room", "counter_circuit_o_1order"]):
    env = VecOvercookedEnv(layout, n, horizon=400, device=dev, auto_reset=True, seed=1)
    env.rollout_random(150)
    feat = torch.empty((n, 2, 96), dtype=torch.float32, device=dev)
    for _ in range(10):
        env.featurize(out=feat)
    torch.cuda.synchronize()
    best = 1e9
    for rep in range(5):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(300):
            env.featurize(out=feat)
        e1.record()
        torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1) / 300 * 1e3)
    b = n * (env.n_planes * 16 + 2 * 96 * 4)
    print("%s: k_featurize %.2f us per launch, %.1f MB, %.2f TB/s = %.3f of 8 TB/s" % (layout, best, b / 1e6, b / best / 1e6, b / best / 1e6 / 8))
