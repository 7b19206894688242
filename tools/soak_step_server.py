"""Soak of the resident batched step around its idle / lifetime windows: python tools/soak_step_server.py [bursts] [idle_ms] [life_s]
Bursts of 1..8 steps separated by pauses around the idle window; every step compared with oc_step on a twin env."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

from overcooked_ai_amd.vec_env import VecOvercookedEnv

bursts = int(sys.argv[1]) if len(sys.argv) > 1 else 2000
idle_ms = float(sys.argv[2]) if len(sys.argv) > 2 else 1.0
life_s = float(sys.argv[3]) if len(sys.argv) > 3 else 0.02
dev = torch.device("cuda:0")
n, horizon = 1500, 11
res = VecOvercookedEnv("coordination_ring", n, horizon=horizon, device=dev, auto_reset=True)
one = VecOvercookedEnv("coordination_ring", n, horizon=horizon, device=dev, auto_reset=True)
rng = np.random.default_rng(7)
g = torch.Generator(device=dev).manual_seed(11)
pauses = [0.0, 0.0, idle_ms * 0.5e-3, idle_ms * 0.9e-3, idle_ms * 1.0e-3, idle_ms * 1.1e-3, idle_ms * 2e-3, life_s * 1.5]
t0 = time.time()
total = 0
with res.step_server(idle_ms=idle_ms, life_s=life_s) as sv:
    for burst in range(bursts):
        K = int(rng.integers(1, 9))
        acts = torch.randint(0, 6, (K, n, 2), dtype=torch.uint8, device=dev, generator=g)
        rew = torch.zeros((K, n, 4), dtype=torch.float32, device=dev)
        fl = torch.zeros((K, n), dtype=torch.uint8, device=dev)
        sv.play(acts, rew, fl)
        for k in range(K):
            r1, f1 = one.step(acts[k])
            assert torch.equal(r1, rew[k]) and torch.equal(f1, fl[k]), (burst, k)
        total += K
        time.sleep(float(rng.choice(pauses)))
    assert sv.steps == total
assert torch.equal(res.state, one.state) and torch.equal(res.ep_returns, one.ep_returns)
print("soak ok: %d bursts, %d steps, idle %.2f ms, life %.3f s, %.1f s" % (bursts, total, idle_ms, life_s, time.time() - t0))
