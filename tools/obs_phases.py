"""Where k_train_step_obs spends its time (tuning build: OC_AMD_LIB=.../obs_tune.so): python tools/obs_phases.py [layout] [u8|f32]"""
import ctypes
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

from overcooked_ai_amd import _lib
from overcooked_ai_amd.multi_agent import VecOvercookedMultiAgent

layout = sys.argv[1] if len(sys.argv) > 1 else "cramped_room"
dt = torch.float32 if len(sys.argv) > 2 and sys.argv[2] == "f32" else torch.uint8
n = 65536
dev = torch.device("cuda:0")
env = VecOvercookedMultiAgent(layout, n, horizon=400, reward_shaping_factor=1.0, device=dev, use_phi=True, obs_dtype=dt)
acts = torch.randint(0, 6, (64, n, 2), dtype=torch.uint8, device=dev)
for i in range(150):
    env.step(acts[i % 64])
torch.cuda.synchronize()
L = _lib.load()
words = 256 * 8 * 8
buf = (ctypes.c_uint32 * words)()
L.oc_debug_train_obs.argtypes = [ctypes.c_void_p, ctypes.c_int]
assert L.oc_debug_train_obs(buf, words) == 0
d = np.frombuffer(buf, dtype=np.uint32).reshape(256, 8, 8).astype(np.float64) * 10.0  # ns
own, hel = d[:, :4, :], d[:, 4:, :]
names = ["start->barrier2", "barrier2->first claim", "copy (sum)", "players (sum)", "objects (sum)", "stream (sum)", "sub-groups x10", "start->done"]
print("%s %s: per wavefront, ns (mean / max over the launch's 1 024 owners, 1 024 helpers)" % (layout, dt))
for k, nm in enumerate(names):
    print("  %-24s owners %8.0f / %8.0f    helpers %8.0f / %8.0f" % (nm, own[..., k].mean(), own[..., k].max(), hel[..., k].mean(), hel[..., k].max()))
ng = d[..., 6] / 10.0
print("  per sub-group: copy %.0f players %.0f objects %.0f stream %.0f ns (owners); kernel = slowest wavefront %.0f ns" % (
    own[..., 2].sum() / ng[:, :4].sum(), own[..., 3].sum() / ng[:, :4].sum(), own[..., 4].sum() / ng[:, :4].sum(), own[..., 5].sum() / ng[:, :4].sum(), d[..., 7].max()))
