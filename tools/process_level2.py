"""Follow-up of tools/process_level.py: rewards and flags carved from ONE buffer with a gap between them — which relative placement
of the two output streams is slow?  python tools/process_level2.py"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from overcooked_ai_amd.vec_env import VecOvercookedEnv

n, T = 65536, 4000
dev = torch.device("cuda:0")
env = VecOvercookedEnv("cramped_room", n, horizon=400, device=dev, auto_reset=True, seed=0)


def level(rew, fl, tiled=True):
    for _ in range(2):
        env.rollout_random(T, rew, fl, flags_tiled8=tiled)
    torch.cuda.synchronize()
    evs = [torch.cuda.Event(enable_timing=True) for _ in range(9)]
    for i in range(8):
        evs[i].record()
        env.rollout_random(T, rew, fl, flags_tiled8=tiled)
    evs[8].record()
    torch.cuda.synchronize()
    ms = sorted(a.elapsed_time(b) for a, b in zip(evs[:-1], evs[1:]))
    return n * T / (ms[len(ms) // 2] * 1e-3) / 1e9


MiB = 1 << 20
big = torch.zeros((T * n * 17 + 2200 * MiB,), dtype=torch.uint8, device=dev)
base = (-big.data_ptr()) % (2 * MiB)
R = T * n * 16
row = []
for gap in (0, 4096, 65536, MiB // 2, MiB, 2 * MiB, 3 * MiB, 4 * MiB, 8 * MiB, 16 * MiB, 32 * MiB, 64 * MiB, 96 * MiB, 128 * MiB, 256 * MiB, 512 * MiB, 1024 * MiB, 2048 * MiB):
    r = big[base: base + R].view(torch.float32).view(T, n, 4)
    f = big[base + R + gap: base + R + gap + T * n].view(T // 8, n, 8)
    row.append("gap %s: %.1f" % (("%d MiB" % (gap // MiB)) if gap >= MiB else ("%d KiB" % (gap // 1024)), level(r, f)))
print("flags behind the rewards, one buffer:  " + "  ".join(row))
row = []
for gap in (0, 2 * MiB, 64 * MiB, 1024 * MiB):  # flags in FRONT of the rewards
    f = big[base: base + T * n].view(T // 8, n, 8)
    r = big[base + 250 * MiB + gap: base + 250 * MiB + gap + R].view(torch.float32).view(T, n, 4)
    row.append("gap %d MiB: %.1f" % (gap // MiB, level(r, f)))
print("flags in front of the rewards:  " + "  ".join(row))
del big, r, f
torch.cuda.empty_cache()
# separate allocations in both orders, and the [step][env] flags layout
rew = torch.zeros((T, n, 4), dtype=torch.float32, device=dev)
fl = torch.zeros((T // 8, n, 8), dtype=torch.uint8, device=dev)
print("separate allocations, rewards first: %.1f  (rewards at %#x, flags at %#x)" % (level(rew, fl), rew.data_ptr(), fl.data_ptr()))
print("   the same arrays, [step][env] flags: %.1f" % level(rew, fl.view(T, n), tiled=False))
del rew, fl
torch.cuda.empty_cache()
fl = torch.zeros((T // 8, n, 8), dtype=torch.uint8, device=dev)
rew = torch.zeros((T, n, 4), dtype=torch.float32, device=dev)
print("separate allocations, flags first:   %.1f  (rewards at %#x, flags at %#x)" % (level(rew, fl), rew.data_ptr(), fl.data_ptr()))
