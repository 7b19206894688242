"""What sets a process's level on the headline launch?  All figures: 4 000-step launches of 65 536 cramped_room
envs with tiled flags (the driver's headline launch), G env-steps/s.  Run several processes, under different HSA_* / GPU_* settings,
on ONE box (tools/gpu_r5_level.sh):
  python tools/process_level.py [tag]        one line: median (slowest..fastest) of 12 launches per placement of the output arrays —
                                             as torch hands them out, carved from one buffer at +4K / +64K / +1M / +0, fresh allocations
  python tools/process_level.py gaps         rewards and flags carved from ONE buffer with a gap between them, both orders; separate allocations
  python tools/process_level.py series       per-launch rate of the first 80 launches after an allocation, three allocations in one process"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from overcooked_ai_amd.vec_env import VecOvercookedEnv

mode = sys.argv[1] if len(sys.argv) > 1 and sys.argv[1] in ("gaps", "series") else "placements"
tag = sys.argv[1] if len(sys.argv) > 1 and mode == "placements" else "run"


def placements():
    n, T = 65536, 4000
    dev = torch.device("cuda:0")
    env = VecOvercookedEnv("cramped_room", n, horizon=400, device=dev, auto_reset=True, seed=0)


    def level(rew, fl):
        for _ in range(3):
            env.rollout_random(T, rew, fl, flags_tiled8=True)
        torch.cuda.synchronize()
        evs = [torch.cuda.Event(enable_timing=True) for _ in range(13)]
        for i in range(12):
            evs[i].record()
            env.rollout_random(T, rew, fl, flags_tiled8=True)
        evs[12].record()
        torch.cuda.synchronize()
        ms = sorted(a.elapsed_time(b) for a, b in zip(evs[:-1], evs[1:]))
        return n * T / (ms[len(ms) // 2] * 1e-3) / 1e9, n * T / (ms[0] * 1e-3) / 1e9, n * T / (ms[-1] * 1e-3) / 1e9


    out = []
    rew = torch.zeros((T, n, 4), dtype=torch.float32, device=dev)
    fl = torch.zeros((T // 8, n, 8), dtype=torch.uint8, device=dev)
    out.append(("base", level(rew, fl)))
    del rew, fl
    torch.cuda.empty_cache()
    big = torch.zeros((T * n * 17 + (8 << 20),), dtype=torch.uint8, device=dev)
    base = (-big.data_ptr()) % (2 << 20)
    for name, off in (("+4K", 4096), ("+64K", 65536), ("+1M", 1 << 20), ("+0", 0)):
        r = big[base + off: base + off + T * n * 16].view(torch.float32).view(T, n, 4)
        f = big[base + off + T * n * 16: base + off + T * n * 17].view(T // 8, n, 8)
        out.append((name, level(r, f)))
    del big, r, f
    torch.cuda.empty_cache()
    for k in range(2):
        pad = torch.empty(((37 + 64 * k) << 20,), dtype=torch.uint8, device=dev)  # shifts where the next segments land
        rew = torch.zeros((T, n, 4), dtype=torch.float32, device=dev)
        fl = torch.zeros((T // 8, n, 8), dtype=torch.uint8, device=dev)
        out.append(("fresh%d" % k, level(rew, fl)))
        del rew, fl, pad
        torch.cuda.empty_cache()
    print("%-22s " % tag + "  ".join("%s %.1f (%.1f..%.1f)" % (k, v[0], v[2], v[1]) for k, v in out))


def gaps():
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


def series():
    n, T = 65536, 4000
    dev = torch.device("cuda:0")
    env = VecOvercookedEnv("cramped_room", n, horizon=400, device=dev, auto_reset=True, seed=0)
    for rnd in range(3):
        rew = torch.zeros((T, n, 4), dtype=torch.float32, device=dev)
        fl = torch.zeros((T // 8, n, 8), dtype=torch.uint8, device=dev)
        torch.cuda.synchronize()
        N = 80
        evs = [torch.cuda.Event(enable_timing=True) for _ in range(N + 1)]
        for i in range(N):
            evs[i].record()
            env.rollout_random(T, rew, fl, flags_tiled8=True)
        evs[N].record()
        torch.cuda.synchronize()
        g = [n * T / (a.elapsed_time(b) * 1e-3) / 1e9 for a, b in zip(evs[:-1], evs[1:])]
        print("allocation %d (rewards at %#x): launches 1-80, G env-steps/s: %s" % (rnd, rew.data_ptr(), " ".join("%.0f" % x for x in g)))
        del rew, fl
        torch.cuda.empty_cache()
        pad = torch.empty(((53 + 64 * rnd) << 20,), dtype=torch.uint8, device=dev)


{"placements": placements, "gaps": gaps, "series": series}[mode]()
