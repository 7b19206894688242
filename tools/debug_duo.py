"""Find the first env-step where the mover / interact rollout differs from the oracle and print what happened there:
python tools/debug_duo.py [layout[,layout...]] [n_envs] [n_steps]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

from oracle import oracle as O
from overcooked_ai_amd.layouts import LayoutTable, spec_from_name
from overcooked_ai_amd.vec_env import VecOvercookedEnv

names = (sys.argv[1] if len(sys.argv) > 1 else "asymmetric_advantages").split(",")
n = int(sys.argv[2]) if len(sys.argv) > 2 else 2048
T = int(sys.argv[3]) if len(sys.argv) > 3 else 400
dev = torch.device("cuda:0")
OLD = os.environ.get("OLD") == "1"  # OLD=1: old dynamics; DRAW=0.35: drawn start states with that rnd_obj_prob_thresh; HORIZON
DRAW = float(os.environ.get("DRAW", "0"))
HZ = int(os.environ.get("HORIZON", "400"))
SEED, OFF, FUSE = int(os.environ.get("SEED", "0")), int(os.environ.get("ENV_OFFSET", "0")), int(os.environ.get("FUSE", "8"))
table = LayoutTable([spec_from_name(nm, old_dynamics=True) if OLD else spec_from_name(nm) for nm in names],
                    pad_to=(9, 5) if len(names) > 1 else None)
lid = ((np.arange(n) + OFF) % len(names)).astype(np.uint16)
kw = dict(random_start_pos=True, rnd_obj_prob_thresh=DRAW) if DRAW else {}
TRACK = os.environ.get("TRACK") == "1"  # TRACK=1: the event log (per-episode counters) against the oracle's event masks
if TRACK:
    kw["track_events"] = True
counts = np.zeros((n, 25, 2), np.int64)
env = VecOvercookedEnv(table, n, horizon=HZ, device=dev, auto_reset=True, seed=SEED, env_offset=OFF, layout_id=lid if len(names) > 1 else None, **kw)
env.one_wavefront = os.environ.get("ONE_WAVEFRONT") == "1"
orc = O.Oracle([O.mdp_from_layout_dict(s.to_layout_dict()) for s in table.specs])
lid_o = lid if len(names) > 1 else None
st = env.get_packed_state().copy() if DRAW else orc.reset(orc.new_state(n), layout_id=lid_o)
rew = torch.zeros((FUSE, n, 4), dtype=torch.float32, device=dev)
fl = torch.zeros((FUSE, n), dtype=torch.uint8, device=dev)
W, H = orc.W, orc.H
for t in range(0, T, FUSE):
    prev = st.copy()
    env.rollout_random(FUSE, rew, fl)
    got = env.get_packed_state()
    # oracle step by step to find the exact step
    cur = prev.copy()
    states = [cur.copy()]
    rews = []
    for k in range(FUSE):
        a = O.random_actions(SEED, OFF, t + k, n)
        sp = O.start_spec(seed=SEED, env_offset=OFF, epoch=1 + t + k, random_start_pos=True, rnd_obj_prob_thresh=DRAW) if DRAW else None
        cur, r, f = orc.step(cur, a, horizon=HZ, options=1, layout_id=lid_o, start=sp)
        states.append(cur.copy())
        rews.append(r)
        if TRACK:
            bits = ((orc.last_events[:, None] >> np.arange(50, dtype=np.uint64)[None, :]) & np.uint64(1)).astype(np.int64)
            counts += bits.reshape(n, 25, 2)
            counts[(f & 4) != 0] = 0
            states[-1] = (states[-1], orc.last_events.copy())
    st = cur
    if TRACK:
        evs = [x[1] for x in states[1:]]
        states = [states[0]] + [x[0] for x in states[1:]]
        got = env.event_counts.cpu().numpy().astype(np.int64)
        got2 = np.stack([got & 0xFFFF, (got >> 16) & 0xFFFF], -1)
        badc = np.argwhere(got2 != counts)
        if len(badc):
            e = int(badc[0][0])
            print("event counters differ at t=%d..%d: env %d layout %s (event, player) %s gpu %s oracle %s" % (
                t, t + FUSE, e, names[lid[e]], badc[badc[:, 0] == e][:, 1:].tolist(), got2[e][got2[e] != counts[e]].tolist(),
                counts[e][got2[e] != counts[e]].tolist()))
            for k in range(FUSE):
                a = O.random_actions(SEED, OFF, t + k, n)[e]
                s0 = states[k][:, e, :]
                cells = np.concatenate([s0[1 + p] for p in range(orc.n_planes - 1)])[:W * H].reshape(H, W)
                nz = [(int(y), int(x), int(cells[y, x])) for y in range(H) for x in range(W) if cells[y, x]]
                evk = int(evs[k][e])
                print(" t=%d actions %s hdr %s objects %s oracle events %s" % (t + k, a.tolist(), s0[0][:10].tolist(), nz,
                                                                             [(b >> 1, b & 1) for b in range(50) if (evk >> b) & 1]))
            print("\n".join("".join(r) for r in table.specs[lid[e]].terrain_mtx))
            break
    rew_g = rew.cpu().numpy()
    bad_r = np.argwhere(np.abs(rew_g - np.stack(rews)) > 1e-6)
    bad_s = np.argwhere(got != st)
    if len(bad_r) or len(bad_s):
        print("block at t=%d: %d reward mismatches, %d state byte mismatches" % (t, len(bad_r), len(bad_s)))
        envs = sorted(set(int(x[1]) for x in bad_r) | set(int(x[1]) for x in bad_s))[:3]
        for e in envs:
            print("== env %d layout %s" % (e, names[lid[e]]))
            ks = sorted(set(int(x[0]) for x in bad_r if int(x[1]) == e))
            k_lo = max(0, (ks[0] if ks else FUSE - 8) - 6)
            for k in range(k_lo, min(FUSE, k_lo + 10)):
                a = O.random_actions(SEED, OFF, t + k, n)[e]
                s0 = states[k][:, e, :]
                print(" t=%d actions %s  hdr %s" % (t + k, a.tolist(), s0[0].tolist()))
                cells = np.concatenate([s0[1 + p] for p in range(orc.n_planes - 1)])[:W * H].reshape(H, W)
                nz = [(int(y), int(x), int(cells[y, x])) for y in range(H) for x in range(W) if cells[y, x]]
                print("      objects (y,x,code) %s  oracle rew %s  gpu rew %s" % (nz, rews[k][e].tolist(), rew_g[k, e].tolist()))
            print(" final oracle hdr %s" % st[0][e].tolist())
            print(" final gpu    hdr %s" % got[0][e].tolist())
            for p in range(1, orc.n_planes):
                if (st[p][e] != got[p][e]).any():
                    print("  plane %d oracle %s\n           gpu    %s" % (p, st[p][e].tolist(), got[p][e].tolist()))
        g = table.specs[lid[envs[0]]].terrain_mtx
        print("\n".join("".join(r) for r in g))
        break
else:
    print("no mismatch in %d steps x %d envs" % (T, n))
