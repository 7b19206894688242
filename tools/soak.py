"""Randomized differential soak test, HIP kernels vs the C oracle (run on a GPU box; not part of the test suite):

    python tools/soak.py [--seeds 40] [--envs 8192] [--steps 900]

Every seed draws a fresh table of generated layouts (random shapes, feature densities, tomatoes, recipe parameters,
old dynamics), random start states from oc_reset_random, and runs fused rollouts (all three kernel families in turn)
and explicit-action steps with event masks across the auto-reset boundary; states, rewards, flags, event masks,
encodings (also as oc_rollout_encode trajectories), features and potentials are compared bit for bit with the oracle."""
import argparse
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import oracle as O  # noqa: E402
from overcooked_ai_amd.layout_gen import generate_layouts  # noqa: E402
from overcooked_ai_amd.layouts import LayoutTable  # noqa: E402
from overcooked_ai_amd.potential import potential_params  # noqa: E402
from overcooked_ai_amd.vec_env import VecOvercookedEnv  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--seeds", type=int, default=40)
    ap.add_argument("--envs", type=int, default=8192)
    ap.add_argument("--steps", type=int, default=900)
    args = ap.parse_args()
    dev = torch.device("cuda:0")
    t_start, total = time.time(), 0
    for seed in range(args.seeds):
        rng = np.random.default_rng(1000 + seed)
        shape = [(5, 4), (5, 5), (7, 5), (9, 5), (8, 6), (12, 7)][seed % 6]
        feats = ("P", "O", "D", "S", "T") if seed % 2 else ("P", "O", "D", "S")
        base = {}
        if seed % 3 == 1:
            base = {"start_all_orders": [{"ingredients": ["onion"] * 3}, {"ingredients": ["onion", "tomato"]},
                                         {"ingredients": ["tomato"] * 2}, {"ingredients": ["onion", "onion", "tomato"]}],
                    "start_bonus_orders": [{"ingredients": ["onion", "tomato"]}], "onion_value": 9, "tomato_value": 6,
                    "onion_time": int(rng.integers(1, 6)), "tomato_time": int(rng.integers(1, 9))}
        elif seed % 3 == 2:
            base = {"cook_time": int(rng.integers(1, 30)), "delivery_reward": int(rng.integers(1, 50))}
        if seed % 5 == 4 and not base.get("start_all_orders"):
            base["old_dynamics"] = True  # needs 3-item orders only (mdp.py:1121-1127)
            base["start_all_orders"] = [{"ingredients": ["onion"] * 3}, {"ingredients": ["onion", "onion", "tomato"]}]
        n_lay = int(rng.choice([1, 3, 17, 40]))
        specs = generate_layouts(n_lay, seed=seed, inner_shape=shape, prop_empty=float(rng.uniform(0.5, 0.95)),
                                 prop_feats=float(rng.uniform(0.1, 0.5)), feature_types=feats, base_params=base)
        table = LayoutTable(specs)
        n = args.envs
        lid = rng.integers(0, n_lay, size=n).astype(np.uint16) if n_lay > 1 else None
        horizon = int(rng.choice([37, 150, 400]))
        # the default kernels (k_rollout4 in its MODE 0 / 1 / 2 instances, k_step3) on most seeds, the cross-check families on
        # the others; every third default seed with a table re-draws the layout of every new episode (regen_mdp)
        mode = [None, None, None, None, "lane_pair", None, "predicate_interact", None][seed % 8]
        regen = (0, n_lay) if (n_lay > 1 and mode is None and seed % 3 == 0) else None
        env = VecOvercookedEnv(table, n, horizon=horizon, device=dev, layout_id=lid, auto_reset=True, seed=seed,
                               regen_layout=bool(regen))
        orc = O.Oracle([O.mdp_from_layout_dict(s.to_layout_dict()) for s in specs])
        spec_now = lambda: O.start_spec(seed, 0, env.reset_epoch, regen=regen) if regen else None
        thr = float(rng.choice([0.0, 0.3, 0.8]))
        epoch = env.reset_epoch
        env.reset(random_start_pos=True, rnd_obj_prob_thresh=thr)
        if regen:
            O.regen_layouts(lid, O.start_spec(seed, 0, epoch, regen=regen))
            assert np.array_equal(env.layout_ids(), lid), ("regen at reset", seed)
        st = orc.reset_random(orc.new_state(n), seed=seed, epoch=epoch, random_start_pos=True, rnd_obj_prob_thresh=thr, layout_id=lid)
        assert np.array_equal(env.get_packed_state(), st), ("reset_random", seed)
        if mode:
            setattr(env, mode, True)
        t0, done = 0, 0
        while done < args.steps:
            k = int(rng.integers(1, 130))
            rew = torch.zeros((k, n, 4), dtype=torch.float32, device=dev)
            fl = torch.zeros((k, n), dtype=torch.uint8, device=dev)
            sp = spec_now()
            env.rollout_random(k, rew, fl)
            r_o, f_o = orc.rollout_random(st, k, horizon=horizon, options=1, seed=seed, t0=t0, layout_id=lid, start=sp)
            assert lid is None or np.array_equal(env.layout_ids(), lid), ("layout ids", seed, done)
            assert np.array_equal(env.get_packed_state(), st), ("rollout state", seed, done)
            assert np.array_equal(rew.cpu().numpy(), r_o) and np.array_equal(fl.cpu().numpy(), f_o), ("rollout outputs", seed, done)
            t0 += k
            done += k
            # explicit actions with event masks
            acts = rng.integers(0, 6, size=(n, 2)).astype(np.uint8)
            ev = torch.zeros((n,), dtype=torch.int64, device=dev)
            sp = spec_now()
            r, f = env.step(torch.from_numpy(acts).to(dev), events_out=ev)
            st2, r2, f2 = orc.step(st, acts, horizon=horizon, options=1, layout_id=lid, start=sp)
            assert np.array_equal(env.get_packed_state(), st2) and np.array_equal(r.cpu().numpy(), r2) and np.array_equal(f.cpu().numpy(), f2), ("step", seed)
            assert np.array_equal(ev.cpu().numpy().view(np.uint64), orc.last_events), ("events", seed)
            st = st2
            # the same call in place without event logging = k_step1 (the transition on the wire format), an illegal action now and then
            acts = rng.integers(0, 6, size=(n, 2)).astype(np.uint8)
            acts[rng.integers(0, n, 2), rng.integers(0, 2, 2)] = 6 + int(rng.integers(0, 200))
            sp = spec_now()
            r, f = env.step(torch.from_numpy(acts).to(dev))
            st, r2, f2 = orc.step(st, acts, horizon=horizon, options=1, layout_id=lid, start=sp)
            assert np.array_equal(env.get_packed_state(), st) and np.array_equal(r.cpu().numpy(), r2) and np.array_equal(f.cpu().numpy(), f2), ("lean step", seed)
            assert lid is None or np.array_equal(env.layout_ids(), lid), ("layout ids after the lean step", seed, done)
            done += 2
            total += n * (k + 2)
        # K caller-supplied-action transitions in one launch (oc_step_many), illegal actions included
        K = int(rng.integers(2, 40))
        a = rng.integers(0, 6, size=(K, n, 2)).astype(np.uint8)
        a[rng.integers(0, K, 8), rng.integers(0, n, 8), rng.integers(0, 2, 8)] = 6
        rew = torch.zeros((K, n, 4), dtype=torch.float32, device=dev)
        fl = torch.zeros((K, n), dtype=torch.uint8, device=dev)
        ep0 = env.reset_epoch
        env.step_many(torch.from_numpy(a).to(dev), rew, fl)
        for k in range(K):
            st, r2, f2 = orc.step(st, a[k], horizon=horizon, options=1, layout_id=lid,
                                  start=O.start_spec(seed, 0, ep0 + k, regen=regen) if regen else None)
            assert np.array_equal(rew[k].cpu().numpy(), r2) and np.array_equal(fl[k].cpu().numpy(), f2), ("step_many", seed, k)
        assert np.array_equal(env.get_packed_state(), st), ("step_many state", seed)
        assert lid is None or np.array_equal(env.layout_ids(), lid), ("layout ids after step_many", seed)
        total += n * K
        enc = env.encode_lossless(torch.uint8).cpu().numpy().astype(np.int32)
        assert np.array_equal(enc, orc.encode_lossless(st, horizon=horizon, layout_id=lid)), ("encode", seed)
        # K transitions with the observation of every step in one call (oc_rollout_encode; k_rollout_encode where the
        # table allows it), random policy, against the oracle step by step
        if all(s.num_players == 2 for s in specs):
            K = int(rng.integers(2, 24))
            n2 = min(n, 1024)  # (the trajectory buffer: K x n2 observations)
            e2 = VecOvercookedEnv(table, n2, horizon=horizon, device=dev, layout_id=None if lid is None else lid[:n2],
                                  auto_reset=True, seed=seed)
            e2.one_kernel = True
            st2 = np.ascontiguousarray(st[:, :n2])
            e2.set_packed_state(st2)
            e2.t_global = 5
            dt = torch.uint8 if seed % 2 else torch.float32
            obs = torch.zeros((K, n2, 2, table.width, table.height, 26), dtype=dt, device=dev)
            rew = torch.zeros((K, n2, 4), dtype=torch.float32, device=dev)
            fl = torch.zeros((K, n2), dtype=torch.uint8, device=dev)
            e2.rollout_encode(K, obs, rew, fl, dtype=dt)
            lid2 = None if lid is None else lid[:n2]
            for k in range(K):
                r_o, f_o = orc.rollout_random(st2, 1, horizon=horizon, options=1, seed=seed, t0=5 + k, layout_id=lid2)
                assert np.array_equal(rew[k].cpu().numpy(), r_o[0]) and np.array_equal(fl[k].cpu().numpy(), f_o[0]), ("rollout_encode outputs", seed, k)
                assert np.array_equal(obs[k].cpu().numpy().astype(np.int32), orc.encode_lossless(st2, horizon=horizon, layout_id=lid2)), ("rollout_encode obs", seed, k)
            assert np.array_equal(e2.get_packed_state(), st2), ("rollout_encode state", seed)
            total += n2 * K
        if all(s.num_players == 2 for s in specs):
            cg = "all" if seed % 2 else "none"
            assert np.array_equal(env.featurize(counter_goals=cg, num_pots=seed % 4).cpu().numpy(),
                                  O.featurize(orc, st, counter_goals=cg, num_pots=seed % 4, layout_id=lid)), ("featurize", seed)
        try:
            pp = [potential_params(s, 0.99) for s in specs]
            phi = env.potential(0.99).cpu().numpy()
            assert np.array_equal(phi, O.potential(orc, st, pp, layout_id=lid), equal_nan=True), ("potential", seed)
        except ValueError:
            pass
        print("seed %d ok: %d layouts %s, horizon %d, mode %s, thr %.1f%s" % (seed, n_lay, shape, horizon, mode, thr,
                                                                            ", layout re-drawn every episode" if regen else ""), flush=True)
    print("soak ok: %d seeds, %.1f M env-steps compared in %.0f s" % (args.seeds, total / 1e6, time.time() - t_start))


if __name__ == "__main__":
    main()
