"""benchlib.legs — the side legs of the default bench line: BASELINE configs[2..4] (`configs`), the general rollout path, the observation
kernels, the one-launch-per-step / resident / single-env step APIs and the training environment."""
import argparse
import json
import os
import sys
import time

from benchlib.common import *  # noqa: F401,F403 (constants and helpers)
from benchlib.common import _StubEnv, _Timer  # noqa: F401
from benchlib.cpu_baseline import _reference_python_stored
from benchlib.traffic import measure_traffic


def side_legs(args, torch, VecOvercookedEnv, sharding, dev):
    """BASELINE configs[2], [3], [4] on this GPU — what `--config 3 / 4 / 5` print, bounded to --leg-seconds each: value,
    median launch duration (HIP events), roofline, parity check against the C oracle (config 3: the observation of every
    step of one launch; configs 4 / 5: 1 200 steps from reset across two restarts).  Config 5 runs the shape ONE rank of
    the 8-GPU config launches: 131 072 envs."""
    legs = {}
    try:
        legs["3"] = encode_measure(torch, VecOvercookedEnv, sharding, dev, 0, 1, N_ENVS_PER_GPU, seconds=args.leg_seconds,
                                   parity=not args.no_parity_check, extras=False)
    except Exception as exc:  # a side leg must never cost the headline line
        legs["3"] = {"error": repr(exc)[:300]}
    for cfg, envs in ((4, N_ENVS_PER_GPU), (5, 2 * N_ENVS_PER_GPU)):
        try:
            a = argparse.Namespace(config=cfg, envs=envs, layout="cramped_room", terrains=4096, stub=False, fuse=DEFAULT_FUSE,
                                   lane_pair=False, predicate_interact=False, one_wavefront=getattr(args, "one_wavefront", False),
                                   flags_layout=getattr(args, "flags_layout", "step"))
            wl = make_workload(a, 0)
            make_env = rollout_workload_env(a, wl, envs, 0, dev, VecOvercookedEnv)
            env = make_env()
            fuse = DEFAULT_FUSE
            rew = torch.zeros((fuse, envs, 4), dtype=torch.float32, device=dev)
            fl = torch.zeros((fuse, envs), dtype=torch.uint8, device=dev)

            tiled8 = flags_tiled8_ok(a, env, fuse, rew, fl)
            fl_t = fl.view(fuse // 8, envs, 8) if tiled8 else None

            def launch():
                if tiled8:
                    env.rollout_random(fuse, rew, fl_t, flags_tiled8=True)
                else:
                    env.rollout_random(fuse, rew, fl)

            launch()
            k = launches_for(torch, dev, launch, args.leg_seconds)
            wall, ms = timed_launches(torch, dev, sharding, launch, k)
            ms = sorted(ms)
            med = ms[len(ms) // 2]
            bpl = envs * (2 * wl["sbytes"] + OUT_BYTES * fuse)
            leg = {"value": envs * fuse * k / wall, "unit": "env steps/s (one GPU)", "envs": envs, "launches": k,
                   "timed_region_s": wall, "launch_ms": med, "launch_ms_min": ms[0], "workload": wl["workload"],
                   "flags_layout": "[steps/8][envs][8] (OC_OPT_FLAGS_TILED8)" if tiled8 else "[steps][envs]",
                   "roofline": {"bound": "hbm", "kernel": "k_rollout5|k_rollout4", "achieved": bpl / (med * 1e-3) / 1e9,
                                "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": bpl / (med * 1e-3) / 1e9 / HBM_PEAK_GBS,
                                "bytes_per_launch": bpl, "traffic": None,
                                "traffic_source": {"how": "not collected", "why": "side leg; `bench.py --config %d` collects it" % cfg},
                                "bytes_model": "n_envs*(2*S + 17*T), S=%d B (SURVEY 8d)" % wl["sbytes"]}}
            if not args.no_parity_check:
                leg["parity_check"] = parity_check(torch, wl, make_env, envs, 0, 1200, rew, fl, usable_cores(), tiled8=tiled8)
            del env, rew, fl
            if not getattr(args, "no_traffic", False):  # two --pmc child passes of this leg's launch shape (no nulls)
                torch.cuda.empty_cache()
                leg["roofline"]["traffic"], leg["roofline"]["traffic_source"] = measure_traffic(a, "k_rollout5|k_rollout4", tiled8)
            legs[str(cfg)] = leg
            continue
        except Exception as exc:
            legs[str(cfg)] = {"error": repr(exc)[:300]}
    try:  # config 3's traffic last (its 7.7 GB trajectory buffer is gone by now): 10-step launches in the child
        if "roofline" in legs.get("3", {}) and not getattr(args, "no_traffic", False):
            torch.cuda.empty_cache()
            a3 = argparse.Namespace(config=3, envs=N_ENVS_PER_GPU, layout="asymmetric_advantages", terrains=4096, fuse=PMC_ENC_FUSE,
                                    lane_pair=False, predicate_interact=False, one_wavefront=False)
            t10, src = measure_traffic(a3, "k_rollout_encode", False)
            rl = legs["3"]["roofline"]
            if t10 is not None:
                n, per_step = N_ENVS_PER_GPU, OUT_BYTES + 2 * 9 * 5 * 26
                b10 = n * (2 * S_ASYM + per_step * PMC_ENC_FUSE)
                src = dict(src, measured_on="launches of %d steps (a %d-step launch wraps WRITE_SIZE): %d bytes measured against %d "
                                            "algorithmic; `traffic` = that ratio x bytes_per_launch" % (PMC_ENC_FUSE, ENC_FUSE, int(t10), b10),
                           traffic_over_algorithmic=t10 / b10)
                rl["traffic"] = rl["bytes_per_launch"] * t10 / b10
            rl["traffic_source"] = src
    except Exception as exc:
        legs["3"]["roofline"]["traffic_source"] = {"how": "not collected", "why": repr(exc)[:200]}
    return legs

def general_case(name):
    """(LayoutTable, VecOvercookedEnv keyword arguments) of a general_path leg — shared with the --pmc child that replays it."""
    from overcooked_ai_amd.layouts import LayoutTable, spec_from_name

    if name == "coordination_ring_old_dynamics":
        return LayoutTable([spec_from_name("coordination_ring", old_dynamics=True)]), {}
    if name == "asymmetric_advantages_old_dynamics":
        return LayoutTable([spec_from_name("asymmetric_advantages", old_dynamics=True)]), {}
    if name == "cramped_room_event_log":
        return LayoutTable([spec_from_name("cramped_room")]), {"track_events": True}
    if name == "marshmallow_experiment":
        return LayoutTable([spec_from_name("marshmallow_experiment")]), {}
    raise ValueError("unknown general_path leg %r" % name)


GENERAL_CASES = ("coordination_ring_old_dynamics", "asymmetric_advantages_old_dynamics", "cramped_room_event_log", "marshmallow_experiment")


def general_legs(args, torch, VecOvercookedEnv, sharding, dev):
    """The batches OUTSIDE "two players, <= 2 pots, <= 64 cells, new dynamics, no event log" (VERDICT r5 #3): old dynamics — what
    the reference's paper-reproduction runs use (human_aware_rl/ppo/run_experiments.sh:4-12; mdp.py:1517-1518, 1696-1701) —,
    per-episode event logging (env.py:382-401 game_stats), and a 13 x 5 layout (65 cells).  Same launch shape as the headline
    (65 536 envs x 4 000 fused steps), each with roofline and an oracle parity check."""
    from overcooked_ai_amd.layouts import LayoutTable, spec_from_name

    legs = {}
    n, fuse = N_ENVS_PER_GPU, DEFAULT_FUSE
    for name in GENERAL_CASES:
        try:
            table, kw = general_case(name)
            wl = {"table": table, "specs": table.specs, "lid": None, "sbytes": 4 * ((table.n_planes * 16) // 4),
                  "workload": "%s x %d envs, random policy, horizon %d auto-reset, outputs every step" % (name, n, HORIZON)}

            def make_env():
                return VecOvercookedEnv(table, n, horizon=HORIZON, device=dev, auto_reset=True, seed=0, **kw)

            env = make_env()
            rew = torch.zeros((fuse, n, 4), dtype=torch.float32, device=dev)
            fl = torch.zeros((fuse, n), dtype=torch.uint8, device=dev)
            a = argparse.Namespace(flags_layout=getattr(args, "flags_layout", "step"), stub=False)
            tiled8 = flags_tiled8_ok(a, env, fuse, rew, fl)
            fl_t = fl.view(fuse // 8, n, 8) if tiled8 else None

            def launch():
                if tiled8:
                    env.rollout_random(fuse, rew, fl_t, flags_tiled8=True)
                else:
                    env.rollout_random(fuse, rew, fl)

            launch()
            k = launches_for(torch, dev, launch, args.leg_seconds)
            wall, ms = timed_launches(torch, dev, sharding, launch, k)
            ms = sorted(ms)
            med = ms[len(ms) // 2]
            bpl = n * (2 * wl["sbytes"] + OUT_BYTES * fuse)
            leg = {"value": n * fuse * k / wall, "unit": "env steps/s (one GPU)", "envs": n, "launches": k, "launch_ms": med,
                   "workload": wl["workload"], "flags_layout": "tiled8" if tiled8 else "[steps][envs]",
                   "roofline": {"bound": "hbm", "achieved": bpl / (med * 1e-3) / 1e9, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                                "frac": bpl / (med * 1e-3) / 1e9 / HBM_PEAK_GBS, "bytes_per_launch": bpl}}
            if not args.no_parity_check:
                leg["parity_check"] = parity_check(torch, wl, make_env, n, 0, 1200, rew, fl, usable_cores(), tiled8=tiled8)
            legs[name] = leg
            del env, rew, fl
            torch.cuda.empty_cache()
            if not getattr(args, "no_traffic", False):  # two --pmc child passes of this leg's launch shape, as for the configs legs
                ca = argparse.Namespace(config=2, envs=n, fuse=fuse, layout="cramped_room", terrains=getattr(args, "terrains", 4096),
                                        lane_pair=False, predicate_interact=False, one_wavefront=False)
                leg["roofline"]["traffic"], leg["roofline"]["traffic_source"] = measure_traffic(
                    ca, "k_rollout5|k_rollout4", tiled8, extra=["--general-leg", name])
        except Exception as exc:
            legs[name] = {"error": repr(exc)[:300]}
    return legs

def encode_measure(torch, VecOvercookedEnv, sharding, dev, rank, world, n, launches=0, warm_launches=2, seconds=0.0,
                   parity=True, extras=True):
    """BASELINE configs[2] (the rollout of configs[1] plus oc_encode_lossless every step) on this rank's
    shard: ENC_FUSE transitions per oc_rollout_encode launch, the observation of every step kept ([ENC_FUSE][n] u8
    trajectory buffer: 7.7 GB at 65 536 9x5 envs).  `launches` fixed (the --config 3 line) or as many as fill `seconds`
    (the side leg of the default line)."""
    import numpy as np

    env = VecOvercookedEnv("asymmetric_advantages", n, horizon=HORIZON, device=dev, auto_reset=True, seed=0,
                           env_offset=rank * n)
    workload, sbytes = ("asymmetric_advantages x %d envs/GPU, random policy (in-kernel Philox actions) + lossless u8 "
                        "encoding of every step into a [steps][envs] trajectory buffer (oc_rollout_encode)" % n), S_ASYM
    fuse = ENC_FUSE
    rew = torch.zeros((fuse, n, 4), dtype=torch.float32, device=dev)
    fl = torch.zeros((fuse, n), dtype=torch.uint8, device=dev)
    obs = torch.empty((fuse, n, 2, env.width, env.height, 26), dtype=torch.uint8, device=dev)

    def launch():  # one `fuse`-step unit of the workload
        env.rollout_encode(fuse, obs, rew, fl)

    for _ in range(max(1, warm_launches)):
        launch()
    if not launches:
        launches = launches_for(torch, dev, launch, seconds)
    wall, per_launch_ms = timed_launches(torch, dev, sharding, launch, launches)
    per_launch = sorted(per_launch_ms)
    unit_med = per_launch[len(per_launch) // 2]
    tmax = torch.tensor([wall], dtype=torch.float64, device=dev)
    sharding.allreduce_max(tmax)
    wall = float(tmax.item())
    unit_bytes = n * (2 * sbytes + OUT_BYTES * fuse) + fuse * n * 2 * env.width * env.height * 26
    # parity of this launch shape: one ENC_FUSE-step launch from reset — every reward quad and flag byte, the final states,
    # and the u8 observation of EVERY step of every env (the oracle's encoder threaded over the host cores; compared on
    # the GPU) against the C oracle
    pc = None
    if parity:
        from oracle import oracle as O
        from overcooked_ai_amd.layouts import spec_from_name

        t_par = time.perf_counter()
        threads = O.set_threads(max(1, usable_cores() // max(1, world)))
        orc = O.Oracle(O.mdp_from_layout_dict(spec_from_name("asymmetric_advantages").to_layout_dict()))
        env2 = VecOvercookedEnv("asymmetric_advantages", n, horizon=HORIZON, device=dev, auto_reset=True, seed=0, env_offset=rank * n)
        env2.rollout_encode(fuse, obs, rew, fl)
        st = orc.reset(orc.new_state(n))
        bad, bad_obs = 0, 0
        rg, fg = rew.cpu().numpy(), fl.cpu().numpy()
        for k in range(fuse):
            r_o, f_o = orc.rollout_random(st, 1, horizon=HORIZON, options=1, seed=0, env_offset=rank * n, t0=k)
            bad += int(((rg[k] != r_o[0]).any(axis=1) | (fg[k] != f_o[0])).sum())
            enc_o = torch.from_numpy(O.encode_lossless_u8(orc, st, horizon=HORIZON)).to(dev)
            bad_obs += int((obs[k] != enc_o).flatten(1).any(dim=1).sum().item())
        bad_states = int((env2.get_packed_state() != st).any(axis=(0, 2)).sum())
        O.set_threads(1)
        del env2
        pc = {"envs": n, "steps": fuse, "mismatches": bad + bad_obs + bad_states, "mismatching_env_steps": bad,
              "mismatching_observations": bad_obs, "mismatching_final_states": bad_states,
              "observations_checked": "every step (%d) x every env (%d)" % (fuse, n),
              "seconds": time.perf_counter() - t_par, "oracle_threads": threads,
              "what": "one %d-step oc_rollout_encode launch from reset: every reward quad and flag byte, the final states and "
                      "the u8 observation of every env-step against oracle/overcooked_oracle.c" % fuse}
    out = {"value": float(world) * n * fuse * launches / wall, "unit": "env steps/s", "envs": n, "launches": launches,
           "timed_region_s": wall, "launch_ms": unit_med, "ms_per_batched_transition": unit_med / fuse, "workload": workload,
           "roofline": {"bound": "hbm", "kernel": "k_rollout_encode", "achieved": unit_bytes / (unit_med * 1e-3) / 1e9,
                        "peak": HBM_PEAK_GBS, "unit": "GB/s",
                        "frac": unit_bytes / (unit_med * 1e-3) / 1e9 / HBM_PEAK_GBS, "traffic": None,
                        "traffic_source": {"how": "not collected", "why": "WRITE_SIZE wraps on multi-GB launches"},
                        "bytes_per_launch": unit_bytes, "launch_ms": unit_med,
                        "note": "algorithmic bytes of one %d-step launch (state in + out, 17 B outputs and 2*W*H*26 observation "
                                "bytes per env-step) / its median duration from HIP events" % fuse},
           "parity_check": pc}
    if extras:
        # the same step with caller-supplied actions, one call per step (oc_step_encode: what a policy in the loop pays)
        acts = torch.randint(0, 6, (64, n, 2), dtype=torch.uint8, device=dev)
        ob1 = obs[0]
        for i in range(20):
            env.step_encode(acts[i % 64], torch.uint8, out=ob1)
        ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        ev0.record()
        for i in range(300):
            env.step_encode(acts[i % 64], torch.uint8, out=ob1)
        ev1.record()
        torch.cuda.synchronize(dev)
        us = ev0.elapsed_time(ev1) / 300 * 1e3
        out["caller_actions_one_step"] = {"us_per_step": us, "value": n / us * 1e6, "unit": "env steps/s (one GPU)",
                                          "note": "oc_step_encode: caller-supplied actions resident in HBM, one C call per batched step"}
        # the f32 variant of the observation (what the reference's RLlib wrapper casts to): 10 steps per launch
        del obs
        K32 = 10
        obs32 = torch.empty((K32, n, 2, env.width, env.height, 26), dtype=torch.float32, device=dev)
        for _ in range(2):
            env.rollout_encode(K32, obs32, rew[:K32], fl[:K32], dtype=torch.float32)
        ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        ev0.record()
        for _ in range(20):
            env.rollout_encode(K32, obs32, rew[:K32], fl[:K32], dtype=torch.float32)
        ev1.record()
        torch.cuda.synchronize(dev)
        us = ev0.elapsed_time(ev1) / (20 * K32) * 1e3
        b32 = n * 2 * env.width * env.height * 26 * 4
        out["f32_observations"] = {"us_per_step": us, "value": n / us * 1e6, "unit": "env steps/s (one GPU)",
                                   "achieved_GBs": b32 / us / 1e3, "frac": b32 / us / 1e3 / HBM_PEAK_GBS,
                                   "note": "oc_rollout_encode with f32 observations, %d steps per launch into a [steps][envs] buffer" % K32}
        del obs32
    return out

def run_encode_config(args, torch, VecOvercookedEnv, sharding, dev, rank, world):
    """--config 3 = BASELINE configs[2]: K bench steps of `--launches-per-step` oc_rollout_encode launches of ENC_FUSE
    transitions + observations each; same timing protocol as the headline, not the headline line."""
    lps = args.launches_per_step or ENC_LAUNCHES_PER_STEP
    m = encode_measure(torch, VecOvercookedEnv, sharding, dev, rank, world, args.envs, launches=args.steps * lps,
                       warm_launches=args.warmup * lps, parity=not args.no_parity_check, extras=True)
    out = {"metric": "env steps/sec (whole node)", "value": m["value"], "unit": "env steps/s",
           "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": m["timed_region_s"] * 1e3 / args.steps,
           "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "u8", "data": "synthetic",
           "timed_region_s": m["timed_region_s"], "timed_launches": m["launches"],
           "ms_per_batched_transition": m["timed_region_s"] * 1e3 / (m["launches"] * ENC_FUSE),
           "ms_per_batched_transition_median": m["launch_ms"] / ENC_FUSE,
           "config": {"workload": m["workload"], "baseline_config": args.config, "envs_per_gpu": args.envs,
                      "fused_transitions_per_launch": ENC_FUSE, "launches_per_step": lps,
                      "step_definition": "one bench step = %d oc_rollout_encode launches of %d transitions + observations" % (lps, ENC_FUSE)},
           "roofline": m["roofline"], "parity_check": m["parity_check"],
           "f32_observations": m.get("f32_observations"), "caller_actions_one_step": m.get("caller_actions_one_step")}
    if rank == 0:
        out["summary"] = summarize(out)  # (last: the driver keeps the line's tail)
        emit(out)
    sharding.barrier()

def bench_step_api(env, dev, torch, iters=2000):
    """The one-launch-per-step API (actions supplied by the caller, resident in HBM): oc_step per batched step."""
    n = env.n_envs
    acts = torch.randint(0, 6, (16, n, 2), dtype=torch.uint8, device=dev)
    for i in range(50):
        env.step(acts[i % 16])
    torch.cuda.synchronize(dev)
    ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    t0 = time.perf_counter()
    ev0.record()
    for i in range(iters):
        env.step(acts[i % 16])
    ev1.record()
    torch.cuda.synchronize(dev)
    wall = time.perf_counter() - t0
    ms = ev0.elapsed_time(ev1) / iters
    b = n * (2 * S_CRAMPED + 2 + OUT_BYTES)
    # oc_step_many: the same K transitions with caller-supplied actions in ONE launch
    K = 500
    acts_k = torch.randint(0, 6, (K, n, 2), dtype=torch.uint8, device=dev)
    rew_k = torch.zeros((K, n, 4), dtype=torch.float32, device=dev)
    fl_k = torch.zeros((K, n), dtype=torch.uint8, device=dev)
    env.step_many(acts_k[:50], rew_k[:50], fl_k[:50])
    torch.cuda.synchronize(dev)
    evm0, evm1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    tm0 = time.perf_counter()
    evm0.record()
    env.step_many(acts_k, rew_k, fl_k)
    evm1.record()
    torch.cuda.synchronize(dev)
    wall_many = time.perf_counter() - tm0
    ms_many = evm0.elapsed_time(evm1) / K
    # the same one-launch-per-step kernels replayed from a HIP graph: 16 captured steps (each reads its own action row, as
    # a captured policy -> step chain would), so the host pays one graph launch per 16 steps instead of 16 kernel launches
    graph_leg = None
    try:
        side = torch.cuda.Stream(device=dev)
        side.wait_stream(torch.cuda.current_stream(dev))
        with torch.cuda.stream(side):
            env.step(acts[0])
        torch.cuda.current_stream(dev).wait_stream(side)
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g):
            for i in range(16):
                env.step(acts[i])
        for _ in range(5):
            g.replay()
        torch.cuda.synchronize(dev)
        reps = max(1, iters // 16)
        tg0 = time.perf_counter()
        for _ in range(reps):
            g.replay()
        torch.cuda.synchronize(dev)
        wall_g = time.perf_counter() - tg0
        graph_leg = {"value": n * 16 * reps / wall_g, "us_per_step": wall_g / (16 * reps) * 1e6,
                     "note": "16 oc_step launches captured in one HIP graph (torch.cuda.graph), replayed: wall clock incl. the replay calls"}
    except Exception as exc:  # (graph capture unavailable: report why, keep the eager numbers)
        graph_leg = {"value": None, "error": repr(exc)[:200]}
    # the resident batched step (oc_step_server_*): the same K steps as DEPENDENT round trips — the client kernel posts step
    # k + 1 only after it has step k's rewards and flags — without a launch per step; checked against oc_step_many from the same states
    resident = None
    try:
        st0, ep0 = env.state.clone(), env.ep_returns.clone()
        rew_r, fl_r = torch.zeros_like(rew_k), torch.zeros_like(fl_k)
        with env.step_server(idle_ms=10.0, life_s=60.0) as sv:
            sv.play(acts_k, rew_r, fl_r)
            us = [sv.last_play_ms / K * 1e3]
            for _ in range(4):
                sv.play(acts_k, rew_k, fl_k)
                us.append(sv.last_play_ms / K * 1e3)
        st1 = env.state.clone()
        env.state.copy_(st0)
        env.ep_returns.copy_(ep0)
        env.step_many(acts_k, rew_k, fl_k)
        mism = int((rew_r != rew_k).any(dim=-1).sum().item() + (fl_r != fl_k).sum().item())
        env.state.copy_(st1)
        best = min(us)
        resident = {"value": n / (best * 1e-6), "us_per_batched_step": best, "us_per_batched_step_each_play": [round(u, 3) for u in us],
                    "achieved_GBs": b / best / 1e3, "frac": b / best / 1e3 / HBM_PEAK_GBS,
                    "parity_check": {"compared_with": "oc_step_many from the same states", "steps": K, "mismatches": mism},
                    "note": "k_step_server + k_step_client: %d dependent steps per client launch, per-env tagged mailboxes in HBM (8 B request, 32 B "
                            "response), no launch / barrier / fence per step; frac on oc_step's 67 B per env-step (the states stay on chip)" % K}
    except Exception as exc:
        resident = {"value": None, "error": repr(exc)[:300]}
    many = {"value": n * K / wall_many, "launch_ms": ms_many, "achieved_GBs": b / (ms_many * 1e-3) / 1e9,
            "frac": b / (ms_many * 1e-3) / 1e9 / HBM_PEAK_GBS, "note": "oc_step_many: K transitions with caller-supplied actions in one launch (envs stay on chip)"}
    return {"value": n * iters / wall, "step_many": many, "resident": resident, "graph_replay": graph_leg, "unit": "env steps/s", "launch_ms": ms, "bytes_per_launch": b,
            "achieved_GBs": b / (ms * 1e-3) / 1e9, "frac": b / (ms * 1e-3) / 1e9 / HBM_PEAK_GBS,
            "note": "oc_step, one launch per batched step incl. Python/ctypes launch overhead; SURVEY 8d: 67 B/env-step"}

def bench_single_env_api(dev, torch, episodes=3):
    """The drop-in single-env surface existing agents hit: OvercookedEnv.step -> OvercookedGridworld.get_state_transition
    (one env per call: pack -> mailbox of the resident kernel -> unpack), same protocol as the reference's CPU measurement
    (cramped_room, horizon 400, random joint actions, info_level 0).  Reported next to the reference's 16.4 k steps/s."""
    import numpy as np

    from overcooked_ai_amd.actions import Action
    from overcooked_ai_amd.env import OvercookedEnv
    from overcooked_ai_amd.mdp import OvercookedGridworld

    mdp = OvercookedGridworld.from_layout_name("cramped_room", device=str(dev))
    env = OvercookedEnv.from_mdp(mdp, horizon=HORIZON, info_level=0)
    rng = np.random.RandomState(0)

    def episode():
        env.reset(regen_mdp=False)
        acts = rng.randint(0, 6, (HORIZON, 2))
        done, k = False, 0
        while not done:
            _, _, done, _ = env.step((Action.INDEX_TO_ACTION[acts[k, 0]], Action.INDEX_TO_ACTION[acts[k, 1]]))
            k += 1
        return k

    episode()  # warm-up
    t0 = time.perf_counter()
    steps = sum(episode() for _ in range(episodes))
    dt = time.perf_counter() - t0
    return {"value": steps / dt, "unit": "env steps/s", "us_per_step": dt / steps * 1e6, "episodes": episodes,
            "reference_python": _reference_python_stored()["value"],
            "note": "OvercookedEnv.step through the single-env drop-in API: state and action written into the pinned mailbox of "
                    "a resident kernel (oc_mailbox_*: no launch per call; ~4 us per transition since round 5 — request granules polled by 8 lanes, the response as one 8-lane store —, the rest is Python: the state comes back as a lazy view of the packed bytes); "
                    "latency-bound by construction - batch with VecOvercookedEnv for throughput"}

def bench_training_env(dev, torch, iters=300):
    """The RLlib-shaped training environment (VecOvercookedMultiAgent.step = oc_multi_agent_step: step, phi(s'),
    shaped rewards, restart of finished envs, observation; use_phi, caller-supplied actions) on 65 536 cramped_room envs."""
    from overcooked_ai_amd.multi_agent import VecOvercookedMultiAgent

    n = N_ENVS_PER_GPU
    out = {}
    for name, dt in (("obs_u8", torch.uint8), ("obs_f32", torch.float32)):
        env = VecOvercookedMultiAgent("cramped_room", n, horizon=HORIZON, reward_shaping_factor=1.0, use_phi=True,
                                      obs_dtype=dt, device=dev)
        acts = torch.randint(0, 6, (16, n, 2), dtype=torch.uint8, device=dev)
        for i in range(20):
            env.step(acts[i % 16])
        torch.cuda.synchronize(dev)
        t0 = time.perf_counter()
        for i in range(iters):
            env.step(acts[i % 16])
        torch.cuda.synchronize(dev)
        wall = time.perf_counter() - t0
        out[name] = {"value": n * iters / wall, "unit": "env steps/s", "us_per_batched_step": wall / iters * 1e6}
        if name == "obs_u8":  # the same chain (k_train_step1 -> k_encode) replayed from a HIP graph of 16 captured steps
            try:
                side = torch.cuda.Stream(device=dev)
                side.wait_stream(torch.cuda.current_stream(dev))
                with torch.cuda.stream(side):
                    env.step(acts[0])
                torch.cuda.current_stream(dev).wait_stream(side)
                g = torch.cuda.CUDAGraph()
                with torch.cuda.graph(g):
                    for i in range(16):
                        env.step(acts[i])
                for _ in range(3):
                    g.replay()
                torch.cuda.synchronize(dev)
                reps = max(1, iters // 16)
                t0 = time.perf_counter()
                for _ in range(reps):
                    g.replay()
                torch.cuda.synchronize(dev)
                wall_g = time.perf_counter() - t0
                out[name]["graph_replay_us_per_batched_step"] = wall_g / (16 * reps) * 1e6
            except Exception as exc:
                out[name]["graph_replay_error"] = repr(exc)[:200]
    for name, dt in (("obs_u8", torch.uint8),):  # roofline of the u8 leg: the bytes one call must move / its wall time
        nbytes = n * (2 * 24 + 2 + 17 + 16 + 1 + 16 + 2 * 5 * 4 * 26)  # state in+out, actions, outputs, shaped, done, phi, observation
        us = out[name]["us_per_batched_step"]
        out[name].update({"bytes_per_step": nbytes, "achieved_GBs": nbytes / us / 1e3, "frac": nbytes / us / 1e3 / HBM_PEAK_GBS})
    out["note"] = ("per batched step: one oc_multi_agent_step call = ONE kernel since round 5 (k_train_step_obs: transition on the wire "
                   "format + phi + shaped rewards + restart + the lossless observation; rounds 2-4: k_train_step1 then k_encode); "
                   "wall clock of back-to-back calls from Python; graph_replay: 16 such calls captured in one HIP graph")
    return out

def bench_encode(dev, torch, VecOvercookedEnv, iters=200):
    """BASELINE configs[2] kernel: lossless_state_encoding of 65 536 asymmetric_advantages envs."""
    n = N_ENVS_PER_GPU
    env = VecOvercookedEnv("asymmetric_advantages", n, horizon=HORIZON, device=dev, auto_reset=True, seed=1)
    env.rollout_random(150)
    res = {}
    for name, dt, elem in (("u8", torch.uint8, 1), ("f32", torch.float32, 4)):
        obs = torch.empty((n, 2, env.width, env.height, 26), dtype=dt, device=dev)
        for _ in range(5):
            env.encode_lossless(dt, out=obs)
        torch.cuda.synchronize(dev)
        ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        ev0.record()
        for _ in range(iters):
            env.encode_lossless(dt, out=obs)
        ev1.record()
        torch.cuda.synchronize(dev)
        ms = ev0.elapsed_time(ev1) / iters
        b = n * (S_ASYM + 2 * env.width * env.height * 26 * elem)
        res[name] = {"launch_ms": ms, "bytes_per_launch": b, "achieved_GBs": b / (ms * 1e-3) / 1e9,
                     "frac": b / (ms * 1e-3) / 1e9 / HBM_PEAK_GBS, "env_encodes_per_s": n / (ms * 1e-3)}
    res["note"] = "k_encode on asymmetric_advantages x 65536; SURVEY 8d: 2384 B (u8) / 9404 B (f32) per env"
    # featurize_state (mdp.py:2579): 2 x 96 float32 per env
    feat = torch.empty((n, 2, 96), dtype=torch.float32, device=dev)
    for _ in range(5):
        env.featurize(out=feat)
    torch.cuda.synchronize(dev)
    ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    ev0.record()
    for _ in range(iters):
        env.featurize(out=feat)
    ev1.record()
    torch.cuda.synchronize(dev)
    ms = ev0.elapsed_time(ev1) / iters
    fb = n * (S_ASYM + 2 * 96 * 4)
    res["featurize_state"] = {"launch_ms": ms, "bytes_per_launch": fb, "achieved_GBs": fb / (ms * 1e-3) / 1e9,
                              "frac": fb / (ms * 1e-3) / 1e9 / HBM_PEAK_GBS, "env_featurizations_per_s": n / (ms * 1e-3)}
    # potential_function (mdp.py:2920): one float64 per env, on mid-episode states
    env.rollout_random(120)
    phi = torch.empty((n,), dtype=torch.float64, device=dev)
    for _ in range(5):
        env.potential(out=phi)
    torch.cuda.synchronize(dev)
    ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    ev0.record()
    for _ in range(iters):
        env.potential(out=phi)
    ev1.record()
    torch.cuda.synchronize(dev)
    ms = ev0.elapsed_time(ev1) / iters
    pb = n * (S_ASYM + 8)
    res["potential_function"] = {"launch_ms": ms, "bytes_per_launch": pb, "achieved_GBs": pb / (ms * 1e-3) / 1e9,
                                 "frac": pb / (ms * 1e-3) / 1e9 / HBM_PEAK_GBS, "env_potentials_per_s": n / (ms * 1e-3)}
    return res
