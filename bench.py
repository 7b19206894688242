#!/usr/bin/env python
"""bench.py — env steps/s of the MI355X Overcooked hot path on BASELINE.json's metric.

Workload (BASELINE.json configs[1] / SURVEY.md §8d-2): 65 536 parallel cramped_room envs PER GPU, uniform random
policy drawn in-kernel with Philox, horizon 400 with auto-reset to the standard start state, outputs written
every step (4 x f32 rewards + 1 flag byte per env-step; the flag bytes as flags[step / 8][env][step % 8], the C-ABI's
OC_OPT_FLAGS_TILED8 layout, wherever the kernel that serves the batch writes it — `config.flags_layout` says which, and
`--flags-layout step` asks for flags[step][env]).  A "step" is one batched transition of all envs of a
GPU.  The timed region launches oc_rollout_random with --fuse transitions per launch; `value` is whole-job env-steps/s
over all ranks (weak scaling: every rank owns 65 536 envs, disjoint Philox streams via env_offset).

What `--steps K` counts (round 4): one bench STEP is one pass of the hot path over the batch long enough for a clock
outside this process to see it: LAUNCHES_PER_STEP (400) back-to-back 4 000-transition launches = 1.6 M batched
transitions of all 65 536 envs of a GPU = 4 000 episodes per env (about a third of a second).  Exactly K such steps are
timed after W warm-up steps; `ms_per_step` is per bench step, `ms_per_batched_transition` per batched env transition
(the figure earlier rounds called ms_per_step), and `ms_per_step_each` lists the K steps one by one.

    python bench.py [--gpus N] [--steps K] [--warmup W]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus N --steps K --warmup W

Prints ONE JSON line on rank 0 (see DESIGN.md §6 for every field).  After the timed region every rank replays one launch
of the timed shape from reset and compares it with the C oracle (`parity_check`); at N = 1 rank 0 also collects the
kernel's HBM traffic with two rocprofv3 --pmc child passes of the same launch shape (`roofline.traffic`, provenance in
`roofline.traffic_source`) and times the oracle on the host cores (`cpu_baseline`).  Fields replayed from files under
profiles/ instead of being measured in the run say so in a `source` / `how` key next to them.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

# (round 6: the parts live in benchlib/ — common: constants, command line, timed region, workloads, parity; cpu_baseline; traffic:
#  PMC passes, SQ counters, the store-only ceiling; legs: the side legs of the default line — behind the same command line)
from benchlib.common import *  # noqa: F401,F403 (constants and helpers)
from benchlib.common import _StubEnv, _Timer  # noqa: F401
from benchlib.cpu_baseline import REFERENCE_PYTHON, _reference_python_stored, cpu_baseline, reference_python  # noqa: F401
from benchlib.legs import (bench_encode, bench_single_env_api, bench_step_api, bench_training_env, encode_measure, general_legs,  # noqa: F401
                           run_encode_config, side_legs)
from benchlib.traffic import issue_counters, measure_store_only, measure_traffic, pmc_child, traffic_from_file  # noqa: F401


def main():
    args = parse()
    if args.single_process:
        return run_single_process(args)
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        raise SystemExit(spawn_ranks(args.gpus))
    quiet_stdout_unless_rank0()
    import torch

    from overcooked_ai_amd import sharding

    if not args.stub:
        from overcooked_ai_amd import build
        from overcooked_ai_amd.vec_env import VecOvercookedEnv

        build.build_extension()  # no-op when liboc_amd.so is up to date (a fresh checkout has none: it is git-ignored);
        # every rank may do it: the build writes a per-process temp file and renames it atomically
        if not torch.cuda.is_available():
            raise SystemExit("bench.py needs an MI355X; there is no CPU path")
        if args.gpus > torch.cuda.device_count() and not args.share_device:
            raise SystemExit("--gpus %d but only %d GPU(s) visible" % (args.gpus, torch.cuda.device_count()))
    else:
        VecOvercookedEnv = None
    if args.pmc_child:
        dev = torch.device("cuda", 0)
        torch.cuda.set_device(dev)
        return pmc_child(args, torch, VecOvercookedEnv, dev)

    # RCCL prints its version banner to C stdio's stdout when the first communicator comes up: bring the process group up
    # (and run one collective) with fd 1 pointed at stderr, so that stdout carries the ONE JSON line and nothing else
    sys.stdout.flush()
    saved_stdout = os.dup(1)
    os.dup2(2, 1)
    try:
        if args.share_device and not args.stub:  # (before the process group: nccl would set_device(local_rank))
            os.environ["LOCAL_RANK"] = str(int(os.environ.get("LOCAL_RANK", "0")) % torch.cuda.device_count())
        rank, local_rank, world = sharding.init_process_group("gloo" if args.stub or args.backend == "gloo" else None)
        if sharding._live():
            if not args.stub:
                torch.cuda.set_device(local_rank)
            warm = torch.zeros((1,), device="cpu" if args.stub else torch.device("cuda", local_rank))
            sharding.allreduce_metrics(warm)
            if not args.stub:
                torch.cuda.synchronize()
        import ctypes

        try:
            ctypes.CDLL(None).fflush(None)
        except OSError:
            pass
    finally:
        os.dup2(saved_stdout, 1)
        os.close(saved_stdout)
    if world != args.gpus:
        raise SystemExit("--gpus %d but WORLD_SIZE=%d" % (args.gpus, world))
    if sharding._live():
        import torch.distributed as dist

        want = "gloo" if args.stub else args.backend
        if dist.get_world_size() != args.gpus or dist.get_backend() != want:
            raise SystemExit("process group has %d ranks over %s; expected %d over %s"
                             % (dist.get_world_size(), dist.get_backend(), args.gpus, want))
    dev = torch.device("cpu") if args.stub else torch.device("cuda", local_rank)
    numa = None
    if not args.stub:
        torch.cuda.set_device(dev)
        if world > 1:
            numa = pin_to_gpu_numa(torch, local_rank)
    if args.config == 3:
        return run_encode_config(args, torch, VecOvercookedEnv, sharding, dev, rank, world)
    return run_rollout_config(args, torch, VecOvercookedEnv, sharding, dev, rank, world, numa)

def run_single_process(args):
    """`--gpus N --single-process`: the product-level sharded env (overcooked_ai_amd.sharded_env) instead of N ranks —
    one process, one VecOvercookedEnv + HIP stream per GPU, every launch fanned out to all shards without a host
    synchronisation in between; same step definition, same JSON keys; the aggregate metrics are summed on the host."""
    import argparse as _ap

    import numpy as np
    import torch

    from overcooked_ai_amd import build
    from overcooked_ai_amd.sharded_env import ShardedVecOvercookedEnv

    build.build_extension()
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X; there is no CPU path")
    N = args.gpus
    devices = ["cuda:%d" % (i % torch.cuda.device_count()) for i in range(N)]  # (more shards than GPUs: several per device)
    n, fuse = args.envs, max(1, args.fuse)
    lps = args.launches_per_step or LAUNCHES_PER_STEP
    # the global batch = the N per-rank workloads side by side (global env e -> layout e % K)
    wls = [make_workload(_ap.Namespace(config=args.config if args.config != 3 else 2, envs=n, layout=args.layout,
                                       terrains=args.terrains), r) for r in range(N)]
    lid = None if wls[0]["lid"] is None else np.concatenate([w["lid"] for w in wls])
    def make():
        return ShardedVecOvercookedEnv(wls[0]["table"], N * n, devices=devices, layout_id=lid, horizon=HORIZON, auto_reset=True, seed=0)

    # the flags layout, as the per-rank path decides it: tiled by 8 steps where the kernels that serve the shards write it
    tiled8 = False
    if args.flags_layout == "tiled8" and fuse % 8 == 0:
        probe = make()
        try:
            pr, pf = probe.alloc_outputs(8, flags_tiled8=True)
            probe.rollout_random(8, pr, pf, flags_tiled8=True)
            probe.synchronize()
            tiled8 = True
        except Exception:
            tiled8 = False
        del probe
    env = make()
    rews, fls = env.alloc_outputs(fuse, flags_tiled8=tiled8)

    def launch():
        env.rollout_random(fuse, rews, fls, flags_tiled8=tiled8)

    for _ in range(max(1, args.warmup * lps)):
        launch()
    env.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps * lps):
        launch()
    env.synchronize()
    wall = time.perf_counter() - t0
    transitions = args.steps * lps * fuse
    agg = env.aggregate(rews, fls)
    parity = None
    if not args.no_parity_check:
        from oracle import oracle as O

        psteps = min(fuse, args.parity_steps or 400)
        psteps -= psteps % 8 if tiled8 else 0
        chk = make()
        fl_chk = [f[:psteps // 8] if tiled8 else f[:psteps] for f in fls]
        chk.rollout_random(psteps, [r[:psteps] for r in rews], fl_chk, flags_tiled8=tiled8)
        if tiled8:  # [steps / 8][envs][8] -> [steps][envs] for the comparison
            fl_chk = [f.permute(0, 2, 1).reshape(psteps, f.shape[1]) for f in fl_chk]
        O.set_threads(usable_cores())
        orc = O.Oracle([O.mdp_from_layout_dict(sp.to_layout_dict()) for sp in wls[0]["specs"]])
        st = orc.reset(orc.new_state(N * n), layout_id=lid)
        ep = np.zeros((N * n, 4), np.float32)
        rew_o, fl_o = orc.rollout_random(st, psteps, horizon=HORIZON, options=1, seed=0, layout_id=lid, ep_returns=ep)
        O.set_threads(1)
        bad = int(((chk.gather([r[:psteps] for r in rews], 1) != rew_o).any(axis=2) | (chk.gather(fl_chk, 1) != fl_o)).sum())
        bad += int((chk.get_packed_state() != st).any(axis=(0, 2)).sum()) + int((chk.ep_returns() != ep).any(axis=1).sum())
        parity = {"envs": N * n, "steps": psteps, "mismatches": bad,
                  "what": "one %d-step launch per shard from reset: rewards, flags, final states and episode returns of all "
                          "%d envs against oracle/overcooked_oracle.c" % (psteps, N * n)}
    emit({"metric": "env steps/sec (whole node), 65k parallel cramped_room envs" if args.config == 2 else "env steps/sec (whole node)",
          "value": float(N) * n * transitions / wall, "unit": "env steps/s", "n_gpus": len(set(devices)), "n_shards": N,
          "steps": args.steps, "warmup": args.warmup,
          "ms_per_step": wall * 1e3 / args.steps, "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "u8",
          "data": "synthetic", "timed_region_s": wall, "timed_transitions_per_env": transitions,
          "ms_per_batched_transition": wall * 1e3 / transitions,
          "config": {"workload": wls[0]["workload"], "baseline_config": args.config, "envs_per_gpu": n,
                     "flags_layout": "[steps/8][envs][8] (OC_OPT_FLAGS_TILED8)" if tiled8 else "[steps][envs]",
                     "fused_transitions_per_launch": fuse, "launches_per_step": lps,
                     "parallelism": "single process, ShardedVecOvercookedEnv, %d shards on %s" % (N, sorted(set(devices)))},
          "roofline": None, "parity_check": parity, "aggregate": dict(agg, reduced_over="host sum over shards")})

def run_rollout_config(args, torch, VecOvercookedEnv, sharding, dev, rank, world, numa):
    """--config 2 (the headline: BASELINE configs[1]), 4 and 5 (configs[3] / [4] on one GPU's shard): K bench steps of
    `--launches-per-step` oc_rollout_random launches of `--fuse` transitions each (module docstring)."""
    n = args.envs
    wl = make_workload(args, rank)
    make_env = rollout_workload_env(args, wl, n, rank, dev, VecOvercookedEnv)
    env = make_env()
    fuse = max(1, args.fuse)
    lps = args.launches_per_step or LAUNCHES_PER_STEP
    rew = torch.zeros((fuse, n, 4), dtype=torch.float32, device=dev)
    fl = torch.zeros((fuse, n), dtype=torch.uint8, device=dev)

    tiled8 = flags_tiled8_ok(args, env, fuse, rew, fl)
    fl_t = fl.view(fuse // 8, n, 8) if tiled8 else None

    def launch():
        if tiled8:
            env.rollout_random(fuse, rew, fl_t, flags_tiled8=True)
        else:
            env.rollout_random(fuse, rew, fl)

    for _ in range(max(1, args.warmup * lps)):  # W bench steps, untimed (at least one launch)
        launch()
    launches = args.steps * lps
    wall, per_launch_ms = timed_launches(torch, dev, sharding, launch, launches)
    step_each = [sum(per_launch_ms[i * lps:(i + 1) * lps]) for i in range(args.steps)]
    per_launch = sorted(per_launch_ms)
    dev_ms = sum(per_launch)
    launch_med, launch_min = per_launch[len(per_launch) // 2], per_launch[0]
    tmax = torch.tensor([wall], dtype=torch.float64, device=dev)
    sharding.allreduce_max(tmax)
    wall_max = float(tmax.item())
    transitions = launches * fuse  # batched transitions of every env in the timed region
    per_rank = torch.zeros((world,), dtype=torch.float64, device=dev)
    per_rank[rank] = wall * 1e3 / args.steps
    sharding.allreduce_metrics(per_rank)  # disjoint slots: the sum is a gather
    # aggregate-return metric: the only collective, outside the hot path (RCCL all-reduce of 3 scalars)
    metrics = torch.stack([rew[..., 0:2].sum().to(torch.float64), rew[..., 2:4].sum().to(torch.float64),
                           ((fl_t[-1, :, 7] if tiled8 else fl[-1]) & 1).sum().to(torch.float64)])
    sharding.allreduce_metrics(metrics)

    value = float(world) * n * transitions / wall_max

    # roofline of the dominant kernel (k_rollout4): algorithmic HBM bytes per launch / median launch duration
    state_bytes = wl["sbytes"]
    bytes_per_launch = n * (2 * state_bytes + OUT_BYTES * fuse)
    achieved = bytes_per_launch / (launch_med * 1e-3) / 1e9
    # (whole 256-env workgroups and 8-step blocks run k_rollout5's mover / interact workgroups, other shapes k_rollout4)
    kernel = "k_rollout" if args.predicate_interact else "k_rollout_pair" if args.lane_pair else "k_rollout5|k_rollout4"
    traffic, traffic_src = None, {"how": "not collected", "why": "only rank 0 of a 1-GPU run collects PMC traffic"}
    if rank == 0 and world == 1 and not args.stub:
        if not args.no_traffic:
            traffic, traffic_src = measure_traffic(args, kernel, tiled8)
        if traffic is None:
            why = traffic_src.get("why")
            traffic, traffic_src = traffic_from_file(kernel, n, fuse, args.layout if args.config == 2 else "", bytes_per_launch)
            if traffic is None and why:
                traffic_src["same_run_attempt"] = why
    issue = issue_counters(kernel, n, args.layout) if args.config == 2 and not args.stub else None

    # parity of the timed launch shape, per rank, against the C oracle (never inside the timed region)
    parity = None
    if not args.no_parity_check:
        psteps = args.parity_steps or (fuse if world == 1 else min(fuse, 1200))
        psteps = min(psteps, fuse)
        mine = parity_check(torch, wl, make_env, n, rank, psteps, rew, fl, max(1, usable_cores() // max(1, world)), tiled8=tiled8)
        pr = torch.zeros((world, 2), dtype=torch.float64, device=dev)
        pr[rank, 0], pr[rank, 1] = mine["mismatches"], mine["seconds"]
        sharding.allreduce_metrics(pr)
        parity = dict(mine, envs=n * world, mismatches=int(pr[:, 0].sum().item()),
                      mismatches_by_rank=[int(x) for x in pr[:, 0].tolist()],
                      seconds_by_rank=[float(x) for x in pr[:, 1].tolist()], envs_per_rank=n)
        if args.stub:
            parity["stub"] = "oracle compared with itself: plumbing only"

    # what the output format itself admits on this device at this batch size: the rollout's output stores and nothing else
    # (oc_output_stores_only, include/oc_amd.h), same arrays, after everything that reads them
    store_only = None
    if rank == 0 and not args.stub:
        try:
            store_only = measure_store_only(torch, dev, env, n, fuse, rew, fl, value / world, launch_med, tiled8=tiled8)
        except Exception as exc:  # an aid, never a reason to lose the line
            store_only = {"error": repr(exc)}

    # the same launches with flags[step][env] (what a caller that does not ask for the tiled layout gets), next to the headline
    step_layout = None
    if rank == 0 and world == 1 and not args.stub and tiled8 and not args.no_extras:
        try:
            for _ in range(3):
                env.rollout_random(fuse, rew, fl)
            tm = _Timer(torch, dev, reserve=41)
            tm.sync()
            for _ in range(40):
                tm.mark()
                env.rollout_random(fuse, rew, fl)
            tm.mark()
            tm.sync()
            ms = sorted(tm.launch_ms())
            med = ms[len(ms) // 2]
            step_layout = {"flags_layout": "[steps][envs]", "launch_ms": med, "env_steps_per_s": n * fuse / (med * 1e-3),
                           "frac": bytes_per_launch / (med * 1e-3) / 1e9 / HBM_PEAK_GBS,
                           "note": "40 launches of the headline shape with the flags as [steps][envs] rows (no OC_OPT_FLAGS_TILED8), median, HIP events"}
        except Exception as exc:
            step_layout = {"error": repr(exc)}

    out = {
        "metric": "env steps/sec (whole node), 65k parallel cramped_room envs" if args.config == 2 else "env steps/sec (whole node)",
        "value": value, "unit": "env steps/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": wall_max * 1e3 / args.steps, "higher_is_better": True, "scaling": "weak",
        "vs_baseline": None, "dtype": "u8", "data": "stub" if args.stub else "synthetic",
        "timed_region_s": wall_max, "timed_launches": launches, "timed_transitions_per_env": transitions,
        "timed_env_steps": float(world) * n * transitions,
        "ms_per_batched_transition": wall_max * 1e3 / transitions,
        "ms_per_batched_transition_median": launch_med / fuse, "ms_per_batched_transition_min": launch_min / fuse,
        "ms_per_step_each": step_each, "ms_per_step_by_rank": [float(x) for x in per_rank.tolist()],
        "warmup_launches_run": max(1, args.warmup * lps),
        "config": {"workload": wl["workload"], "baseline_config": args.config,
                   "flags_layout": "[steps/8][envs][8] (OC_OPT_FLAGS_TILED8: the flag bytes of 8 steps of an env side by side; "
                                   "17 B per env-step as before; parity_check untiles them)" if tiled8 else "[steps][envs]",
                   "envs_per_gpu": n, "fused_transitions_per_launch": fuse, "launches_per_step": lps, "launches": launches,
                   "parallelism": "env-shard x%d" % world + (
                       " — %d ranks SHARING %d GPU(s) over %s: a rehearsal of the per-rank path (process group, env_offset, NUMA "
                       "pinning, per-rank parity, metric reductions) with real kernels; NOT a scaling measurement, nothing about xGMI"
                       % (world, torch.cuda.device_count(), args.backend) if getattr(args, "share_device", False) else ""),
                   "backend": ("gloo" if args.stub else args.backend) if world > 1 else None, "numa_node_rank0": numa,
                   "step_definition": "one bench step = %d back-to-back oc_rollout_random launches of %d transitions = %d "
                                      "batched transitions of all %d envs of a GPU (%d episodes per env); exactly --steps of "
                                      "them are timed after --warmup untimed ones; value = n_gpus x envs x transitions / wall "
                                      "clock (max over ranks); ms_per_step_each = the steps one by one (HIP events, rank 0)"
                                      % (lps, fuse, lps * fuse, n, lps * fuse // HORIZON)},
        "roofline": {"bound": "hbm", "kernel": kernel, "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                     "frac": achieved / HBM_PEAK_GBS, "traffic": traffic, "traffic_source": traffic_src,
                     "bytes_per_launch": bytes_per_launch, "launch_ms": launch_med, "launch_ms_min": launch_min,
                     "launch_ms_mean": dev_ms / max(1, launches), "launch_timing": "per-launch HIP events on the launch stream, median",
                     "bytes_model": "n_envs*(2*S + 17*T): S=%d B state in+out once per launch, 17 B outputs per env-step, actions in-kernel" % state_bytes,
                     "survey_8d_per_step_model_GBs": n * (2 * state_bytes + OUT_BYTES) * fuse / (launch_med * 1e-3) / 1e9,
                     "issue_bound": issue, "store_only": store_only, "step_layout": step_layout},
        "parity_check": parity,
        "device_ms_timed_region": dev_ms,
        "aggregate": {"sparse_return_last_launch": float(metrics[0]), "shaped_return_last_launch": float(metrics[1]),
                      "episodes_done_last_step": float(metrics[2]),
                      "reduced_over": ("RCCL all-reduce" if not args.stub and args.backend == "nccl" else "gloo all-reduce") if sharding._live() else "single rank"},
    }

    side = rank == 0 and world == 1 and not args.no_extras and not args.stub and args.config == 2
    if side:
        out["step_api"] = bench_step_api(env, dev, torch)
    del env, rew, fl
    if side:
        out["single_env_api"] = bench_single_env_api(dev, torch)
        out["encode"] = bench_encode(dev, torch, VecOvercookedEnv)
        out["training_env"] = bench_training_env(dev, torch)
        # the other BASELINE configs, each with its own roofline and parity check, in this same line
        out["configs"] = side_legs(args, torch, VecOvercookedEnv, sharding, dev)
        out["general_path"] = general_legs(args, torch, VecOvercookedEnv, sharding, dev)
    if rank == 0 and world == 1 and not args.no_cpu_baseline and not args.stub:  # the CPU leg runs at N = 1 only
        wl_cpu = make_workload(args, rank)
        port = cpu_baseline(wl_cpu, n, args.cpu_seconds)
        ref = reference_python(args)
        if ref.get("same_run") and args.config == 2:
            # north_star: "next to the reference Python OvercookedEnv.step timed on the same box's host cores (core count
            # stated)": the reference itself (oracle/_ref, byte-compiled) is the CPU baseline, the C oracle sits beside it
            out["cpu_baseline"] = {
                "value": ref["all_cores"]["value"], "unit": "env steps/s", "cores": ref["all_cores"]["cores"], "kind": "reference",
                "single_core": ref["value"], "with_lossless_encoding": ref["with_lossless_encoding"],
                "sample": "the reference's own OvercookedEnv.step (overcooked_env.py:244) on cramped_room, horizon 400, random joint "
                          "actions: 50 episodes per process after 1 warm-up (BASELINE.md 3.2), one env per process on every usable "
                          "core (multiprocessing.Pool), and on one core; both layouts, with and without the encoding: %.1f s in "
                          "this run" % ref.get("seconds", 0.0),
                "where": ref["where"], "source": ref["source"], "same_run": True, "same_box": True, "port": port}
            if ref.get("asymmetric_advantages") and "3" in out.get("configs", {}) and "error" not in out["configs"]["3"]:
                out["configs"]["3"]["cpu_baseline"] = ref["asymmetric_advantages"]  # configs[2]: the same workload on the host
        else:
            out["cpu_baseline"] = dict(port, reference_python=ref)
        if "single_env_api" in out:
            out["single_env_api"]["reference_python"] = ref["value"]
    if rank == 0:
        out["summary"] = summarize(out)  # (last: the driver keeps the line's tail)
        emit(out)
    sharding.barrier()

def _finish():
    try:
        import torch.distributed as dist

        if dist.is_available() and dist.is_initialized():
            dist.destroy_process_group()
    except Exception:
        pass


if __name__ == "__main__":
    try:
        main()
    finally:
        _finish()
