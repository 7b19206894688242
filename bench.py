#!/usr/bin/env python
"""bench.py — env steps/s of the MI355X Overcooked hot path on BASELINE.json's metric.

Workload (BASELINE.json configs[1] / SURVEY.md §8d-2): 65 536 parallel cramped_room envs PER GPU, uniform random
policy drawn in-kernel with Philox, horizon 400 with auto-reset to the standard start state, outputs written
every step (4 x f32 rewards + 1 flag byte per env-step).  A "step" is one batched transition of all envs of a
GPU.  The timed region launches oc_rollout_random with --fuse steps per launch; `value` is whole-job env-steps/s
over all ranks (weak scaling: every rank owns 65 536 envs, disjoint Philox streams via env_offset).

    python bench.py [--gpus N] [--steps K] [--warmup W]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus N --steps K --warmup W

Prints ONE JSON line on rank 0 (see DESIGN.md §6 for every field).
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # MI355X HBM3E spec peak, /opt/skills/guides/MI355X_MICROARCH.md
N_ENVS_PER_GPU = 65536
HORIZON = 400
# env steps per oc_rollout_random launch: ten whole episodes.  A launch pays ~16 us before its first step and after its
# last (LUT staging, joint move table build, state load / store, the gap to the next launch): 12 % of a 400-step launch
# (126 us), 1.5 % of a 4 000-step one (1.07 ms) — measured 203.9 / 222.6 / 230.3 / 236.5 / 241.7 / 243.2 G env-steps/s at
# 400 / 800 / 1 200 / 2 000 / 4 000 / 8 000 steps per launch (before the last scheduling changes).  4 000 = 4.5 GB of
# per-step outputs per launch; the PMC byte counters were verified up to 8 000.
DEFAULT_FUSE = 10 * HORIZON

# SURVEY.md §8d algorithmic bytes.  S = minimal state of cramped_room (2 players x 3 B + 14 non-floor cells
# + 1 pot tick + 2 B timestep -> 24 B), outputs 17 B per env-step, actions generated in-kernel (0 B).
S_CRAMPED = 24
OUT_BYTES = 17
S_ASYM = 44


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20000)
    ap.add_argument("--warmup", type=int, default=2000)
    ap.add_argument("--fuse", type=int, default=DEFAULT_FUSE, help="env steps fused per oc_rollout_random launch (default: ten 400-step episodes)")
    ap.add_argument("--envs", type=int, default=N_ENVS_PER_GPU, help="envs per GPU")
    ap.add_argument("--layout", default="cramped_room")
    ap.add_argument("--config", type=int, default=2, choices=[2, 3, 4, 5],
                    help="BASELINE.json config index (1-based): 2 = headline (default); 3 = asymmetric_advantages + "
                         "lossless encoding every step; 4 = 5-layout mix padded to 9x5; 5 = 4096 generated 9x5 terrains")
    ap.add_argument("--lane-per-env", action="store_true", help="force the one-lane-per-env rollout kernel")
    ap.add_argument("--lane-pair", action="store_true", help="force the two-lanes-per-env rollout kernel")
    ap.add_argument("--predicate-interact", action="store_true", help="lane-per-env kernel with the predicate-network interact (v2)")
    ap.add_argument("--rollout-v3", action="store_true", help="the previous table-driven rollout kernel (k_rollout3)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-extras", action="store_true", help="skip the step-API and encode side measurements")
    ap.add_argument("--cpu-seconds", type=float, default=10.0)
    ap.add_argument("--min-seconds", type=float, default=0.25,
                    help="minimum length of the timed region: the --steps-step region is repeated back to back until it lasts this long")
    ap.add_argument("--stub", action="store_true",
                    help="CPU-only plumbing test (gloo, no kernels): exercises rank spawning and the reductions; never a measurement")
    return ap.parse_args()


def run_fused(env, n_steps, fuse, rew, fl):
    """n_steps batched steps as ceil(n_steps / fuse) launches; returns number of launches."""
    launches = 0
    left = n_steps
    while left > 0:
        k = min(fuse, left)
        env.rollout_random(k, rew[:k], fl[:k])
        left -= k
        launches += 1
    return launches


def usable_cores():
    """Host cores this process may actually use: the affinity mask, capped by the cgroup CPU quota (the GPU boxes show
    256 logical CPUs but run the job under a 16-CPU quota; oversubscribing it makes the threaded oracle slower)."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        if quota != "max":
            n = min(n, max(1, -(-int(quota) // int(period))))
    except (OSError, ValueError):
        pass
    return max(1, n)


def cpu_baseline(layout, seconds):
    """The C oracle (a scalar port of the reference's algorithm) on the host: a bounded sample of the same workload
    (same layout, random policy, horizon 400 with auto-reset, outputs written every step), first on one core, then
    with the independent envs spread over all cores (OpenMP).  `value` is the all-cores figure."""
    import numpy as np

    from oracle import oracle as O
    from overcooked_ai_amd.layouts import spec_from_name

    spec = spec_from_name(layout)
    orc = O.Oracle(O.mdp_from_layout_dict(spec.to_layout_dict()))
    n, T = 8192, 100
    st = orc.reset(orc.new_state(n))
    ep = np.zeros((n, 4), np.float32)
    tg = 0

    def timed(budget):
        nonlocal tg
        orc.rollout_random(st, 10, horizon=HORIZON, options=1, seed=0, t0=tg, ep_returns=ep)  # warm
        tg += 10
        t0 = time.perf_counter()
        orc.rollout_random(st, T, horizon=HORIZON, options=1, seed=0, t0=tg, ep_returns=ep)
        tg += T
        probe = time.perf_counter() - t0
        reps = max(1, int(budget / max(probe, 1e-6)))
        t0 = time.perf_counter()
        for _ in range(reps):
            orc.rollout_random(st, T, horizon=HORIZON, options=1, seed=0, t0=tg, ep_returns=ep)
            tg += T
        dt = time.perf_counter() - t0
        return reps * n * T / dt, reps * T, dt

    O.set_threads(1)
    one, steps1, dt1 = timed(seconds * 0.4)
    cores = O.set_threads(usable_cores())
    allc, steps_all, dt_all = (one, steps1, dt1) if cores == 1 else timed(seconds * 0.6)
    O.set_threads(1)
    return {
        "value": allc, "unit": "env steps/s", "cores": cores, "kind": "port", "single_core": one,
        "sample": "%d envs x %d steps of %s (C oracle, %d threads, %.1f s); single core: %d steps in %.1f s"
                  % (n, steps_all, layout, cores, dt_all, steps1, dt1),
    }


# The reference's own rate (north_star: "next to the reference Python OvercookedEnv.step"): the Python reference cannot
# travel to the GPU box (/root/reference does not exist there), so the figure measured in the build container
# (SURVEY §8d-1 protocol: cramped_room, horizon 400, random joint actions, info_level 0) is carried in
# the JSON with its provenance.
REFERENCE_PYTHON = {
    "value": 16400.0, "unit": "env steps/s", "cores": 1,
    "all_cores": {"value": 74000.0, "cores": 8, "note": "one env per process, multiprocessing.Pool(8)"},
    "where": "build container (8 vCPU Xeon 2.1 GHz, CPython 3.10.12, numpy 2.2.6), not the GPU box",
    "what": "reference OvercookedEnv.step (src/overcooked_ai_py/mdp/overcooked_env.py:244), cramped_room, horizon 400, "
            "np.random.RandomState joint actions, >= 50 episodes after 1 warm-up",
    "source": "BASELINE.md §2 / SURVEY.md §8d-1",
}


def lcm(a, b):
    import math

    return a * b // math.gcd(a, b)


def plan_repeats(steps, fuse, est_ms_per_step, min_seconds):
    """The timed region is R back-to-back repetitions of the --steps-step region, launched as whole `fuse`-step
    launches: R is the smallest count that (a) makes steps * R a multiple of `fuse` and (b) lasts >= min_seconds at the
    rate estimated during warm-up."""
    unit = lcm(steps, fuse) // steps  # repetitions per whole number of launches
    need = max(1, int(-(-min_seconds * 1.05e3 // max(est_ms_per_step * steps, 1e-9))))  # 5 % margin over the estimate
    return -(-need // unit) * unit


class _StubEnv:
    """CPU stand-in for VecOvercookedEnv used ONLY by `--stub` (tests of the rank-spawning / reduction plumbing on a
    box without GPUs, gloo backend).  It steps nothing; the JSON it yields says data: "stub"."""

    n_planes = 3

    def __init__(self, n):
        self.n_envs, self.t_global = n, 0

    def rollout_random(self, k, rew=None, fl=None):
        self.t_global += k
        if rew is not None:
            rew[:k].fill_(1.0 / 16)


class _Timer:
    """Device-side timing of each launch: HIP events on the stream the kernels are launched on (torch's current
    stream — VecOvercookedEnv launches there); wall clock on CPU for the stub."""

    def __init__(self, torch, dev, reserve=0):
        self.torch, self.gpu, self.dev, self.ev = torch, dev.type == "cuda", dev, []
        # events are created up front: creating one per launch inside the timed loop costs host time per launch
        self.pool = [torch.cuda.Event(enable_timing=True) for _ in range(reserve)] if self.gpu else []

    def mark(self):
        if self.gpu:
            e = self.pool.pop() if self.pool else self.torch.cuda.Event(enable_timing=True)
            e.record()
            self.ev.append(e)
        else:
            self.ev.append(time.perf_counter())

    def sync(self):
        if self.gpu:
            self.torch.cuda.synchronize(self.dev)

    def launch_ms(self):
        if self.gpu:
            return [a.elapsed_time(b) for a, b in zip(self.ev[:-1], self.ev[1:])]
        return [(b - a) * 1e3 for a, b in zip(self.ev[:-1], self.ev[1:])]


def emit(out):
    """The ONE JSON line of rank 0.  Native libraries (RCCL's version banner) write to C stdio's stdout, which is
    block-buffered when redirected: flush it first so that nothing of theirs lands after — or inside — the line."""
    import ctypes

    try:
        ctypes.CDLL(None).fflush(None)
    except OSError:
        pass
    print(json.dumps(out), flush=True)


def quiet_stdout_unless_rank0():
    """Ranks other than 0 never print the result; send whatever their native libraries write to stdout to stderr."""
    if int(os.environ.get("RANK", "0")) != 0:
        sys.stdout.flush()
        os.dup2(2, 1)


def spawn_ranks(n):
    """`python bench.py --gpus N` without a launcher: start N ranks of this same command (one per GPU) with the
    torch.distributed rendezvous environment on 127.0.0.1; rank 0's JSON line is the output."""
    import socket
    import subprocess

    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    procs = []
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), LOCAL_WORLD_SIZE=str(n),
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), OC_BENCH_SPAWNED="1")
        env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env))
    rc = 0
    try:
        for p in procs:
            rc = p.wait() or rc
    finally:
        for p in procs:
            if p.poll() is None:
                p.kill()
    return rc


def main():
    args = parse()
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
        if args.gpus > torch.cuda.device_count():
            raise SystemExit("--gpus %d but only %d GPU(s) visible" % (args.gpus, torch.cuda.device_count()))
    else:
        VecOvercookedEnv = None

    rank, local_rank, world = sharding.init_process_group("gloo" if args.stub else None)
    if world != args.gpus:
        raise SystemExit("--gpus %d but WORLD_SIZE=%d" % (args.gpus, world))
    dev = torch.device("cpu") if args.stub else torch.device("cuda", local_rank)
    if not args.stub:
        torch.cuda.set_device(dev)

    n = args.envs
    if args.config != 2:
        return run_other_config(args, torch, VecOvercookedEnv, sharding, dev, rank, world)
    if args.stub:
        env = _StubEnv(n)
    else:
        env = VecOvercookedEnv(args.layout, n, horizon=HORIZON, device=dev, auto_reset=True, seed=0,
                               env_offset=rank * n)
        env.lane_per_env = args.lane_per_env
        env.lane_pair = args.lane_pair
        env.predicate_interact = args.predicate_interact
        env.rollout_v3 = args.rollout_v3
    fuse = max(1, args.fuse)  # launch shape: independent of --steps (a 20-step --steps must not shrink the launches)
    rew = torch.zeros((fuse, n, 4), dtype=torch.float32, device=dev)
    fl = torch.zeros((fuse, n), dtype=torch.uint8, device=dev)
    tm = _Timer(torch, dev)

    # warm-up: at least the W steps asked for, rounded up to whole launches of the timed shape (so every launch of the
    # kernel in a profile of this command is the same `fuse`-step launch), then 3 more that calibrate R
    warm_launches = max(1, -(-args.warmup // fuse))
    for _ in range(warm_launches):
        env.rollout_random(fuse, rew, fl)
    cal = _Timer(torch, dev)
    for _ in range(3):
        cal.mark()
        env.rollout_random(fuse, rew, fl)
    cal.mark()
    cal.sync()
    est = torch.tensor([min(cal.launch_ms()[1:]) / fuse], dtype=torch.float64, device=dev)
    sharding.allreduce_max(est)  # every rank must pick the same R
    repeats = plan_repeats(args.steps, fuse, float(est.item()), args.min_seconds)
    total_steps = args.steps * repeats
    launches = total_steps // fuse
    tm = _Timer(torch, dev, reserve=launches + 1)

    tm.sync()
    sharding.barrier()
    tm.sync()
    t0 = time.perf_counter()
    for _ in range(launches):
        tm.mark()
        env.rollout_random(fuse, rew, fl)
    tm.mark()
    tm.sync()
    sharding.barrier()
    tm.sync()
    wall = time.perf_counter() - t0
    per_launch = sorted(tm.launch_ms())
    dev_ms = sum(per_launch)
    launch_med, launch_min = per_launch[len(per_launch) // 2], per_launch[0]
    tmax = torch.tensor([wall], dtype=torch.float64, device=dev)
    sharding.allreduce_max(tmax)
    wall_max = float(tmax.item())
    per_rank = torch.zeros((world,), dtype=torch.float64, device=dev)
    per_rank[rank] = wall * 1e3 / total_steps
    sharding.allreduce_metrics(per_rank)  # disjoint slots: the sum is a gather
    # aggregate-return metric: the only collective, outside the hot path (RCCL all-reduce of 3 scalars)
    metrics = torch.stack([rew[..., 0:2].sum().to(torch.float64), rew[..., 2:4].sum().to(torch.float64),
                           (fl[-1] & 1).sum().to(torch.float64)])
    sharding.allreduce_metrics(metrics)

    value = float(world) * n * total_steps / wall_max

    # roofline of the dominant kernel (k_rollout4): algorithmic HBM bytes per launch / median launch duration
    state_bytes = S_CRAMPED if args.layout == "cramped_room" else 4 * ((env.n_planes * 16) // 4)
    bytes_per_launch = n * (2 * state_bytes + OUT_BYTES * fuse)
    achieved = bytes_per_launch / (launch_med * 1e-3) / 1e9
    kernel = ("k_rollout" if args.predicate_interact else "k_rollout_pair" if args.lane_pair
              else "k_rollout3" if args.rollout_v3 else "k_rollout4")
    traffic = None
    try:  # PMC HBM bytes per launch measured by tools/profile_round.sh on this same launch shape (profiles/traffic.json)
        with open(os.path.join(ROOT, "profiles", "traffic.json")) as f:
            tj = json.load(f)
        best = 0
        for k, v in tj.items():  # the template instance that ran the headline launches (most dispatches)
            if k.startswith(kernel + "<") and n == N_ENVS_PER_GPU and fuse == DEFAULT_FUSE and args.layout == "cramped_room" \
                    and v.get("launches", 0) > best:
                best, traffic = v["launches"], v["hbm_bytes_per_launch"]
    except (OSError, ValueError, KeyError):
        pass
    if traffic is not None and not 0.5 < traffic / bytes_per_launch < 2.0:
        traffic = None  # profiles/traffic.json was collected for another launch length: not this launch's traffic
    issue = None
    try:  # SQ counters of the same kernel (tools/pmc_rollout.sh): what bounds it is instruction issue, not HBM
        with open(os.path.join(ROOT, "profiles", "sq_counters.json")) as f:
            sq = json.load(f)
        if kernel == sq.get("kernel", "").split("<")[0] and n == N_ENVS_PER_GPU and args.layout == "cramped_room":
            issue = {"valu_per_env_step": sq["valu_per_env_step"], "salu_per_env_step": sq["salu_per_env_step"],
                     "lds_per_env_step": sq["lds_per_env_step"], "valu_busy_frac": sq["valu_busy_frac"],
                     "wait_frac": sq["wait_any_frac"],
                     "wave_clk_per_env_step": sq.get("wave_clk_per_env_step"),
                     "note": "one wavefront per SIMD at 65 536 envs: every instruction of the wavefront issues in turn (~4 clk "
                             "each), so (VALU + SALU + LDS + VMEM per env-step) * 4 clk is the floor of a batched step "
                             "whatever the bytes moved"}
    except (OSError, ValueError, KeyError):
        pass
    out = {
        "metric": "env steps/sec (whole node), 65k parallel cramped_room envs",
        "value": value, "unit": "env steps/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": wall_max * 1e3 / total_steps, "higher_is_better": True, "scaling": "weak",
        "vs_baseline": None, "dtype": "u8", "data": "stub" if args.stub else "synthetic",
        "repeats": repeats, "timed_steps": total_steps, "timed_region_s": wall_max,
        "warmup_steps_run": (warm_launches + 3) * fuse,
        "ms_per_step_median": launch_med / fuse, "ms_per_step_min": launch_min / fuse,
        "ms_per_step_by_rank": [float(x) for x in per_rank.tolist()],
        "config": {"workload": "%s x %d envs/GPU, in-kernel Philox random policy, horizon %d auto-reset, outputs every step"
                               % (args.layout, n, HORIZON),
                   "envs_per_gpu": n, "fused_steps_per_launch": fuse, "launches": launches, "parallelism": "env-shard x%d" % world,
                   "timing_rule": "timed region = `repeats` back-to-back repetitions of the --steps-step region (steps x repeats "
                                  "batched steps, issued as whole %d-step launches whatever --steps is), repeats = smallest count "
                                  "with steps*repeats a multiple of %d and a region >= %.2f s at the warm-up rate; value and "
                                  "ms_per_step are over the whole region (wall clock, max over ranks)" % (fuse, fuse, args.min_seconds)},
        "roofline": {"bound": "hbm", "kernel": kernel, "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                     "frac": achieved / HBM_PEAK_GBS, "traffic": traffic,
                     "bytes_per_launch": bytes_per_launch, "launch_ms": launch_med, "launch_ms_min": launch_min,
                     "launch_ms_mean": dev_ms / max(1, launches), "launch_timing": "per-launch HIP events on the launch stream, median",
                     "bytes_model": "n_envs*(2*S + 17*T): S=%d B state in+out once per launch, 17 B outputs per env-step, actions in-kernel" % state_bytes,
                     "survey_8d_per_step_model_GBs": n * (2 * state_bytes + OUT_BYTES) * fuse / (launch_med * 1e-3) / 1e9,
                     "issue_bound": issue},
        "device_ms_timed_region": dev_ms,
        "aggregate": {"sparse_return_last_launch": float(metrics[0]), "shaped_return_last_launch": float(metrics[1]),
                      "episodes_done_last_step": float(metrics[2]),
                      "reduced_over": ("RCCL all-reduce" if not args.stub else "gloo all-reduce") if sharding._live() else "single rank"},
    }

    if rank == 0 and world == 1 and not args.no_extras and not args.stub:
        out["step_api"] = bench_step_api(env, dev, torch)
        out["single_env_api"] = bench_single_env_api(dev, torch)
        out["encode"] = bench_encode(dev, torch, VecOvercookedEnv)
        out["training_env"] = bench_training_env(dev, torch)
    if rank == 0 and world == 1 and not args.no_cpu_baseline and not args.stub:  # the CPU leg runs at N = 1 only
        out["cpu_baseline"] = cpu_baseline(args.layout, args.cpu_seconds)
        out["cpu_baseline"]["reference_python"] = REFERENCE_PYTHON
    if rank == 0:
        emit(out)
    sharding.barrier()


def run_other_config(args, torch, VecOvercookedEnv, sharding, dev, rank, world):
    """Side measurements for BASELINE.json configs 3-5 (same timing protocol; not the headline line)."""
    import numpy as np

    from overcooked_ai_amd.layouts import LayoutTable, spec_from_name

    n = args.envs
    encode = False
    if args.config == 3:
        env = VecOvercookedEnv("asymmetric_advantages", n, horizon=HORIZON, device=dev, auto_reset=True, seed=0,
                               env_offset=rank * n)
        encode, workload, sbytes = True, ("asymmetric_advantages x %d envs/GPU, random policy (in-kernel Philox actions) + lossless u8 "
                                          "encoding of every step into a [steps][envs] trajectory buffer (oc_rollout_encode)" % n), S_ASYM
    elif args.config == 4:
        names = ["cramped_room", "asymmetric_advantages", "coordination_ring", "forced_coordination", "counter_circuit"]
        table = LayoutTable([spec_from_name(nm) for nm in names], pad_to=(9, 5))
        lid = ((np.arange(n) + rank * n) % 5).astype(np.uint16)
        env = VecOvercookedEnv(table, n, horizon=HORIZON, device=dev, auto_reset=True, seed=0, env_offset=rank * n,
                               layout_id=lid)
        workload, sbytes = "5 canonical layouts padded to 9x5 (env e -> layout e %% 5) x %d envs/GPU, random policy" % n, 34
    else:
        from overcooked_ai_amd.layout_gen import reference_generated_layouts

        K = 4096  # the reference LayoutGenerator's own terrains (np.random.seed(0)), recorded as package data
        table = LayoutTable(reference_generated_layouts(K))
        lid = ((np.arange(n) + rank * n) % K).astype(np.uint16)
        env = VecOvercookedEnv(table, n, horizon=HORIZON, device=dev, auto_reset=True, seed=0, env_offset=rank * n,
                               layout_id=lid)
        workload, sbytes = "%d LayoutGenerator 9x5 terrains (reference generator, seed 0; env e -> terrain e %% %d) x %d envs/GPU, random policy" % (K, K, n), 36
    # configs[2] (the rollout of configs[1] plus oc_encode_lossless every step): ENC_FUSE steps per launch, the
    # observation of every step kept ([ENC_FUSE][n] u8 trajectory buffer: 7.7 GB at 65 536 9x5 envs)
    ENC_FUSE = 50
    fuse = ENC_FUSE if encode else max(1, args.fuse)
    rew = torch.zeros((fuse, n, 4), dtype=torch.float32, device=dev)
    fl = torch.zeros((fuse, n), dtype=torch.uint8, device=dev)
    obs = torch.empty((fuse, n, 2, env.width, env.height, 26), dtype=torch.uint8, device=dev) if encode else None

    def launch():  # one `fuse`-step unit of the workload
        if encode:
            env.rollout_encode(fuse, obs, rew, fl)
        else:
            env.rollout_random(fuse, rew, fl)

    for _ in range(-(-args.warmup // fuse)):
        launch()
    cal = _Timer(torch, dev)
    for _ in range(4):
        cal.mark()
        launch()
    cal.mark()
    cal.sync()
    est = torch.tensor([min(cal.launch_ms()[1:]) / fuse], dtype=torch.float64, device=dev)
    sharding.allreduce_max(est)
    repeats = plan_repeats(args.steps, fuse, float(est.item()), args.min_seconds)
    total_steps = args.steps * repeats
    tm = _Timer(torch, dev, reserve=total_steps // fuse + 1)
    tm.sync()
    sharding.barrier()
    tm.sync()
    t0 = time.perf_counter()
    for _ in range(total_steps // fuse):
        tm.mark()
        launch()
    tm.mark()
    tm.sync()
    sharding.barrier()
    tm.sync()
    wall = time.perf_counter() - t0
    per_launch = sorted(tm.launch_ms())
    unit_med = per_launch[len(per_launch) // 2]
    tmax = torch.tensor([wall], dtype=torch.float64, device=dev)
    sharding.allreduce_max(tmax)
    wall = float(tmax.item())
    unit_bytes = n * (2 * sbytes + OUT_BYTES * fuse) + (fuse * n * 2 * env.width * env.height * 26 if encode else 0)
    one_step = None
    if encode:  # the same step with caller-supplied actions, one call per step (oc_step_encode: what a policy in the loop pays)
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
        one_step = {"us_per_step": us, "value": n / us * 1e6, "unit": "env steps/s (one GPU)",
                    "note": "oc_step_encode: caller-supplied actions resident in HBM, one C call per batched step"}
    out = {"metric": "env steps/sec (whole node)", "value": float(world) * n * total_steps / wall, "unit": "env steps/s",
           "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": wall * 1e3 / total_steps,
           "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "u8", "data": "synthetic",
           "repeats": repeats, "timed_steps": total_steps, "timed_region_s": wall, "ms_per_step_median": unit_med / fuse,
           "config": {"workload": workload, "baseline_config": args.config, "envs_per_gpu": n,
                      "fused_steps_per_launch": fuse},
           "roofline": {"bound": "hbm", "achieved": unit_bytes / (unit_med * 1e-3) / 1e9, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                        "frac": unit_bytes / (unit_med * 1e-3) / 1e9 / HBM_PEAK_GBS, "traffic": None,
                        "bytes_per_launch": unit_bytes, "launch_ms": unit_med,
                        "note": "algorithmic bytes of one %d-step unit (all its kernels) / its median duration from HIP events" % fuse}}
    if encode:  # the f32 variant of the observation (what the reference's RLlib wrapper casts to): 10 steps per launch
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
    if one_step is not None:
        out["caller_actions_one_step"] = one_step
    if rank == 0:
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
    many = {"value": n * K / wall_many, "launch_ms": ms_many, "achieved_GBs": b / (ms_many * 1e-3) / 1e9,
            "frac": b / (ms_many * 1e-3) / 1e9 / HBM_PEAK_GBS, "note": "oc_step_many: K transitions with caller-supplied actions in one launch (envs stay on chip)"}
    return {"value": n * iters / wall, "step_many": many, "unit": "env steps/s", "launch_ms": ms, "bytes_per_launch": b,
            "achieved_GBs": b / (ms * 1e-3) / 1e9, "frac": b / (ms * 1e-3) / 1e9 / HBM_PEAK_GBS,
            "note": "oc_step, one launch per batched step incl. Python/ctypes launch overhead; SURVEY 8d: 67 B/env-step"}


def bench_single_env_api(dev, torch, episodes=3):
    """The drop-in single-env surface existing agents hit: OvercookedEnv.step -> OvercookedGridworld.get_state_transition
    (one env per call: pack -> H2D -> k_step -> D2H -> unpack), same protocol as the reference's CPU measurement
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
            "reference_python": REFERENCE_PYTHON["value"],
            "note": "OvercookedEnv.step through the single-env drop-in API: state and action written into a pinned host buffer "
                    "the kernel reads and writes in place (no staging copies), one launch + one stream wait per call; "
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
    out["note"] = "per batched step: one oc_multi_agent_step call = k_train_step (step + phi + shaped rewards + restart, fused) + k_encode"
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
