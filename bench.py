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

HBM_PEAK_GBS = 8000.0  # MI355X HBM3E spec peak, /opt/skills/guides/MI355X_MICROARCH.md
N_ENVS_PER_GPU = 65536
HORIZON = 400
# env steps per oc_rollout_random launch: ten whole episodes.  A launch pays ~16 us before its first step and after its
# last (LUT staging, joint move table build, state load / store, the gap to the next launch): 12 % of a 400-step launch
# (126 us), 1.5 % of a 4 000-step one (1.07 ms) — measured 203.9 / 222.6 / 230.3 / 236.5 / 241.7 / 243.2 G env-steps/s at
# 400 / 800 / 1 200 / 2 000 / 4 000 / 8 000 steps per launch (before the last scheduling changes).  4 000 = 4.5 GB of
# per-step outputs per launch; the PMC byte counters were verified up to 8 000.
DEFAULT_FUSE = 10 * HORIZON
# launches per bench step (see the module docstring): 400 x 4 000 transitions = 0.34 s at 300 G env-steps/s, so the
# driver's `--steps 20` is a ~7 s timed region its utilisation sampler and its own clock can see
LAUNCHES_PER_STEP = 400
ENC_FUSE = 50                 # --config 3: transitions (+ observations) per oc_rollout_encode launch
ENC_LAUNCHES_PER_STEP = 200   # ... 200 x 50 = 10 000 transitions + observations per bench step (~0.3 s)
PMC_ENC_FUSE = 10             # --config 3: steps per launch inside the --pmc child passes (a 50-step launch wraps WRITE_SIZE)

# SURVEY.md §8d algorithmic bytes.  S = minimal state of cramped_room (2 players x 3 B + 14 non-floor cells
# + 1 pot tick + 2 B timestep -> 24 B), outputs 17 B per env-step, actions generated in-kernel (0 B).
S_CRAMPED = 24
OUT_BYTES = 17
S_ASYM = 44


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20, help="bench steps timed (one step = --launches-per-step launches)")
    ap.add_argument("--warmup", type=int, default=2, help="bench steps run before the timed region")
    ap.add_argument("--launches-per-step", type=int, default=0,
                    help="launches per bench step (default: %d rollout launches of --fuse transitions; %d oc_rollout_encode "
                         "launches of %d transitions for --config 3)" % (LAUNCHES_PER_STEP, ENC_LAUNCHES_PER_STEP, ENC_FUSE))
    ap.add_argument("--fuse", type=int, default=DEFAULT_FUSE, help="env steps fused per oc_rollout_random launch (default: ten 400-step episodes)")
    ap.add_argument("--envs", type=int, default=N_ENVS_PER_GPU, help="envs per GPU")
    ap.add_argument("--layout", default="cramped_room")
    ap.add_argument("--config", type=int, default=2, choices=[2, 3, 4, 5],
                    help="BASELINE.json config index (1-based): 2 = headline (default); 3 = asymmetric_advantages + "
                         "lossless encoding every step; 4 = 5-layout mix padded to 9x5; 5 = 4096 generated 9x5 terrains")
    ap.add_argument("--lane-pair", action="store_true", help="force the two-lanes-per-env rollout kernel")
    ap.add_argument("--predicate-interact", action="store_true", help="lane-per-env kernel with the predicate-network interact (v2)")
    ap.add_argument("--one-wavefront", action="store_true",
                    help="OC_OPT_ONE_WAVEFRONT: keep every env-step in one wavefront (no mover / interact split of the per-env-terrain step)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-extras", action="store_true", help="skip the step-API and encode side measurements")
    ap.add_argument("--cpu-seconds", type=float, default=10.0)
    ap.add_argument("--leg-seconds", type=float, default=1.2,
                    help="length of the timed region of each `configs` side leg (BASELINE configs[2..4]) of the default line")
    ap.add_argument("--terrains", type=int, default=4096,
                    help="--config 5: size of the LayoutGenerator terrain table (first 4 096 = the grids recorded from the "
                         "reference; more are generated on this host by the draw-exact restatement, up to 65 536)")
    ap.add_argument("--single-process", action="store_true",
                    help="--gpus N from ONE process: ShardedVecOvercookedEnv drives one shard per visible GPU on its own "
                         "stream (no ranks, no process group); the default for N > 1 stays one process per GPU")
    ap.add_argument("--stub", action="store_true",
                    help="CPU-only plumbing test (gloo, no kernels): exercises rank spawning and the reductions; never a measurement")
    ap.add_argument("--no-parity-check", action="store_true", help="skip the oracle comparison of one launch after the timed region")
    ap.add_argument("--flags-layout", choices=("tiled8", "step"), default="tiled8",
                    help="layout of the flags output of the headline's oc_rollout_random launches: tiled8 = [steps/8][envs][8] "
                         "(OC_OPT_FLAGS_TILED8, where the batch allows it; default), step = [steps][envs]")
    ap.add_argument("--parity-steps", type=int, default=0,
                    help="steps of the launch the parity check replays from reset (default: one whole --fuse launch at 1 GPU, "
                         "1 200 steps per rank otherwise)")
    ap.add_argument("--no-traffic", action="store_true",
                    help="do not collect roofline.traffic with rocprofv3 --pmc child passes of this same launch shape")
    ap.add_argument("--backend", choices=("nccl", "gloo"), default="nccl",
                    help="process-group backend of a multi-rank run (nccl = RCCL: what the driver's scaling runs use; gloo: rehearsals)")
    ap.add_argument("--share-device", action="store_true",
                    help="rehearsal on a box with fewer GPUs than ranks: rank r runs on GPU r %% device_count (with --backend gloo: RCCL "
                         "refuses two ranks on one device).  Exercises the per-rank path — process group, env_offset, NUMA pinning, "
                         "per-rank parity, the metric reductions — with real kernels; it says NOTHING about scaling or xGMI")
    ap.add_argument("--pmc-child", action="store_true", help=argparse.SUPPRESS)  # internal: the process rocprofv3 wraps
    return ap.parse_args()


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


def cpu_baseline(wl, n_envs, seconds):
    """The C oracle (a scalar port of the reference's algorithm) on the host: a bounded sample of the same workload
    (same layout table and env -> layout map, random policy, horizon 400 with auto-reset, outputs written every step),
    first on one core, then with the independent envs spread over all cores (OpenMP).  `value` is the all-cores figure."""
    import numpy as np

    from oracle import oracle as O

    orc = O.Oracle([O.mdp_from_layout_dict(sp.to_layout_dict()) for sp in wl["specs"]])
    n, T = min(8192, n_envs), 100
    lid = None if wl["lid"] is None else np.ascontiguousarray(wl["lid"][:n])
    st = orc.reset(orc.new_state(n), layout_id=lid)
    ep = np.zeros((n, 4), np.float32)
    tg = 0

    def timed(budget):
        nonlocal tg
        orc.rollout_random(st, 10, horizon=HORIZON, options=1, seed=0, t0=tg, ep_returns=ep, layout_id=lid)  # warm
        tg += 10
        t0 = time.perf_counter()
        orc.rollout_random(st, T, horizon=HORIZON, options=1, seed=0, t0=tg, ep_returns=ep, layout_id=lid)
        tg += T
        probe = time.perf_counter() - t0
        reps = max(1, int(budget / max(probe, 1e-6)))
        t0 = time.perf_counter()
        for _ in range(reps):
            orc.rollout_random(st, T, horizon=HORIZON, options=1, seed=0, t0=tg, ep_returns=ep, layout_id=lid)
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
        "sample": "%d envs x %d steps of the bench workload (C oracle, %d threads, %.1f s); single core: %d steps in %.1f s"
                  % (n, steps_all, cores, dt_all, steps1, dt1),
    }


def reference_python(args=None):
    """The reference's own rate (north_star: "next to the reference Python OvercookedEnv.step timed on the same box's host
    cores").  /root/reference does not exist on the GPU box and its sources are never copied into this repo; what travels
    is oracle/_ref/src — the reference's hot-path modules byte-compiled by oracle/build_ref.py in the build container
    (build output, git-ignored, like liboc_amd.so).  When it is there, tools/time_reference_python.py times it IN THIS RUN
    on this box (1 core and all usable cores, with and without the lossless encoding): same_run / same_box true.
    Otherwise the stored figure of an earlier box is replayed and labelled as such."""
    import subprocess

    src = os.path.join(ROOT, "oracle", "_ref", "src")
    if os.path.exists(os.path.join(src, "overcooked_ai_py", "mdp", "overcooked_env.pyc")):
        try:
            # >= 50 timed episodes, cramped_room and asymmetric_advantages (config 3), with and without the encoding
            env = dict(os.environ, OVERCOOKED_REFERENCE_SRC=src, LAYOUTS="cramped_room,asymmetric_advantages", EPISODES="50",
                       PYTHONDONTWRITEBYTECODE="1")
            for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT"):
                env.pop(k, None)
            t0 = time.perf_counter()
            p = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "time_reference_python.py")], env=env, cwd=ROOT,
                               capture_output=True, text=True, timeout=400)
            j = json.loads([l for l in p.stdout.splitlines() if l.startswith("{")][-1])
            cr, aa = j["cramped_room"], j.get("asymmetric_advantages")
            aa_leg = None
            if aa:  # BASELINE configs[2]'s CPU side: asymmetric_advantages with lossless_state_encoding_mdp every step
                aa_leg = {"value": aa["step_encode_allcores"]["steps_per_s"], "unit": "env steps/s", "kind": "reference",
                          "cores": aa["step_encode_allcores"]["processes"], "single_core": aa["step_encode_1core"]["steps_per_s"],
                          "without_encoding": {"value": aa["step_allcores"]["steps_per_s"], "single_core": aa["step_1core"]["steps_per_s"]},
                          "sample": "the reference's OvercookedEnv.step + lossless_state_encoding_mdp (overcooked_env.py:244, 276) on "
                                    "asymmetric_advantages, horizon 400, %s episodes per process after 1 warm-up, one env per process, "
                                    "same run, same box" % j.get("episodes_per_process")}
            return {
                "asymmetric_advantages": aa_leg,
                "value": cr["step_1core"]["steps_per_s"], "unit": "env steps/s", "cores": 1,
                "all_cores": {"value": cr["step_allcores"]["steps_per_s"], "cores": cr["step_allcores"]["processes"],
                              "note": "one env per process, multiprocessing.Pool"},
                "with_lossless_encoding": {"value": cr["step_encode_1core"]["steps_per_s"], "cores": 1,
                                           "all_cores": cr["step_encode_allcores"]["steps_per_s"]},
                "same_run": True, "same_box": True, "seconds": time.perf_counter() - t0,
                "where": "this box, this run (%s, %s usable cores, CPython %s, numpy %s)"
                         % (j.get("cpu_model"), j.get("usable_cores"), j.get("python"), j.get("numpy")),
                "what": j.get("what", "") + ": cramped_room, horizon 400, np.random.RandomState joint actions, %s episodes per "
                                            "process after 1 warm-up (%d timed steps on one core)"
                                            % (j.get("episodes_per_process"), cr["step_1core"]["steps"]),
                "source": "tools/time_reference_python.py on oracle/_ref/src: the reference's own modules (overcooked_env.py, "
                          "overcooked_mdp.py, actions.py, ...) byte-compiled from /root/reference by oracle/build_ref.py",
            }
        except Exception as exc:  # (byte code of another CPython, a missing module: fall back, say why)
            stored = _reference_python_stored()
            stored["same_run_attempt"] = repr(exc)[:300]
            return stored
    return _reference_python_stored()


def _reference_python_stored():
    """Fallback: the figure of tools/time_reference_python.py run once on an MI355X box of this pool in round 3
    (profiles/r03_reference_python_gpubox.json), else the build container's figure — NOT this run."""
    path = os.path.join(ROOT, "profiles", "r03_reference_python_gpubox.json")
    try:
        with open(path) as f:
            j = json.load(f)
        cr = j["cramped_room"]
        return {
            "value": cr["step_1core"]["steps_per_s"], "unit": "env steps/s", "cores": 1,
            "all_cores": {"value": cr["step_allcores"]["steps_per_s"], "cores": cr["step_allcores"]["processes"],
                          "note": "one env per process, multiprocessing.Pool"},
            "with_lossless_encoding": {"value": cr["step_encode_1core"]["steps_per_s"], "cores": 1,
                                       "all_cores": cr["step_encode_allcores"]["steps_per_s"]},
            "same_run": False, "same_box": False,
            "where": "an MI355X box of this pool (%s, %s usable cores, CPython %s, numpy %s) in a separate gpurun call, NOT this run"
                     % (j.get("cpu_model"), j.get("usable_cores"), j.get("python"), j.get("numpy")),
            "what": j.get("what", "") + ": cramped_room, horizon 400, np.random.RandomState joint actions, %s episodes per process "
                                        "after 1 warm-up" % j.get("episodes_per_process"),
            "source": "profiles/r03_reference_python_gpubox.json (tools/time_reference_python.py)",
        }
    except (OSError, ValueError, KeyError):
        return dict(REFERENCE_PYTHON)


REFERENCE_PYTHON = {
    "value": 16400.0, "unit": "env steps/s", "cores": 1,
    "all_cores": {"value": 74000.0, "cores": 8, "note": "one env per process, multiprocessing.Pool(8)"},
    "same_run": False, "same_box": False,
    "where": "build container (8 vCPU Xeon 2.1 GHz, CPython 3.10.12, numpy 2.2.6), not the GPU box",
    "what": "reference OvercookedEnv.step (src/overcooked_ai_py/mdp/overcooked_env.py:244), cramped_room, horizon 400, "
            "np.random.RandomState joint actions, >= 50 episodes after 1 warm-up",
    "source": "BASELINE.md 2 / SURVEY.md 8d-1",
}


def timed_launches(torch, dev, sharding, launch, n_launches):
    """The timed region: barrier + synchronize, `n_launches` back-to-back calls of `launch()` with a HIP event before each
    (and one after the last) on the launch stream, synchronize + barrier.  Returns (wall seconds of this rank, per-launch
    milliseconds in issue order)."""
    tm = _Timer(torch, dev, reserve=n_launches + 1)
    tm.sync()
    sharding.barrier()
    tm.sync()
    t0 = time.perf_counter()
    for _ in range(n_launches):
        tm.mark()
        launch()
    tm.mark()
    tm.sync()
    sharding.barrier()
    tm.sync()
    return time.perf_counter() - t0, tm.launch_ms()


def measure_store_only(torch, dev, env, n, fuse, rew, fl, rate_per_gpu, launch_med_ms, reps=12, tiled8=False):
    """The ceiling of the output format: `reps` launches of oc_output_stores_only — per env-step one 16-byte reward quad and
    one flag byte into the same arrays the rollout writes, in the flags layout the rollout was timed in ([step][env] rows, or
    the OC_OPT_FLAGS_TILED8 tiles: one 8-byte store per env and 8-step block), no state, no game — timed with HIP events."""
    import ctypes

    from overcooked_ai_amd import _lib

    lib = env.lib
    stream = ctypes.c_void_p(torch.cuda.current_stream(dev).cuda_stream)
    opt = _lib.OPT_FLAGS_TILED8 if tiled8 else 0

    def launch():
        rc = lib.oc_output_stores_only(n, fuse, rew.data_ptr(), fl.data_ptr(), opt, stream)
        if rc:
            raise RuntimeError("oc_output_stores_only: rc %d" % rc)

    for _ in range(3):
        launch()
    tm = _Timer(torch, dev, reserve=reps + 1)
    tm.sync()
    for _ in range(reps):
        tm.mark()
        launch()
    tm.mark()
    tm.sync()
    ms = sorted(tm.launch_ms())
    med = ms[len(ms) // 2]
    rate = n * fuse / (med * 1e-3)
    return {"what": "oc_output_stores_only: nothing but the rollout's output stores (16-byte quad + flag byte per env-step, same "
                    "arrays, same launch shape, flags layout %s), median of %d launches"
                    % ("[steps/8][envs][8] (OC_OPT_FLAGS_TILED8), as timed" if tiled8 else "[steps][envs], as timed", reps),
            "flags_layout": "tiled8" if tiled8 else "step",
            "launch_ms": med, "env_steps_per_s": rate, "GBs": n * fuse * OUT_BYTES / (med * 1e-3) / 1e9,
            "frac_of_peak": n * fuse * OUT_BYTES / (med * 1e-3) / 1e9 / HBM_PEAK_GBS,
            "rollout_over_store_only": (n * fuse / (launch_med_ms * 1e-3)) / rate}


def launches_for(torch, dev, launch, seconds, lo=3):
    """How many launches fill `seconds` (side legs: a bounded region, not a step count): 3 calibration launches."""
    cal = _Timer(torch, dev)
    for _ in range(3):
        cal.mark()
        launch()
    cal.mark()
    cal.sync()
    return max(lo, int(seconds * 1e3 / max(min(cal.launch_ms()[1:]), 1e-6)))


class _StubEnv:
    """CPU stand-in for VecOvercookedEnv used ONLY by `--stub` (tests of the rank-spawning / reduction / parity-check
    plumbing on a box without GPUs, gloo backend): it steps the C oracle where the product steps the HIP kernels, so the
    JSON it yields says data: "stub" and is never a measurement."""

    def __init__(self, wl, n, rank):
        import numpy as np

        from oracle import oracle as O

        self.n_envs, self.t_global, self.env_offset, self.lid = n, 0, rank * n, wl["lid"]
        self.orc = O.Oracle([O.mdp_from_layout_dict(sp.to_layout_dict()) for sp in wl["specs"]])
        self.n_planes = self.orc.n_planes
        self.st = self.orc.reset(self.orc.new_state(n), layout_id=self.lid)
        self.ep = np.zeros((n, 4), np.float32)

    def rollout_random(self, k, rew=None, fl=None):
        import torch

        r, f = self.orc.rollout_random(self.st, k, horizon=HORIZON, options=1, seed=0, env_offset=self.env_offset,
                                       t0=self.t_global, layout_id=self.lid, ep_returns=self.ep)
        self.t_global += k
        if rew is not None:
            rew[:k].copy_(torch.from_numpy(r))
            fl[:k].copy_(torch.from_numpy(f))

    def get_packed_state(self):
        return self.st

    @property
    def ep_returns(self):
        import torch

        return torch.from_numpy(self.ep)


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


def pin_to_gpu_numa(torch, local_rank):
    """Keep this rank's host threads (launch loop, the oracle of the parity check) on the NUMA node its GPU hangs off:
    PCI bus id of the HIP device -> /sys/bus/pci/devices/<id>/numa_node -> that node's cpulist, intersected with the
    affinity the process already has.  Returns the node (None when the topology cannot be read: nothing is changed)."""
    try:
        props = torch.cuda.get_device_properties(local_rank)
        bus = None
        if hasattr(props, "pci_bus_id") and hasattr(props, "pci_device_id"):
            bus = "%04x:%02x:%02x.0" % (getattr(props, "pci_domain_id", 0), props.pci_bus_id, props.pci_device_id)
        if bus is None or not os.path.exists("/sys/bus/pci/devices/%s/numa_node" % bus):
            import ctypes

            buf = ctypes.create_string_buffer(64)
            hip = ctypes.CDLL("libamdhip64.so")
            if hip.hipDeviceGetPCIBusId(buf, 64, int(local_rank)) != 0:
                return None
            bus = buf.value.decode().lower()
        node = int(open("/sys/bus/pci/devices/%s/numa_node" % bus).read())
        if node < 0:
            return None
        cpus = set()
        for part in open("/sys/devices/system/node/node%d/cpulist" % node).read().strip().split(","):
            a, _, b = part.partition("-")
            cpus.update(range(int(a), int(b or a) + 1))
        cpus &= os.sched_getaffinity(0)
        if cpus:
            os.sched_setaffinity(0, cpus)
        return node
    except Exception:
        return None


def src_hash():
    from overcooked_ai_amd import build

    return build.source_hash()


def make_workload(args, rank):
    """The batch a rank owns for --config 2 / 4 / 5 (BASELINE configs[1] / [3] / [4]): layout table, per-env layout ids
    of ITS global env range, minimal-state bytes of SURVEY 8d, and a description."""
    import numpy as np

    from overcooked_ai_amd.layouts import LayoutTable, spec_from_name

    n = args.envs
    if args.config == 2:
        table = LayoutTable([spec_from_name(args.layout)])
        return {"table": table, "specs": table.specs, "lid": None,
                "sbytes": S_CRAMPED if args.layout == "cramped_room" else 4 * ((table.n_planes * 16) // 4),
                "workload": "%s x %d envs/GPU, in-kernel Philox random policy, horizon %d auto-reset, outputs every step"
                            % (args.layout, n, HORIZON)}
    if args.config == 4:
        names = ["cramped_room", "asymmetric_advantages", "coordination_ring", "forced_coordination", "counter_circuit"]
        table = LayoutTable([spec_from_name(nm) for nm in names], pad_to=(9, 5))
        lid = ((np.arange(n) + rank * n) % 5).astype(np.uint16)
        return {"table": table, "specs": table.specs, "lid": lid, "sbytes": 34,
                "workload": "5 canonical layouts padded to 9x5 (global env e -> layout e %% 5) x %d envs/GPU, random policy, "
                            "horizon %d auto-reset, outputs every step" % (n, HORIZON)}
    from overcooked_ai_amd.layout_gen import generate_reference_layouts, reference_generated_layouts

    # the reference LayoutGenerator's own terrains (np.random.seed(0)): recorded as package data up to 4 096, generated here
    # by its draw-exact restatement (layout_gen.generate_reference_layouts) beyond
    K = int(getattr(args, "terrains", 4096))
    table = LayoutTable(reference_generated_layouts(K) if K <= 4096 else generate_reference_layouts(K, seed=0))
    lid = ((np.arange(n) + rank * n) % K).astype(np.uint16)
    return {"table": table, "specs": table.specs, "lid": lid, "sbytes": 36,
            "workload": "%d LayoutGenerator 9x5 terrains (reference generator, seed 0; global env e -> terrain e %% %d) x %d "
                        "envs/GPU, random policy, horizon %d auto-reset, outputs every step" % (K, K, n, HORIZON)}


def parity_check(torch, wl, make_env, n, rank, steps, rew, fl, threads, tiled8=False):
    """Replay ONE launch of the timed shape from reset and compare every reward row, every flag byte, the final packed
    states and the episode returns with the C oracle (the checker, not the thing measured), in 400-step chunks."""
    import numpy as np

    from oracle import oracle as O

    t_start = time.perf_counter()
    env = make_env()
    if tiled8:  # the timed launches' own flags layout: [steps / 8][envs][8], untiled for the comparison
        steps -= steps % 8
        flt = fl.view(-1)[:steps * n].view(steps // 8, n, 8)
        env.rollout_random(steps, rew[:steps], flt, flags_tiled8=True)
        fl = env.untile_flags(flt)
    else:
        env.rollout_random(steps, rew[:steps], fl[:steps])
    threads = O.set_threads(max(1, threads))
    orc = O.Oracle([O.mdp_from_layout_dict(sp.to_layout_dict()) for sp in wl["specs"]])
    lid = wl["lid"]
    st = orc.reset(orc.new_state(n), layout_id=lid)
    ep = np.zeros((n, 4), np.float32)
    bad_steps, restarts, chunk = 0, 0, 400
    for c0 in range(0, steps, chunk):
        k = min(chunk, steps - c0)
        rew_o, fl_o = orc.rollout_random(st, k, horizon=HORIZON, options=1, seed=0, env_offset=rank * n, t0=c0,
                                         layout_id=lid, ep_returns=ep)
        rg, fg = rew[c0:c0 + k].cpu().numpy(), fl[c0:c0 + k].cpu().numpy()
        bad_steps += int(((rg != rew_o).any(axis=2) | (fg != fl_o)).sum())
        restarts += int(((fl_o & 4) != 0).sum())
    bad_states = int((np.asarray(env.get_packed_state()) != st).any(axis=(0, 2)).sum())
    bad_returns = int((env.ep_returns.cpu().numpy() != ep).any(axis=1).sum())
    O.set_threads(1)
    return {"envs": n, "steps": steps, "mismatches": bad_steps + bad_states + bad_returns,
            "mismatching_env_steps": bad_steps, "mismatching_final_states": bad_states,
            "mismatching_episode_returns": bad_returns, "restarts_covered": restarts,
            "seconds": time.perf_counter() - t_start, "oracle_threads": threads,
            "what": "one %d-step oc_rollout_random launch from reset (seed 0, global env offset %d): every reward quad and "
                    "flag byte of every env-step, the final packed states and the episode returns, bit for bit against "
                    "oracle/overcooked_oracle.c" % (steps, rank * n)}


def flags_tiled8_ok(args, env, fuse, rew, fl):
    """Does this batch / launch shape take the tiled flags layout (OC_OPT_FLAGS_TILED8: the pipelined joint-table kernel,
    launches of whole 8-step blocks)?  Asked by trying one launch; the env is put back to where it was."""
    if getattr(args, "flags_layout", "step") != "tiled8" or args.stub or fuse % 8 or not hasattr(env, "untile_flags"):
        return False
    n = fl.shape[1]
    saved = (env.state.clone(), env.t_global, env.steps_done, env._epoch, env.ep_returns.clone() if env.ep_returns is not None else None)
    try:
        env.rollout_random(8, rew[:8], fl.view(-1)[:8 * n].view(1, n, 8), flags_tiled8=True)
        ok = True
    except Exception:
        ok = False
    env.state.copy_(saved[0])
    env.t_global, env.steps_done, env._epoch = saved[1], saved[2], saved[3]
    if saved[4] is not None:
        env.ep_returns.copy_(saved[4])
    return ok


def pmc_child(args, torch, VecOvercookedEnv, dev):
    """The process the --pmc passes wrap: the same batch, reset, then 3 launches of the timed shape and nothing else.
    --config 3: launches of PMC_ENC_FUSE (10) steps instead of ENC_FUSE (50) — WRITE_SIZE wraps on the 7.7 GB a 50-step
    launch writes; the kernel streams the same bytes per step whatever the launch length."""
    if args.config == 3:
        n, fuse = args.envs, PMC_ENC_FUSE
        env = VecOvercookedEnv("asymmetric_advantages", n, horizon=HORIZON, device=dev, auto_reset=True, seed=0)
        rew = torch.zeros((fuse, n, 4), dtype=torch.float32, device=dev)
        fl = torch.zeros((fuse, n), dtype=torch.uint8, device=dev)
        obs = torch.empty((fuse, n, 2, env.width, env.height, 26), dtype=torch.uint8, device=dev)
        for _ in range(3):
            env.rollout_encode(fuse, obs, rew, fl)
        torch.cuda.synchronize(dev)
        return
    wl = make_workload(args, 0)
    n, fuse = args.envs, max(1, args.fuse)
    env = rollout_workload_env(args, wl, n, 0, dev, VecOvercookedEnv)()
    rew = torch.zeros((fuse, n, 4), dtype=torch.float32, device=dev)
    fl = torch.zeros((fuse, n), dtype=torch.uint8, device=dev)
    tiled8 = args.flags_layout == "tiled8"  # (decided by the parent, which has tried it)
    for _ in range(3):
        if tiled8:
            env.rollout_random(fuse, rew, fl.view(fuse // 8, n, 8), flags_tiled8=True)
        else:
            env.rollout_random(fuse, rew, fl)
    torch.cuda.synchronize(dev)


def measure_traffic(args, kernel, tiled8=False):
    """roofline.traffic measured by THIS run: two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE — they do not fit one
    pass, MI355X_MICROARCH.md 'rocprofv3 PMC slots') over a child of this same command that runs 3 launches of the timed
    shape.  KiB -> bytes; FETCH_SIZE doubled (gfx950 tallies the 128-byte requests of wide coalesced reads at 64 B, same
    guide, 'HBM').  Returns (bytes per launch or None, provenance dict)."""
    import csv
    import glob
    import shutil
    import sqlite3
    import subprocess
    import tempfile

    rocprof = shutil.which("rocprofv3") or "/opt/rocm/bin/rocprofv3"
    if not os.path.exists(rocprof):
        return None, {"how": "not collected", "why": "rocprofv3 not found"}
    child = [sys.executable, os.path.abspath(__file__), "--pmc-child", "--config", str(args.config), "--envs", str(args.envs),
             "--fuse", str(args.fuse), "--layout", args.layout, "--terrains", str(args.terrains),
             "--flags-layout", "tiled8" if tiled8 else "step"]  # (the child takes the parent's decision: no probe launch in the counters)
    for flag, on in (("--lane-pair", args.lane_pair), ("--predicate-interact", args.predicate_interact),
                     ("--one-wavefront", args.one_wavefront)):
        if on:
            child.append(flag)
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
    env["TMPDIR"] = "/tmp"
    got, launches = {}, 0
    for counter, scale in (("FETCH_SIZE", 2.0), ("WRITE_SIZE", 1.0)):
        d = tempfile.mkdtemp(prefix="oc_pmc_", dir="/tmp")
        try:
            p = subprocess.run([rocprof, "--pmc", counter, "-d", d, "-o", "p", "--output-format", "csv", "rocpd", "--"] + child,
                               cwd="/tmp", env=env, capture_output=True, text=True, timeout=300)
            vals = []
            for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
                for row in csv.DictReader(open(f, newline="")):
                    low = {k.lower(): v for k, v in row.items()}
                    if any(k + "<" in low.get("kernel_name", "") for k in kernel.split("|")) and low.get("counter_name") == counter:
                        vals.append(float(low["counter_value"]))
            if not vals:
                for f in glob.glob(os.path.join(d, "**", "*.db"), recursive=True):
                    db = sqlite3.connect(f)
                    try:
                        for k in kernel.split("|"):
                            vals += [float(r[0]) for r in db.execute(
                                "select counter_value from pmc_events where counter_name=? and name like ?", (counter, "%" + k + "<%"))]
                    finally:
                        db.close()
            if not vals:
                return None, {"how": "not collected", "why": "no %s rows for %s (rocprofv3 rc %d): %s"
                                                              % (counter, kernel, p.returncode, (p.stderr or "")[-300:])}
            got[counter] = sum(vals) / len(vals) * 1024.0 * scale
            launches = len(vals)
        except Exception as e:  # a profiler problem must never cost the measurement
            return None, {"how": "not collected", "why": "%s pass failed: %r" % (counter, e)}
        finally:
            shutil.rmtree(d, ignore_errors=True)
    return got["FETCH_SIZE"] + got["WRITE_SIZE"], {
        "how": "same run: rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE in separate child passes of this command's launch "
               "shape (%d launches each, mean per launch); KiB -> bytes, FETCH_SIZE x 2 (gfx950), WRITE_SIZE as reported"
               % launches,
        "fetch_bytes": got["FETCH_SIZE"], "write_bytes": got["WRITE_SIZE"], "kernel_source_sha": src_hash()}


def traffic_from_file(kernel, n, fuse, layout, bytes_per_launch):
    """Fallback provenance: the PMC figure tools/profile_round.sh stored for this launch shape — only when the kernel
    sources are the ones that were profiled."""
    try:
        with open(os.path.join(ROOT, "profiles", "traffic.json")) as f:
            tj = json.load(f)
    except (OSError, ValueError):
        return None, {"how": "not collected", "why": "no profiles/traffic.json"}
    if tj.get("_kernel_source_sha") != src_hash():
        return None, {"how": "not collected", "why": "profiles/traffic.json was recorded for other kernel sources (sha %s)"
                                                      % tj.get("_kernel_source_sha")}
    best, traffic = 0, None
    for k, v in tj.items():
        if any(k.startswith(kn + "<") for kn in kernel.split("|")) and n == N_ENVS_PER_GPU and fuse == DEFAULT_FUSE and layout == "cramped_room" \
                and v.get("launches", 0) > best:
            best, traffic = v["launches"], v["hbm_bytes_per_launch"]
    if traffic is None or not 0.5 < traffic / bytes_per_launch < 2.0:
        return None, {"how": "not collected", "why": "profiles/traffic.json holds no entry for this launch shape"}
    return traffic, {"how": "replayed from profiles/traffic.json (an earlier rocprofv3 --pmc run of the same kernel sources), NOT "
                            "measured in this run", "kernel_source_sha": src_hash()}


def issue_counters(kernel, n, layout):
    """SQ counters of the headline kernel from profiles/sq_counters.json — a stored profile, labelled as such, and dropped
    when the kernel sources have changed since it was taken."""
    try:
        with open(os.path.join(ROOT, "profiles", "sq_counters.json")) as f:
            sq = json.load(f)
    except (OSError, ValueError):
        return None
    if sq.get("kernel", "").split("<")[0] not in kernel.split("|") or n != N_ENVS_PER_GPU or layout != "cramped_room":
        return None
    if sq.get("kernel_source_sha") != src_hash():
        return None
    return {"valu_per_env_step": sq["valu_per_env_step"], "salu_per_env_step": sq["salu_per_env_step"],
            "lds_per_env_step": sq["lds_per_env_step"], "valu_busy_frac": sq["valu_busy_frac"],
            "wait_frac": sq["wait_any_frac"], "wave_clk_per_env_step": sq.get("wave_clk_per_env_step"),
            "wavefronts_per_64_envs": sq.get("wavefronts_per_64_envs", 1),
            "source": "replayed from profiles/sq_counters.json (rocprofv3 --pmc SQ_* passes of tools/pmc_rollout.sh on the same "
                      "kernel sources, sha %s), NOT measured in this run" % sq.get("kernel_source_sha"),
            "note": "instructions per env-step of a 64-env group; two wavefronts share them (k_rollout5: a mover and an interact "
                    "wavefront per 64 envs, two wavefronts per SIMD at 65 536 envs), and a batched step costs about the interact "
                    "wavefront's own instruction stream x the ~5 clk one wavefront needs per instruction (profiles/"
                    "r06_interact_stream.txt) — wave_clk_per_env_step is what one wavefront measured — whatever the bytes moved"}


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


def rollout_workload_env(args, wl, n, rank, dev, VecOvercookedEnv):
    def make_env():
        if args.stub:
            return _StubEnv(wl, n, rank)
        env = VecOvercookedEnv(wl["table"], n, horizon=HORIZON, device=dev, auto_reset=True, seed=0, env_offset=rank * n,
                               layout_id=wl["lid"])
        env.lane_pair = getattr(args, "lane_pair", False)
        env.predicate_interact = getattr(args, "predicate_interact", False)
        env.one_wavefront = getattr(args, "one_wavefront", False)
        return env
    return make_env


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


def general_legs(args, torch, VecOvercookedEnv, sharding, dev):
    """The batches OUTSIDE "two players, <= 2 pots, <= 64 cells, new dynamics, no event log" (VERDICT r5 #3): old dynamics — what
    the reference's paper-reproduction runs use (human_aware_rl/ppo/run_experiments.sh:4-12; mdp.py:1517-1518, 1696-1701) —,
    per-episode event logging (env.py:382-401 game_stats), and a 13 x 5 layout (65 cells).  Same launch shape as the headline
    (65 536 envs x 4 000 fused steps), each with roofline and an oracle parity check."""
    from overcooked_ai_amd.layouts import LayoutTable, spec_from_name

    legs = {}
    n, fuse = N_ENVS_PER_GPU, DEFAULT_FUSE
    cases = (("coordination_ring_old_dynamics", lambda: LayoutTable([spec_from_name("coordination_ring", old_dynamics=True)]), {}),
             ("asymmetric_advantages_old_dynamics", lambda: LayoutTable([spec_from_name("asymmetric_advantages", old_dynamics=True)]), {}),
             ("cramped_room_event_log", lambda: LayoutTable([spec_from_name("cramped_room")]), {"track_events": True}),
             ("marshmallow_experiment", lambda: LayoutTable([spec_from_name("marshmallow_experiment")]), {}))
    for name, make_table, kw in cases:
        try:
            table = make_table()
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
        except Exception as exc:
            legs[name] = {"error": repr(exc)[:300]}
    return legs


def summarize(out):
    """The line's claims in <= 1.2 KB, emitted as its LAST key so that a 2 000-character tail shows every one of them: per leg
    G env-steps/s (1e9), roofline fraction, PMC traffic over algorithmic bytes, parity mismatches."""
    def leg(d, scale=1e9):
        if not isinstance(d, dict) or "error" in d or "roofline" not in d:
            return {"error": True}
        rl = d["roofline"]
        r = {"G": round(d["value"] / scale, 2), "frac": round(rl["frac"], 3)}
        if rl.get("traffic"):
            r["t/a"] = round(rl["traffic"] / rl["bytes_per_launch"], 3)
        pc = d.get("parity_check") or {}
        if "mismatches" in pc or "mismatching_observations" in pc:
            r["mis"] = pc.get("mismatches", pc.get("mismatching_observations"))
        return r

    sm = {"headline": leg(out)}
    so = (out.get("roofline") or {}).get("store_only") or {}
    if "rollout_over_store_only" in so:
        sm["headline"]["over_store_only"] = round(so["rollout_over_store_only"], 3)
    ib = (out.get("roofline") or {}).get("issue_bound")
    if ib:
        sm["wave_clk_per_env_step"] = ib.get("wave_clk_per_env_step")
    for k, v in (out.get("configs") or {}).items():
        sm["cfg" + k] = leg(v)
    for k, v in (out.get("general_path") or {}).items():
        sm[k] = leg(v)
    try:
        sa = out.get("step_api") or {}
        if "launch_ms" in sa:
            sm["step_api"] = {"us": round(sa["launch_ms"] * 1e3, 2), "frac": round(sa["frac"], 3),
                              "many_frac": round((sa.get("step_many") or {}).get("frac", 0.0), 3)}
            if "resident" in sa and "us_per_batched_step" in sa["resident"]:
                sm["step_api"]["resident_us"] = round(sa["resident"]["us_per_batched_step"], 2)
                sm["step_api"]["resident_mism"] = sa["resident"]["parity_check"]["mismatches"]
        enc = out.get("encode") or {}
        sm["encode_frac"] = {k: round(v["frac"], 3) for k, v in enc.items() if isinstance(v, dict) and "frac" in v}
        tr = out.get("training_env") or {}
        sm["train_us"] = {k: round(v["us_per_batched_step"], 1) for k, v in tr.items() if isinstance(v, dict) and "us_per_batched_step" in v}
        se = out.get("single_env_api") or {}
        if "value" in se:
            sm["single_env_steps_s"] = round(se["value"])
    except Exception:
        pass
    cb = out.get("cpu_baseline") or {}
    if cb:
        sm["cpu"] = {"kind": cb.get("kind"), "value": round(cb.get("value", 0)), "cores": cb.get("cores"),
                     "single_core": round(cb.get("single_core", 0)), "port": round((cb.get("port") or {}).get("value", 0))}
    return sm


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
