"""benchlib.common — what every part of bench.py shares: the workload constants (BASELINE.json configs), the command line, the timed
region (per-launch HIP events), rank spawning / NUMA pinning, the workload tables, the oracle replay (`parity_check`) and the
JSON line's `summary`.  bench.py is the entry point; see its docstring for the contract."""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BENCH_PY = os.path.join(ROOT, "bench.py")  # the entry point: what rank spawning and the --pmc child passes re-execute
if ROOT not in sys.path:
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
    ap.add_argument("--general-leg", default="",
                    help="(with --pmc-child) the general_path leg whose launch shape the child replays: one of benchlib.legs.GENERAL_CASES")
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
        procs.append(subprocess.Popen([sys.executable, BENCH_PY] + sys.argv[1:], env=env))
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
