"""benchlib.traffic — `roofline.traffic` (HBM bytes from rocprofv3 --pmc child passes of the timed launch shape), `roofline.issue_bound`
(stored SQ counters keyed to the source hash) and `roofline.store_only` (the output stores alone, oc_output_stores_only)."""
import argparse
import json
import os
import sys
import time

from benchlib.common import *  # noqa: F401,F403 (constants and helpers)
from benchlib.common import _StubEnv, _Timer  # noqa: F401


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
    n, fuse = args.envs, max(1, args.fuse)
    if getattr(args, "general_leg", ""):  # a general_path leg: its own table and env options (benchlib.legs.general_case)
        from benchlib.legs import general_case

        table, kw = general_case(args.general_leg)
        env = VecOvercookedEnv(table, n, horizon=HORIZON, device=dev, auto_reset=True, seed=0, **kw)
    else:
        wl = make_workload(args, 0)
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

def measure_traffic(args, kernel, tiled8=False, extra=None):
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
    child = [sys.executable, BENCH_PY, "--pmc-child", "--config", str(args.config), "--envs", str(args.envs),
             "--fuse", str(args.fuse), "--layout", args.layout, "--terrains", str(args.terrains),
             "--flags-layout", "tiled8" if tiled8 else "step"]  # (the child takes the parent's decision: no probe launch in the counters)
    child += list(extra or [])
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
