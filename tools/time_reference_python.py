#!/usr/bin/env python
"""Time the REFERENCE's own Python `OvercookedEnv.step` (SURVEY.md 8d-1 / BASELINE.md 3 protocol) on this machine.

TEST / MEASUREMENT INFRASTRUCTURE: imports the upstream package through oracle/ref_harness.py from
$OVERCOOKED_REFERENCE_SRC (default /root/reference/src).  The reference's sources are never copied into the repo; on a
GPU box OVERCOOKED_REFERENCE_SRC points at oracle/_ref/src, the reference byte-compiled by oracle/build_ref.py in the
build container (git-ignored build output that travels with the snapshot like liboc_amd.so) — bench.py's cpu_baseline
leg runs this script that way, in its own run, on the box's host cores.  Output: one JSON object.
LAYOUTS=cramped_room[,asymmetric_advantages] and EPISODES=N bound the run.

Protocol: cramped_room (and asymmetric_advantages), OvercookedEnv.from_mdp(mdp, horizon=400, info_level=0), env._mp = object()
(no MotionPlanner), actions np.random.RandomState(seed).randint(0, 6, (400, 2)) through Action.INDEX_TO_ACTION,
env.reset(regen_mdp=False) per episode, 1 warm-up episode, then EPISODES timed episodes; timer around the step loop only.
1 process, then multiprocessing.Pool(all usable cores) with one independent env per process (aggregate = sum of steps / max
worker time); both with and without env.lossless_state_encoding_mdp(next_state) per step.
"""
import json
import multiprocessing as mp
import os
import platform
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ.setdefault("PYTHONDONTWRITEBYTECODE", "1")
sys.dont_write_bytecode = True


def usable_cores():
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        if quota != "max":
            n = min(n, max(1, -(-int(quota) // int(period))))
    except (OSError, ValueError):
        pass
    return max(1, n)


def cpu_model():
    try:
        for l in open("/proc/cpuinfo"):
            if l.startswith("model name"):
                return l.split(":", 1)[1].strip()
    except OSError:
        pass
    return platform.processor() or "unknown"


def ref_mdp(R, layout):
    """The reference's OvercookedGridworld for `layout`, built by ITS from_grid (mdp.py:1103) from this repo's own layout
    data — the byte-compiled reference of oracle/build_ref.py ships no data files, so from_layout_name cannot be used."""
    from overcooked_ai_amd import layouts as L

    d = dict(L.read_layout_dict(layout))
    d.pop("grid")
    return R.OvercookedGridworld.from_grid(L.spec_from_name(layout).grid_rows(), base_layout_params=d)


def worker(job):
    layout, episodes, encode, seed = job
    import numpy as np

    from oracle import ref_harness

    R = ref_harness.load()
    mdp = ref_mdp(R, layout)
    env = R.OvercookedEnv.from_mdp(mdp, horizon=400, info_level=0)
    rng = np.random.RandomState(seed)
    I2A = R.Action.INDEX_TO_ACTION

    def episode():
        env.reset(regen_mdp=False)
        env._mp = object()  # never compute / pickle a MotionPlanner (overcooked_env.py:102-115)
        acts = rng.randint(0, 6, (400, 2))
        t0 = time.perf_counter()
        done, k = False, 0
        while not done:
            s, _, done, _ = env.step((I2A[acts[k, 0]], I2A[acts[k, 1]]))
            if encode:
                env.lossless_state_encoding_mdp(s)
            k += 1
        return k, time.perf_counter() - t0

    episode()
    steps = secs = 0
    for _ in range(episodes):
        k, dt = episode()
        steps += k
        secs += dt
    return steps, secs


def measure(layout, episodes, encode, procs):
    jobs = [(layout, episodes, encode, 1000 + i) for i in range(procs)]
    if procs == 1:
        res = [worker(jobs[0])]
    else:
        with mp.Pool(procs) as pool:
            res = pool.map(worker, jobs)
    steps = sum(r[0] for r in res)
    return {"steps_per_s": steps / max(r[1] for r in res), "processes": procs, "steps": steps,
            "slowest_worker_s": max(r[1] for r in res)}


def main():
    import numpy as np

    from oracle import ref_harness

    if not ref_harness.available():
        raise SystemExit("reference sources not found at %s" % ref_harness.REFERENCE_SRC)
    episodes = int(os.environ.get("EPISODES", "50"))
    cores = usable_cores()
    out = {"what": "reference OvercookedEnv.step (src/overcooked_ai_py/mdp/overcooked_env.py:244), SURVEY.md 8d-1 protocol",
           "reference_src": ref_harness.REFERENCE_SRC, "cpu_model": cpu_model(), "logical_cpus": os.cpu_count(),
           "usable_cores": cores, "python": platform.python_version(), "numpy": np.__version__,
           "episodes_per_process": episodes, "horizon": 400, "host": platform.node()}
    for layout in os.environ.get("LAYOUTS", "cramped_room,asymmetric_advantages").split(","):
        out[layout] = {
            "step_1core": measure(layout, episodes, False, 1),
            "step_allcores": measure(layout, episodes, False, cores),
            "step_encode_1core": measure(layout, max(10, episodes // 3), True, 1),
            "step_encode_allcores": measure(layout, max(10, episodes // 3), True, cores),
        }
    print(json.dumps(out))


if __name__ == "__main__":
    main()
