"""benchlib.cpu_baseline — the `cpu_baseline` object of the bench line: the reference's own OvercookedEnv.step timed on the GPU box's
host cores (oracle/_ref, `kind: "reference"`) with the C oracle on the same workload beside it (`port`)."""
import argparse
import json
import os
import sys
import time

from benchlib.common import *  # noqa: F401,F403 (constants and helpers)
from benchlib.common import _StubEnv, _Timer  # noqa: F401


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
