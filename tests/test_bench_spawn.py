"""bench.py's multi-rank plumbing on CPU: `python bench.py --gpus 2` without a launcher spawns its own ranks
(VERDICT r1 #2); `--stub` swaps the GPU env for a stand-in that steps the C oracle and RCCL for gloo, so the rank spawning, the per-rank shard maps, the parity check and the reductions are exercised.  Nothing here is a measurement."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def _run(cmd, env=None):
    e = dict(os.environ)
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT"):
        e.pop(k, None)
    e.update(env or {})
    p = subprocess.run(cmd, cwd=ROOT, env=e, capture_output=True, text=True, timeout=300)
    assert p.returncode == 0, p.stderr[-2000:]
    lines = [l for l in p.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, p.stdout  # exactly ONE JSON line, from rank 0
    return json.loads(lines[0])


def _check_two_rank_line(out, envs, fuse, lps):
    assert out["n_gpus"] == 2 and out["data"] == "stub" and out["steps"] == 3 and out["warmup"] == 1
    # exactly K bench steps of `lps` launches of `fuse` transitions are timed; ms_per_step is per bench step
    assert out["timed_launches"] == 3 * lps and out["timed_transitions_per_env"] == 3 * lps * fuse
    assert out["config"]["launches_per_step"] == lps and out["config"]["fused_transitions_per_launch"] == fuse
    assert abs(out["ms_per_step"] * out["steps"] - out["timed_region_s"] * 1e3) < 1e-6 * out["timed_region_s"] * 1e3
    assert len(out["ms_per_step_each"]) == 3 and all(x > 0 for x in out["ms_per_step_each"])
    assert len(out["ms_per_step_by_rank"]) == 2 and all(x > 0 for x in out["ms_per_step_by_rank"])
    assert out["aggregate"]["reduced_over"] == "gloo all-reduce"
    assert abs(out["value"] - 2 * envs * out["timed_transitions_per_env"] / out["timed_region_s"]) < 1e-6 * out["value"]
    # every rank replayed a launch of its own shard (global env offset rank * envs) against the oracle
    pc = out["parity_check"]
    assert pc["mismatches"] == 0 and pc["mismatches_by_rank"] == [0, 0] and pc["envs"] == 2 * envs
    assert pc["envs_per_rank"] == envs and pc["steps"] == min(fuse, 1200) and len(pc["seconds_by_rank"]) == 2
    assert out["roofline"]["traffic"] is None and out["roofline"]["traffic_source"]["how"] == "not collected"


def test_self_spawn_two_ranks_gloo():
    import bench

    out = _run([sys.executable, "bench.py", "--gpus", "2", "--steps", "3", "--warmup", "1", "--stub", "--envs", "64",
                "--launches-per-step", "2", "--fuse", "800"])
    _check_two_rank_line(out, 64, 800, 2)
    assert bench.LAUNCHES_PER_STEP * bench.DEFAULT_FUSE == 1_600_000  # the default bench step (module docstring)
    # the stub steps the oracle: rank 1 owns global envs 64..127, so the all-reduced returns differ from 2 x rank 0's
    assert out["aggregate"]["sparse_return_last_launch"] >= 0 and out["aggregate"]["shaped_return_last_launch"] > 0
    assert out["config"]["baseline_config"] == 2 and "cramped_room" in out["config"]["workload"]


def test_self_spawn_other_baseline_configs_gloo():
    """BASELINE configs[3] / configs[4] as the 8-GPU runs shard them (--config 4: env e -> layout e % 5;
    --config 5 --envs 131072: env e -> terrain e % 4096, here with a small batch) — per-rank layout ids follow the
    GLOBAL env index, and the per-rank parity_check and ms_per_step_by_rank are in the line."""
    out = _run([sys.executable, "bench.py", "--gpus", "2", "--steps", "3", "--warmup", "1", "--stub", "--envs", "35",
                "--config", "4", "--fuse", "600", "--launches-per-step", "1"])
    _check_two_rank_line(out, 35, 600, 1)
    assert out["config"]["baseline_config"] == 4 and "5 canonical layouts" in out["config"]["workload"]
    out = _run([sys.executable, "bench.py", "--gpus", "2", "--steps", "3", "--warmup", "1", "--stub", "--envs", "96",
                "--config", "5", "--fuse", "500", "--launches-per-step", "3"])
    _check_two_rank_line(out, 96, 500, 3)
    assert out["config"]["baseline_config"] == 5 and "4096 LayoutGenerator" in out["config"]["workload"]


def test_workload_layout_ids_follow_the_global_env_index():
    """Rank r's slice of the env -> layout map equals the unsharded map's slice (what makes a shard reproduce its slice
    of the single-GPU run)."""
    import argparse

    import numpy as np

    import bench

    for config, envs in ((4, 65536), (5, 131072)):
        args = argparse.Namespace(config=config, envs=envs, layout="cramped_room")
        k = 5 if config == 4 else 4096
        for rank in (0, 3, 7):
            wl = bench.make_workload(args, rank)
            assert np.array_equal(wl["lid"], (np.arange(rank * envs, (rank + 1) * envs) % k).astype(np.uint16))
            assert len(wl["specs"]) == k


def test_launcher_env_is_respected():
    """Under torch.distributed.run (WORLD_SIZE set by the launcher) bench.py must NOT spawn again."""
    out = _run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr",
                "127.0.0.1", "--master-port", "29631", "bench.py", "--gpus", "2", "--steps", "3", "--warmup", "1",
                "--stub", "--envs", "64", "--launches-per-step", "1", "--fuse", "400"])
    assert out["n_gpus"] == 2 and len(out["ms_per_step_by_rank"]) == 2


def test_eight_stub_ranks_rehearse_the_scaling_run():
    """The driver's N = 8 command on CPU: eight self-spawned stub ranks over gloo for the headline
    config and for BASELINE configs[3] / [4] — per-rank layout-id slices of the GLOBAL env index, max-over-ranks timing, a
    parity check on every rank, ONE JSON line.  A rehearsal of the plumbing, never a measurement: no 8-GPU run stands behind
    the multi-GPU path in this repo's rounds."""
    for cfg, envs, marker in ((2, 16, "cramped_room"), (4, 15, "5 canonical layouts"), (5, 24, "4096 LayoutGenerator")):
        out = _run([sys.executable, "bench.py", "--gpus", "8", "--steps", "2", "--warmup", "1", "--stub", "--envs", str(envs),
                    "--config", str(cfg), "--fuse", "400", "--launches-per-step", "1"])
        assert out["n_gpus"] == 8 and out["data"] == "stub" and out["scaling"] == "weak" and marker in out["config"]["workload"]
        assert len(out["ms_per_step_by_rank"]) == 8 and all(x > 0 for x in out["ms_per_step_by_rank"])
        # the timed region is the slowest rank's: ms_per_step x steps == timed_region_s >= every rank's own time
        assert abs(out["ms_per_step"] * out["steps"] - out["timed_region_s"] * 1e3) < 1e-6 * out["timed_region_s"] * 1e3
        assert max(out["ms_per_step_by_rank"]) <= out["ms_per_step"] * (1 + 1e-6)
        assert abs(out["value"] - 8 * envs * out["timed_transitions_per_env"] / out["timed_region_s"]) < 1e-6 * out["value"]
        pc = out["parity_check"]
        assert pc["mismatches"] == 0 and pc["mismatches_by_rank"] == [0] * 8 and pc["envs"] == 8 * envs
        assert out["aggregate"]["reduced_over"] == "gloo all-reduce"
