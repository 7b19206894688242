"""bench.py's multi-rank plumbing on CPU: `python bench.py --gpus 2` without a launcher spawns its own ranks
(VERDICT r1 #2); `--stub` swaps the GPU env for a no-op stand-in and RCCL for gloo so only the rank spawning, the
repeat planning and the reductions are exercised.  Nothing here is a measurement."""
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


def test_plan_repeats():
    import bench

    # 20-step regions, 400-step launches: repeats must be a multiple of 20 and cover >= 0.25 s at 0.5 us/step
    r = bench.plan_repeats(20, 400, 0.0005, 0.25)
    assert r % 20 == 0 and r * 20 * 0.0005e-3 >= 0.25 * 1.05 and (r - 20) * 20 * 0.0005e-3 < 0.25 * 1.05
    assert bench.plan_repeats(20000, 400, 0.0005, 0.25) == 27  # 5 % margin over the warm-up estimate
    assert bench.plan_repeats(1000, 400, 1.0, 0.25) == 2  # lcm(1000, 400) = 2000 steps = 2 repeats
    assert bench.plan_repeats(400, 400, 100.0, 0.25) == 1


def test_self_spawn_two_ranks_gloo():
    import bench

    out = _run([sys.executable, "bench.py", "--gpus", "2", "--steps", "20", "--warmup", "5", "--stub", "--envs", "64",
                "--min-seconds", "0.02"])
    assert out["n_gpus"] == 2 and out["data"] == "stub" and out["steps"] == 20 and out["warmup"] == 5
    assert out["timed_steps"] == 20 * out["repeats"] and out["timed_steps"] % bench.DEFAULT_FUSE == 0
    assert len(out["ms_per_step_by_rank"]) == 2 and all(x > 0 for x in out["ms_per_step_by_rank"])
    # the stub writes 1/16 into every reward slot: the all-reduced sums prove both ranks took part
    assert out["aggregate"]["sparse_return_last_launch"] == 2 * bench.DEFAULT_FUSE * 64 * 2 / 16
    assert out["aggregate"]["reduced_over"] == "gloo all-reduce"
    assert abs(out["value"] - 2 * 64 * out["timed_steps"] / out["timed_region_s"]) < 1e-6 * out["value"]


def test_launcher_env_is_respected():
    """Under torch.distributed.run (WORLD_SIZE set by the launcher) bench.py must NOT spawn again."""
    out = _run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr",
                "127.0.0.1", "--master-port", "29631", "bench.py", "--gpus", "2", "--steps", "20", "--warmup", "5",
                "--stub", "--envs", "64", "--min-seconds", "0.02"])
    assert out["n_gpus"] == 2 and len(out["ms_per_step_by_rank"]) == 2
