"""Multi-GPU path on CPU: world_size-2 gloo process group.  The envs shard as contiguous ranges with no
data-path collective; each rank steps its shard using its global env offset for the Philox streams, and only the
aggregate metrics are all-reduced.

What this does and does not show: there is no GPU here, so every rank steps the C ORACLE, not the HIP path — the test
covers the partition arithmetic, the env_offset / layout-id conventions and the collectives.  That a HIP shard equals its
slice of the unsharded batch is shown on one GPU (the sharding property checked in tests/test_gpu_parity.py's full-size rollout test and the
inner-rank launch shapes of tests/test_gpu_launch_shapes.py); a multi-GPU run has not been measured on hardware."""
import os
import socket
import sys

import numpy as np
import pytest

from conftest import ROOT
from overcooked_ai_amd.sharding import shard_range


def test_shard_range_partitions_exactly():
    for n in (0, 1, 7, 65536, 524288, 1000003):
        for w in (1, 2, 3, 8):
            spans = [shard_range(n, r, w) for r in range(w)]
            assert spans[0][0] == 0 and spans[-1][1] == n
            assert all(a[1] == b[0] for a, b in zip(spans, spans[1:]))
            sizes = [b - a for a, b in spans]
            assert max(sizes) - min(sizes) <= 1
    with pytest.raises(ValueError):
        shard_range(10, 2, 2)


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, n_global, steps, seed, out_dir):
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                      LOCAL_RANK=str(rank))
    import torch

    from oracle import oracle as O
    from overcooked_ai_amd import sharding
    from overcooked_ai_amd.layouts import spec_from_name

    r, lr, w = sharding.init_process_group(backend="gloo")
    assert (r, w) == (rank, world)
    a, b = sharding.shard_range(n_global, r, w)
    orc = O.Oracle(O.mdp_from_layout_dict(spec_from_name("cramped_room").to_layout_dict()))
    st = orc.reset(orc.new_state(b - a))
    ep = np.zeros((b - a, 4), np.float32)
    rew, fl = orc.rollout_random(st, steps, horizon=50, options=1, seed=seed, env_offset=a, ep_returns=ep)
    metrics = torch.tensor([rew[..., :2].sum(), rew[..., 2:].sum(), float((fl & 1).sum()), float((b - a) * steps)],
                           dtype=torch.float64)
    sharding.allreduce_metrics(metrics)
    tmax = torch.tensor([float(rank)], dtype=torch.float64)
    sharding.allreduce_max(tmax)
    sharding.barrier()
    np.savez(os.path.join(out_dir, "rank%d.npz" % rank), state=st, metrics=metrics.numpy(), span=np.array([a, b]),
             tmax=tmax.numpy())
    torch.distributed.destroy_process_group()


def test_two_rank_gloo_shards_reproduce_the_unsharded_batch(tmp_path):
    import torch.multiprocessing as mp

    from oracle import oracle as O
    from overcooked_ai_amd.layouts import spec_from_name

    n_global, steps, seed, world = 1001, 120, 42, 2
    port = _free_port()
    ctx = mp.get_context("spawn")
    procs = [ctx.Process(target=_worker, args=(r, world, port, n_global, steps, seed, str(tmp_path))) for r in range(world)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(180)
        assert p.exitcode == 0
    orc = O.Oracle(O.mdp_from_layout_dict(spec_from_name("cramped_room").to_layout_dict()))
    st = orc.reset(orc.new_state(n_global))
    rew, fl = orc.rollout_random(st, steps, horizon=50, options=1, seed=seed)
    expect = np.array([rew[..., :2].sum(), rew[..., 2:].sum(), float((fl & 1).sum()), float(n_global * steps)])
    for r in range(world):
        d = np.load(os.path.join(str(tmp_path), "rank%d.npz" % r))
        a, b = d["span"]
        assert np.array_equal(d["state"], st[:, a:b])          # shard == slice of the unsharded run
        assert np.allclose(d["metrics"], expect)                # all-reduced aggregate metrics
        assert d["tmax"][0] == world - 1


def test_shard_plan_of_the_sharded_env_covers_the_batch_once():
    """ShardedVecOvercookedEnv's partition arithmetic (the class itself needs GPUs: tests/test_gpu_sharded_env.py): for any
    batch size, device count and world size the shards of all ranks tile [0, n_global) exactly once, in order."""
    torch = __import__("pytest").importorskip("torch")  # (the module imports torch at the top)
    from overcooked_ai_amd.sharded_env import shard_plan

    for n_global in (1, 7, 4099, 65536 * 8, 1048576):
        for n_dev in (1, 2, 8):
            for world in (1, 2, 3):
                got = [r for rank in range(world) for r in shard_plan(n_global, n_dev, rank, world)]
                assert got[0][0] == 0 and got[-1][1] == n_global
                assert all(a[1] == b[0] for a, b in zip(got, got[1:]))
                sizes = [b - a for a, b in got]
                assert max(sizes) - min(sizes) <= 1 and min(sizes) >= 0
    assert shard_plan(524288, 8) == [(i * 65536, (i + 1) * 65536) for i in range(8)]   # BASELINE configs[3]
    assert shard_plan(1048576, 1, 3, 8) == [(3 * 131072, 4 * 131072)]                   # configs[4], rank 3 of 8
