"""Multi-GPU sharding of a batch of envs: one process per GPU, envs partitioned into contiguous index ranges.

Envs are fully independent (no cross-env data flow anywhere in the reference's mdp.py), so there is NO
collective on the step path.  The only communication is an optional sum-reduction of a handful of aggregate
metrics (returns, episodes, steps) once per reporting interval — `allreduce_metrics`, which runs over RCCL/xGMI
when the process group backend is "nccl" (and over gloo in the CPU tests).  The Philox streams of
`oc_rollout_random` are keyed by the GLOBAL env index (`env_offset`), so a sharded run reproduces exactly the
slice of the unsharded run it owns.
"""
import os


def shard_range(n_global, rank, world_size):
    """Contiguous [start, stop) of global env indices owned by `rank`; sizes differ by at most one."""
    if not 0 <= rank < world_size:
        raise ValueError("rank %d out of range for world size %d" % (rank, world_size))
    base, rem = divmod(int(n_global), int(world_size))
    start = rank * base + min(rank, rem)
    return start, start + base + (1 if rank < rem else 0)


def dist_env():
    """(rank, local_rank, world_size) from the torch.distributed.run environment (defaults: single process)."""
    return (int(os.environ.get("RANK", "0")), int(os.environ.get("LOCAL_RANK", "0")),
            int(os.environ.get("WORLD_SIZE", "1")))


def init_process_group(backend=None):
    """Initialise torch.distributed from the launcher's env (MASTER_ADDR/PORT, RANK, WORLD_SIZE)."""
    import torch
    import torch.distributed as dist

    rank, local_rank, world = dist_env()
    if world == 1 and not os.environ.get("OC_FORCE_DIST"):  # OC_FORCE_DIST: exercise the RCCL path with one rank
        return rank, local_rank, world
    if backend is None:
        backend = "nccl" if torch.cuda.is_available() else "gloo"
    if backend == "nccl":
        torch.cuda.set_device(local_rank)
    if not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group(backend=backend, rank=rank, world_size=world)
    return rank, local_rank, world


def _live():
    import torch.distributed as dist

    return dist.is_available() and dist.is_initialized() and (dist.get_world_size() > 1 or bool(os.environ.get("OC_FORCE_DIST")))


def _all_reduce(t, op):
    """all_reduce in place; a device tensor under a gloo group (CPU tests, one-GPU rehearsals) goes through a host copy."""
    import torch.distributed as dist

    if t.is_cuda and dist.get_backend() == "gloo":
        h = t.detach().cpu()
        dist.all_reduce(h, op=op)
        t.copy_(h)
    else:
        dist.all_reduce(t, op=op)
    return t


def allreduce_metrics(t):
    """In-place SUM of a small metrics tensor over all ranks (no-op for a single process)."""
    import torch.distributed as dist

    if _live():
        _all_reduce(t, dist.ReduceOp.SUM)
    return t


def allreduce_max(t):
    import torch.distributed as dist

    if _live():
        _all_reduce(t, dist.ReduceOp.MAX)
    return t


def barrier():
    import torch.distributed as dist

    if _live():
        dist.barrier()
