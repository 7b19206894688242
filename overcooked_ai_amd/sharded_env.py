"""ShardedVecOvercookedEnv — one batch of envs partitioned over the GPUs of a node as independent shards.

north_star: "env instances partition across the 8 GPUs of one node as independent shards (no collectives on the hot
path, an optional RCCL reduce over xGMI only for aggregate-return metrics)".  The reference's analogue is the fan-out of
30 single-env rollout workers (src/human_aware_rl/ppo/ppo_rllib_client.py:97-117); here a shard is a contiguous range of
global env indices resident in one GPU's HBM (`sharding.shard_range`), stepped by that GPU's own kernels on its own HIP
stream.  Envs are independent (no cross-env data flow anywhere in mdp.py) and every random stream — the Philox actions
of `rollout_random`, drawn start states, per-episode layout re-draws — is keyed by the GLOBAL env index
(`env_offset`), so the shards together reproduce the unsharded batch bit for bit, whatever the number of shards.

Two ways to run it, same class:
  * single process, several devices (default): one `VecOvercookedEnv` + one stream per device; every call fans out to
    all shards without a host synchronisation in between (the launches of different GPUs overlap); `aggregate()` sums
    the per-shard metric vectors on the host side of the first device.
  * one process per GPU under `torch.distributed` (`from_process_group`): this process owns only its rank's shard;
    `aggregate()` all-reduces the metric vector (RCCL over xGMI when the backend is nccl).

No scaling curve has been measured on hardware with it in this repo's rounds (no multi-GPU lease): the one-GPU parity
test (two or three shards on cuda:0 == the unsharded batch) and the CPU gloo tests are what stands behind it.
"""
import numpy as np
import torch

from . import sharding
from .vec_env import VecOvercookedEnv, as_layout_table


def shard_plan(n_global, n_devices, rank=0, world=1):
    """[(start, stop)] of the shards a process with `n_devices` local devices owns as rank `rank` of `world` processes:
    the global env range is cut into world * n_devices contiguous parts (sizes differ by at most one) and this process
    takes parts rank * n_devices .. rank * n_devices + n_devices - 1.  Pure arithmetic (CPU-testable)."""
    parts = int(n_devices) * int(world)
    return [sharding.shard_range(n_global, int(rank) * int(n_devices) + i, parts) for i in range(int(n_devices))]


class _Shard:
    __slots__ = ("env", "stream", "start", "stop", "device")

    def __init__(self, env, stream, start, stop):
        self.env, self.stream, self.start, self.stop, self.device = env, stream, start, stop, env.device


class ShardedVecOvercookedEnv:
    def __init__(self, layouts, n_global, devices=None, layout_id=None, env_offset=0, pad_to=None, ranks=None, **env_kw):
        """layouts / layout_id / env_kw as for VecOvercookedEnv (layout_id covers all n_global envs).
        devices: one entry per shard (default: every visible GPU once); a device may appear more than once (several
        shards on one GPU: what the one-GPU parity test does).  ranks: (rank, world) restricts this object to the shard
        that rank owns of a world-wide partition (the multi-process form, see from_process_group)."""
        self.table = as_layout_table(layouts, pad_to)
        self.n_global = int(n_global)
        self.env_offset = int(env_offset)
        if devices is None:
            devices = ["cuda:%d" % i for i in range(max(1, torch.cuda.device_count()))]
        devices = [torch.device(d) for d in devices]
        if layout_id is not None:
            layout_id = np.asarray(layout_id)
            if layout_id.shape != (self.n_global,):
                raise ValueError("layout_id must cover all %d global envs" % self.n_global)
        self._rank, self._world = (0, 1) if ranks is None else (int(ranks[0]), int(ranks[1]))
        self.shards = []
        for dev, (start, stop) in zip(devices, shard_plan(self.n_global, len(devices), self._rank, self._world)):
            with torch.cuda.device(dev):
                stream = torch.cuda.Stream(device=dev)
                with torch.cuda.stream(stream):
                    env = VecOvercookedEnv(self.table, stop - start, device=dev, env_offset=self.env_offset + start,
                                           layout_id=None if layout_id is None else layout_id[start:stop], **env_kw)
            self.shards.append(_Shard(env, stream, start, stop))
        self.width, self.height, self.n_planes = self.table.width, self.table.height, self.table.n_planes
        self.horizon = self.shards[0].env.horizon

    @classmethod
    def from_process_group(cls, layouts, n_global, device=None, **kw):
        """One process per GPU (torch.distributed.run): this process owns the shard of its RANK on cuda:LOCAL_RANK."""
        rank, local_rank, world = sharding.dist_env()
        return cls(layouts, n_global, devices=[device or "cuda:%d" % local_rank], ranks=(rank, world), **kw)

    # ------------------------------------------------------------------ fan-out
    @property
    def n_local(self):
        """Envs this object owns (all of them in the single-process form)."""
        return sum(s.stop - s.start for s in self.shards)

    def ranges(self):
        return [(s.start, s.stop) for s in self.shards]

    def _each(self, fn):
        """fn(shard, index) on every shard, on the shard's stream, without host synchronisation in between.  Each shard
        stream first waits (on the device, an event) for what the caller has enqueued on ITS current stream of that device —
        buffers it has just filled or zeroed, action tensors it has just produced — so a call is ordered after the caller's
        earlier work exactly as a call on one stream would be."""
        out = []
        for i, s in enumerate(self.shards):
            with torch.cuda.device(s.device):
                s.stream.wait_stream(torch.cuda.current_stream(s.device))
                with torch.cuda.stream(s.stream):
                    out.append(fn(s, i))
        return out

    def synchronize(self):
        """Host-side wait for every shard's stream."""
        for s in self.shards:
            s.stream.synchronize()

    def join(self):
        """Device-side: the caller's current stream of every shard's device waits for that shard's stream (no host
        synchronisation) — call it before consuming per-shard results on the caller's own stream."""
        for s in self.shards:
            with torch.cuda.device(s.device):
                torch.cuda.current_stream(s.device).wait_stream(s.stream)

    def alloc_outputs(self, n_steps, flags_tiled8=False):
        """Per-shard (rewards [n_steps, n, 4] f32, flags [n_steps, n] u8) buffers on the shards' devices; flags_tiled8: the
        flags as [n_steps // 8, n, 8] (OC_OPT_FLAGS_TILED8, see VecOvercookedEnv.rollout_random)."""
        def fl(s, i):
            n = s.stop - s.start
            return torch.zeros((n_steps // 8, n, 8) if flags_tiled8 else (n_steps, n), dtype=torch.uint8, device=s.device)
        return (self._each(lambda s, i: torch.zeros((n_steps, s.stop - s.start, 4), dtype=torch.float32, device=s.device)),
                self._each(fl))

    def alloc_observations(self, n_steps=None, dtype=torch.uint8):
        """Per-shard observation buffers: [n, 2, W, H, 26] (n_steps None: one observation per env) or the trajectory form
        [n_steps, n, 2, W, H, 26] of rollout_encode."""
        def ob(s, i):
            shape = (s.stop - s.start, 2, self.width, self.height, 26)
            return torch.empty(shape if n_steps is None else (int(n_steps),) + shape, dtype=dtype, device=s.device)
        return self._each(ob)

    def _split(self, t, per_env_dim=0):
        """A caller tensor over this object's envs -> per-shard tensors on the shards' devices (async copies)."""
        if isinstance(t, (list, tuple)):
            return list(t)
        base = self.shards[0].start
        out = []
        for s in self.shards:
            sl = [slice(None)] * t.dim()
            sl[per_env_dim] = slice(s.start - base, s.stop - base)
            with torch.cuda.device(s.device):
                if t.is_cuda:  # (ordered after whatever produced `t` on its device's current stream)
                    s.stream.wait_stream(torch.cuda.current_stream(t.device))
                with torch.cuda.stream(s.stream):
                    piece = t[tuple(sl)].to(s.device, non_blocking=True).contiguous()
                if piece.is_cuda:
                    # a view of the caller's tensor (same device, already contiguous: no copy was made), or a fresh copy made on
                    # the shard's stream: either way the caching allocator must not hand the memory to somebody else on the
                    # caller's stream while the shard's kernels still read it.  (Refilling `t` before join() / synchronize()
                    # is still the caller's race: keep inputs untouched until then.)
                    piece.record_stream(s.stream)
                out.append(piece)
        return out

    # ------------------------------------------------------------------ env API (lists hold one entry per shard)
    def reset(self, mask=None, **kw):
        masks = None if mask is None else self._split(mask)
        self._each(lambda s, i: s.env.reset(None if masks is None else masks[i], **kw))

    def rollout_random(self, n_steps, rewards_out=None, flags_out=None, events_out=None, flags_tiled8=False):
        """n_steps fused random-policy transitions on every shard (one oc_rollout_random launch per shard, all in flight
        together).  rewards_out / flags_out (/ events_out: int64 [n_steps, n] masks): per-shard lists (alloc_outputs), or
        None.  flags_tiled8: the OC_OPT_FLAGS_TILED8 flags layout on every shard (alloc_outputs(n_steps, flags_tiled8=True);
        ValueError from the first shard whose batch / launch shape no tiled instance serves)."""
        self._each(lambda s, i: s.env.rollout_random(n_steps, None if rewards_out is None else rewards_out[i],
                                                     None if flags_out is None else flags_out[i],
                                                     None if events_out is None else events_out[i], flags_tiled8=flags_tiled8))
        return rewards_out, flags_out

    def rollout_encode(self, n_steps, obs_out, rewards_out=None, flags_out=None, actions=None, dtype=torch.uint8):
        """n_steps transitions with the lossless observation after every step on every shard (VecOvercookedEnv.rollout_encode).
        obs_out: per-shard list (alloc_observations(n_steps) for the trajectory, alloc_observations() for the last observation
        only); actions: None (the random policy, the global Philox stream) or uint8 [n_steps, n_local, 2] / a per-shard list."""
        acts = None if actions is None else self._split(actions, per_env_dim=1)
        self._each(lambda s, i: s.env.rollout_encode(n_steps, obs_out[i], None if rewards_out is None else rewards_out[i],
                                                     None if flags_out is None else flags_out[i],
                                                     None if acts is None else acts[i], dtype=dtype))
        return obs_out, rewards_out, flags_out

    def step_encode(self, actions, dtype=torch.uint8, out=None):
        """step(actions) + the lossless observation of the new states in one C call per shard (VecOvercookedEnv.step_encode).
        Returns (rewards list, flags list, obs list)."""
        acts = self._split(actions)
        res = self._each(lambda s, i: s.env.step_encode(acts[i], dtype, out=None if out is None else out[i]))
        return [r for r, _, _ in res], [f for _, f, _ in res], [o for _, _, o in res]

    def event_stats(self, finished=False):
        """{event name: per-shard list of int32 [n, 2] tensors} — the per-episode event counters of every shard (construct
        with track_events=True); `gather(stats[name])` gives the [n_local, 2] host array."""
        per = self._each(lambda s, i: s.env.event_stats(finished))
        return {name: [d[name] for d in per] for name in per[0]}

    def step(self, actions):
        """actions: uint8 [n_local, 2] (any device / host; split and copied asynchronously) or a per-shard list.
        Returns (rewards list, flags list) — each shard's own reused buffers, valid after its stream's work is done."""
        acts = self._split(actions)
        res = self._each(lambda s, i: s.env.step(acts[i]))
        return [r for r, _ in res], [f for _, f in res]

    def step_many(self, actions, rewards_out, flags_out):
        """actions uint8 [K, n_local, 2] (or per-shard list) -> the alloc_outputs(K) buffers."""
        acts = self._split(actions, per_env_dim=1)
        self._each(lambda s, i: s.env.step_many(acts[i], rewards_out[i], flags_out[i]))
        return rewards_out, flags_out

    def encode_lossless(self, dtype=torch.uint8, out=None):
        """Per-shard [n, 2, W, H, 26] observations (mdp.py:2385)."""
        return self._each(lambda s, i: s.env.encode_lossless(dtype, out=None if out is None else out[i]))

    def featurize(self, **kw):
        return self._each(lambda s, i: s.env.featurize(**kw))

    def potential(self, gamma=0.99):
        return self._each(lambda s, i: s.env.potential(gamma))

    # ------------------------------------------------------------------ gathered views (host; synchronise)
    def get_packed_state(self):
        """[n_planes, n_local, 16] — equal to the unsharded batch's slice, plane for plane."""
        self.synchronize()
        return np.concatenate([s.env.get_packed_state() for s in self.shards], axis=1)

    def layout_ids(self):
        self.synchronize()
        return np.concatenate([s.env.layout_ids() for s in self.shards])

    def ep_returns(self):
        self.synchronize()
        return np.concatenate([s.env.ep_returns.cpu().numpy() for s in self.shards], axis=0)

    def gather(self, per_shard, dim=0):
        """Per-shard tensors -> one host numpy array along the env axis `dim`."""
        self.synchronize()
        return np.concatenate([t.cpu().numpy() for t in per_shard], axis=dim)

    # ------------------------------------------------------------------ the only communication: aggregate metrics
    def aggregate(self, rewards=None, flags=None):
        """Sum-reduced metrics over every env of every shard (and every rank): running episode returns
        (sparse0, sparse1, shaped0, shaped1), batched steps done, and — when the output buffers of the last launch are
        given — the rewards and finished episodes in them.  Each shard reduces on its own GPU (float64), the partial
        vectors are added on the host side (8 scalars per shard); with a live process group the result is all-reduced
        (RCCL over xGMI for backend nccl).  Never on the step path."""
        def partial(s, i):
            v = torch.zeros((8,), dtype=torch.float64, device=s.device)
            if s.env.ep_returns is not None:
                v[0:4] = s.env.ep_returns.sum(dim=0, dtype=torch.float64)
            if rewards is not None:
                v[4] = rewards[i][..., 0:2].sum(dtype=torch.float64)
                v[5] = rewards[i][..., 2:4].sum(dtype=torch.float64)
            if flags is not None:
                v[6] = (flags[i] & 1).sum(dtype=torch.float64)
            v[7] = float(s.env.steps_done) * (s.stop - s.start)
            return v
        parts = self._each(partial)
        self.synchronize()
        total = torch.stack([p.cpu() for p in parts]).sum(dim=0)
        if sharding._live():
            import torch.distributed as dist

            t = total.to(self.shards[0].device) if dist.get_backend() == "nccl" else total
            sharding.allreduce_metrics(t)
            total = t.cpu()
        names = ("ep_sparse_0", "ep_sparse_1", "ep_shaped_0", "ep_shaped_1", "sparse_in_buffers", "shaped_in_buffers",
                 "episodes_done_in_buffers", "env_steps")
        return dict(zip(names, (float(x) for x in total)))


class ShardedVecOvercookedMultiAgent:
    """VecOvercookedMultiAgent (the batched rllib.py:293-342 training env) over the same env-range shards: one
    VecOvercookedMultiAgent per device on its own stream, `step` fans the actions out and returns per-shard lists of
    (obs, shaped rewards, dones, infos) — device tensors that stay on their GPUs (a data-parallel learner consumes each
    shard's where it lives).  Drawn start states are keyed by the global env index, so the shards reproduce the unsharded
    batch."""

    def __init__(self, layouts, n_global, devices=None, layout_id=None, env_offset=0, pad_to=None, ranks=None, **ma_kw):
        from .multi_agent import VecOvercookedMultiAgent

        self.table = as_layout_table(layouts, pad_to)
        self.n_global, self.env_offset = int(n_global), int(env_offset)
        if devices is None:
            devices = ["cuda:%d" % i for i in range(max(1, torch.cuda.device_count()))]
        devices = [torch.device(d) for d in devices]
        if layout_id is not None:
            layout_id = np.asarray(layout_id)
            if layout_id.shape != (self.n_global,):
                raise ValueError("layout_id must cover all %d global envs" % self.n_global)
        rank, world = (0, 1) if ranks is None else (int(ranks[0]), int(ranks[1]))
        self.shards = []
        for dev, (start, stop) in zip(devices, shard_plan(self.n_global, len(devices), rank, world)):
            with torch.cuda.device(dev):
                stream = torch.cuda.Stream(device=dev)
                with torch.cuda.stream(stream):
                    ma = VecOvercookedMultiAgent(self.table, stop - start, device=dev, env_offset=self.env_offset + start,
                                                 layout_id=None if layout_id is None else layout_id[start:stop], **ma_kw)
            sh = _Shard(ma.venv, stream, start, stop)
            self.shards.append((sh, ma))
        self._fan = ShardedVecOvercookedEnv.__new__(ShardedVecOvercookedEnv)  # the fan-out helpers over the same shards
        self._fan.shards = [sh for sh, _ in self.shards]

    @classmethod
    def from_process_group(cls, layouts, n_global, device=None, **kw):
        rank, local_rank, world = sharding.dist_env()
        return cls(layouts, n_global, devices=[device or "cuda:%d" % local_rank], ranks=(rank, world), **kw)

    @property
    def agents(self):
        return [ma for _, ma in self.shards]

    def reset(self):
        """Per-shard observations of the first states."""
        return self._fan._each(lambda s, i: self.shards[i][1].reset())

    def step(self, actions):
        """actions uint8 [n_local, 2] (or a per-shard list) -> per-shard lists (obs, shaped rewards, dones, infos)."""
        acts = self._fan._split(actions)
        res = self._fan._each(lambda s, i: self.shards[i][1].step(acts[i]))
        return [r[0] for r in res], [r[1] for r in res], [r[2] for r in res], [r[3] for r in res]

    def set_reward_shaping_factor(self, factor):
        for _, ma in self.shards:
            ma.set_reward_shaping_factor(factor)

    def anneal_reward_shaping_factor(self, timesteps):
        for _, ma in self.shards:
            ma.anneal_reward_shaping_factor(timesteps)

    def synchronize(self):
        self._fan.synchronize()

    def join(self):
        self._fan.join()

    def gather(self, per_shard, dim=0):
        return self._fan.gather(per_shard, dim)

    def get_packed_state(self):
        return self._fan.get_packed_state()
