"""ShardedVecOvercookedEnv on ONE device: two / three shards on cuda:0 reproduce the unsharded batch bit for bit
(states, rewards, flags, episode returns, layout ids with per-episode re-draws), whatever the number of shards —
every random stream is keyed by the global env index.  The multi-process form is covered on CPU over gloo
(tests/test_sharding_gloo.py); no multi-GPU run stands behind this class (no multi-GPU lease in this repo's rounds)."""
import numpy as np
import pytest

from helpers import CANONICAL_5

pytestmark = pytest.mark.gpu

torch = pytest.importorskip("torch")


@pytest.fixture(scope="module")
def gpu():
    if not torch.cuda.is_available():
        pytest.fail("gpu-marked test run without a GPU")
    from overcooked_ai_amd import _lib

    _lib.load()
    return torch.device("cuda:0")


def _cat(parts, dim):
    return torch.cat([p.cpu() for p in parts], dim=dim).numpy()


@pytest.mark.parametrize("n_shards", [2, 3])
def test_shards_on_one_device_equal_the_unsharded_batch(n_shards, gpu):
    from overcooked_ai_amd import ShardedVecOvercookedEnv, VecOvercookedEnv

    n, K = 5003, 170  # (not a multiple of the shard count, of the wavefront or of the workgroup)
    kw = dict(horizon=60, auto_reset=True, seed=42, random_start_pos=True, rnd_obj_prob_thresh=0.5)
    whole = VecOvercookedEnv("cramped_room", n, device=gpu, env_offset=1000, **kw)
    sh = ShardedVecOvercookedEnv("cramped_room", n, devices=[gpu] * n_shards, env_offset=1000, **kw)
    assert sh.n_local == n and sh.ranges()[0][0] == 0 and sh.ranges()[-1][1] == n
    assert np.array_equal(sh.get_packed_state(), whole.get_packed_state())  # drawn start states of the first episode
    rew = torch.zeros((K, n, 4), dtype=torch.float32, device=gpu)
    fl = torch.zeros((K, n), dtype=torch.uint8, device=gpu)
    whole.rollout_random(K, rew, fl)
    rews, fls = sh.alloc_outputs(K)
    sh.rollout_random(K, rews, fls)
    sh.synchronize()
    assert np.array_equal(sh.get_packed_state(), whole.get_packed_state())
    assert np.array_equal(_cat(rews, 1), rew.cpu().numpy()) and np.array_equal(_cat(fls, 1), fl.cpu().numpy())
    assert np.array_equal(sh.ep_returns(), whole.ep_returns.cpu().numpy())
    assert (fl.cpu().numpy() & 4).any()  # restarts from drawn states happened inside the launch
    # explicit actions: one tensor over all envs, split by the sharded env
    acts = torch.from_numpy(np.random.default_rng(1).integers(0, 6, size=(n, 2)).astype(np.uint8))
    r0, f0 = whole.step(acts.to(gpu))
    r1, f1 = sh.step(acts)
    sh.synchronize()
    assert np.array_equal(_cat(r1, 0), r0.cpu().numpy()) and np.array_equal(_cat(f1, 0), f0.cpu().numpy())
    assert np.array_equal(sh.get_packed_state(), whole.get_packed_state())
    # observations and the aggregate metrics
    assert np.array_equal(_cat(sh.encode_lossless(torch.uint8), 0), whole.encode_lossless(torch.uint8).cpu().numpy())
    agg = sh.aggregate(rews, fls)
    assert agg["ep_sparse_0"] == float(whole.ep_returns[:, 0].sum(dtype=torch.float64))
    assert agg["ep_shaped_1"] == float(whole.ep_returns[:, 3].sum(dtype=torch.float64))
    assert agg["sparse_in_buffers"] == float(rew[..., 0:2].sum(dtype=torch.float64))
    assert agg["episodes_done_in_buffers"] == float((fl & 1).sum()) and agg["env_steps"] == float(n * (K + 1))


def test_sharded_mixed_table_with_layout_redraws(gpu):
    """BASELINE configs[3]'s table with regen_mdp semantics: every new episode of an env runs on a layout drawn from
    the table; the drawn ids follow the global env index, so the shards' ids equal the unsharded batch's."""
    from overcooked_ai_amd import ShardedVecOvercookedEnv, VecOvercookedEnv
    from overcooked_ai_amd.layouts import LayoutTable, spec_from_name

    table = LayoutTable([spec_from_name(nm) for nm in CANONICAL_5], pad_to=(9, 5))
    n, K = 4099, 130
    lid = (np.arange(n) % 5).astype(np.uint16)
    kw = dict(horizon=50, auto_reset=True, seed=9, regen_layout=True)
    whole = VecOvercookedEnv(table, n, device=gpu, layout_id=lid, **kw)
    sh = ShardedVecOvercookedEnv(table, n, devices=[gpu, gpu], layout_id=lid, **kw)
    rew = torch.zeros((K, n, 4), dtype=torch.float32, device=gpu)
    fl = torch.zeros((K, n), dtype=torch.uint8, device=gpu)
    whole.rollout_random(K, rew, fl)
    rews = [torch.full((K, b - a, 4), 7.0, dtype=torch.float32, device=gpu) for a, b in sh.ranges()]  # caller-made buffers, filled
    fls = [torch.full((K, b - a), 9, dtype=torch.uint8, device=gpu) for a, b in sh.ranges()]          # on the caller's stream
    sh.rollout_random(K, rews, fls)
    assert np.array_equal(sh.layout_ids(), whole.layout_ids()) and not np.array_equal(whole.layout_ids(), lid)
    assert np.array_equal(sh.get_packed_state(), whole.get_packed_state())
    assert np.array_equal(_cat(rews, 1), rew.cpu().numpy()) and np.array_equal(_cat(fls, 1), fl.cpu().numpy())
    sh.reset()  # explicit reset: layouts drawn again, same epochs on both sides
    whole.reset()
    assert np.array_equal(sh.layout_ids(), whole.layout_ids())
    assert np.array_equal(sh.get_packed_state(), whole.get_packed_state())


def test_rank_restricted_form_owns_its_slice(gpu):
    """from_process_group's arithmetic without a launcher: (rank, world) = (1, 3) owns the middle third."""
    from overcooked_ai_amd import ShardedVecOvercookedEnv, VecOvercookedEnv
    from overcooked_ai_amd.sharding import shard_range

    n = 1000
    whole = VecOvercookedEnv("asymmetric_advantages", n, device=gpu, horizon=400, auto_reset=True, seed=3)
    whole.rollout_random(40)
    mine = ShardedVecOvercookedEnv("asymmetric_advantages", n, devices=[gpu], ranks=(1, 3), horizon=400, auto_reset=True, seed=3)
    a, b = shard_range(n, 1, 3)
    assert mine.ranges() == [(a, b)] and mine.n_local == b - a
    mine.rollout_random(40)
    assert np.array_equal(mine.get_packed_state(), whole.get_packed_state()[:, a:b])


def test_sharded_surface_tiled_flags_observations_events(gpu):
    """The rest of VecOvercookedEnv's batched surface through the shards: the OC_OPT_FLAGS_TILED8 flags layout
    (the mover / interact kernel on whole-workgroup shards), rollout_encode (trajectory of observations), step_encode, and
    the per-episode event counters — two shards on cuda:0 against the unsharded batch."""
    from overcooked_ai_amd import ShardedVecOvercookedEnv, VecOvercookedEnv
    from overcooked_ai_amd.layouts import LayoutTable, spec_from_name

    table = LayoutTable([spec_from_name(nm) for nm in CANONICAL_5], pad_to=(9, 5))
    n, K = 4096, 96
    lid = ((np.arange(n) * 3 + 1) % 5).astype(np.uint16)
    kw = dict(horizon=40, auto_reset=True, seed=21)
    whole = VecOvercookedEnv(table, n, device=gpu, layout_id=lid, **kw)
    sh = ShardedVecOvercookedEnv(table, n, devices=[gpu, gpu], layout_id=lid, **kw)
    rew = torch.zeros((K, n, 4), dtype=torch.float32, device=gpu)
    fl_t = torch.zeros((K // 8, n, 8), dtype=torch.uint8, device=gpu)
    whole.rollout_random(K, rew, fl_t, flags_tiled8=True)
    rews, fls = sh.alloc_outputs(K, flags_tiled8=True)
    sh.rollout_random(K, rews, fls, flags_tiled8=True)
    sh.synchronize()
    assert np.array_equal(_cat(rews, 1), rew.cpu().numpy()) and np.array_equal(_cat(fls, 1), fl_t.cpu().numpy())
    assert np.array_equal(sh.get_packed_state(), whole.get_packed_state())
    assert (VecOvercookedEnv.untile_flags(fl_t).cpu().numpy() & 4).any()
    # rollout_encode on one layout (k_rollout_encode's own conditions hold on each shard or on none: same results either way)
    kw = dict(horizon=30, auto_reset=True, seed=5)
    whole = VecOvercookedEnv("asymmetric_advantages", 1500, device=gpu, **kw)
    sh = ShardedVecOvercookedEnv("asymmetric_advantages", 1500, devices=[gpu, gpu, gpu], **kw)
    T = 40
    obs = torch.empty((T, 1500, 2, whole.width, whole.height, 26), dtype=torch.uint8, device=gpu)
    rew = torch.zeros((T, 1500, 4), dtype=torch.float32, device=gpu)
    fl = torch.zeros((T, 1500), dtype=torch.uint8, device=gpu)
    whole.rollout_encode(T, obs, rew, fl)
    obss, (rews, fls) = sh.alloc_observations(T), sh.alloc_outputs(T)
    sh.rollout_encode(T, obss, rews, fls)
    sh.synchronize()
    assert np.array_equal(_cat(obss, 1), obs.cpu().numpy()) and np.array_equal(_cat(rews, 1), rew.cpu().numpy())
    assert np.array_equal(_cat(fls, 1), fl.cpu().numpy()) and np.array_equal(sh.get_packed_state(), whole.get_packed_state())
    acts = torch.from_numpy(np.random.default_rng(2).integers(0, 6, size=(1500, 2)).astype(np.uint8)).to(gpu)
    r0, f0, o0 = whole.step_encode(acts)
    r1, f1, o1 = sh.step_encode(acts)
    sh.synchronize()
    assert np.array_equal(_cat(o1, 0), o0.cpu().numpy()) and np.array_equal(_cat(r1, 0), r0.cpu().numpy())
    assert np.array_equal(_cat(f1, 0), f0.cpu().numpy())
    # per-episode event counters
    kw = dict(horizon=50, auto_reset=True, seed=8, track_events=True)
    whole = VecOvercookedEnv("cramped_room", 999, device=gpu, **kw)
    sh = ShardedVecOvercookedEnv("cramped_room", 999, devices=[gpu, gpu], **kw)
    whole.rollout_random(120)
    sh.rollout_random(120)
    sh.synchronize()
    for finished in (False, True):
        a, b = whole.event_stats(finished), sh.event_stats(finished)
        assert set(a) == set(b)
        for name in a:
            assert np.array_equal(_cat(b[name], 0), a[name].cpu().numpy()), name
    assert sum(int(v.sum()) for v in whole.event_stats(True).values()) > 0


def test_sharded_multi_agent_equals_the_unsharded_training_env(gpu):
    """ShardedVecOvercookedMultiAgent (the batched rllib.py:293-342 env over env-range shards): observations, shaped rewards,
    dones and phi of three shards on cuda:0 == VecOvercookedMultiAgent on the whole batch, across episode ends with drawn
    start states."""
    from overcooked_ai_amd.multi_agent import VecOvercookedMultiAgent
    from overcooked_ai_amd.sharded_env import ShardedVecOvercookedMultiAgent

    n = 1000
    kw = dict(horizon=12, reward_shaping_factor=0.7, use_phi=True, obs_dtype=torch.uint8, random_start_pos=True,
              rnd_obj_prob_thresh=0.3, seed=13)
    whole = VecOvercookedMultiAgent("asymmetric_advantages", n, device=gpu, env_offset=500, **kw)
    sh = ShardedVecOvercookedMultiAgent("asymmetric_advantages", n, devices=[gpu, gpu, gpu], env_offset=500, **kw)
    o0, o1 = whole.reset(), sh.reset()
    sh.synchronize()
    assert np.array_equal(_cat(o1, 0), o0.cpu().numpy())
    rng = np.random.default_rng(6)
    for t in range(30):
        acts = torch.from_numpy(rng.integers(0, 6, size=(n, 2)).astype(np.uint8)).to(gpu)
        ob0, r0, d0, i0 = whole.step(acts)
        ob1, r1, d1, i1 = sh.step(acts)
        sh.synchronize()
        assert np.array_equal(_cat(ob1, 0), ob0.cpu().numpy()), t
        assert np.array_equal(_cat(r1, 0), r0.cpu().numpy()) and np.array_equal(_cat(d1, 0), d0.cpu().numpy()), t
        assert np.array_equal(_cat([i["phi_s_prime"] for i in i1], 0), i0["phi_s_prime"].cpu().numpy()), t
        assert np.array_equal(_cat([i["ep_returns"] for i in i1], 0), i0["ep_returns"].cpu().numpy()), t
    assert np.array_equal(sh.get_packed_state(), whole.venv.get_packed_state())
