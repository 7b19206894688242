"""GPU parity on BASELINE.json's configs exactly as `bench.py --config N` runs them, at config scale (65 536 envs per GPU):
   configs[2]  asymmetric_advantages, one fused rollout step + lossless u8 encoding per iteration
   configs[3]  the five canonical layouts padded to 9x5, env e -> layout e % 5
   configs[4]  the 4 096 terrains of the reference's own LayoutGenerator (layout_generator.py:110-160, recorded with
               np.random.seed(0)), env e -> terrain e % 4096: per-env divergent terrain from a table in HBM
Checked against the C oracle (pinned to the reference by tests/test_oracle_golden.py): packed state, rewards, flags
and the u8 observation, bit for bit, across an episode boundary, for every rollout kernel family."""
import os

import numpy as np
import pytest

from helpers import CANONICAL_5, select_kernel

pytestmark = pytest.mark.gpu

torch = pytest.importorskip("torch")
N = 65536
KERNELS = ["default", "lane_pair", "predicate_interact"]


@pytest.fixture(scope="module")
def gpu():
    if not torch.cuda.is_available():
        pytest.fail("gpu-marked test run without a GPU")
    from overcooked_ai_amd import _lib

    _lib.load()
    return torch.device("cuda:0")


def _oracle(specs):
    from oracle import oracle as O

    O.set_threads(min(16, len(os.sched_getaffinity(0))))  # the envs are independent
    return O.Oracle([O.mdp_from_layout_dict(s.to_layout_dict()) for s in specs])


def _env(layouts, gpu, **kw):
    from overcooked_ai_amd.vec_env import VecOvercookedEnv

    return VecOvercookedEnv(layouts, N, device=gpu, auto_reset=True, **kw)


def test_config2_asymmetric_advantages_step_plus_encoding(gpu):
    """bench.py --config 3: rollout of ONE step, then oc_encode_lossless, per iteration; horizon 400."""
    from overcooked_ai_amd.layouts import spec_from_name

    spec = spec_from_name("asymmetric_advantages")
    orc = _oracle([spec])
    env = _env(spec, gpu, horizon=400, seed=0)
    st = orc.reset(orc.new_state(N))
    rew = torch.zeros((1, N, 4), dtype=torch.float32, device=gpu)
    fl = torch.zeros((1, N), dtype=torch.uint8, device=gpu)
    obs = torch.empty((N, 2, env.width, env.height, 26), dtype=torch.uint8, device=gpu)
    # jump close to the horizon so that the boundary is crossed: timestep 380 for every env, then 40 iterations
    st[0, :, 6] = 380 & 0xFF
    st[0, :, 7] = 380 >> 8
    env.set_packed_state(st)
    for it in range(40):
        env.rollout_random(1, rew, fl)
        env.encode_lossless(torch.uint8, out=obs)
        rew_o, fl_o = orc.rollout_random(st, 1, horizon=400, options=1, seed=0, t0=it)
        assert np.array_equal(fl.cpu().numpy(), fl_o) and np.array_equal(rew.cpu().numpy(), rew_o), it
        if it % 8 == 7 or it in (19, 20):  # the encodings around the restart (step 20) and a sample elsewhere
            assert np.array_equal(env.get_packed_state(), st), it
            assert np.array_equal(obs.cpu().numpy().astype(np.int32), orc.encode_lossless(st, horizon=400)), it
    assert (fl_o >= 0).all() and np.array_equal(env.get_packed_state(), st)


def test_config2_one_kernel_trajectory(gpu):
    """bench.py --config 3 as it runs now: oc_rollout_encode (k_rollout_encode) at 65 536 asymmetric_advantages envs —
    40 steps across the horizon in ONE launch, observations of all steps in a trajectory buffer; rewards, flags, final
    state and sampled observations against the C oracle."""
    from overcooked_ai_amd.layouts import spec_from_name

    spec = spec_from_name("asymmetric_advantages")
    orc = _oracle([spec])
    env = _env(spec, gpu, horizon=400, seed=0)
    st = orc.reset(orc.new_state(N))
    st[0, :, 6] = 380 & 0xFF
    st[0, :, 7] = 380 >> 8
    env.set_packed_state(st)
    K = 40
    rew = torch.zeros((K, N, 4), dtype=torch.float32, device=gpu)
    fl = torch.zeros((K, N), dtype=torch.uint8, device=gpu)
    obs = torch.empty((K, N, 2, env.width, env.height, 26), dtype=torch.uint8, device=gpu)
    env.rollout_encode(K, obs, rew, fl)
    rew_h, fl_h = rew.cpu().numpy(), fl.cpu().numpy()
    for it in range(K):
        rew_o, fl_o = orc.rollout_random(st, 1, horizon=400, options=1, seed=0, t0=it)
        assert np.array_equal(fl_h[it], fl_o[0]) and np.array_equal(rew_h[it], rew_o[0]), it
        if it % 8 == 7 or it in (19, 20):
            assert np.array_equal(obs[it].cpu().numpy().astype(np.int32), orc.encode_lossless(st, horizon=400)), it
    assert (fl_h[19] & 4).all() and np.array_equal(env.get_packed_state(), st)


@pytest.mark.parametrize("kernel", KERNELS)
def test_config3_five_layout_mix(kernel, gpu):
    """bench.py --config 4: 65 536 envs, env e -> layout e % 5 of the canonical five padded to 9x5."""
    from overcooked_ai_amd.layouts import LayoutTable, spec_from_name

    table = LayoutTable([spec_from_name(nm) for nm in CANONICAL_5], pad_to=(9, 5))
    lid = (np.arange(N) % 5).astype(np.uint16)
    orc = _oracle(table.specs)
    env = _env(table, gpu, horizon=120, seed=0, layout_id=lid)
    select_kernel(env, kernel)
    st = orc.reset(orc.new_state(N), layout_id=lid)
    ep_o = np.zeros((N, 4), np.float32)
    T = 160  # crosses the horizon at step 120
    rew = torch.zeros((T, N, 4), dtype=torch.float32, device=gpu)
    fl = torch.zeros((T, N), dtype=torch.uint8, device=gpu)
    env.rollout_random(T, rew, fl)
    rew_o, fl_o = orc.rollout_random(st, T, horizon=120, options=1, seed=0, layout_id=lid, ep_returns=ep_o)
    assert np.array_equal(env.get_packed_state(), st)
    assert np.array_equal(fl.cpu().numpy(), fl_o) and (fl_o[119] == 5).all()
    assert np.array_equal(rew.cpu().numpy(), rew_o) and np.array_equal(env.ep_returns.cpu().numpy(), ep_o)
    enc = env.encode_lossless(torch.uint8).cpu().numpy().astype(np.int32)
    assert np.array_equal(enc, orc.encode_lossless(st, horizon=120, layout_id=lid))


@pytest.mark.parametrize("kernel", KERNELS)
def test_config4_4096_generated_terrains(kernel, gpu):
    """bench.py --config 5: the reference LayoutGenerator's 4 096 9x5 terrains, env e -> terrain e % 4096 (table in HBM)."""
    from overcooked_ai_amd.layout_gen import reference_generated_layouts
    from overcooked_ai_amd.layouts import LayoutTable

    K = 4096
    table = LayoutTable(reference_generated_layouts(K))
    assert len(table) == K and (table.width, table.height) == (9, 5)
    lid = (np.arange(N) % K).astype(np.uint16)
    orc = _oracle(table.specs)
    env = _env(table, gpu, horizon=100, seed=0, layout_id=lid)
    select_kernel(env, kernel)
    st = orc.reset(orc.new_state(N), layout_id=lid)
    ep_o = np.zeros((N, 4), np.float32)
    T = 150  # crosses the horizon at step 100
    rew = torch.zeros((T, N, 4), dtype=torch.float32, device=gpu)
    fl = torch.zeros((T, N), dtype=torch.uint8, device=gpu)
    env.rollout_random(T, rew, fl)
    rew_o, fl_o = orc.rollout_random(st, T, horizon=100, options=1, seed=0, layout_id=lid, ep_returns=ep_o)
    assert np.array_equal(env.get_packed_state(), st)
    assert np.array_equal(fl.cpu().numpy(), fl_o) and (fl_o[99] == 5).all() and fl_o[:99].sum() == 0
    assert np.array_equal(rew.cpu().numpy(), rew_o) and np.array_equal(env.ep_returns.cpu().numpy(), ep_o)
    assert np.abs(rew_o).sum() > 0  # something was potted / picked up somewhere in 9.8 M env-steps
    if kernel == "default":
        enc = env.encode_lossless(torch.uint8).cpu().numpy().astype(np.int32)
        assert np.array_equal(enc, orc.encode_lossless(st, horizon=100, layout_id=lid))
        # the step API (k_step3) on the same table
        acts = np.random.default_rng(5).integers(0, 6, size=(N, 2)).astype(np.uint8)
        r, f = env.step(torch.from_numpy(acts).to(gpu))
        st2, r_o, f_o = orc.step(st, acts, horizon=100, options=1, layout_id=lid)
        assert np.array_equal(env.get_packed_state(), st2) and np.array_equal(r.cpu().numpy(), r_o)
        assert np.array_equal(f.cpu().numpy(), f_o)


def test_big_batch_variant_of_the_rollout_kernel(gpu):
    """More than ~1.5 wavefronts per SIMD (here 131 072 cramped_room envs) selects k_rollout4's lean instance (no
    one-step-ahead cell reads, cooking starts in the rare branch): same results as the oracle across a restart."""
    from overcooked_ai_amd.layouts import spec_from_name
    from overcooked_ai_amd.vec_env import VecOvercookedEnv

    n = 131072
    spec = spec_from_name("cramped_room")
    orc = _oracle([spec])
    env = VecOvercookedEnv(spec, n, device=gpu, auto_reset=True, horizon=70, seed=9)
    st = orc.reset(orc.new_state(n))
    ep_o = np.zeros((n, 4), np.float32)
    T = 100
    rew = torch.zeros((T, n, 4), dtype=torch.float32, device=gpu)
    fl = torch.zeros((T, n), dtype=torch.uint8, device=gpu)
    env.rollout_random(T, rew, fl)
    rew_o, fl_o = orc.rollout_random(st, T, horizon=70, options=1, seed=9, ep_returns=ep_o)
    assert np.array_equal(env.get_packed_state(), st) and np.array_equal(fl.cpu().numpy(), fl_o)
    assert np.array_equal(rew.cpu().numpy(), rew_o) and np.array_equal(env.ep_returns.cpu().numpy(), ep_o)
    assert (fl_o[69] == 5).all() and np.abs(rew_o).sum() > 0
