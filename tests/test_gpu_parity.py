"""Parity of the HIP kernels (through the C-ABI of liboc_amd.so) against (a) fixtures generated from the real
reference and (b) the C oracle on seeded random inputs.  Integer state: bit-exact; rewards: |diff| <= 1e-6."""
import os

import numpy as np
import pytest

from conftest import GOLDEN
from helpers import CANONICAL_5, random_packed_states, select_kernel

pytestmark = pytest.mark.gpu

torch = pytest.importorskip("torch")

TRANSITION_CONFIGS = [
    "cramped_room", "asymmetric_advantages", "coordination_ring", "forced_coordination", "counter_circuit",
    "mdp_test", "cramped_room_old_dynamics", "bonus_order_test", "cramped_room_tomato", "cramped_room_single",
    "cramped_room_padded_9x5",
]
ROLLOUT_CONFIGS = ["cramped_room", "asymmetric_advantages", "counter_circuit", "mdp_test", "cramped_room_old_dynamics"]


@pytest.fixture(scope="module")
def gpu():
    if not torch.cuda.is_available():
        pytest.fail("gpu-marked test run without a GPU")
    from overcooked_ai_amd import _lib

    _lib.load()  # fail loudly if the HIP extension is missing
    return torch.device("cuda:0")


def make_env(layouts, n, gpu, **kw):
    from overcooked_ai_amd.vec_env import VecOvercookedEnv

    return VecOvercookedEnv(layouts, n, device=gpu, **kw)


def oracle_for(specs):
    from oracle import oracle as O

    if not isinstance(specs, (list, tuple)):
        specs = [specs]
    return O.Oracle([O.mdp_from_layout_dict(s.to_layout_dict()) for s in specs])


def u8(t):
    return t.cpu().numpy()


@pytest.mark.parametrize("interact", ["table", "predicate"])
@pytest.mark.parametrize("name", TRANSITION_CONFIGS)
def test_golden_transitions(name, interact, manifest, gpu):
    """8 400 (state, joint action) -> (next state, rewards, event_infos) transitions of the reference per configuration,
    through oc_step with event logging: the table-driven kernel (k_step3) and the predicate-network one (k_step)."""
    from overcooked_ai_amd.layouts import LayoutSpec

    cfg = manifest["configs"][name]["transitions"]
    spec = LayoutSpec(cfg["layout"])
    d = np.load(os.path.join(GOLDEN, "transitions_%s.npz" % name))
    n = d["actions"].shape[0]
    env = make_env(spec, n, gpu, horizon=65535)
    env.predicate_interact = interact == "predicate"
    env.set_packed_state(d["state_in"])
    ev = torch.zeros((n,), dtype=torch.int64, device=gpu)
    rew, flags = env.step(torch.from_numpy(d["actions"]).to(gpu), events_out=ev)
    assert np.array_equal(env.get_packed_state(), d["state_out"])
    assert np.max(np.abs(u8(rew).astype(np.float64) - d["rewards"])) <= 1e-6
    assert not u8(flags).any()
    assert np.array_equal(u8(ev).view(np.uint64), d["events"])  # event_infos bit masks of the reference
    # pure-function form: separate output buffer leaves the input untouched (mdp.py:1400 deep-copies)
    env.set_packed_state(d["state_in"])
    out = torch.zeros_like(env.state)
    env.step(torch.from_numpy(d["actions"]).to(gpu), state_out=out)
    assert np.array_equal(u8(env.state), d["state_in"]) and np.array_equal(u8(out), d["state_out"])


@pytest.mark.parametrize("name", [n for n in TRANSITION_CONFIGS if n != "cramped_room_single"])
def test_golden_lossless_encoding(name, manifest, gpu):
    from overcooked_ai_amd.layouts import LayoutSpec

    cfg = manifest["configs"][name]["transitions"]
    spec = LayoutSpec(cfg["layout"])
    d = np.load(os.path.join(GOLDEN, "transitions_%s.npz" % name))
    st = np.ascontiguousarray(d["state_in"][:, d["enc_index"], :])
    for n in (st.shape[1], 1, 5, 63):  # ragged tails of the encode kernel's env groups
        env = make_env(spec, n, gpu, horizon=int(d["enc_horizon"]))
        env.set_packed_state(st[:, :n])
        a = u8(env.encode_lossless(torch.uint8))
        b = u8(env.encode_lossless(torch.float32))
        assert a.shape == (n, 2, spec.width, spec.height, 26)
        assert np.array_equal(a.astype(np.int16), d["enc"][:n])
        assert np.array_equal(b, d["enc"][:n].astype(np.float32))


def test_reference_golden_trajectory(manifest, gpu):
    from overcooked_ai_amd.layouts import LayoutSpec

    cfg = manifest["ref_mdp_dynamics"]
    spec = LayoutSpec(cfg["layout"])
    d = np.load(os.path.join(GOLDEN, "ref_mdp_dynamics.npz"))
    states, actions = d["states"], d["actions"]
    n = states.shape[1]
    env = make_env(spec, n, gpu, horizon=65535)
    env.set_packed_state(states)
    rew, _ = env.step(torch.from_numpy(actions).to(gpu))
    out = env.get_packed_state()
    assert np.array_equal(out[:, : n - 1], states[:, 1:])
    r = u8(rew)
    assert np.array_equal(r[:, 0] + r[:, 1], d["ep_rewards"]) and np.array_equal(r[:, 2:], d["shaped"])
    # sequential replay of the 1499 transitions on one env
    env1 = make_env(spec, 1, gpu, horizon=65535)
    env1.set_packed_state(states[:, :1])
    acts = torch.from_numpy(actions).to(gpu)
    total = 0.0
    for t in range(n - 1):
        r1, _ = env1.step(acts[t:t + 1])
        total += float(r1[0, 0] + r1[0, 1])
    assert np.array_equal(env1.get_packed_state()[:, 0], states[:, n - 1])
    assert total == d["ep_rewards"][: n - 1].sum()


@pytest.mark.parametrize("kernel", ["default", "lane_pair", "predicate_interact"])
@pytest.mark.parametrize("name", ROLLOUT_CONFIGS)
def test_golden_rollouts_fused(name, kernel, manifest, gpu):
    """Both fused Philox rollout kernels against episodes run through the reference's OvercookedEnv.step."""
    from overcooked_ai_amd.layouts import LayoutSpec

    cfg = manifest["configs"][name]["rollouts"]
    spec = LayoutSpec(cfg["layout"])
    d = np.load(os.path.join(GOLDEN, "rollouts_%s.npz" % name))
    n, seed, horizon = cfg["n_envs"], int(d["seed"]), int(d["horizon"])
    env = make_env(spec, n, gpu, horizon=horizon, seed=seed)
    select_kernel(env, kernel)
    for k in range(4):
        rew = torch.zeros((100, n, 4), dtype=torch.float32, device=gpu)
        fl = torch.zeros((100, n), dtype=torch.uint8, device=gpu)
        env.rollout_random(100, rew, fl)
        assert np.array_equal(u8(rew).astype(np.float64), d["rewards"][100 * k:100 * (k + 1)])
        assert np.array_equal(env.get_packed_state(), d["checkpoints"][k])
        f = u8(fl)
        assert f[:-1].sum() == 0 and (f[-1] == (1 if k == 3 else 0)).all()
    assert np.array_equal(u8(env.ep_returns).astype(np.float64), d["rewards"].sum(axis=0))


def test_small_fixtures_and_micro_cases(small_fixtures, gpu):
    from overcooked_ai_amd import state as S
    from overcooked_ai_amd.layouts import LayoutSpec

    cases = list(small_fixtures["micro"])
    for key in ("old_dynamics_cook_test_old0", "old_dynamics_cook_test_old1", "old_dynamics_put_test_old0",
                "old_dynamics_put_test_old1"):
        c = dict(small_fixtures[key])
        c["label"] = key
        cases.append(c)
    for case in cases:
        spec = LayoutSpec(case["layout"])
        env = make_env(spec, 1, gpu, horizon=65535)
        env.set_states([case["state"]])
        rew, _ = env.step(torch.tensor([case["actions"]], dtype=torch.uint8, device=gpu))
        got = env.get_states(as_dict=True)[0]
        assert S.canonical_state_dict(got) == S.canonical_state_dict(case["expected_state"]), case["label"]
        if "sparse" in case:
            r = u8(rew)[0]
            assert list(r[:2]) == case["sparse"] and list(r[2:]) == case["shaped"], case["label"]
        if "after_more_stay" in case:
            stay = torch.tensor([[4, 4]], dtype=torch.uint8, device=gpu)
            for _ in range(case["n_more_stay"]):
                env.step(stay)
            assert S.canonical_state_dict(env.get_states(as_dict=True)[0]) == S.canonical_state_dict(
                case["after_more_stay"])
        if "encoding_layer_sums_p0" in case:
            env.set_states([case["state"]])
            env.horizon = 400
            enc = u8(env.encode_lossless(torch.uint8)).astype(np.int64)
            assert [int(v) for v in enc[0, 0].sum(axis=(0, 1))] == case["encoding_layer_sums_p0"]
            assert [int(v) for v in enc[0, 1].sum(axis=(0, 1))] == case["encoding_layer_sums_p1"]
    # scripted bonus-order episode: total sparse reward 50 (overcooked_test.py:994-998)
    ep = small_fixtures["mdp_test_bonus_episode"]
    spec = LayoutSpec(ep["layout"])
    env = make_env(spec, 1, gpu, horizon=65535)
    env.set_states([ep["start_state"]])
    total = 0.0
    for step in ep["steps"]:
        rew, _ = env.step(torch.tensor([step["actions"]], dtype=torch.uint8, device=gpu))
        r = u8(rew)[0]
        assert list(r[:2]) == step["sparse"] and list(r[2:]) == step["shaped"]
        total += r[:2].sum()
    assert total == 50
    assert S.canonical_state_dict(env.get_states(as_dict=True)[0]) == S.canonical_state_dict(
        ep["steps"][-1]["next_state"])


@pytest.mark.parametrize("n_envs", [1, 63, 257, 4096])
def test_step_vs_oracle_random_states(n_envs, gpu):
    """Ragged batch sizes, every canonical layout, explicit actions incl. illegal ones and auto-reset."""
    from overcooked_ai_amd.layouts import spec_from_name

    rng = np.random.default_rng(n_envs)
    for name in CANONICAL_5 + ["mdp_test"]:
        spec = spec_from_name(name)
        orc = oracle_for(spec)
        st = random_packed_states(spec, n_envs, rng)
        acts = rng.integers(0, 6, size=(n_envs, 2)).astype(np.uint8)
        acts[rng.random(n_envs) < 0.02, 0] = 6 + rng.integers(0, 200)
        if n_envs > 1:
            acts[1, 1] = 9
        horizon = 200
        env = make_env(spec, n_envs, gpu, horizon=horizon, auto_reset=True)
        env.set_packed_state(st)
        ep0 = rng.integers(0, 50, size=(n_envs, 4)).astype(np.float32)
        env.ep_returns.copy_(torch.from_numpy(ep0))
        ev = torch.zeros((n_envs,), dtype=torch.int64, device=gpu)
        rew, flags = env.step(torch.from_numpy(acts).to(gpu), events_out=ev)
        ep_o = ep0.copy()
        out_o, rew_o, fl_o = orc.step(st, acts, horizon=horizon, options=1, ep_returns=ep_o)
        assert np.array_equal(env.get_packed_state(), out_o), name
        assert np.array_equal(u8(ev).view(np.uint64), orc.last_events), name
        assert np.array_equal(u8(rew), rew_o) and np.array_equal(u8(flags), fl_o)
        assert np.array_equal(u8(env.ep_returns), ep_o)
        assert (fl_o & 2).any() or n_envs == 1
        # the same step in place without event logging = k_step1 (the transition on the wire format itself), and through
        # OC_STEP's out-of-place form = k_step3
        lean = make_env(spec, n_envs, gpu, horizon=horizon, auto_reset=True)
        lean.set_packed_state(st)
        lean.ep_returns.copy_(torch.from_numpy(ep0))
        rew2, flags2 = lean.step(torch.from_numpy(acts).to(gpu))
        assert np.array_equal(lean.get_packed_state(), out_o), name
        assert np.array_equal(u8(rew2), rew_o) and np.array_equal(u8(flags2), fl_o) and np.array_equal(u8(lean.ep_returns), ep_o)


@pytest.mark.parametrize("kernel", ["default", "lane_pair", "predicate_interact"])
def test_full_size_rollout_vs_oracle(kernel, gpu):
    """BASELINE config 2 at full size: 65 536 cramped_room envs, random policy, horizon 400 with auto-reset."""
    from overcooked_ai_amd.layouts import spec_from_name

    n = 65536
    spec = spec_from_name("cramped_room")
    orc = oracle_for(spec)
    env = make_env(spec, n, gpu, horizon=400, auto_reset=True, seed=1234)
    select_kernel(env, kernel)
    st_o = orc.reset(orc.new_state(n))
    ep_o = np.zeros((n, 4), np.float32)
    t0 = 0
    for chunk in (7, 64, 129, 200, 100):  # crosses the 400-step reset boundary
        rew = torch.zeros((chunk, n, 4), dtype=torch.float32, device=gpu)
        fl = torch.zeros((chunk, n), dtype=torch.uint8, device=gpu)
        env.rollout_random(chunk, rew, fl)
        rew_o, fl_o = orc.rollout_random(st_o, chunk, horizon=400, options=1, seed=1234, t0=t0, ep_returns=ep_o)
        t0 += chunk
        assert np.array_equal(env.get_packed_state(), st_o)
        assert np.array_equal(u8(rew), rew_o) and np.array_equal(u8(fl), fl_o)
        assert np.array_equal(u8(env.ep_returns), ep_o)
    # step-by-step API with the same Philox actions reaches the same state as the fused kernel
    from oracle import oracle as O

    env2 = make_env(spec, n, gpu, horizon=400, auto_reset=True)
    for t in range(20):
        env2.step(torch.from_numpy(O.random_actions(1234, 0, t, n)).to(gpu))
    env3 = make_env(spec, n, gpu, horizon=400, auto_reset=True, seed=1234)
    env3.rollout_random(20)
    assert torch.equal(env2.state, env3.state)
    # ... and so does oc_step_many on the same action tensor
    env4 = make_env(spec, n, gpu, horizon=400, auto_reset=True)
    acts = torch.from_numpy(np.stack([O.random_actions(1234, 0, t, n) for t in range(20)])).to(gpu)
    rew4 = torch.zeros((20, n, 4), dtype=torch.float32, device=gpu)
    fl4 = torch.zeros((20, n), dtype=torch.uint8, device=gpu)
    env4.step_many(acts, rew4, fl4)
    rew3 = torch.zeros((20, n, 4), dtype=torch.float32, device=gpu)
    env5 = make_env(spec, n, gpu, horizon=400, auto_reset=True, seed=1234)
    env5.rollout_random(20, rew3, None)
    assert torch.equal(env4.state, env3.state) and torch.equal(rew4, rew3)
    # sharding property: a shard that owns envs [a, b) reproduces exactly that slice
    a, b = 12345, 12345 + 777
    shard = make_env(spec, b - a, gpu, horizon=400, auto_reset=True, seed=1234, env_offset=a)
    shard.rollout_random(20)
    assert torch.equal(shard.state, env3.state[:, a:b])


@pytest.mark.parametrize("kernel", ["default", "lane_pair", "predicate_interact"])
def test_mixed_layout_batch_vs_oracle(kernel, gpu):
    """BASELINE config 4 (one GPU's shard): env e uses canonical layout e % 5, all padded to 9x5."""
    from overcooked_ai_amd.layouts import LayoutTable, spec_from_name

    n = 20000
    table = LayoutTable([spec_from_name(nm) for nm in CANONICAL_5], pad_to=(9, 5))
    lid = (np.arange(n) % 5).astype(np.uint16)
    orc = oracle_for(table.specs)
    env = make_env(table, n, gpu, horizon=400, auto_reset=True, seed=99, layout_id=lid)
    select_kernel(env, kernel)
    rng = np.random.default_rng(5)
    st = np.zeros((table.n_planes, n, 16), np.uint8)
    for l in range(5):
        idx = np.nonzero(lid == l)[0]
        st[:, idx] = random_packed_states(table.specs[l], len(idx), rng)
    env.set_packed_state(st)
    st_o = st.copy()
    ep_o = np.zeros((n, 4), np.float32)
    rew = torch.zeros((150, n, 4), dtype=torch.float32, device=gpu)
    fl = torch.zeros((150, n), dtype=torch.uint8, device=gpu)
    env.rollout_random(150, rew, fl)
    rew_o, fl_o = orc.rollout_random(st_o, 150, horizon=400, options=1, seed=99, layout_id=lid, ep_returns=ep_o)
    assert np.array_equal(env.get_packed_state(), st_o)
    assert np.array_equal(u8(rew), rew_o) and np.array_equal(u8(fl), fl_o)
    assert np.abs(rew_o).sum() > 0
    enc = u8(env.encode_lossless(torch.uint8))
    assert np.array_equal(enc.astype(np.int32), orc.encode_lossless(st_o, horizon=400, layout_id=lid))


def test_large_layout_table_global_path(gpu):
    """More than 32 layouts: the table is read from HBM/L2 instead of LDS."""
    from overcooked_ai_amd.layouts import LayoutTable, spec_from_name

    names = ["cramped_room", "cramped_room_tomato", "bonus_order_test", "mdp_test", "simple_o", "simple_o_t",
             "simple_tomato", "m_shaped_s", "cramped_room_o_3orders"]
    specs = [spec_from_name(nm) for nm in names] * 5  # 45 entries
    table = LayoutTable(specs)
    n = 9000
    lid = (np.arange(n) * 7 % len(specs)).astype(np.uint16)
    orc = oracle_for(table.specs)
    env = make_env(table, n, gpu, horizon=100, auto_reset=True, seed=5, layout_id=lid)
    rng = np.random.default_rng(11)
    st = np.zeros((table.n_planes, n, 16), np.uint8)
    for l in range(len(specs)):
        idx = np.nonzero(lid == l)[0]
        st[:, idx] = random_packed_states(table.specs[l], len(idx), rng, timestep_max=99)
    st_o = st.copy()
    orc.rollout_random(st_o, 120, horizon=100, options=1, seed=5, layout_id=lid, want_outputs=False)
    for table_interact in (True, False):
        env.predicate_interact = table_interact
        env.set_packed_state(st)
        env.t_global = 0
        env.rollout_random(120)
        assert np.array_equal(env.get_packed_state(), st_o)
    acts = rng.integers(0, 6, size=(n, 2)).astype(np.uint8)
    env.step(torch.from_numpy(acts).to(gpu))
    st_o2, _, _ = orc.step(st_o, acts, horizon=100, options=1, layout_id=lid)
    assert np.array_equal(env.get_packed_state(), st_o2)
    enc = u8(env.encode_lossless(torch.float32))
    assert np.array_equal(enc, orc.encode_lossless(st_o2, horizon=100, layout_id=lid).astype(np.float32))


def test_one_step_kernels_on_a_table_read_from_hbm(gpu):
    """oc_step in place (k_step1), out of place and with event logging (the event kernel) on a table of more than 32
    layouts: those instances read the layout records through L2 and stage only the interact LUT in LDS — the
    workgroup barrier between that staging and the first look-up is part of the prologue whatever the table's
    size.  Fresh envs every iteration so that every launch is a cold one."""
    from overcooked_ai_amd.layouts import LayoutTable, spec_from_name

    names = ["cramped_room", "cramped_room_tomato", "bonus_order_test", "mdp_test", "simple_o", "simple_o_t",
             "simple_tomato", "m_shaped_s", "cramped_room_o_3orders"]
    specs = [spec_from_name(nm) for nm in names] * 4  # 36 entries
    table = LayoutTable(specs)
    n = 70000
    lid = (np.arange(n) * 5 % len(specs)).astype(np.uint16)
    orc = oracle_for(table.specs)
    rng = np.random.default_rng(23)
    st = np.zeros((table.n_planes, n, 16), np.uint8)
    for l in range(len(specs)):
        idx = np.nonzero(lid == l)[0]
        st[:, idx] = random_packed_states(table.specs[l], len(idx), rng, timestep_max=99)
    for it in range(6):
        acts = rng.integers(0, 6, size=(n, 2)).astype(np.uint8)
        st_o, rew_o, fl_o = orc.step(st, acts, horizon=100, options=1, layout_id=lid)
        env = make_env(table, n, gpu, horizon=100, auto_reset=True, seed=5, layout_id=lid)
        env.set_packed_state(st)
        if it % 2 == 0:
            rew, fl = env.step(torch.from_numpy(acts).to(gpu))
        else:
            ev = torch.zeros((n,), dtype=torch.int64, device=gpu)
            rew, fl = env.step(torch.from_numpy(acts).to(gpu), events_out=ev)
            assert np.array_equal(u8(ev).view(np.uint64), orc.last_events)
        assert np.array_equal(env.get_packed_state(), st_o), it
        assert np.array_equal(u8(rew), rew_o) and np.array_equal(u8(fl), fl_o), it
        st = st_o


def test_every_registry_layout_vs_oracle(gpu):
    """All 1- and 2-player layouts shipped by the reference (grids up to 14x9 = 8 object planes)."""
    from overcooked_ai_amd.layouts import layout_names, spec_from_name

    rng = np.random.default_rng(2024)
    n = 1500
    for name in layout_names():
        if name == "multiplayer_schelling":
            continue
        spec = spec_from_name(name)
        orc = oracle_for(spec)
        st = random_packed_states(spec, n, rng)
        env = make_env(spec, n, gpu, horizon=400, auto_reset=True, seed=3)
        st_o = st.copy()
        rew_o, _ = orc.rollout_random(st_o, 60, horizon=400, options=1, seed=3)
        for kernel in ("default", "lane_pair", "predicate_interact"):  # lane_pair falls back where it does not apply
            select_kernel(env, kernel)
            env.set_packed_state(st)
            env.t_global = 0
            rew = torch.zeros((60, n, 4), dtype=torch.float32, device=gpu)
            env.rollout_random(60, rew, None)
            assert np.array_equal(env.get_packed_state(), st_o), (name, kernel)
            assert np.array_equal(u8(rew), rew_o), (name, kernel)
        if spec.num_players == 2:
            enc = u8(env.encode_lossless(torch.uint8))
            assert np.array_equal(enc.astype(np.int32), orc.encode_lossless(st_o, horizon=400)), name


def test_full_size_encoding_properties(gpu):
    """BASELINE config 3 at full size: 65 536 asymmetric_advantages envs. Size-independent properties:
    f32 == u8, player-swap symmetry (overcooked_test.py:1112-1128), static-layer checksums; oracle on a slice."""
    from overcooked_ai_amd.layouts import spec_from_name

    n = 65536
    spec = spec_from_name("asymmetric_advantages")
    env = make_env(spec, n, gpu, horizon=400, auto_reset=True, seed=8)
    rng = np.random.default_rng(0)
    st = random_packed_states(spec, 4096, rng)
    env.set_packed_state(np.tile(st, (1, n // 4096, 1)))
    env.rollout_random(37)
    a = env.encode_lossless(torch.uint8)
    b = env.encode_lossless(torch.float32)
    assert torch.equal(a.float(), b)
    sw = env.state.clone()
    sw[0, :, 0:3], sw[0, :, 3:6] = env.state[0, :, 3:6], env.state[0, :, 0:3]
    c = env.encode_lossless(torch.uint8, state=sw)
    assert torch.equal(a[:, 0], c[:, 1]) and torch.equal(a[:, 1], c[:, 0])
    sums = a.sum(dim=(2, 3), dtype=torch.int64)  # [n, 2, 26]
    assert (sums[:, :, 0] == 1).all() and (sums[:, :, 1] == 1).all()
    assert (sums[:, :, 2:6].sum(-1) == 1).all() and (sums[:, :, 6:10].sum(-1) == 1).all()
    static = torch.tensor([2, 23, 2, 0, 2, 2], device=gpu)  # P, X, O, T, D, S cells of asymmetric_advantages
    assert (sums[:, :, 10:16] == static).all()
    orc = oracle_for(spec)
    host = env.get_packed_state()[:, :3000]
    assert np.array_equal(u8(a[:3000]).astype(np.int32), orc.encode_lossless(np.ascontiguousarray(host), horizon=400))


def test_abi_argument_errors(gpu):
    import ctypes

    from overcooked_ai_amd import _lib

    env = make_env("cramped_room", 8, gpu)
    L = _lib.load()
    rc = L.oc_step(env._bref, None, None, None, None, None, None, None, 400, 0, None, None, None)
    assert rc == -1 and b"NULL" in L.oc_last_error()
    rc = L.oc_step(env._bref, env.state.data_ptr(), env.state.data_ptr(), env.flags.data_ptr(),
                   env.rewards.data_ptr(), env.flags.data_ptr(), None, None, 0, 0, None, None, None)
    assert rc == -1 and b"horizon" in L.oc_last_error()
    bad = _lib.OcBatch(d_layouts=env.d_layouts.data_ptr(), d_layout_id=None, n_envs=8, n_layouts=2, width=5, height=4,
                       max_pots=1)
    rc = L.oc_reset(ctypes.byref(bad), env.state.data_ptr(), None, None, None)
    assert rc == -1 and b"layout_id" in L.oc_last_error()
    with pytest.raises(ValueError):
        env.step(torch.zeros((8, 2), dtype=torch.int64, device=gpu))
    with pytest.raises(ValueError):  # host tensors are refused: the kernels write through raw device pointers
        env.rollout_random(3, torch.zeros((3, env.n_envs, 4)), None)
    with pytest.raises(ValueError):
        env.encode_lossless(torch.uint8, out=torch.zeros((5,), dtype=torch.uint8, device=gpu))


def test_many_pots_layout_vs_oracle(gpu):
    """A custom 7-pot layout: exercises the 8-pot-slot kernels (every layout shipped by the reference has <= 2)."""
    from overcooked_ai_amd.layouts import LayoutSpec, LayoutTable

    spec = LayoutSpec({
        "grid": "XPPPPPX\nO 1 2 O\nX     X\nXDPSPTX",
        "start_all_orders": [{"ingredients": ["onion", "onion", "tomato"]}, {"ingredients": ["onion"]},
                             {"ingredients": ["tomato", "tomato"]}],
        "start_bonus_orders": [{"ingredients": ["onion"]}],
        "onion_value": 7, "tomato_value": 4, "onion_time": 3, "tomato_time": 5,
    })
    assert len(spec.cells_of("P")) == 7
    orc = oracle_for(spec)
    rng = np.random.default_rng(77)
    n = 6000
    st = random_packed_states(spec, n, rng)
    env = make_env(spec, n, gpu, horizon=400, auto_reset=True, seed=21)
    st_o = st.copy()
    rew_o, _ = orc.rollout_random(st_o, 90, horizon=400, options=1, seed=21)
    for table_interact in (False, True):
        env.predicate_interact = table_interact
        env.set_packed_state(st)
        env.t_global = 0
        rew = torch.zeros((90, n, 4), dtype=torch.float32, device=gpu)
        env.rollout_random(90, rew, None)
        assert np.array_equal(env.get_packed_state(), st_o) and np.array_equal(u8(rew), rew_o)
    env.predicate_interact = False
    acts = rng.integers(0, 6, size=(n, 2)).astype(np.uint8)
    acts[rng.random(n) < 0.5] = 5
    ev = torch.zeros((n,), dtype=torch.int64, device=gpu)
    env.set_packed_state(st)
    r, f = env.step(torch.from_numpy(acts).to(gpu), events_out=ev)
    out_o, r_o, f_o = orc.step(st, acts, horizon=400, options=1)
    assert np.array_equal(env.get_packed_state(), out_o) and np.array_equal(u8(r), r_o)
    assert np.array_equal(u8(ev).view(np.uint64), orc.last_events)
    enc = u8(env.encode_lossless(torch.uint8))
    assert np.array_equal(enc.astype(np.int32), orc.encode_lossless(out_o, horizon=400))
    # mixed with a 2-pot layout in one table (per-lane pot counts differ)
    from overcooked_ai_amd.layouts import spec_from_name
    table = LayoutTable([spec, spec_from_name("scenario1_s")])
    lid = (np.arange(n) % 2).astype(np.uint16)
    orc2 = oracle_for(table.specs)
    st2 = np.zeros((table.n_planes, n, 16), np.uint8)
    for l in range(2):
        idx = np.nonzero(lid == l)[0]
        st2[:, idx] = random_packed_states(table.specs[l], len(idx), rng)
    env2 = make_env(table, n, gpu, horizon=400, auto_reset=True, seed=4, layout_id=lid)
    env2.set_packed_state(st2)
    env2.rollout_random(70)
    orc2.rollout_random(st2, 70, horizon=400, options=1, seed=4, layout_id=lid, want_outputs=False)
    assert np.array_equal(env2.get_packed_state(), st2)


def test_featurize_state_golden_and_oracle(gpu):
    """featurize_state (mdp.py:2579): reference fixtures (incl. the states behind the reference's own golden
    expected_2.pickle) and, on every 2-player layout, the oracle — for both counter_goals settings."""
    import json

    from oracle import oracle as O
    from overcooked_ai_amd.layouts import LayoutSpec, LayoutTable, layout_names, spec_from_name

    with open(os.path.join(GOLDEN, "featurize_manifest.json")) as f:
        man = json.load(f)
    for key, cfg in man.items():
        spec = LayoutSpec(cfg["layout"])
        d = np.load(os.path.join(GOLDEN, "featurize_%s.npz" % key))
        env = make_env(spec, d["states"].shape[1], gpu)
        env.set_packed_state(d["states"])
        got = u8(env.featurize(num_pots=2, counter_goals=cfg["counter_goals"]))
        assert got.shape == d["features"].shape and np.array_equal(got, d["features"]), key
    d = np.load(os.path.join(GOLDEN, "ref_greedy_rollouts.npz"))
    env = make_env(LayoutSpec(man["cramped_room_none"]["layout"]), d["states"].shape[1], gpu)
    env.set_packed_state(d["states"])
    assert np.array_equal(u8(env.featurize()), d["features"])  # == the reference's expected_2.pickle
    for num_pots in (0, 1):  # == expected_0.pickle / expected_1.pickle (overcooked_test.py:1069-1093 loops range(3))
        assert np.array_equal(u8(env.featurize(num_pots=num_pots)), d["features_num_pots_%d" % num_pots]), num_pots
    # symmetry (overcooked_test.py:1094-1110): swapping the players swaps the two feature rows
    sw = env.state.clone()
    sw[0, :, 0:3], sw[0, :, 3:6] = env.state[0, :, 3:6], env.state[0, :, 0:3]
    a, b = env.featurize(), env.featurize(state=sw)
    assert torch.equal(a[:, 0], b[:, 1]) and torch.equal(a[:, 1], b[:, 0])
    rng = np.random.default_rng(99)
    for name in layout_names():
        spec = spec_from_name(name)
        if spec.num_players != 2:
            continue
        orc = oracle_for(spec)
        for n, num_pots in ((700, 2), (65, 0), (129, 3)):
            st = random_packed_states(spec, n, rng)
            env = make_env(spec, n, gpu)
            env.set_packed_state(st)
            for cg in ("none", "all"):
                got = u8(env.featurize(num_pots=num_pots, counter_goals=cg))
                assert np.array_equal(got, O.featurize(orc, st, counter_goals=cg, num_pots=num_pots)), (name, cg, num_pots)
    # mixed-layout table
    table = LayoutTable([spec_from_name(nm) for nm in CANONICAL_5], pad_to=(9, 5))
    n = 5000
    lid = (np.arange(n) % 5).astype(np.uint16)
    st = np.zeros((table.n_planes, n, 16), np.uint8)
    for l in range(5):
        idx = np.nonzero(lid == l)[0]
        st[:, idx] = random_packed_states(table.specs[l], len(idx), rng)
    env = make_env(table, n, gpu, layout_id=lid)
    env.set_packed_state(st)
    orc = oracle_for(table.specs)
    assert np.array_equal(u8(env.featurize(counter_goals="all")), O.featurize(orc, st, counter_goals="all", layout_id=lid))


def test_potential_function_golden_and_oracle(gpu):
    """potential_function (mdp.py:2920-3238): float64 phi bit-identical to the reference fixtures (8 layouts incl. a
    7-pot layout with tomatoes and a bonus order, gamma 0.99 and 0.9) and to the oracle on every layout."""
    import json

    from oracle import oracle as O
    from overcooked_ai_amd.layouts import LayoutSpec, LayoutTable, layout_names, spec_from_name
    from overcooked_ai_amd.potential import potential_params

    with open(os.path.join(GOLDEN, "potential_manifest.json")) as f:
        man = json.load(f)
    for key, cfg in man.items():
        spec = LayoutSpec(dict(cfg["layout"]))
        d = np.load(os.path.join(GOLDEN, "potential_%s.npz" % key))
        env = make_env(spec, d["states"].shape[1], gpu)
        env.set_packed_state(d["states"])
        for gi, gamma in enumerate(cfg["gammas"]):
            got = env.potential(gamma).cpu().numpy()
            assert got.dtype == np.float64 and np.array_equal(got, d["phi"][gi]), (key, gamma, np.abs(got - d["phi"][gi]).max())
    rng = np.random.default_rng(2024)
    for name in layout_names():
        spec = spec_from_name(name)
        if spec.num_players not in (1, 2):
            continue
        try:
            pp = potential_params(spec, 0.97)
            from overcooked_ai_amd.potential import phi_record
            phi_record(spec, 0.97)
        except ValueError:
            continue  # no order with a positive value: the reference divides by zero there
        orc = oracle_for(spec)
        st = random_packed_states(spec, 1500, rng)
        env = make_env(spec, 1500, gpu)
        env.set_packed_state(st)
        got = env.potential(0.97).cpu().numpy()
        assert np.array_equal(got, O.potential(orc, st, [pp]), equal_nan=True), name  # tutorial_3's infinite recipe value gives nan in the reference too
    # mixed-layout table, a few million values at BASELINE size: phi(s) >= steady-state > 0 and finite
    table = LayoutTable([spec_from_name(nm) for nm in CANONICAL_5], pad_to=(9, 5))
    n = 5000
    lid = (np.arange(n) % 5).astype(np.uint16)
    st = np.zeros((table.n_planes, n, 16), np.uint8)
    for l in range(5):
        idx = np.nonzero(lid == l)[0]
        st[:, idx] = random_packed_states(table.specs[l], len(idx), rng)
    env = make_env(table, n, gpu, layout_id=lid)
    env.set_packed_state(st)
    orc = oracle_for(table.specs)
    got = env.potential(0.99).cpu().numpy()
    assert np.array_equal(got, O.potential(orc, st, [potential_params(s_, 0.99) for s_ in table.specs], layout_id=lid))
    big = make_env("cramped_room", 65536, gpu, horizon=400, auto_reset=True, seed=1)
    big.rollout_random(150)
    phi = big.potential().cpu().numpy()
    sub = big.get_packed_state()[:, :4096]
    assert np.isfinite(phi).all() and phi.min() > 0
    assert np.array_equal(phi[:4096], O.potential(oracle_for(spec_from_name("cramped_room")), sub,
                                                  [potential_params(spec_from_name("cramped_room"), 0.99)]))


def test_reset_random_matches_oracle(gpu):
    """oc_reset_random: bit-exact against the oracle's restatement of the same Philox stream — single layouts
    (1 and 2 players, 7 pots), a mixed table, masks, shards and epochs; ep_returns cleared for the reset envs."""
    from overcooked_ai_amd.layouts import LayoutSpec, LayoutTable, spec_from_name

    seven = LayoutSpec({"grid": "XPPPPPX\nO 1 2 O\nX     X\nXDPSPTX", "onion_time": 3, "tomato_time": 5,
                        "onion_value": 7, "tomato_value": 4})
    specs = [spec_from_name(nm) for nm in ("cramped_room", "asymmetric_advantages", "cramped_room_single", "corridor",
                                           "counter_circuit")] + [seven]
    for spec in specs:
        n = 3000
        orc = oracle_for(spec)
        env = make_env(spec, n, gpu, seed=21, env_offset=700)
        for pos, t in ((True, 0.0), (False, 0.4), (True, 1.0), (True, 0.27)):
            epoch = env.reset_epoch
            env.reset(random_start_pos=pos, rnd_obj_prob_thresh=t)
            want = orc.reset_random(orc.new_state(n), seed=21, env_offset=700, epoch=epoch, random_start_pos=pos,
                                    rnd_obj_prob_thresh=t)
            assert np.array_equal(env.get_packed_state(), want), (spec.layout_name, pos, t)
        assert env.reset_epoch == 5  # epoch 0 belongs to the constructor's reset, then one per explicit reset
    table = LayoutTable([spec_from_name(nm) for nm in CANONICAL_5], pad_to=(9, 5))
    n = 4097
    lid = (np.arange(n) % 5).astype(np.uint16)
    env = make_env(table, n, gpu, layout_id=lid, seed=5)
    env.rollout_random(37)
    before = env.get_packed_state()
    mask = (np.arange(n) % 3 == 0)
    env.ep_returns.fill_(1.0)
    epoch = env.reset_epoch  # the launch consumed one epoch per step (explicit resets and in-kernel restarts share ONE counter)
    assert epoch == 1 + 37
    env.reset(mask=torch.from_numpy(mask), random_start_pos=True, rnd_obj_prob_thresh=0.5)
    want = oracle_for(table.specs).reset_random(before.copy(), seed=5, epoch=epoch, random_start_pos=True, rnd_obj_prob_thresh=0.5,
                                                layout_id=lid, mask=mask.astype(np.uint8))
    assert np.array_equal(env.get_packed_state(), want)
    ep = env.ep_returns.cpu().numpy()
    assert (ep[mask] == 0).all() and (ep[~mask] == 1).all()
    # a randomized batch steps like any other: fused rollout == oracle from those states
    st = env.get_packed_state()
    env.rollout_random(25)
    oracle_for(table.specs).rollout_random(st, 25, horizon=400, options=0, seed=5, t0=37, layout_id=lid, want_outputs=False)
    assert np.array_equal(env.get_packed_state(), st)


def test_hip_graph_capture_of_step_and_encode(gpu):
    """The C-ABI launches on the caller's stream without synchronising, so a step + encode pair can be captured in
    a HIP graph (torch.cuda.graph) and replayed with new actions: same states, rewards and observations as eager."""
    n = 4096
    env = make_env("asymmetric_advantages", n, gpu, horizon=50, auto_reset=True, seed=0)
    ref = make_env("asymmetric_advantages", n, gpu, horizon=50, auto_reset=True, seed=0)
    acts = torch.zeros((n, 2), dtype=torch.uint8, device=gpu)
    obs, obs_ref = (torch.empty((n, 2, 9, 5, 26), dtype=torch.uint8, device=gpu) for _ in range(2))
    side = torch.cuda.Stream(device=gpu)
    side.wait_stream(torch.cuda.current_stream(gpu))
    with torch.cuda.stream(side):
        env.step(acts)
        env.encode_lossless(out=obs)
    torch.cuda.current_stream(gpu).wait_stream(side)
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph):
        env.step(acts)
        env.encode_lossless(out=obs)
    env.reset()
    gen = torch.Generator(device=gpu).manual_seed(3)
    for _ in range(60):  # crosses the horizon: auto-reset inside the captured step
        acts.copy_(torch.randint(0, 6, (n, 2), dtype=torch.uint8, device=gpu, generator=gen))
        graph.replay()
        ref.step(acts)
        ref.encode_lossless(out=obs_ref)
        assert torch.equal(env.rewards, ref.rewards) and torch.equal(env.flags, ref.flags)
    assert torch.equal(env.state, ref.state) and torch.equal(obs, obs_ref) and torch.equal(env.ep_returns, ref.ep_returns)


def test_hip_graph_capture_of_the_training_step(gpu):
    """oc_multi_agent_step (k_train_step1 + k_encode: step, phi, shaped rewards, restart, observation) enqueues on the
    caller's stream without synchronising: captured in a HIP graph and replayed with new actions it gives what the eager
    calls give — the (actions -> step -> observation) chain of a policy loop without per-kernel launches."""
    from overcooked_ai_amd.multi_agent import VecOvercookedMultiAgent

    n = 3000
    kw = dict(horizon=40, reward_shaping_factor=0.7, use_phi=True, obs_dtype=torch.uint8, device=gpu)
    env = VecOvercookedMultiAgent("cramped_room", n, **kw)
    ref = VecOvercookedMultiAgent("cramped_room", n, **kw)
    acts = torch.zeros((n, 2), dtype=torch.uint8, device=gpu)
    side = torch.cuda.Stream(device=gpu)
    side.wait_stream(torch.cuda.current_stream(gpu))
    with torch.cuda.stream(side):
        env.step(acts)  # (builds the potential tables outside the capture)
    torch.cuda.current_stream(gpu).wait_stream(side)
    ref.step(acts)
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph):
        obs, shaped, done, infos = env.step(acts)
    graph.replay()  # (capturing recorded the call without running it)
    ref.step(acts)
    gen = torch.Generator(device=gpu).manual_seed(5)
    for k in range(55):  # crosses the horizon: finished envs restart inside the captured call
        acts.copy_(torch.randint(0, 6, (n, 2), dtype=torch.uint8, device=gpu, generator=gen))
        graph.replay()
        obs_r, shaped_r, done_r, infos_r = ref.step(acts)
        assert torch.equal(shaped, shaped_r) and torch.equal(done, done_r) and torch.equal(obs, obs_r), k
        assert torch.equal(infos["phi_s_prime"], infos_r["phi_s_prime"]), k
    assert torch.equal(env.venv.state, ref.venv.state) and done_r.sum() >= 0


def test_fused_training_step_equals_the_kernel_sequence(gpu):
    """oc_multi_agent_step runs k_train_step (one kernel) for two-player tables with <= 2 pots and the sequence
    oc_step -> oc_potential -> oc_shape_rewards -> copy -> oc_reset otherwise; both must produce identical states,
    rewards, flags, episode returns, potentials, shaped rewards and done masks, with and without use_phi, across
    episode ends and illegal actions."""
    from overcooked_ai_amd.layouts import LayoutSpec, LayoutTable, spec_from_name
    from overcooked_ai_amd.multi_agent import VecOvercookedMultiAgent

    seven = LayoutSpec({"grid": "XPPPPPX\nO 1 2 O\nX     X\nXDPSPTX", "onion_time": 3, "tomato_time": 5,
                        "onion_value": 7, "tomato_value": 4})
    cases = [("cramped_room", None), ("coordination_ring", None), ("corridor", None), ("counter_circuit", None),
             (LayoutTable([spec_from_name(nm) for nm in CANONICAL_5], pad_to=(9, 5)), 5), (seven, None)]
    rng = np.random.default_rng(8)
    for layouts, n_lay in cases:
        for use_phi in (True, False):
            n = 3000
            lid = (np.arange(n) % n_lay).astype(np.uint16) if n_lay else None
            env = VecOvercookedMultiAgent(layouts, n, horizon=23, reward_shaping_factor=0.37, use_phi=use_phi, device=gpu,
                                          layout_id=lid, seed=1)
            ref = VecOvercookedMultiAgent(layouts, n, horizon=23, reward_shaping_factor=0.37, use_phi=use_phi, device=gpu,
                                          layout_id=lid, seed=1)
            v = ref.venv
            env.venv.reset(random_start_pos=True, rnd_obj_prob_thresh=0.5)
            v.set_packed_state(env.venv.get_packed_state())
            if use_phi:
                env.phi_cur.copy_(env.venv.potential(0.99))
                ref.phi_cur.copy_(env.phi_cur)
            for t in range(60):
                a = rng.integers(0, 6, size=(n, 2)).astype(np.uint8)
                a[rng.integers(0, n, size=5), rng.integers(0, 2, size=5)] = 9  # illegal actions: env untouched, flagged
                acts = torch.from_numpy(a).to(gpu)
                env.step(acts)
                # the same step through the separate entry points
                rew, fl = v.step(acts)
                if use_phi:
                    v.potential(0.99, out=ref.phi_next)
                rc = v.lib.oc_shape_rewards(v._bref, rew.data_ptr(), fl.data_ptr(), ref.phi_next.data_ptr() if use_phi else None,
                                            ref.phi_cur.data_ptr(), ref.phi_start.data_ptr(), 0.37, ref.shaped.data_ptr(),
                                            ref.done.data_ptr(), torch.cuda.current_stream().cuda_stream)
                assert rc == 0
                ref.ep_returns.copy_(v.ep_returns)
                v.reset(mask=ref.done)
                for name, x, y in (("state", env.venv.state, v.state), ("rewards", env.venv.rewards, v.rewards),
                                   ("flags", env.venv.flags, v.flags), ("ep", env.venv.ep_returns, v.ep_returns),
                                   ("ep_out", env.ep_returns, ref.ep_returns), ("shaped", env.shaped, ref.shaped),
                                   ("done", env.done, ref.done), ("phi_next", env.phi_next, ref.phi_next),
                                   ("phi_cur", env.phi_cur, ref.phi_cur)):
                    assert torch.equal(x, y), (getattr(layouts, "layout_name", layouts), use_phi, t, name)
            assert int(env.done.sum()) >= 0 and (env.venv.flags & 2).any()


def test_training_step_with_its_observation_in_one_kernel(gpu):
    """Single-layout batches of >= 32 768 envs run oc_multi_agent_step as ONE kernel (k_train_step_obs: transition, phi,
    shaped rewards, restart AND the lossless observation).  The same envs over a table that holds the layout TWICE (a mixed
    table: k_train_step1's per-env-layout instance + the generic observation kernel) must give identical states, rewards,
    flags, episode returns, potentials, shaped rewards, done masks and observations — u8 and f32, one and two pots, with
    and without use_phi, standard and drawn restarts, across episode ends, illegal actions and a ragged last workgroup."""
    from overcooked_ai_amd.layouts import LayoutTable, spec_from_name
    from overcooked_ai_amd.multi_agent import VecOvercookedMultiAgent

    rng = np.random.default_rng(21)
    n = 128 * 256 + 232  # at least half of the CUs get a workgroup; the last workgroup is ragged
    cases = [("cramped_room", torch.uint8, True, True), ("cramped_room", torch.float32, False, False),
             ("coordination_ring", torch.uint8, True, False), ("asymmetric_advantages", torch.uint8, False, True),
             ("asymmetric_advantages", torch.float32, True, True)]
    for name, dt, use_phi, random_starts in cases:
        spec = spec_from_name(name)
        kw = dict(horizon=11, reward_shaping_factor=0.37, use_phi=use_phi, device=gpu, obs_dtype=dt, seed=5,
                  random_start_pos=random_starts, rnd_obj_prob_thresh=0.4 if random_starts else 0.0)
        one = VecOvercookedMultiAgent(spec, n, **kw)
        two = VecOvercookedMultiAgent(LayoutTable([spec, spec]), n, layout_id=(np.arange(n) % 2).astype(np.uint16), **kw)
        one.venv.reset(random_start_pos=True, rnd_obj_prob_thresh=0.5)
        two.venv.set_packed_state(one.venv.get_packed_state())
        if use_phi:
            one.phi_cur.copy_(one.venv.potential(0.99))
            two.phi_cur.copy_(one.phi_cur)
        two.venv._epoch = one.venv._epoch  # (the restart draws are keyed by (seed, env, epoch))
        for t in range(25):
            a = rng.integers(0, 6, size=(n, 2)).astype(np.uint8)
            a[rng.integers(0, n, size=7), rng.integers(0, 2, size=7)] = 9  # illegal actions: env untouched, flagged
            acts = torch.from_numpy(a).to(gpu)
            obs1 = one.step(acts)[0]
            obs2 = two.step(acts)[0]
            for what, x, y in (("state", one.venv.state, two.venv.state), ("rewards", one.venv.rewards, two.venv.rewards),
                               ("flags", one.venv.flags, two.venv.flags), ("ep", one.venv.ep_returns, two.venv.ep_returns),
                               ("ep_out", one.ep_returns, two.ep_returns), ("shaped", one.shaped, two.shaped),
                               ("done", one.done, two.done), ("phi_next", one.phi_next, two.phi_next),
                               ("phi_cur", one.phi_cur, two.phi_cur), ("obs", obs1, obs2)):
                assert torch.equal(x, y), (name, dt, use_phi, random_starts, t, what)
        assert int(one.done.sum()) >= 0 and (one.venv.flags & 2).any()
        # ... and the observation is the encoding of the stored state
        assert torch.equal(obs1, one.venv.encode_lossless(dt))


def test_no_out_of_bounds_writes_with_ragged_batches(gpu):
    """Every caller-owned buffer sits between two guard regions; after all kernels ran on ragged batch sizes (not a
    multiple of the 256-lane workgroup, of the encode group or of the 128-env featurize block) the guards are intact."""
    from overcooked_ai_amd.layouts import LayoutTable, spec_from_name
    from overcooked_ai_amd.multi_agent import VecOvercookedMultiAgent

    GUARD = 4096

    def guarded(shape, dtype):
        n = int(np.prod(shape)) * torch.empty((), dtype=dtype).element_size()
        pad = (-n) % 16
        raw = torch.full((GUARD + n + pad + GUARD,), 0xAB, dtype=torch.uint8, device=gpu)
        return raw, raw[GUARD:GUARD + n].view(dtype).view(*shape)

    def intact(raw, body_bytes):
        return bool((raw[:GUARD] == 0xAB).all()) and bool((raw[GUARD + body_bytes:] == 0xAB).all())

    table5 = LayoutTable([spec_from_name(nm) for nm in CANONICAL_5], pad_to=(9, 5))
    for layouts, n, n_lay in (("cramped_room", 1, 0), ("cramped_room", 1001, 0), ("asymmetric_advantages", 777, 0),
                              ("corridor", 333, 0), (table5, 1283, 5)):
        lid = (np.arange(n) % n_lay).astype(np.uint16) if n_lay else None
        env = make_env(layouts, n, gpu, horizon=9, auto_reset=True, seed=3, layout_id=lid)
        bufs = []
        K = 7
        r_raw, rew = guarded((K, n, 4), torch.float32); bufs.append((r_raw, rew))
        f_raw, fl = guarded((K, n), torch.uint8); bufs.append((f_raw, fl))
        for mode in (None, "default", "lane_pair", "predicate_interact"):
            select_kernel(env, mode)
            env.rollout_random(K, rew, fl)
        s_raw, st_out = guarded(tuple(env.state.shape), torch.uint8); bufs.append((s_raw, st_out))
        e_raw, ev = guarded((n,), torch.int64); bufs.append((e_raw, ev))
        acts = torch.randint(0, 6, (n, 2), dtype=torch.uint8, device=gpu)
        env.step(acts, state_out=st_out, events_out=ev)
        env.step(acts, state_out=st_out)
        a_raw, acts_k = guarded((K, n, 2), torch.uint8); acts_k.copy_(torch.randint(0, 6, (K, n, 2), dtype=torch.uint8, device=gpu))
        env.step_many(acts_k, rew, fl)
        for dt in (torch.uint8, torch.float32):
            o_raw, obs = guarded((n, 2, env.width, env.height, 26), dt); bufs.append((o_raw, obs))
            env.encode_lossless(dt, out=obs)
        ft_raw, feat = guarded((n, 2, 96), torch.float32); bufs.append((ft_raw, feat))
        env.featurize(out=feat)
        p_raw, phi = guarded((n,), torch.float64); bufs.append((p_raw, phi))
        env.potential(out=phi)
        env.reset(mask=torch.ones(n, dtype=torch.uint8, device=gpu), random_start_pos=True, rnd_obj_prob_thresh=0.5)
        ma = VecOvercookedMultiAgent(layouts, n, horizon=5, reward_shaping_factor=1.0, use_phi=True, device=gpu, layout_id=lid)
        for _ in range(7):
            ma.step(acts)
        torch.cuda.synchronize()
        for raw, body in bufs:
            assert intact(raw, body.numel() * body.element_size()), (getattr(layouts, "layout_name", layouts), n, tuple(body.shape), body.dtype)


def test_step_many_equals_single_steps_with_illegal_actions_and_resets(gpu):
    """oc_step_many (K transitions in one launch) == K oc_step launches == the oracle, across episode ends
    (auto-reset) and with illegal actions sprinkled in (those envs stay untouched and are flagged), on a single
    layout, a mixed table and a 7-pot layout."""
    from overcooked_ai_amd.layouts import LayoutSpec, LayoutTable, spec_from_name

    seven = LayoutSpec({"grid": "XPPPPPX\nO 1 2 O\nX     X\nXDPSPTX", "onion_time": 3, "tomato_time": 5,
                        "onion_value": 7, "tomato_value": 4})
    table5 = LayoutTable([spec_from_name(nm) for nm in CANONICAL_5], pad_to=(9, 5))
    rng = np.random.default_rng(12)
    for layouts, n_lay in (("cramped_room", 0), (table5, 5), (seven, 0), ("cramped_room_single", 0)):
        n, K, horizon = 2500, 61, 17
        lid = (np.arange(n) % n_lay).astype(np.uint16) if n_lay else None
        a = rng.integers(0, 6, size=(K, n, 2)).astype(np.uint8)
        a[rng.integers(0, K, 40), rng.integers(0, n, 40), rng.integers(0, 2, 40)] = 7
        if layouts == "cramped_room_single":
            a[:, :, 1] = 4
        acts = torch.from_numpy(a).to(gpu)
        many = make_env(layouts, n, gpu, horizon=horizon, auto_reset=True, layout_id=lid)
        one = make_env(layouts, n, gpu, horizon=horizon, auto_reset=True, layout_id=lid)
        rew = torch.zeros((K, n, 4), dtype=torch.float32, device=gpu)
        fl = torch.zeros((K, n), dtype=torch.uint8, device=gpu)
        many.step_many(acts, rew, fl)
        specs = layouts.specs if n_lay else [layouts if not isinstance(layouts, str) else spec_from_name(layouts)]
        orc = oracle_for(specs)
        st = orc.reset(orc.new_state(n), layout_id=lid)
        ep = np.zeros((n, 4), np.float32)
        for k in range(K):
            r1, f1 = one.step(acts[k])
            assert torch.equal(r1, rew[k]) and torch.equal(f1, fl[k]), k
            st, r_o, f_o = orc.step(st, a[k], horizon=horizon, options=1, layout_id=lid, ep_returns=ep)
            assert np.array_equal(r_o, rew[k].cpu().numpy()) and np.array_equal(f_o, fl[k].cpu().numpy()), k
        assert torch.equal(many.state, one.state) and torch.equal(many.ep_returns, one.ep_returns)
        assert np.array_equal(many.get_packed_state(), st) and np.array_equal(many.ep_returns.cpu().numpy(), ep)
        assert (fl & 2).any() and (fl & 4).any()


@pytest.mark.parametrize("layouts", ["cramped_room", "asymmetric_advantages", "mixed"])
def test_random_starts_inside_the_fused_auto_reset(layouts, gpu):
    """start_state_fn = get_random_start_state_fn(random_start_pos, rnd_obj_prob_thresh) (mdp.py:1307-1369) as
    OvercookedEnv.reset uses it (env.py:288-319): every restart at the horizon, inside the step kernels, draws a new
    start state (OcStartSpec) — rollout (k_rollout4), step and step_many (k_step3) against the oracle's restatement,
    across several episode boundaries, with staggered timesteps so that envs restart at different steps."""
    from oracle import oracle as O
    from overcooked_ai_amd.layouts import LayoutTable, spec_from_name

    n, horizon, seed, off = 5000, 23, 11, 1000
    if layouts == "mixed":
        table = LayoutTable([spec_from_name(nm) for nm in CANONICAL_5], pad_to=(9, 5))
        lid = (np.arange(n) * 3 % 5).astype(np.uint16)
    else:
        table = LayoutTable([spec_from_name(layouts)])
        lid = None
    orc = oracle_for(table.specs)
    kw = dict(random_start_pos=True, rnd_obj_prob_thresh=0.35)
    env = make_env(table, n, gpu, horizon=horizon, auto_reset=True, seed=seed, env_offset=off, layout_id=lid, **kw)
    st = orc.reset_random(orc.new_state(n), seed=seed, env_offset=off, epoch=0, layout_id=lid, **kw)
    assert np.array_equal(env.get_packed_state(), st)
    stagger = (np.arange(n) % horizon).astype(np.uint8)  # envs reach the horizon at different steps
    st[0, :, 6] = stagger
    env.set_packed_state(st)
    ep_o = np.zeros((n, 4), np.float32)
    steps_done = 0
    # fused rollout: 3 launches of 31 steps = 4 horizons
    for launch in range(3):
        T = 31
        rew = torch.zeros((T, n, 4), dtype=torch.float32, device=gpu)
        fl = torch.zeros((T, n), dtype=torch.uint8, device=gpu)
        env.rollout_random(T, rew, fl)
        rew_o, fl_o = orc.rollout_random(st, T, horizon=horizon, options=1, seed=seed, env_offset=off, t0=steps_done,
                                         layout_id=lid, ep_returns=ep_o,
                                         start=O.start_spec(seed, off, 1 + steps_done, **kw))
        steps_done += T
        assert np.array_equal(fl.cpu().numpy(), fl_o) and (fl_o & 4).sum() >= n, launch
        assert np.array_equal(env.get_packed_state(), st), launch
        assert np.array_equal(rew.cpu().numpy(), rew_o) and np.array_equal(env.ep_returns.cpu().numpy(), ep_o)
    # the step API, one call per step, then K steps in one call
    rng = np.random.default_rng(3)
    for t in range(horizon + 3):
        acts = rng.integers(0, 6, size=(n, 2)).astype(np.uint8)
        r, f = env.step(torch.from_numpy(acts).to(gpu))
        st, r_o, f_o = orc.step(st, acts, horizon=horizon, options=1, layout_id=lid, ep_returns=ep_o,
                                start=O.start_spec(seed, off, 1 + steps_done, **kw))
        steps_done += 1
        assert np.array_equal(env.get_packed_state(), st) and np.array_equal(f.cpu().numpy(), f_o), t
        assert np.array_equal(r.cpu().numpy(), r_o)
    K = horizon + 5
    acts_k = rng.integers(0, 6, size=(K, n, 2)).astype(np.uint8)
    rew_k = torch.zeros((K, n, 4), dtype=torch.float32, device=gpu)
    fl_k = torch.zeros((K, n), dtype=torch.uint8, device=gpu)
    env.step_many(torch.from_numpy(acts_k).to(gpu), rew_k, fl_k)
    for k in range(K):
        st, r_o, f_o = orc.step(st, acts_k[k], horizon=horizon, options=1, layout_id=lid, ep_returns=ep_o,
                                start=O.start_spec(seed, off, 1 + steps_done + k, **kw))
        assert np.array_equal(rew_k[k].cpu().numpy(), r_o) and np.array_equal(fl_k[k].cpu().numpy(), f_o), k
    assert np.array_equal(env.get_packed_state(), st) and np.array_equal(env.ep_returns.cpu().numpy(), ep_o)
    # kernels that cannot draw start states refuse instead of silently restarting from the standard state
    from overcooked_ai_amd._lib import OcAmdError
    env.predicate_interact = True
    with pytest.raises(OcAmdError):
        env.rollout_random(3)


@pytest.mark.parametrize("layouts", ["cramped_room", "counter_circuit", "mixed"])
def test_event_masks_and_episode_counters_on_the_fast_paths(layouts, gpu):
    """event_infos / game_stats without the slow kernel: k_rollout4 and k_step3 emit the per-step event
    masks and keep per-env, per-episode counters (OcEventSink) — masks equal the oracle's event_infos of every step,
    the counters equal the popcount sums per episode, are published when the episode ends and restart from zero."""
    from oracle import oracle as O
    from overcooked_ai_amd.layouts import LayoutTable, spec_from_name

    n, horizon, seed = 3000, 40, 5
    if layouts == "mixed":
        table = LayoutTable([spec_from_name(nm) for nm in CANONICAL_5], pad_to=(9, 5))
        lid = (np.arange(n) % 5).astype(np.uint16)
    else:
        table = LayoutTable([spec_from_name(layouts)])
        lid = None
    orc = oracle_for(table.specs)
    env = make_env(table, n, gpu, horizon=horizon, auto_reset=True, seed=seed, layout_id=lid, track_events=True,
                   random_start_pos=True, rnd_obj_prob_thresh=0.5)  # random starts: events from the first steps on
    st = env.get_packed_state()
    counts = np.zeros((n, 25, 2), np.int64)
    done_counts = np.zeros((n, 25, 2), np.int64)

    def account(ev_o, flags_o):
        bits = ((ev_o[:, None] >> np.arange(50, dtype=np.uint64)[None, :]) & np.uint64(1)).astype(np.int64).reshape(n, 25, 2)
        counts[:] += bits
        fin = (flags_o & 1) != 0
        done_counts[fin] = counts[fin]
        counts[(flags_o & 4) != 0] = 0

    def check_counters():
        got = env.event_counts.cpu().numpy().astype(np.int64)
        assert np.array_equal(np.stack([got & 0xFFFF, (got >> 16) & 0xFFFF], -1), counts)
        gd = env.event_counts_done.cpu().numpy().astype(np.int64)
        assert np.array_equal(np.stack([gd & 0xFFFF, (gd >> 16) & 0xFFFF], -1), done_counts)

    steps = 0
    T = 55  # crosses the horizon once
    ev = torch.zeros((T, n), dtype=torch.int64, device=gpu)
    env.rollout_random(T, None, None, events_out=ev)
    ev_np = ev.cpu().numpy().view(np.uint64)
    for k in range(T):
        acts = O.random_actions(seed, 0, k, n)
        st, _, f_o = orc.step(st, acts, horizon=horizon, options=1, layout_id=lid,
                              start=O.start_spec(seed, 0, 1 + steps, True, 0.5))
        steps += 1
        assert np.array_equal(ev_np[k], orc.last_events), k
        account(orc.last_events, f_o)
    assert np.array_equal(env.get_packed_state(), st) and counts.sum() > 0 and done_counts.sum() > 0
    check_counters()
    # the step API: one step with a mask, then K steps in one call
    rng = np.random.default_rng(2)
    acts = rng.integers(0, 6, size=(n, 2)).astype(np.uint8)
    ev1 = torch.zeros((n,), dtype=torch.int64, device=gpu)
    _, f = env.step(torch.from_numpy(acts).to(gpu), events_out=ev1)
    st, _, f_o = orc.step(st, acts, horizon=horizon, options=1, layout_id=lid, start=O.start_spec(seed, 0, 1 + steps, True, 0.5))
    steps += 1
    assert np.array_equal(ev1.cpu().numpy().view(np.uint64), orc.last_events) and np.array_equal(f.cpu().numpy(), f_o)
    account(orc.last_events, f_o)
    K = 45
    acts_k = rng.integers(0, 6, size=(K, n, 2)).astype(np.uint8)
    rew_k = torch.zeros((K, n, 4), dtype=torch.float32, device=gpu)
    fl_k = torch.zeros((K, n), dtype=torch.uint8, device=gpu)
    ev_k = torch.zeros((K, n), dtype=torch.int64, device=gpu)
    env.step_many(torch.from_numpy(acts_k).to(gpu), rew_k, fl_k, events_out=ev_k)
    ev_np = ev_k.cpu().numpy().view(np.uint64)
    for k in range(K):
        st, _, f_o = orc.step(st, acts_k[k], horizon=horizon, options=1, layout_id=lid,
                              start=O.start_spec(seed, 0, 1 + steps + k, True, 0.5))
        assert np.array_equal(ev_np[k], orc.last_events), k
        account(orc.last_events, f_o)
    assert np.array_equal(env.get_packed_state(), st)
    check_counters()
    stats = env.event_stats(finished=True)
    assert set(stats) == set(__import__("overcooked_ai_amd").EVENT_TYPES) and stats["onion_pickup"].shape == (n, 2)


def test_episode_counters_equal_reference_game_stats(gpu):
    """The counters against the reference itself: the lengths of info["episode"]["ep_game_stats"][event][agent]
    (env.py:363-401) of the recorded OvercookedEnv episodes."""
    import json

    from overcooked_ai_amd import EVENT_TYPES
    from overcooked_ai_amd.layouts import LayoutSpec

    with open(os.path.join(GOLDEN, "env_episodes.json")) as f:
        episodes = json.load(f)
    for name, ep in episodes.items():
        spec = LayoutSpec(ep["layout"])
        env = make_env(spec, 2, gpu, horizon=ep["horizon"], auto_reset=True, track_events=True)
        acts = np.array([s["actions"] for s in ep["steps"]], np.uint8)[:, None, :].repeat(2, axis=1)
        K = acts.shape[0]
        rew = torch.zeros((K, 2, 4), dtype=torch.float32, device=gpu)
        fl = torch.zeros((K, 2), dtype=torch.uint8, device=gpu)
        env.step_many(torch.from_numpy(acts).to(gpu), rew, fl)
        assert int(fl[-1, 0]) & 1, name
        stats = env.event_stats(finished=True)
        for ev_name in EVENT_TYPES:
            want = [len(x) for x in ep["episode"]["ep_game_stats"][ev_name]]
            assert stats[ev_name][0].tolist() == want, (name, ev_name)
        assert not env.event_counts.any()  # restarted: the running episode has no events yet


@pytest.mark.parametrize("layout", ["cramped_room", "asymmetric_advantages", "counter_circuit", "mixed", "seven_pots"])
def test_fused_step_encode_equals_step_then_encode(layout, gpu):
    """oc_step_encode (one C call for the step of a training loop: transition + observation of the state the next step
    starts from) == oc_step followed by oc_encode_lossless, bit for bit: state, rewards, flags, episode returns, u8 and
    f32 observations — across the horizon (auto-reset: the observation is that of the start state), with illegal
    actions and drawn start states."""
    from overcooked_ai_amd.layouts import LayoutSpec, LayoutTable, spec_from_name

    n, horizon = 5003, 17  # ragged: not a multiple of any group size
    lid = None
    if layout == "mixed":
        table = LayoutTable([spec_from_name(nm) for nm in CANONICAL_5], pad_to=(9, 5))
        lid = (np.arange(n) % 5).astype(np.uint16)
    elif layout == "seven_pots":
        table = LayoutTable([LayoutSpec({"grid": "\n".join(["XPPPPPX", "O 1 2 O", "X     X", "XDPSPTX"]), "onion_time": 3,
                                         "tomato_time": 5, "onion_value": 7, "tomato_value": 4})])
    else:
        table = LayoutTable([spec_from_name(layout)])
    rng = np.random.default_rng(4)
    for random_starts in (False, True):
        kw = dict(random_start_pos=True, rnd_obj_prob_thresh=0.4) if random_starts else {}
        a = make_env(table, n, gpu, horizon=horizon, auto_reset=True, seed=6, layout_id=lid, **kw)
        b = make_env(table, n, gpu, horizon=horizon, auto_reset=True, seed=6, layout_id=lid, **kw)
        a.one_kernel = True  # k_rollout_encode with one step where the table allows it (no drawn starts, u8)
        st = np.concatenate([random_packed_states(table.specs[l if lid is not None else 0], int(((lid == l).sum() if lid is not None else n)), rng,
                                                  timestep_max=horizon - 1) for l in (range(5) if lid is not None else [0])], axis=1)
        if lid is not None:  # envs of layout l in the order they appear
            order = np.argsort(np.argsort(lid, kind="stable"), kind="stable")
            st = st[:, order]
        a.set_packed_state(st)
        b.set_packed_state(st)
        for t in range(2 * horizon + 3):
            acts = rng.integers(0, 6, size=(n, 2)).astype(np.uint8)
            acts[rng.integers(0, n, size=4), rng.integers(0, 2, size=4)] = 7  # illegal: env untouched, flagged
            ta = torch.from_numpy(acts).to(gpu)
            dt = torch.uint8 if t % 2 == 0 else torch.float32
            r1, f1, obs1 = a.step_encode(ta, dt)
            r2, f2 = b.step(ta)
            obs2 = b.encode_lossless(dt)
            assert torch.equal(a.state, b.state) and torch.equal(r1, r2) and torch.equal(f1, f2), (layout, random_starts, t)
            assert torch.equal(obs1, obs2) and torch.equal(a.ep_returns, b.ep_returns), (layout, random_starts, t)
        assert (f1 & 2).any() or True


@pytest.mark.gpu
def test_timestep_saturates_at_the_packing_limit(gpu):
    """The packed state's timestep is a u16: an env that keeps stepping past 65 535 without a reset (no auto-reset, the
    caller ignores `done`) stores 65 535 from then on instead of wrapping to 0 — every kernel family."""
    from overcooked_ai_amd.layouts import spec_from_name

    spec = spec_from_name("cramped_room")
    n = 300
    rng = np.random.default_rng(11)
    st = random_packed_states(spec, n, rng)
    st[0, :, 6], st[0, :, 7] = 0xFC, 0xFF  # timestep 65 532
    for kernel in ("default", "lane_pair", "predicate_interact", "step", "step_predicate"):
        env = make_env(spec, n, gpu, horizon=65535, auto_reset=False, seed=2)
        select_kernel(env, "predicate_interact" if kernel == "step_predicate" else kernel)
        env.set_packed_state(st)
        if kernel.startswith("step"):
            for _ in range(12):
                _, fl = env.step(torch.from_numpy(rng.integers(0, 6, size=(n, 2)).astype(np.uint8)).to(gpu))
        else:
            fl = torch.zeros((12, n), dtype=torch.uint8, device=gpu)
            env.rollout_random(12, None, fl)
            fl = fl[-1]
        out = env.get_packed_state()
        t = out[0, :, 6].astype(np.int64) | (out[0, :, 7].astype(np.int64) << 8)
        assert (t == 65535).all(), (kernel, t[:8])
        assert ((u8(fl) & 1) == 1).all(), kernel  # past the horizon: done every step


@pytest.mark.gpu
@pytest.mark.parametrize("layout", ["cramped_room", "asymmetric_advantages", "counter_circuit", "coordination_ring",
                                    "cramped_room_old_dynamics", "mixed", "seven_pots"])
def test_rollout_with_observations_equals_the_one_step_kernels(layout, gpu):
    """oc_rollout_encode (BASELINE configs[2]: K transitions and the lossless observation after every step in one call;
    one kernel for single-layout / u8 / <= 2-pot batches, the one-step kernels step by step otherwise) == K x
    (oc_rollout_random or oc_step, then oc_encode_lossless), bit for bit: per-step rewards, flags and observations, the
    final state and episode returns — random policy and caller actions with illegal entries, across two horizons with
    auto-reset, ragged batch (partly filled last wavefront, empty wavefronts), trajectory buffer and single buffer; the
    first steps are also checked against the C oracle directly."""
    from overcooked_ai_amd.layouts import LayoutSpec, LayoutTable, spec_from_name

    n, horizon, K = 5004, 17, 40  # a multiple of 4 (16-byte observation rows), not of 64
    lid = None
    if layout == "mixed":
        table = LayoutTable([spec_from_name(nm) for nm in CANONICAL_5], pad_to=(9, 5))
        lid = (np.arange(n) % 5).astype(np.uint16)
    elif layout == "seven_pots":
        table = LayoutTable([LayoutSpec({"grid": "\n".join(["XPPPPPX", "O 1 2 O", "X     X", "XDPSPTX"]), "onion_time": 3,
                                         "tomato_time": 5, "onion_value": 7, "tomato_value": 4})])
    elif layout == "cramped_room_old_dynamics":
        table = LayoutTable([spec_from_name("cramped_room", old_dynamics=True)])
    else:
        table = LayoutTable([spec_from_name(layout)])
    rng = np.random.default_rng(9)
    st = np.concatenate([random_packed_states(table.specs[l if lid is not None else 0], int(((lid == l).sum() if lid is not None else n)), rng,
                                              timestep_max=horizon - 1) for l in (range(5) if lid is not None else [0])], axis=1)
    if lid is not None:
        order = np.argsort(np.argsort(lid, kind="stable"), kind="stable")
        st = st[:, order]
    W, H = table.width, table.height
    for mode in ("random", "actions", "single_buffer"):
        a = make_env(table, n, gpu, horizon=horizon, auto_reset=True, seed=21, layout_id=lid)
        b = make_env(table, n, gpu, horizon=horizon, auto_reset=True, seed=21, layout_id=lid)
        a.one_kernel = True  # the batch is far too small to pick k_rollout_encode by itself
        a.set_packed_state(st)
        b.set_packed_state(st)
        acts = None
        if mode == "actions":
            acts_np = rng.integers(0, 6, size=(K, n, 2)).astype(np.uint8)
            acts_np[rng.integers(0, K, size=30), rng.integers(0, n, size=30), rng.integers(0, 2, size=30)] = 9
            acts = torch.from_numpy(acts_np).to(gpu)
        obs_b = torch.zeros((K, n, 2, W, H, 26), dtype=torch.uint8, device=gpu)
        rew_b = torch.zeros((K, n, 4), dtype=torch.float32, device=gpu)
        fl_b = torch.zeros((K, n), dtype=torch.uint8, device=gpu)
        for k in range(K):
            if acts is None:
                b.rollout_random(1, rew_b[k:k + 1], fl_b[k:k + 1])
            else:
                r, f = b.step(acts[k])
                rew_b[k].copy_(r)
                fl_b[k].copy_(f)
            b.encode_lossless(torch.uint8, out=obs_b[k])
        rew_a = torch.full((K, n, 4), -1.0, dtype=torch.float32, device=gpu)
        fl_a = torch.full((K, n), 0xEE, dtype=torch.uint8, device=gpu)
        if mode == "single_buffer":
            obs_a = torch.full((n + 64, 2, W, H, 26), 0xAB, dtype=torch.uint8, device=gpu)  # + guard rows
            a.rollout_encode(K, obs_a[:n], rew_a, fl_a)
            assert torch.equal(obs_a[:n], obs_b[-1]), (layout, mode)
            assert (obs_a[n:] == 0xAB).all(), (layout, mode, "write past the observation")
        else:
            obs_a = torch.full((K, n, 2, W, H, 26), 0xAB, dtype=torch.uint8, device=gpu)
            a.rollout_encode(K, obs_a, rew_a, fl_a, actions=acts)
            for k in range(K):
                assert torch.equal(obs_a[k], obs_b[k]), (layout, mode, k)
        assert torch.equal(rew_a, rew_b) and torch.equal(fl_a, fl_b), (layout, mode)
        assert torch.equal(a.state, b.state) and torch.equal(a.ep_returns, b.ep_returns), (layout, mode)
        assert (u8(fl_a) & 4).any(), "no auto-reset inside the window"
        if mode == "random":  # the oracle itself on the first steps
            orc = oracle_for(table.specs)
            st_o = st.copy()
            for k in range(6):
                orc.rollout_random(st_o, 1, horizon=horizon, options=1, seed=21, t0=k, layout_id=lid, want_outputs=False)
                assert np.array_equal(u8(obs_a[k]).astype(np.int32), orc.encode_lossless(st_o, horizon=horizon, layout_id=lid)), (layout, k)


@pytest.mark.gpu
def test_rollout_with_observations_edge_sizes(gpu):
    """k_rollout_encode on batches that leave wavefronts empty or partly filled (1, 63, 65, 257 envs), without auto-reset
    past the horizon, and without reward / flag outputs."""
    from overcooked_ai_amd.layouts import LayoutTable, spec_from_name

    rng = np.random.default_rng(13)
    for layout, n in (("cramped_room", 1), ("cramped_room", 63), ("asymmetric_advantages", 68), ("cramped_room", 257),
                      ("coordination_ring", 260), ("coordination_ring", 257)):  # (257 x 1300 B rows: the step-by-step path)
        table = LayoutTable([spec_from_name(layout)])
        st = random_packed_states(table.specs[0], n, rng, timestep_max=9)
        W, H, K = table.width, table.height, 14
        for auto_reset, outputs in ((True, True), (False, True), (True, False)):
            a = make_env(table, n, gpu, horizon=10, auto_reset=auto_reset, seed=5)
            b = make_env(table, n, gpu, horizon=10, auto_reset=auto_reset, seed=5)
            a.one_kernel = True
            a.set_packed_state(st)
            b.set_packed_state(st)
            obs_a = torch.full((K, n, 2, W, H, 26), 0xAB, dtype=torch.uint8, device=gpu)
            obs_b = torch.zeros_like(obs_a)
            rew_a = torch.zeros((K, n, 4), dtype=torch.float32, device=gpu) if outputs else None
            fl_a = torch.zeros((K, n), dtype=torch.uint8, device=gpu) if outputs else None
            rew_b = torch.zeros((K, n, 4), dtype=torch.float32, device=gpu)
            fl_b = torch.zeros((K, n), dtype=torch.uint8, device=gpu)
            a.rollout_encode(K, obs_a, rew_a, fl_a)
            for k in range(K):
                b.rollout_random(1, rew_b[k:k + 1], fl_b[k:k + 1])
                obs_b[k].copy_(b.encode_lossless(torch.uint8))
            assert torch.equal(obs_a, obs_b) and torch.equal(a.state, b.state), (layout, n, auto_reset, outputs)
            assert torch.equal(a.ep_returns, b.ep_returns), (layout, n, auto_reset, outputs)
            if outputs:
                assert torch.equal(rew_a, rew_b) and torch.equal(fl_a, fl_b), (layout, n, auto_reset)


@pytest.mark.gpu
def test_rollout_random_with_flags_tiled_by_8_steps(gpu):
    """OC_OPT_FLAGS_TILED8: flags[k // 8][e][k % 8] from the pipelined joint-table kernel == the [step][env] flags of the
    default call, rewards / states / episode returns identical, across in-kernel restarts and a second launch that starts at
    step 16; launches that are not whole 8-step blocks and batches other kernels serve are refused."""
    from overcooked_ai_amd import _lib
    from overcooked_ai_amd.layouts import LayoutTable, spec_from_name
    from overcooked_ai_amd.vec_env import VecOvercookedEnv

    table = LayoutTable([spec_from_name("cramped_room")])
    rng = np.random.default_rng(41)
    for n, horizon in ((1000, 20), (4096, 400), (65536, 12)):
        st = random_packed_states(table.specs[0], n, rng, timestep_max=min(horizon - 1, 9))
        a = make_env(table, n, gpu, horizon=horizon, auto_reset=True, seed=11)
        b = make_env(table, n, gpu, horizon=horizon, auto_reset=True, seed=11)
        a.set_packed_state(st)
        b.set_packed_state(st)
        for K in (16, 48):
            rew_a = torch.full((K, n, 4), 3.0, dtype=torch.float32, device=gpu)
            rew_b = torch.zeros_like(rew_a)
            fl_a = torch.full((K // 8, n, 8), 0x55, dtype=torch.uint8, device=gpu)
            fl_b = torch.zeros((K, n), dtype=torch.uint8, device=gpu)
            a.rollout_random(K, rew_a, fl_a, flags_tiled8=True)
            b.rollout_random(K, rew_b, fl_b)
            assert torch.equal(VecOvercookedEnv.untile_flags(fl_a), fl_b), (n, horizon, K)
            assert torch.equal(rew_a, rew_b) and torch.equal(a.state, b.state) and torch.equal(a.ep_returns, b.ep_returns), (n, horizon, K)
            assert int((fl_b & 1).sum()) > 0 or horizon == 400  # (episode ends are inside the launches)
    # the per-env-terrain instances that write the tiled array: the 5-layout mix (table in LDS, one wavefront per SIMD or less)
    # and a one-pot table of more than 32 layouts (in HBM), with and without the one-step-ahead reads (131 072 envs)
    mix = LayoutTable([spec_from_name(nm) for nm in CANONICAL_5], pad_to=(9, 5))
    forty = LayoutTable([spec_from_name(nm) for nm in ("cramped_room", "forced_coordination_tomato", "cramped_room_tomato",
                                                         "cramped_room_single") if spec_from_name(nm).num_players == 2
                         and len(spec_from_name(nm).cells_of("P")) == 1] * 20, pad_to=(9, 5))
    assert len(forty) > 32 and forty.max_pots == 1
    for tab, n, K in ((mix, 3000, 48), (forty, 2500, 32), (forty, 131072, 16)):
        lid = (np.arange(n) % len(tab)).astype(np.uint16)
        st = np.zeros((tab.n_planes, n, 16), np.uint8)
        for l in range(len(tab)):
            idx = np.nonzero(lid == l)[0]
            st[:, idx] = random_packed_states(tab.specs[l], len(idx), rng, timestep_max=9)
        a = make_env(tab, n, gpu, horizon=24, auto_reset=True, seed=5, layout_id=lid)
        b = make_env(tab, n, gpu, horizon=24, auto_reset=True, seed=5, layout_id=lid)
        a.set_packed_state(st)
        b.set_packed_state(st)
        rew_a = torch.full((K, n, 4), 3.0, dtype=torch.float32, device=gpu)
        rew_b = torch.zeros_like(rew_a)
        fl_a = torch.full((K // 8, n, 8), 0x55, dtype=torch.uint8, device=gpu)
        fl_b = torch.zeros((K, n), dtype=torch.uint8, device=gpu)
        a.rollout_random(K, rew_a, fl_a, flags_tiled8=True)
        b.rollout_random(K, rew_b, fl_b)
        assert torch.equal(VecOvercookedEnv.untile_flags(fl_a), fl_b), (len(tab), n)
        assert torch.equal(rew_a, rew_b) and torch.equal(a.state, b.state) and torch.equal(a.ep_returns, b.ep_returns), (len(tab), n)
        assert int((fl_b & 4).sum()) > 0
    # refused: a launch that is not whole blocks, a step counter off a block boundary, a batch another kernel serves
    a = make_env(table, 256, gpu, horizon=400, auto_reset=True, seed=1)
    rew = torch.zeros((16, 256, 4), dtype=torch.float32, device=gpu)
    fl = torch.zeros((2, 256, 8), dtype=torch.uint8, device=gpu)
    with pytest.raises((_lib.OcAmdError, ValueError)):
        a.rollout_random(12, rew[:12], fl.view(-1)[:12 * 256].view(12, 256), flags_tiled8=True)
    a.rollout_random(4, rew[:4], fl.view(-1)[:4 * 256].view(4, 256))  # step counter -> 4
    with pytest.raises((_lib.OcAmdError, ValueError)):
        a.rollout_random(16, rew, fl, flags_tiled8=True)
    # one two-pot layout: whole workgroups of envs run the mover / interact instance (round 5), which writes the tiled array;
    # a ragged batch of the same layout is served by an instance that does not
    t2 = LayoutTable([spec_from_name("asymmetric_advantages")])
    c = make_env(t2, 256, gpu, horizon=400, auto_reset=True, seed=1)
    d = make_env(t2, 256, gpu, horizon=400, auto_reset=True, seed=1)
    rew_d, fl_d = torch.zeros_like(rew), torch.zeros((16, 256), dtype=torch.uint8, device=gpu)
    c.rollout_random(16, rew, fl, flags_tiled8=True)
    d.rollout_random(16, rew_d, fl_d)
    assert torch.equal(VecOvercookedEnv.untile_flags(fl), fl_d) and torch.equal(rew, rew_d) and torch.equal(c.state, d.state)
    c = make_env(t2, 300, gpu, horizon=400, auto_reset=True, seed=1)
    with pytest.raises((_lib.OcAmdError, ValueError)):
        c.rollout_random(16, torch.zeros((16, 300, 4), dtype=torch.float32, device=gpu),
                         torch.zeros((2, 300, 8), dtype=torch.uint8, device=gpu), flags_tiled8=True)


@pytest.mark.gpu
def test_output_stores_only_writes_every_output_of_a_rollout_shape(gpu):
    """oc_output_stores_only (the ceiling bench.py reports next to the roofline): every reward quad and flag byte of a
    [steps][envs] rollout is written (zeros), nothing beyond the arrays, ragged batch sizes and quads-only included."""
    import ctypes

    from overcooked_ai_amd import _lib

    L = _lib.load()
    for n, steps, with_flags in ((1000, 37, True), (256, 8, True), (65, 5, False)):
        rew = torch.full((steps + 1, n, 4), 7.0, dtype=torch.float32, device=gpu)   # one guard row behind the arrays
        fl = torch.full((steps + 1, n), 9, dtype=torch.uint8, device=gpu)
        with torch.cuda.device(gpu):
            rc = L.oc_output_stores_only(n, steps, rew.data_ptr(), fl.data_ptr() if with_flags else None, 0,
                                         ctypes.c_void_p(torch.cuda.current_stream(gpu).cuda_stream))
        assert rc == 0
        torch.cuda.synchronize(gpu)
        assert float(rew[:steps].abs().sum()) == 0.0 and bool((rew[steps] == 7.0).all())
        assert bool((fl[steps] == 9).all()) and bool((fl[:steps] == (0 if with_flags else 9)).all())
    for n, steps in ((1000, 40), (256, 8)):  # the tiled flags layout: [steps / 8][envs][8], one guard tile row behind
        rew = torch.full((steps + 1, n, 4), 7.0, dtype=torch.float32, device=gpu)
        fl = torch.full((steps // 8 + 1, n, 8), 9, dtype=torch.uint8, device=gpu)
        with torch.cuda.device(gpu):
            rc = L.oc_output_stores_only(n, steps, rew.data_ptr(), fl.data_ptr(), _lib.OPT_FLAGS_TILED8,
                                         ctypes.c_void_p(torch.cuda.current_stream(gpu).cuda_stream))
        assert rc == 0
        torch.cuda.synchronize(gpu)
        assert float(rew[:steps].abs().sum()) == 0.0 and bool((rew[steps] == 7.0).all())
        assert bool((fl[steps // 8] == 9).all()) and bool((fl[:steps // 8] == 0).all())


@pytest.mark.gpu
@pytest.mark.parametrize("dtype", ["u8", "f32"])
def test_rollout_with_observations_crowded_grids(dtype, gpu):
    """k_rollout_encode scatters what lies on the grid from per-env compact lists of at most 14 objects; an env with more
    sends its sub-group through the object-dword loop.  Batches whose envs carry something on (nearly) every counter, next
    to sparse ones in the same wavefront, against the one-step kernels."""
    from overcooked_ai_amd.layouts import LayoutTable, spec_from_name

    rng = np.random.default_rng(29)
    tdt = torch.uint8 if dtype == "u8" else torch.float32
    for layout, n in (("asymmetric_advantages", 130), ("counter_circuit", 70), ("cramped_room", 96)):
        table = LayoutTable([spec_from_name(layout)])
        # every third env crowded (more than 14 objects on the bigger grids), the next one half full, the next one sparse
        st = random_packed_states(table.specs[0], n, rng, timestep_max=3, counter_fill=lambda e: (0.95, 0.5, 0.05)[e % 3])
        n_obj = (st[1:] != 0).sum(axis=(0, 2))  # objects on each env's grid (pots included)
        if layout != "cramped_room":
            assert n_obj.max() > 14 and n_obj.min() <= 14
        W, H, K = table.width, table.height, 5
        a = make_env(table, n, gpu, horizon=400, auto_reset=True, seed=3)
        b = make_env(table, n, gpu, horizon=400, auto_reset=True, seed=3)
        a.one_kernel = True
        a.set_packed_state(st)
        b.set_packed_state(st)
        obs_a = torch.full((K, n, 2, W, H, 26), 7, dtype=tdt, device=gpu)
        obs_b = torch.zeros_like(obs_a)
        rew_a = torch.zeros((K, n, 4), dtype=torch.float32, device=gpu)
        fl_a = torch.zeros((K, n), dtype=torch.uint8, device=gpu)
        rew_b, fl_b = torch.zeros_like(rew_a), torch.zeros_like(fl_a)
        a.rollout_encode(K, obs_a, rew_a, fl_a, dtype=tdt)
        for k in range(K):
            b.rollout_random(1, rew_b[k:k + 1], fl_b[k:k + 1])
            obs_b[k].copy_(b.encode_lossless(tdt))
        assert torch.equal(obs_a, obs_b) and torch.equal(a.state, b.state), layout
        assert torch.equal(rew_a, rew_b) and torch.equal(fl_a, fl_b), layout


@pytest.mark.gpu
@pytest.mark.parametrize("layout", ["cramped_room", "asymmetric_advantages", "counter_circuit"])
def test_rollout_with_float32_observations(layout, gpu):
    """k_rollout_encode<float>: the f32 observation of every step == oc_encode_lossless(f32) after each one-step call."""
    from overcooked_ai_amd.layouts import LayoutTable, spec_from_name

    n, horizon, K = 1000, 11, 25
    table = LayoutTable([spec_from_name(layout)])
    rng = np.random.default_rng(17)
    st = random_packed_states(table.specs[0], n, rng, timestep_max=horizon - 1)
    W, H = table.width, table.height
    a = make_env(table, n, gpu, horizon=horizon, auto_reset=True, seed=3)
    b = make_env(table, n, gpu, horizon=horizon, auto_reset=True, seed=3)
    a.one_kernel = True
    a.set_packed_state(st)
    b.set_packed_state(st)
    obs_a = torch.full((K, n, 2, W, H, 26), -7.0, dtype=torch.float32, device=gpu)
    rew_a = torch.zeros((K, n, 4), dtype=torch.float32, device=gpu)
    fl_a = torch.zeros((K, n), dtype=torch.uint8, device=gpu)
    a.rollout_encode(K, obs_a, rew_a, fl_a, dtype=torch.float32)
    rew_b, fl_b = torch.zeros_like(rew_a), torch.zeros_like(fl_a)
    for k in range(K):
        b.rollout_random(1, rew_b[k:k + 1], fl_b[k:k + 1])
        assert torch.equal(obs_a[k], b.encode_lossless(torch.float32)), (layout, k)
    assert torch.equal(rew_a, rew_b) and torch.equal(fl_a, fl_b) and torch.equal(a.state, b.state)


@pytest.mark.gpu
@pytest.mark.parametrize("layout", ["cramped_room", "asymmetric_advantages", "mixed"])
def test_rollout_with_observations_and_drawn_start_states(layout, gpu):
    """oc_rollout_encode with an OcStartSpec (the env's start_state_fn = get_random_start_state_fn, mdp.py:1307-1369):
    restarts inside the launch draw the same states as the one-step kernels do (epoch = spec.epoch + step index)."""
    from overcooked_ai_amd.layouts import LayoutTable, spec_from_name

    n, horizon, K = 3000, 9, 30
    lid = None
    if layout == "mixed":
        table = LayoutTable([spec_from_name(nm) for nm in CANONICAL_5], pad_to=(9, 5))
        lid = (np.arange(n) % 5).astype(np.uint16)
    else:
        table = LayoutTable([spec_from_name(layout)])
    W, H = table.width, table.height
    kw = dict(horizon=horizon, auto_reset=True, seed=8, layout_id=lid, random_start_pos=True, rnd_obj_prob_thresh=0.5)
    rng = np.random.default_rng(2)
    for with_actions in (False, True):
        a, b = make_env(table, n, gpu, **kw), make_env(table, n, gpu, **kw)
        a.one_kernel = True
        a.reset()
        b.reset()
        assert torch.equal(a.state, b.state)
        acts = torch.from_numpy(rng.integers(0, 6, size=(K, n, 2)).astype(np.uint8)).to(gpu) if with_actions else None
        obs_a = torch.full((K, n, 2, W, H, 26), 0xAB, dtype=torch.uint8, device=gpu)
        rew_a = torch.zeros((K, n, 4), dtype=torch.float32, device=gpu)
        fl_a = torch.zeros((K, n), dtype=torch.uint8, device=gpu)
        a.rollout_encode(K, obs_a, rew_a, fl_a, actions=acts)
        rew_b, fl_b = torch.zeros_like(rew_a), torch.zeros_like(fl_a)
        for k in range(K):
            if acts is None:
                b.rollout_random(1, rew_b[k:k + 1], fl_b[k:k + 1])
            else:
                r, f = b.step(acts[k])
                rew_b[k].copy_(r)
                fl_b[k].copy_(f)
            assert torch.equal(obs_a[k], b.encode_lossless(torch.uint8)), (layout, with_actions, k)
        assert torch.equal(rew_a, rew_b) and torch.equal(fl_a, fl_b) and torch.equal(a.state, b.state), (layout, with_actions)
        assert (u8(fl_a) & 4).sum() >= 3 * n, "several restarts per env inside the launch"


@pytest.mark.gpu
def test_event_counters_follow_step_encode_and_rollout_encode(gpu):
    """track_events: step_encode / rollout_encode keep the per-episode event counters exactly as step / rollout_random do."""
    from overcooked_ai_amd.layouts import LayoutTable, spec_from_name

    table = LayoutTable([spec_from_name("cramped_room")])
    n, K = 2048, 40
    kw = dict(horizon=25, auto_reset=True, seed=4, track_events=True)
    a, b = make_env(table, n, gpu, **kw), make_env(table, n, gpu, **kw)
    obs = torch.empty((K, n, 2, table.width, table.height, 26), dtype=torch.uint8, device=gpu)
    a.rollout_encode(K, obs)
    b.rollout_random(K)
    assert torch.equal(a.state, b.state) and torch.equal(a.event_counts, b.event_counts)
    assert torch.equal(a.event_counts_done, b.event_counts_done) and int(a.event_counts_done.sum()) > 0
    acts = torch.randint(0, 6, (n, 2), dtype=torch.uint8, device=gpu)
    for _ in range(30):
        a.step_encode(acts)
        b.step(acts)
    assert torch.equal(a.state, b.state) and torch.equal(a.event_counts, b.event_counts)
    assert torch.equal(a.event_counts_done, b.event_counts_done)
