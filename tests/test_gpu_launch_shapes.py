"""GPU parity at the shapes that actually launch: the 4 000-step launch `bench.py` times — ten episodes
per launch, nine in-kernel restarts plus the one at the last step — and the per-env-terrain instances BASELINE
configs[3] / configs[4] put on EACH of 8 GPUs (65 536 and 131 072 envs per GPU; `k_rollout4` MODE 0, layout table in LDS
or read through L2, global env offset of an inner rank), compared with the C oracle (pinned to the reference by
tests/test_oracle_golden.py): every reward quad and flag byte of every env-step, the final packed states and the
episode returns, bit for bit.  The oracle runs in 400-step chunks to bound host memory."""
import os

import numpy as np
import pytest

from helpers import CANONICAL_5

pytestmark = pytest.mark.gpu

torch = pytest.importorskip("torch")
HORIZON, T = 400, 4000


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


def _long_launch_against_oracle(gpu, table, n, lid=None, env_offset=0, seed=0, steps=T, horizon=HORIZON, start=None,
                                flags_tiled8=False, one_wavefront=False, expect_shaped=True, **env_kw):
    """flags_tiled8: the launch writes the OC_OPT_FLAGS_TILED8 layout (the instances bench.py times), untiled before the
    comparison; one_wavefront: OC_OPT_ONE_WAVEFRONT (no mover / interact split where the batch would get it)."""
    from overcooked_ai_amd.vec_env import VecOvercookedEnv

    env = VecOvercookedEnv(table, n, horizon=horizon, device=gpu, auto_reset=True, seed=seed, env_offset=env_offset,
                           layout_id=lid, **env_kw)
    env.one_wavefront = one_wavefront
    orc = _oracle(env.table.specs)
    rew = torch.zeros((steps, n, 4), dtype=torch.float32, device=gpu)
    fl = torch.zeros((steps // 8, n, 8) if flags_tiled8 else (steps, n), dtype=torch.uint8, device=gpu)
    st = env.get_packed_state().copy()
    st_o = orc.reset(orc.new_state(n), layout_id=lid)
    if start is None:
        assert np.array_equal(st, st_o)
    else:
        st_o = st
    env.rollout_random(steps, rew, fl, flags_tiled8=flags_tiled8)  # ONE launch
    if flags_tiled8:
        fl = VecOvercookedEnv.untile_flags(fl)
    ep_o = np.zeros((n, 4), np.float32)
    restarts = shaped = sparse = 0
    for c0 in range(0, steps, 400):
        k = min(400, steps - c0)
        sp = None
        if start is not None:
            from oracle import oracle as O

            sp = O.start_spec(seed=seed, env_offset=env_offset, epoch=1 + c0, **start)
        rew_o, fl_o = orc.rollout_random(st_o, k, horizon=horizon, options=1, seed=seed, env_offset=env_offset, t0=c0,
                                         layout_id=lid, ep_returns=ep_o, start=sp)
        assert np.array_equal(fl[c0:c0 + k].cpu().numpy(), fl_o), "flags differ in steps %d..%d" % (c0, c0 + k)
        assert np.array_equal(rew[c0:c0 + k].cpu().numpy(), rew_o), "rewards differ in steps %d..%d" % (c0, c0 + k)
        restarts += int(((fl_o & 4) != 0).sum())
        sparse += float(rew_o[..., :2].sum())
        shaped += float(rew_o[..., 2:].sum())
    assert np.array_equal(env.get_packed_state(), st_o), "final states differ"
    assert np.array_equal(env.ep_returns.cpu().numpy(), ep_o), "episode returns differ"
    assert restarts == n * (steps // horizon) and (shaped > 0 or not expect_shaped)
    return sparse, shaped


def test_bench_launch_shape_cramped_room_65536_x_4000(gpu):
    """Exactly bench.py's launch: 65 536 cramped_room envs from the standard start, seed 0, horizon 400, auto-reset,
    4 000 fused steps (k_rollout4, joint move table, one-step-ahead cell reads)."""
    sparse, _ = _long_launch_against_oracle(gpu, "cramped_room", 65536)
    assert sparse > 0  # soups were delivered somewhere in 262 M env-steps


def test_config4_launch_shape_131072_generated_terrains_inner_rank(gpu):
    """BASELINE configs[4] as rank 3 of 8 launches it: 131 072 envs per GPU, the reference LayoutGenerator's 4 096 terrains
    (global env e -> terrain e % 4096, table read through L2), global env offset 3 x 131 072, 4 000 fused steps."""
    from overcooked_ai_amd.layout_gen import reference_generated_layouts
    from overcooked_ai_amd.layouts import LayoutTable

    n, K, rank = 131072, 4096, 3
    table = LayoutTable(reference_generated_layouts(K))
    lid = ((np.arange(n) + rank * n) % K).astype(np.uint16)
    _long_launch_against_oracle(gpu, table, n, lid=lid, env_offset=rank * n)


def test_config3_launch_shape_five_layout_mix_65536_x_4000(gpu):
    """BASELINE configs[3] as rank 5 of 8 launches it: 65 536 envs per GPU, the five canonical layouts padded to 9x5
    (global env e -> layout e % 5, table staged in LDS), 4 000 fused steps."""
    from overcooked_ai_amd.layouts import LayoutTable, spec_from_name

    n, rank = 65536, 5
    table = LayoutTable([spec_from_name(nm) for nm in CANONICAL_5], pad_to=(9, 5))
    lid = ((np.arange(n) + rank * n) % 5).astype(np.uint16)
    sparse, _ = _long_launch_against_oracle(gpu, table, n, lid=lid, env_offset=rank * n)
    assert sparse > 0


def _mix_table_and_ids(n, rank):
    from overcooked_ai_amd.layouts import LayoutTable, spec_from_name

    table = LayoutTable([spec_from_name(nm) for nm in CANONICAL_5], pad_to=(9, 5))
    return table, ((np.arange(n) + rank * n) % 5).astype(np.uint16)


@pytest.mark.parametrize("shape", ["cramped_room_65536", "five_layout_mix_65536", "generated_4096_131072",
                                   "cramped_room_131072_two_rounds", "five_layout_mix_196608_three_rounds"])
def test_tiled_flags_launch_shapes_against_oracle(shape, gpu):
    """The OC_OPT_FLAGS_TILED8 instances — the ones every number of bench.py's default line is timed on (headline: joint-table
    kernel; configs[3]: the mover / interact kernel on the 5-layout mix; configs[4]: the one-pot per-env-terrain kernel at
    131 072 envs) — against the ORACLE at exactly those launch shapes: 4 000 fused steps, flags untiled, every reward quad and
    flag byte, final states, episode returns (they used to be compared with their [step][env] siblings only)."""
    if shape == "cramped_room_65536":
        sparse, _ = _long_launch_against_oracle(gpu, "cramped_room", 65536, flags_tiled8=True)
    elif shape == "cramped_room_131072_two_rounds":  # (round 5: the mover / interact workgroups of a bigger batch run in rounds of one per CU)
        sparse, _ = _long_launch_against_oracle(gpu, "cramped_room", 131072, flags_tiled8=True, steps=400)
    elif shape == "five_layout_mix_196608_three_rounds":
        table, lid = _mix_table_and_ids(196608, 1)
        sparse, _ = _long_launch_against_oracle(gpu, table, 196608, lid=lid, env_offset=196608, flags_tiled8=True, steps=400)
    elif shape == "five_layout_mix_65536":
        table, lid = _mix_table_and_ids(65536, 5)
        sparse, _ = _long_launch_against_oracle(gpu, table, 65536, lid=lid, env_offset=5 * 65536, flags_tiled8=True)
    else:
        from overcooked_ai_amd.layout_gen import reference_generated_layouts
        from overcooked_ai_amd.layouts import LayoutTable

        n, K, rank = 131072, 4096, 3
        lid = ((np.arange(n) + rank * n) % K).astype(np.uint16)
        sparse, _ = _long_launch_against_oracle(gpu, LayoutTable(reference_generated_layouts(K)), n, lid=lid, env_offset=rank * n,
                                                flags_tiled8=True)
    assert sparse > 0


@pytest.mark.parametrize("tiled", [False, True])
@pytest.mark.parametrize("layout", ["cramped_room", "asymmetric_advantages", "coordination_ring", "forced_coordination",
                                    "counter_circuit", "mix5", "generated_4096", "cramped_room_old", "coordination_ring_old",
                                    "asymmetric_advantages_old", "mix4_old", "marshmallow_experiment", "corridor", "long_cook_time",
                                    "small_corridor_old", "big_mix"])
def test_mover_interact_split_against_oracle(layout, tiled, gpu):
    """k_rollout5 (two wavefronts per 64 envs: a mover running ahead of an interact wavefront through a ring in LDS) on every batch
    kind it serves — cramped_room, single two-player layouts, the 5-layout table in LDS, 4 096 generated terrains read through L2,
    the same with OLD dynamics (pots that start by themselves with their third item; drawn start states bring pots that arrive
    idle and full), and grids of 65..126 cells — against the oracle over episodes of 23 steps with DRAWN start states (so that every restart exercises the
    mover's own draw of the start pose), tiled and [step][env] flags; then the same launch with OC_OPT_ONE_WAVEFRONT must agree."""
    from overcooked_ai_amd.layout_gen import reference_generated_layouts
    from overcooked_ai_amd.layouts import LayoutTable, spec_from_name

    n, rank = 4096, 2
    if layout == "mix5":
        table, lid = _mix_table_and_ids(n, rank)
    elif layout == "mix4_old":
        # (the canonical layouts whose orders all have three items: old dynamics accepts no others, mdp.py:1121-1127)
        names = ("cramped_room", "asymmetric_advantages", "coordination_ring", "forced_coordination")
        table = LayoutTable([spec_from_name(nm, old_dynamics=True) for nm in names], pad_to=(9, 5))
        lid = ((np.arange(n) + rank * n) % len(names)).astype(np.uint16)
    elif layout == "big_mix":  # 13 x 5 layouts (65 cells: 16-bit cell words, the 128-bit floor mask) in one table
        names = ("marshmallow_experiment", "inverse_marshmallow_experiment", "marshmallow_experiment_coordination", "small_corridor")
        table = LayoutTable([spec_from_name(nm) for nm in names])
        lid = ((np.arange(n) + rank * n) % len(names)).astype(np.uint16)
    elif layout.endswith("_old"):
        table, lid = LayoutTable([spec_from_name(layout[:-4], old_dynamics=True)]), None
    elif layout == "generated_4096":
        table, lid = LayoutTable(reference_generated_layouts(4096)), ((np.arange(n) * 5 + 1) % 4096).astype(np.uint16)
    else:
        table, lid = layout, None
    kw = dict(lid=lid, env_offset=rank * n, seed=11, steps=96, horizon=23, expect_shaped=layout != "corridor",  # (its pots are far away)
              start={"random_start_pos": True, "rnd_obj_prob_thresh": 0.35}, random_start_pos=True, rnd_obj_prob_thresh=0.35)
    a = _long_launch_against_oracle(gpu, table, n, flags_tiled8=tiled, **kw)
    # (single layouts get the tiled flags from the split kernel only: their one-wavefront run writes [step][env] rows)
    # (... and old dynamics or more than 64 cells in one wavefront is MODE 0: [step][env] rows as well)
    b = _long_launch_against_oracle(gpu, table, n, one_wavefront=True, **kw,
                                    flags_tiled8=tiled and lid is not None and not layout.endswith("_old") and layout != "big_mix")
    assert a == b
    # standard start states, a launch that begins in the middle of an episode (t0 = 96 from the env's own counter)
    _long_launch_against_oracle(gpu, table, n, lid=lid, env_offset=rank * n, seed=3, steps=416, horizon=HORIZON, flags_tiled8=tiled,
                                expect_shaped=layout != "corridor")


@pytest.mark.parametrize("table_kind", ["generated_4096", "canonical_5"])
def test_mover_interact_split_with_layouts_redrawn_at_every_restart(table_kind, gpu):
    """MODE 3 with regen_layout: at a restart BOTH wavefronts of an env draw the next layout — the interact wavefront records
    it and resets the grid, the mover takes its floor mask and start pose — from the same counter-based stream.  Whole
    workgroups (16 384 envs), launches of whole 8-step blocks (32 + 40 + 48 steps), horizon 23: five boundaries, layout ids,
    states, rewards, flags and returns against the oracle, standard and drawn start states, tiled and row flags."""
    from oracle import oracle as O
    from overcooked_ai_amd.layout_gen import reference_generated_layouts
    from overcooked_ai_amd.layouts import LayoutTable, spec_from_name
    from overcooked_ai_amd.vec_env import VecOvercookedEnv

    if table_kind == "generated_4096":
        table = LayoutTable(reference_generated_layouts(4096))
    else:
        table = LayoutTable([spec_from_name(nm) for nm in CANONICAL_5], pad_to=(9, 5))
    K, n, horizon, seed, off = len(table), 16384, 23, 29, 7 * 16384
    orc = _oracle(table.specs)
    for start_kw in ({}, {"random_start_pos": True, "rnd_obj_prob_thresh": 0.3}):
        lid = ((np.arange(n) * 11 + 5) % K).astype(np.uint16)
        lid_o = lid.copy()
        env = VecOvercookedEnv(table, n, horizon=horizon, device=gpu, auto_reset=True, seed=seed, env_offset=off, layout_id=lid,
                               regen_layout=True, **start_kw)
        st = env.get_packed_state().copy()
        ep_o = np.zeros((n, 4), np.float32)
        steps = 0
        for T_, tiled in ((32, False), (40, True), (48, False)):
            rew = torch.zeros((T_, n, 4), dtype=torch.float32, device=gpu)
            fl = torch.zeros((T_ // 8, n, 8) if tiled else (T_, n), dtype=torch.uint8, device=gpu)
            env.rollout_random(T_, rew, fl, flags_tiled8=tiled)
            if tiled:
                fl = VecOvercookedEnv.untile_flags(fl)
            sp = O.start_spec(seed, off, 1 + steps, regen=(0, K), **start_kw)
            rew_o, fl_o = orc.rollout_random(st, T_, horizon=horizon, options=1, seed=seed, env_offset=off, t0=steps,
                                             layout_id=lid_o, ep_returns=ep_o, start=sp)
            steps += T_
            assert np.array_equal(env.layout_ids(), lid_o), "layout ids differ after %d steps" % steps
            assert np.array_equal(fl.cpu().numpy(), fl_o) and np.array_equal(rew.cpu().numpy(), rew_o), steps
            assert np.array_equal(env.get_packed_state(), st) and np.array_equal(env.ep_returns.cpu().numpy(), ep_o)
        assert (lid_o != lid).mean() > 0.75


def test_long_launch_with_drawn_start_states_131072(gpu):
    """The big-batch (lean) joint-table instance with the env's start_state_fn drawn inside the fused auto-reset:
    131 072 cramped_room envs, 2 000 steps = five restarts per env, each from a state drawn with its own epoch."""
    _long_launch_against_oracle(gpu, "cramped_room", 131072, seed=5, steps=2000,
                                start={"random_start_pos": True, "rnd_obj_prob_thresh": 0.4},
                                random_start_pos=True, rnd_obj_prob_thresh=0.4)


@pytest.mark.parametrize("table_kind", ["generated_4096", "canonical_5"])
def test_layout_redrawn_every_episode_inside_the_fused_auto_reset(table_kind, gpu):
    """regen_layout (OvercookedEnv.reset(regen_mdp=True) over a layout generator, env.py:288-302 — the reference meaning of
    BASELINE configs[4]): every restart inside oc_rollout_random / oc_step / oc_step_many moves the env to a layout drawn from
    the table, then starts the episode there.  Layout ids, states, rewards and flags against the oracle's restatement across
    more than three episode boundaries, on the 4 096-terrain table (read through L2) and the 5-layout table (in LDS), with
    standard and with drawn start states; then an explicit reset with the same semantics."""
    from oracle import oracle as O
    from overcooked_ai_amd.layout_gen import reference_generated_layouts
    from overcooked_ai_amd.layouts import LayoutTable, spec_from_name
    from overcooked_ai_amd.vec_env import VecOvercookedEnv

    if table_kind == "generated_4096":
        table = LayoutTable(reference_generated_layouts(4096))
    else:
        table = LayoutTable([spec_from_name(nm) for nm in CANONICAL_5], pad_to=(9, 5))
    K, n, horizon, seed, off = len(table), 70000 if table_kind == "generated_4096" else 30000, 23, 17, 5000
    orc = _oracle(table.specs)
    for start_kw in ({}, {"random_start_pos": True, "rnd_obj_prob_thresh": 0.3}):
        lid = ((np.arange(n) * 7 + 3) % K).astype(np.uint16)
        lid_o = lid.copy()
        env = VecOvercookedEnv(table, n, horizon=horizon, device=gpu, auto_reset=True, seed=seed, env_offset=off, layout_id=lid,
                               regen_layout=True, **start_kw)
        st = env.get_packed_state().copy()
        if not start_kw:
            assert np.array_equal(st, orc.reset(orc.new_state(n), layout_id=lid_o))
        ep_o = np.zeros((n, 4), np.float32)
        steps = 0
        for T in (31, 40):  # the fused rollout: three boundaries (steps 23, 46, 69)
            rew = torch.zeros((T, n, 4), dtype=torch.float32, device=gpu)
            fl = torch.zeros((T, n), dtype=torch.uint8, device=gpu)
            env.rollout_random(T, rew, fl)
            sp = O.start_spec(seed, off, 1 + steps, regen=(0, K), **start_kw)
            rew_o, fl_o = orc.rollout_random(st, T, horizon=horizon, options=1, seed=seed, env_offset=off, t0=steps,
                                             layout_id=lid_o, ep_returns=ep_o, start=sp)
            steps += T
            assert np.array_equal(env.layout_ids(), lid_o), "layout ids differ after %d steps" % steps
            assert np.array_equal(fl.cpu().numpy(), fl_o) and np.array_equal(rew.cpu().numpy(), rew_o)
            assert np.array_equal(env.get_packed_state(), st) and np.array_equal(env.ep_returns.cpu().numpy(), ep_o)
        assert (lid_o != lid).mean() > 0.75 and len(np.unique(lid_o)) > min(K, 1000) * 0.9  # (5 layouts: one restart in five redraws its own)
        rng = np.random.default_rng(4)
        for t in range(horizon + 2):  # the step API, one call per step, across a fourth boundary
            acts = rng.integers(0, 6, size=(n, 2)).astype(np.uint8)
            r, f = env.step(torch.from_numpy(acts).to(gpu))
            sp = O.start_spec(seed, off, 1 + steps, regen=(0, K), **start_kw)
            st, r_o, f_o = orc.step(st, acts, horizon=horizon, options=1, layout_id=lid_o, ep_returns=ep_o, start=sp)
            steps += 1
            assert np.array_equal(f.cpu().numpy(), f_o) and np.array_equal(r.cpu().numpy(), r_o), t
        assert np.array_equal(env.layout_ids(), lid_o) and np.array_equal(env.get_packed_state(), st)
        Ks = horizon + 4  # K steps in one launch (oc_step_many), a fifth boundary
        acts_k = rng.integers(0, 6, size=(Ks, n, 2)).astype(np.uint8)
        rew_k = torch.zeros((Ks, n, 4), dtype=torch.float32, device=gpu)
        fl_k = torch.zeros((Ks, n), dtype=torch.uint8, device=gpu)
        env.step_many(torch.from_numpy(acts_k).to(gpu), rew_k, fl_k)
        for k in range(Ks):
            sp = O.start_spec(seed, off, 1 + steps + k, regen=(0, K), **start_kw)
            st, r_o, f_o = orc.step(st, acts_k[k], horizon=horizon, options=1, layout_id=lid_o, ep_returns=ep_o, start=sp)
            assert np.array_equal(rew_k[k].cpu().numpy(), r_o) and np.array_equal(fl_k[k].cpu().numpy(), f_o), k
        steps += Ks
        assert np.array_equal(env.layout_ids(), lid_o) and np.array_equal(env.get_packed_state(), st)
        # an explicit reset of a third of the envs: new layouts first, then their start states
        mask = (np.arange(n) % 3 == 1)
        epoch = env.reset_epoch
        env.reset(mask=torch.from_numpy(mask))
        O.regen_layouts(lid_o, O.start_spec(seed, off, epoch, regen=(0, K)), mask=mask.astype(np.uint8))
        if start_kw:
            st = orc.reset_random(st, seed=seed, env_offset=off, epoch=epoch, layout_id=lid_o, mask=mask.astype(np.uint8), **start_kw)
        else:
            st = orc.reset(st, layout_id=lid_o, mask=mask.astype(np.uint8))
        assert np.array_equal(env.layout_ids(), lid_o) and np.array_equal(env.get_packed_state(), st)
        # the object view follows the moved layouts
        some = env.get_states()[:5]
        assert [s.timestep for s in some] == [int(st[0, e, 6]) for e in range(5)]


@pytest.mark.parametrize("n", [8 * 256 - 48, 16 * 256 - 1, 24 * 256])
def test_xcd_contiguous_block_mapping_on_ragged_batches(n, gpu):
    """xcd_block() (round 5): grids that are a multiple of 8 workgroups deal block (b % 8) * grid / 8 + b / 8 to workgroup b, so
    the ragged last block of a batch is no longer the last workgroup.  Every kernel that takes the mapping, on batches whose
    grid is 8, 16 and 24 workgroups with a ragged or a full last block: the rollout (one layout: joint-table instances; the
    5-layout mix: per-env terrain; with and without whole workgroups, i.e. mover / interact or one wavefront), K caller-action
    steps (k_step3) and the rollout with observations (k_rollout_encode), against the C oracle."""
    from overcooked_ai_amd.layouts import LayoutTable, spec_from_name
    from overcooked_ai_amd.vec_env import VecOvercookedEnv

    steps = 48
    _long_launch_against_oracle(gpu, LayoutTable([spec_from_name("cramped_room")]), n, steps=steps, horizon=20, seed=3)
    mix = LayoutTable([spec_from_name(nm) for nm in CANONICAL_5], pad_to=(9, 5))
    lid = (np.arange(n) % 5).astype(np.uint16)
    _long_launch_against_oracle(gpu, mix, n, lid=lid, steps=steps, horizon=20, seed=4)
    if n % 256 == 0:
        _long_launch_against_oracle(gpu, mix, n, lid=lid, steps=steps, horizon=20, seed=4, flags_tiled8=True)
    # K caller-action steps in one launch + the rollout with the observation of every step, one layout
    rng = np.random.default_rng(n)
    for name in ("cramped_room", "asymmetric_advantages"):
        spec = spec_from_name(name)
        env = VecOvercookedEnv(spec, n, horizon=9, device=gpu, auto_reset=True, seed=2)
        orc = _oracle(env.table.specs)
        st = orc.reset(orc.new_state(n))
        K = 12
        acts = rng.integers(0, 6, size=(K, n, 2)).astype(np.uint8)
        rew = torch.zeros((K, n, 4), dtype=torch.float32, device=gpu)
        fl = torch.zeros((K, n), dtype=torch.uint8, device=gpu)
        env.step_many(torch.from_numpy(acts).to(gpu), rew, fl)
        for k in range(K):
            st, r_o, f_o = orc.step(st, acts[k], horizon=9, options=1)
            assert np.array_equal(rew[k].cpu().numpy(), r_o) and np.array_equal(fl[k].cpu().numpy(), f_o), (name, n, k)
        assert np.array_equal(env.get_packed_state(), st), (name, n)
        env.one_kernel = True
        obs = torch.zeros((K, n, 2, env.width, env.height, 26), dtype=torch.uint8, device=gpu)
        env.rollout_encode(K, obs, rew, fl)
        for k in range(K):
            r_o, f_o = orc.rollout_random(st, 1, horizon=9, options=1, seed=2, t0=k)  # (the Philox clock counts random-policy steps only)
            assert np.array_equal(rew[k].cpu().numpy(), r_o[0]) and np.array_equal(fl[k].cpu().numpy(), f_o[0]), (name, n, k)
            assert np.array_equal(obs[k].cpu().numpy().astype(np.int32), orc.encode_lossless(st, horizon=9)), (name, n, k)
        assert np.array_equal(env.get_packed_state(), st), (name, n)


@pytest.mark.parametrize("layouts", ["cramped_room", "asymmetric_advantages", "coordination_ring", "mix5", "cramped_room_tomato",
                                     "coordination_ring_old"])
def test_mover_interact_event_log_against_oracle(layouts, gpu):
    """k_rollout5 with the event log (track_events, whole workgroups and 8-step blocks): the per-episode counters — the entries'
    event kinds, the USEFUL_* variants from the full-pot count and the other player's hand, useful dish pick-ups and the potting
    classes from the rare branch — equal the per-episode popcounts of the oracle's event_infos, are published when an episode
    ends and restart from zero; drawn start states, three episode boundaries inside the launches; rewards and states as well."""
    from oracle import oracle as O
    from overcooked_ai_amd.layouts import LayoutTable, spec_from_name
    from overcooked_ai_amd.vec_env import VecOvercookedEnv

    n, horizon, seed, off = 2048, 37, 5, 4096
    if layouts == "mix5":
        table = LayoutTable([spec_from_name(nm) for nm in CANONICAL_5], pad_to=(9, 5))
        lid = ((np.arange(n) + off) % 5).astype(np.uint16)
    elif layouts.endswith("_old"):
        table, lid = LayoutTable([spec_from_name(layouts[:-4], old_dynamics=True)]), None
    else:
        table, lid = LayoutTable([spec_from_name(layouts)]), None
    orc = _oracle(table.specs)
    env = VecOvercookedEnv(table, n, horizon=horizon, device=gpu, auto_reset=True, seed=seed, env_offset=off, layout_id=lid,
                           track_events=True, random_start_pos=True, rnd_obj_prob_thresh=0.5)
    st = env.get_packed_state().copy()
    counts = np.zeros((n, 25, 2), np.int64)
    done_counts = np.zeros((n, 25, 2), np.int64)
    steps = 0
    for T_, tiled in ((64, False), (56, True)):
        rew = torch.zeros((T_, n, 4), dtype=torch.float32, device=gpu)
        fl = torch.zeros((T_ // 8, n, 8) if tiled else (T_, n), dtype=torch.uint8, device=gpu)
        env.rollout_random(T_, rew, fl, flags_tiled8=tiled)
        fl_np = (VecOvercookedEnv.untile_flags(fl) if tiled else fl).cpu().numpy()
        rew_np = rew.cpu().numpy()
        for k in range(T_):
            acts = O.random_actions(seed, off, steps, n)
            st, r_o, f_o = orc.step(st, acts, horizon=horizon, options=1, layout_id=lid,
                                    start=O.start_spec(seed, off, 1 + steps, True, 0.5))
            steps += 1
            assert np.array_equal(fl_np[k], f_o) and np.array_equal(rew_np[k], r_o), steps
            bits = ((orc.last_events[:, None] >> np.arange(50, dtype=np.uint64)[None, :]) & np.uint64(1)).astype(np.int64)
            counts += bits.reshape(n, 25, 2)
            fin = (f_o & 1) != 0
            done_counts[fin] = counts[fin]
            counts[(f_o & 4) != 0] = 0
        assert np.array_equal(env.get_packed_state(), st)
        got = env.event_counts.cpu().numpy().astype(np.int64)
        got2 = np.stack([got & 0xFFFF, (got >> 16) & 0xFFFF], -1)
        bad = np.argwhere(got2 != counts)
        assert len(bad) == 0, "running counters after %d steps: %d differ, first (env, event, player) %s: %d vs %d; events that differ %s, layouts %s" % (
            steps, len(bad), bad[0].tolist(), got2[tuple(bad[0])], counts[tuple(bad[0])], sorted(set(bad[:, 1].tolist())),
            sorted(set((lid[bad[:, 0]] if lid is not None else np.zeros(1, int)).tolist())))
        gd = env.event_counts_done.cpu().numpy().astype(np.int64)
        assert np.array_equal(np.stack([gd & 0xFFFF, (gd >> 16) & 0xFFFF], -1), done_counts), "published counters after %d steps" % steps
    assert done_counts.sum() > 0 and done_counts[:, 1].sum() + done_counts[:, 6].sum() > 0  # useful pick-ups were logged


def test_rollouts_without_output_arrays_equal_the_ones_with(gpu):
    """A launch without rewards / flags arrays (a rollout run for its final states, returns or event counters) takes the mover /
    interact kernel's store-free instances (round 6; before: the general one-wavefront path): states, episode returns and the
    per-episode event counters equal those of the same launch WITH output arrays — new and old dynamics, a 65-cell grid, a table
    read through L2, the event log."""
    from overcooked_ai_amd.layout_gen import reference_generated_layouts
    from overcooked_ai_amd.layouts import LayoutTable, spec_from_name
    from overcooked_ai_amd.vec_env import VecOvercookedEnv

    n, T = 4096, 808
    cases = [(LayoutTable([spec_from_name("cramped_room")]), None, {}),
             (LayoutTable([spec_from_name("coordination_ring", old_dynamics=True)]), None, {}),
             (LayoutTable([spec_from_name("marshmallow_experiment")]), None, {}),
             (LayoutTable(reference_generated_layouts(40)), (np.arange(n) * 3 % 40).astype(np.uint16), {}),
             (LayoutTable([spec_from_name("asymmetric_advantages")]), None, {"track_events": True})]
    for ci, (table, lid, kw) in enumerate(cases):
        a = VecOvercookedEnv(table, n, horizon=100, device=gpu, auto_reset=True, seed=2, layout_id=lid, **kw)
        b = VecOvercookedEnv(table, n, horizon=100, device=gpu, auto_reset=True, seed=2, layout_id=lid, **kw)
        rew = torch.zeros((T, n, 4), dtype=torch.float32, device=gpu)
        fl = torch.zeros((T, n), dtype=torch.uint8, device=gpu)
        a.rollout_random(T)
        b.rollout_random(T, rew, fl)
        assert torch.equal(a.state, b.state) and torch.equal(a.ep_returns, b.ep_returns), ci
        if kw:
            assert torch.equal(a.event_counts, b.event_counts) and torch.equal(a.event_counts_done, b.event_counts_done)
            assert int(a.event_counts_done.sum()) > 0


def test_long_launches_off_the_eight_step_grid_split_around_whole_blocks(gpu):
    """A launch whose first step or length is not a multiple of 8 runs its whole 8-step blocks on the mover / interact kernel and
    the few steps around them on the one-wavefront instances (three launches inside one call): rewards, flags, states, returns,
    event counters and layout re-draws equal the same calls served by the one-wavefront instances alone (OC_OPT_ONE_WAVEFRONT) —
    drawn start states and per-episode layout re-draws included, whose draws are keyed by the step's epoch."""
    from overcooked_ai_amd.layout_gen import reference_generated_layouts
    from overcooked_ai_amd.layouts import LayoutTable, spec_from_name
    from overcooked_ai_amd.vec_env import VecOvercookedEnv

    n = 2048
    gen = LayoutTable(reference_generated_layouts(24))
    cases = [(LayoutTable([spec_from_name("asymmetric_advantages")]), None, dict(random_start_pos=True, rnd_obj_prob_thresh=0.3)),
             (LayoutTable([spec_from_name("cramped_room")]), None, dict(track_events=True)),
             (gen, (np.arange(n) * 5 % 24).astype(np.uint16), dict(regen_layout=True, random_start_pos=True))]
    for ci, (table, lid, kw) in enumerate(cases):
        a = VecOvercookedEnv(table, n, horizon=37, device=gpu, auto_reset=True, seed=6, layout_id=lid, **kw)
        b = VecOvercookedEnv(table, n, horizon=37, device=gpu, auto_reset=True, seed=6, layout_id=lid, **kw)
        b.one_wavefront = True
        for T in (3, 1000, 13, 407, 800):  # first steps 0, 3, 1003, 1016, 1423: off the grid from the second call on
            ra, fa = torch.zeros((T, n, 4), dtype=torch.float32, device=gpu), torch.zeros((T, n), dtype=torch.uint8, device=gpu)
            rb, fb = torch.zeros_like(ra), torch.zeros_like(fa)
            a.rollout_random(T, ra, fa)
            b.rollout_random(T, rb, fb)
            assert torch.equal(ra, rb) and torch.equal(fa, fb), (ci, T)
            assert torch.equal(a.state, b.state) and torch.equal(a.ep_returns, b.ep_returns), (ci, T)
        if "track_events" in kw:
            assert torch.equal(a.event_counts, b.event_counts) and torch.equal(a.event_counts_done, b.event_counts_done)
        if "regen_layout" in kw:
            assert np.array_equal(a.layout_ids(), b.layout_ids())
