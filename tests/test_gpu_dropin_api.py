"""The reference-shaped API (OvercookedGridworld / OvercookedEnv / Overcooked) running on the HIP kernels,
against whole episodes recorded from the reference's own OvercookedEnv (tests/golden/env_episodes.json) and the
reference's test fixtures.  These tests read like testing/overcooked_test.py on purpose."""
import json
import os

import numpy as np
import pytest

from conftest import GOLDEN

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def episodes():
    with open(os.path.join(GOLDEN, "env_episodes.json")) as f:
        return json.load(f)


def test_env_episodes_match_reference(episodes):
    from overcooked_ai_amd import Action, OvercookedEnv, OvercookedGridworld
    from overcooked_ai_amd import state as S
    from overcooked_ai_amd.layouts import LayoutSpec

    for name, ep in episodes.items():
        mdp = OvercookedGridworld.from_spec(LayoutSpec(ep["layout"]))
        env = OvercookedEnv.from_mdp(mdp, horizon=ep["horizon"], info_level=0)
        info = None
        for t, step in enumerate(ep["steps"]):
            ja = [Action.INDEX_TO_ACTION[a] for a in step["actions"]]
            state, reward, done, info = env.step(ja)
            assert reward == step["reward"] and done == step["done"], (name, t)
            assert info["sparse_r_by_agent"] == step["sparse_r_by_agent"]
            assert info["shaped_r_by_agent"] == step["shaped_r_by_agent"]
        assert S.canonical_state_dict(env.state) == S.canonical_state_dict(ep["final_state"])
        got, exp = info["episode"], ep["episode"]
        assert got["ep_length"] == exp["ep_length"] and got["ep_sparse_r"] == exp["ep_sparse_r"]
        assert got["ep_shaped_r"] == exp["ep_shaped_r"]
        assert list(got["ep_sparse_r_by_agent"]) == exp["ep_sparse_r_by_agent"]
        assert list(got["ep_shaped_r_by_agent"]) == exp["ep_shaped_r_by_agent"]
        for k, v in exp["ep_game_stats"].items():  # event timestep lists per agent + cumulative rewards
            assert [list(x) if not np.isscalar(x) else x for x in np.asarray(got["ep_game_stats"][k], dtype=object)] == v \
                or list(got["ep_game_stats"][k]) == v, (name, k)
        with pytest.raises(AssertionError):
            env.step([Action.STAY, Action.STAY])  # env.py:255: stepping a done env
        enc = np.stack(env.lossless_state_encoding_mdp(env.state))
        assert list(enc.shape) == ep["encoding_shape"] and enc.dtype == np.int64
        nz = [[int(i) for i in idx] + [int(enc[tuple(idx)])] for idx in np.argwhere(enc)]
        assert nz == ep["final_encoding_nonzero"]


def test_gridworld_api_mirrors_reference(small_fixtures):
    from overcooked_ai_amd import Action, Direction, OvercookedGridworld, OvercookedState, PlayerState
    from overcooked_ai_amd import state as S

    mdp = OvercookedGridworld.from_layout_name("mdp_test")
    assert mdp.shape == (5, 4) and mdp.num_players == 2 and mdp.num_pots == 2
    assert list(mdp.get_lossless_state_encoding_shape()) == [5, 4, 26]
    start = mdp.get_standard_start_state()
    assert S.canonical_state_dict(start) == S.canonical_state_dict(small_fixtures["mdp_test_start_state"]["expected_state"])
    # test_transitions_and_environment (overcooked_test.py:468-514)
    bad_state = OvercookedState([PlayerState((0, 0), Direction.SOUTH), PlayerState((3, 1), Direction.SOUTH)], {})
    with pytest.raises(AssertionError):
        mdp.get_state_transition(bad_state, [Action.STAY, Action.STAY])
    with pytest.raises(ValueError):
        mdp.get_state_transition(start, [Action.STAY, "jump"])
    case = small_fixtures["mdp_test_one_transition"]
    before = start.to_dict()
    new_state, infos = mdp.get_state_transition(start, [Direction.NORTH, Direction.EAST])
    assert start.to_dict() == before  # input state untouched
    exp = OvercookedState.from_dict(case["expected_state"])
    assert new_state.time_independent_equal(exp) and sum(infos["sparse_reward_by_agent"]) == case["expected_reward"]
    assert set(infos) == {"event_infos", "sparse_reward_by_agent", "shaped_reward_by_agent"}
    assert len(infos["event_infos"]) == 25 and not any(any(v) for v in infos["event_infos"].values())
    # old dynamics (overcooked_test.py:527-563)
    with pytest.raises(AssertionError):
        OvercookedGridworld.from_layout_name("mdp_test", old_dynamics=True)
    for lname, new_cooking, old_cooking in (("old_dynamics_cook_test", True, False), ("old_dynamics_put_test", False, True)):
        new_mdp = OvercookedGridworld.from_layout_name(lname, old_dynamics=False)
        old_mdp = OvercookedGridworld.from_layout_name(lname, old_dynamics=True)
        s_new, _ = new_mdp.get_state_transition(new_mdp.start_state, [Action.INTERACT])
        s_old, _ = old_mdp.get_state_transition(old_mdp.start_state, [Action.INTERACT])
        assert s_new.get_object((2, 0)).is_cooking == new_cooking
        assert s_old.get_object((2, 0)).is_cooking == old_cooking
    # encoding symmetry under player reversal (overcooked_test.py:1112-1128)
    st = new_state.deepcopy()
    a = mdp.lossless_state_encoding(st, horizon=400)
    b = mdp.lossless_state_encoding(st.deepcopy().reverse_players(), horizon=400)
    assert np.array_equal(a[0], b[1]) and np.array_equal(a[1], b[0]) and a[0].shape == (5, 4, 26)


def test_batched_transitions_and_gym_wrapper():
    from overcooked_ai_amd import Action, Overcooked, OvercookedEnv, OvercookedGridworld

    mdp = OvercookedGridworld.from_layout_name("asymmetric_advantages")
    start = mdp.get_standard_start_state()
    states, infos = mdp.get_state_transitions([start] * 6, [[Action.INDEX_TO_ACTION[a], Action.STAY] for a in range(6)])
    singles = [mdp.get_state_transition(start, [Action.INDEX_TO_ACTION[a], Action.STAY])[0] for a in range(6)]
    assert states == singles and len(infos) == 6
    base_env = OvercookedEnv.from_mdp(mdp, horizon=20, info_level=0)
    env = Overcooked(base_env, base_env.lossless_state_encoding_mdp)
    assert env.action_space.n == 6 and env.observation_space.shape == (9, 5, 26)
    np.random.seed(3)
    obs = env.reset()
    assert set(obs) == {"both_agent_obs", "overcooked_state", "other_agent_env_idx"}
    done, n = False, 0
    while not done:
        obs, reward, done, info = env.step((np.random.randint(6), np.random.randint(6)))
        n += 1
        assert info["policy_agent_idx"] == env.agent_idx
        ob_main = base_env.lossless_state_encoding_mdp(obs["overcooked_state"])[env.agent_idx]
        assert np.array_equal(obs["both_agent_obs"][0], ob_main)
    assert n == 20 and info["episode"]["ep_length"] == 20 and info["episode"]["policy_agent_idx"] == env.agent_idx
    # env_params / copy() (env.py:229-242): a copy is a fresh episode over the same generator and parameters
    assert base_env.env_params == {"start_state_fn": None, "horizon": 20, "info_level": 0, "num_mdp": 1}
    twin = base_env.copy()
    assert twin.mdp is base_env.mdp and twin.horizon == 20 and twin.state.timestep == 0 and base_env.state.timestep == 20
    assert twin.is_done() is False and base_env.is_done() is True
    with pytest.raises(AssertionError):
        OvercookedEnv(mdp)  # a gridworld instead of a generator function


def test_featurize_state_api_matches_reference_golden():
    """OvercookedEnv.featurize_state_mdp (env.py:282) on states of the reference's own featurization test."""
    from overcooked_ai_amd import OvercookedEnv, OvercookedGridworld
    from overcooked_ai_amd import state as S

    with open(os.path.join(GOLDEN, "featurize_manifest.json")) as f:
        man = json.load(f)
    from overcooked_ai_amd.layouts import LayoutSpec

    spec = LayoutSpec(man["cramped_room_none"]["layout"])
    mdp = OvercookedGridworld.from_spec(spec)
    env = OvercookedEnv.from_mdp(mdp, horizon=400, info_level=0)
    d = np.load(os.path.join(GOLDEN, "ref_greedy_rollouts.npz"))
    assert mdp.get_featurize_state_shape() == (96,)
    for e in (0, 57, 203, 399, 1500, 1999):
        state = S.unpack_states(spec, d["states"][:, e:e + 1])[0]
        f0, f1 = env.featurize_state_mdp(state)
        assert f0.shape == (96,) and np.array_equal(f0, d["features"][e, 0]) and np.array_equal(f1, d["features"][e, 1])
    batch = mdp.featurize_states(S.unpack_states(spec, d["states"][:, :64]))
    assert np.array_equal(batch, d["features"][:64])


def test_potential_function_api_matches_reference_golden():
    """OvercookedGridworld.potential_function / get_state_transition(display_phi=True) (mdp.py:1421-1429, 2920)."""
    from overcooked_ai_amd import Action, OvercookedEnv, OvercookedGridworld
    from overcooked_ai_amd import state as S
    from overcooked_ai_amd.layouts import LayoutSpec

    with open(os.path.join(GOLDEN, "potential_manifest.json")) as f:
        man = json.load(f)
    spec = LayoutSpec(dict(man["counter_circuit"]["layout"]))
    mdp = OvercookedGridworld.from_spec(spec)
    d = np.load(os.path.join(GOLDEN, "potential_counter_circuit.npz"))
    states = S.unpack_states(spec, d["states"][:, :32])
    for e in (0, 7, 31):
        assert mdp.potential_function(states[e], None) == d["phi"][0, e]
        assert mdp.potential_function(states[e], None, gamma=0.9) == d["phi"][1, e]
    assert np.array_equal(mdp.potential_functions(states), d["phi"][0, :32])
    nxt, infos = mdp.get_state_transition(states[3], (Action.STAY, Action.INTERACT), display_phi=True)
    assert infos["phi_s"] == d["phi"][0, 3] and infos["phi_s_prime"] == mdp.potential_function(nxt, None)
    env = OvercookedEnv.from_mdp(mdp, horizon=400, info_level=0)
    _, _, _, info = env.step((Action.STAY, Action.STAY), display_phi=True)
    assert info["phi_s"] == info["phi_s_prime"] == mdp.potential_function(mdp.get_standard_start_state(), None)


def _multi_agent_fixture():
    with open(os.path.join(GOLDEN, "multi_agent.json")) as f:
        man = json.load(f)
    for name, case in man.items():
        z = np.load(os.path.join(GOLDEN, "multi_agent_%s.npz" % name))
        yield name, case, z["obs"].astype(np.float32), z["obs_len"]


def test_vec_multi_agent_matches_reference_episodes():
    """VecOvercookedMultiAgent (oc_step + oc_potential + oc_shape_rewards + masked oc_reset + encode per batched step)
    on the same recorded episodes, and its restart-on-done behaviour on a batch."""
    import torch

    from overcooked_ai_amd import VecOvercookedMultiAgent
    from overcooked_ai_amd.layouts import LayoutSpec

    dev = torch.device("cuda:0")
    for name, case, obs, obs_len in _multi_agent_fixture():
        spec = LayoutSpec(dict(case["layout"]))
        kw = case["kwargs"]
        env = VecOvercookedMultiAgent(spec, 3, horizon=case["horizon"], reward_shaping_factor=kw["reward_shaping_factor"],
                                      reward_shaping_horizon=kw["reward_shaping_horizon"], use_phi=kw["use_phi"], device=dev)
        row = 0
        for ep in case["episodes"]:
            ob = env.reset().cpu().numpy()
            for j, a in enumerate(ep["agents"]):
                if a.startswith("ppo"):
                    assert np.array_equal(ob[1, j].ravel(), obs[row, j, :obs_len[row, j]])
            row += 1
            for t, st in enumerate(ep["steps"]):
                env.set_reward_shaping_factor(st["factor"])
                acts = torch.tensor([st["actions"]] * 3, dtype=torch.uint8, device=dev)
                ob, rew, done, infos = env.step(acts)
                assert rew.dtype == torch.float64 and rew[2].tolist() == st["rewards"], (name, t)
                assert bool(done[0]) == st["done"]
                if kw["use_phi"]:
                    assert float(infos["phi_s_prime"][1]) == st["phi_s_prime"]
                if not st["done"]:  # after the last step the batch already shows the restarted episode
                    o = ob.cpu().numpy()
                    bc = env.observations("bc").cpu().numpy()
                    for j, a in enumerate(ep["agents"]):
                        want = obs[row, j, :obs_len[row, j]]
                        got = o[1, j].ravel() if a.startswith("ppo") else bc[1, j]
                        assert np.array_equal(got, want), (name, t, a)
                else:
                    assert float(infos["ep_returns"][0, 2:4].sum()) == ep["ep_shaped_r"]
                    fresh = VecOvercookedMultiAgent(spec, 3, horizon=case["horizon"], use_phi=False, device=dev)
                    assert torch.equal(env.venv.state, fresh.venv.state) and not env.venv.ep_returns.any()
                row += 1
    # a batch with staggered episodes: envs finish at different steps and restart on their own
    env = VecOvercookedMultiAgent("cramped_room", 4096, horizon=30, reward_shaping_factor=1.0, use_phi=True, device=dev)
    stagger = torch.arange(4096, device=dev) % 30
    hdr = env.venv.state[0]
    hdr[:, 6] = stagger.to(torch.uint8)  # timestep low byte
    gen = torch.Generator(device=dev).manual_seed(0)
    n_done = 0
    for t in range(45):
        acts = torch.randint(0, 6, (4096, 2), dtype=torch.uint8, device=dev, generator=gen)
        ob, rew, done, infos = env.step(acts)
        n_done += int(done.sum())
        assert ob.shape == (4096, 2, 5, 4, 26) and torch.isfinite(rew).all()
        ts = env.venv.state[0][:, 6].long() + 256 * env.venv.state[0][:, 7].long()
        assert int(ts.max()) < 30
    assert n_done == int(((stagger + 45) // 30).sum())


def test_vec_multi_agent_random_start_states():
    """start_state_fn = get_random_start_state_fn(...) for the batched training env: the first states and every
    restart are drawn by oc_reset_random (checked against the oracle's restatement), phi is carried correctly."""
    import torch

    from oracle import oracle as O
    from overcooked_ai_amd import VecOvercookedMultiAgent
    from overcooked_ai_amd.layouts import spec_from_name
    from overcooked_ai_amd.potential import potential_params

    dev = torch.device("cuda:0")
    n, horizon = 2048, 9
    spec = spec_from_name("coordination_ring")
    env = VecOvercookedMultiAgent(spec, n, horizon=horizon, reward_shaping_factor=1.0, use_phi=True, device=dev, seed=4,
                                  random_start_pos=True, rnd_obj_prob_thresh=0.4)
    orc = O.Oracle([O.mdp_from_layout_dict(spec.to_layout_dict())])
    pp = [potential_params(spec, 0.99)]
    st = orc.reset_random(orc.new_state(n), seed=4, epoch=0, random_start_pos=True, rnd_obj_prob_thresh=0.4)
    assert np.array_equal(env.venv.get_packed_state(), st)
    assert np.array_equal(env.phi_cur.cpu().numpy(), O.potential(orc, st, pp))
    gen = torch.Generator(device=dev).manual_seed(1)
    epoch = 1
    for t in range(2 * horizon + 3):
        acts = torch.randint(0, 6, (n, 2), dtype=torch.uint8, device=dev, generator=gen)
        phi_before = env.phi_cur.clone()
        ob, rew, done, infos = env.step(acts)
        st, r, f = orc.step(st, acts.cpu().numpy(), horizon=horizon, options=0)
        phi_next = O.potential(orc, st, pp)
        want = (r[:, 0] + r[:, 1]).astype(np.float64)[:, None] + 1.0 * (phi_next - phi_before.cpu().numpy())[:, None]
        assert np.array_equal(rew.cpu().numpy(), np.repeat(want, 2, axis=1)), t
        d = (f & 1).astype(np.uint8)
        assert np.array_equal(done.cpu().numpy(), d)
        st = orc.reset_random(st, seed=4, epoch=epoch, random_start_pos=True, rnd_obj_prob_thresh=0.4, mask=d)
        epoch += 1
        assert np.array_equal(env.venv.get_packed_state(), st), t
        assert np.array_equal(env.phi_cur.cpu().numpy(), O.potential(orc, st, pp)), t
        assert np.array_equal(ob.cpu().numpy().astype(np.int32), orc.encode_lossless(st, horizon=horizon)), t


def test_infinite_order_bonus_delivery():
    """tutorial_3 ships order_bonus = inf: delivering its bonus order pays an infinite sparse reward in the reference
    (get_recipe_value, mdp.py:1595-1602) — the drop-in API must return inf, not crash on int(inf)."""
    from overcooked_ai_amd import Action, Direction, OvercookedEnv, OvercookedGridworld
    from overcooked_ai_amd.state import OvercookedState, PlayerState, SoupState

    mdp = OvercookedGridworld.from_layout_name("tutorial_3")
    ct = mdp.spec.recipe_time((1, 2))
    soup = SoupState((5, 1), ["onion", "tomato", "tomato"], ct, ct)
    state = OvercookedState([PlayerState((5, 3), Direction.NORTH), PlayerState((5, 1), Direction.EAST, soup)], {},
                            bonus_orders=mdp.start_bonus_orders, all_orders=mdp.start_all_orders)
    new_state, infos = mdp.get_state_transition(state, (Action.STAY, Action.INTERACT))
    assert infos["sparse_reward_by_agent"] == [0, float("inf")] and not new_state.players[1].has_object()
    # the same through the batched general path and through OvercookedEnv.step
    _, infos_b = mdp.get_state_transitions([state], [(Action.STAY, Action.INTERACT)])
    assert infos_b[0]["sparse_reward_by_agent"] == [0, float("inf")]
    env = OvercookedEnv.from_mdp(mdp, horizon=10, info_level=0)
    env.state = state
    _, reward, _, info = env.step((Action.STAY, Action.INTERACT))
    assert reward == float("inf") and info["sparse_r_by_agent"] == [0, float("inf")]


def test_vec_multi_agent_event_counters():
    """The batched training env keeps the per-agent event counters RLlib reports (rllib.py:453-483, env.py:382-401):
    fused kernel (k_train_step) and the general sequence, against the oracle's event_infos summed per episode."""
    import torch

    from oracle import oracle as O
    from overcooked_ai_amd import VecOvercookedMultiAgent
    from overcooked_ai_amd.layouts import LayoutSpec, spec_from_name

    dev = torch.device("cuda:0")
    seven = LayoutSpec({"grid": "XPPPPPX\nO 1 2 O\nX     X\nXDPSPTX", "onion_time": 3, "tomato_time": 5,
                        "onion_value": 7, "tomato_value": 4})
    for spec in (spec_from_name("cramped_room"), seven):
        n, horizon = 1500, 30
        env = VecOvercookedMultiAgent(spec, n, horizon=horizon, use_phi=False, reward_shaping_factor=1.0, device=dev,
                                      track_events=True)
        orc = O.Oracle([O.mdp_from_layout_dict(spec.to_layout_dict())])
        st = orc.reset(orc.new_state(n))
        counts = np.zeros((n, 25, 2), np.int64)
        done_counts = np.zeros((n, 25, 2), np.int64)
        gen = torch.Generator(device=dev).manual_seed(3)
        for t in range(75):
            acts = torch.randint(0, 6, (n, 2), dtype=torch.uint8, device=dev, generator=gen)
            env.step(acts)
            st, _, f = orc.step(st, acts.cpu().numpy(), horizon=horizon, options=0)
            ev = orc.last_events
            counts += ((ev[:, None] >> np.arange(50, dtype=np.uint64)[None, :]) & np.uint64(1)).astype(np.int64).reshape(n, 25, 2)
            fin = (f & 1) != 0
            done_counts[fin] = counts[fin]
            counts[fin] = 0
            st = orc.reset(st, mask=fin.astype(np.uint8))
        assert np.array_equal(env.venv.get_packed_state(), st)
        got = env.venv.event_counts.cpu().numpy().astype(np.int64)
        assert np.array_equal(np.stack([got & 0xFFFF, (got >> 16) & 0xFFFF], -1), counts) and counts.sum() > 0
        gd = env.venv.event_counts_done.cpu().numpy().astype(np.int64)
        assert np.array_equal(np.stack([gd & 0xFFFF, (gd >> 16) & 0xFFFF], -1), done_counts) and done_counts.sum() > 0


class _ReplayAgentPair:
    """Stands where the reference's AgentPair(RandomAgent, RandomAgent) stood when the fixture was recorded
    (agents/agent.py:137, 223): `joint_action(state)` hands back that episode's recorded (action, info) pairs."""

    def __init__(self, steps):
        self.steps, self.t, self.mdp, self.resets = steps, 0, None, 0

    def joint_action(self, state):
        from overcooked_ai_amd import Action

        st = self.steps[self.t]
        self.t += 1
        return tuple((Action.INDEX_TO_ACTION[a], {k: np.asarray(v) for k, v in info.items()})
                     for a, info in zip(st["action"], st["agent_infos"]))

    def set_mdp(self, mdp):
        self.mdp = mdp

    def reset(self):
        self.t, self.resets = 0, self.resets + 1


def test_run_agents_replays_the_reference_agent_pair_episodes():
    """Episodes the reference's own AgentPair(RandomAgent(all_actions=True) x 2) played through the reference's
    OvercookedEnv.run_agents (recorded by oracle/gen_golden.py gen_agent_pair_rollouts): the mirror env's run_agents, fed
    the same joint actions through the agent-pair interface, yields the same trajectory rows, totals, episode summary —
    and get_rollouts the codebase's trajectory dict of it."""
    from overcooked_ai_amd import OvercookedEnv, OvercookedGridworld
    from overcooked_ai_amd import state as S
    from overcooked_ai_amd.env import DEFAULT_TRAJ_KEYS
    from overcooked_ai_amd.layouts import LayoutSpec

    with open(os.path.join(GOLDEN, "agent_pair_rollouts.json")) as f:
        fixtures = json.load(f)
    for name, fx in fixtures.items():
        mdp = OvercookedGridworld.from_spec(LayoutSpec(fx["layout"]))
        env = OvercookedEnv.from_mdp(mdp, horizon=fx["horizon"], info_level=0)
        pair = _ReplayAgentPair(fx["steps"])
        traj, t_elapsed, total_sparse, total_shaped = env.run_agents(pair, include_final_state=True)
        assert traj.shape == (len(fx["steps"]) + 1, 5) and t_elapsed == fx["t_elapsed"]
        assert total_sparse == fx["total_sparse"] and total_shaped == fx["total_shaped"]
        for row, st in zip(traj[:-1], fx["steps"]):
            s_t, a_t, r_t, done, info = row
            assert S.canonical_state_dict(s_t) == S.canonical_state_dict(st["state"])
            assert r_t == st["reward"] and done == st["done"]
            assert info["sparse_r_by_agent"] == st["sparse_r_by_agent"] and info["shaped_r_by_agent"] == st["shaped_r_by_agent"]
            assert [ai["action_probs"].tolist() for ai in info["agent_infos"]] == [ai["action_probs"] for ai in st["agent_infos"]]
        assert S.canonical_state_dict(traj[-1][0]) == S.canonical_state_dict(fx["final_state"]) and traj[-1][1] == (None, None)
        ep = traj[-2][4]["episode"]
        assert ep["ep_length"] == fx["ep_length"]
        for key, want in fx["ep_game_stats"].items():
            got = ep["ep_game_stats"][key]
            assert (got.tolist() if hasattr(got, "tolist") else [list(x) for x in got]) == want, (name, key)
        # the same through get_rollouts (two games of the same recorded actions: the env restarts from the standard state)
        env.reset(regen_mdp=False)
        pair.reset()
        out = env.get_rollouts(pair, 2, info=False)
        assert sorted(out) == sorted(DEFAULT_TRAJ_KEYS) and pair.mdp is mdp and pair.resets == 3
        assert out["ep_returns"].tolist() == [fx["total_sparse"]] * 2 and out["ep_lengths"].tolist() == [fx["ep_length"]] * 2
        assert out["ep_states"].shape == (2, fx["ep_length"]) and out["ep_rewards"][1].tolist() == [s["reward"] for s in fx["steps"]]
        assert out["mdp_params"][0]["layout_name"] == mdp.layout_name and out["env_params"][0]["horizon"] == fx["horizon"]


def test_env_planner_properties_and_limits():
    """OvercookedEnv.mp is the MotionPlanner of the current mdp (rebuilt when reset regenerates the mdp); .mlam carries it
    and names what it cannot do; execute_plan leaves the env at a start state like the reference's; a step at the
    packing limit of the timestep raises instead of freezing it."""
    from overcooked_ai_amd import Action, Direction, OvercookedEnv, OvercookedGridworld

    mdp = OvercookedGridworld.from_layout_name("cramped_room")
    env = OvercookedEnv.from_mdp(mdp, horizon=400, info_level=0)
    mp = env.mp
    assert env.mlam.motion_planner is mp and env.mp is mp
    start = mdp.get_standard_start_state()
    cost, where = mp.min_cost_to_feature(start.players[0].pos_and_or, mdp.get_pot_locations(), with_argmin=True)
    assert where == (2, 0) and cost == 3  # the reference planner's figure for the start pose: 2 motion actions + the interact
    with pytest.raises(NotImplementedError):
        env.mlam.joint_ml_actions(start)
    env.reset(regen_mdp=True)
    assert env.mp is not mp
    end, done = env.execute_plan(start, [(Direction.NORTH, Action.STAY)] * 3)
    assert end.timestep == 3 and not done and env.state.timestep == 0
    assert "^0" in repr(env) or "0" in repr(env)
    late = start.deepcopy()
    late.timestep = 65535
    env2 = OvercookedEnv.from_mdp(mdp, info_level=0)  # the reference's default horizon
    env2.state = late
    with pytest.raises(ValueError, match="16 bits"):
        env2.step((Action.STAY, Action.STAY))


def test_overcooked_multi_agent_matches_reference_episodes():
    """The RLlib environment class (human_aware_rl/rllib/rllib.py:112-438) replayed against episodes recorded from the
    reference: agent roles (same np.random stream), per-agent observations (lossless / featurize_state), rewards
    sparse + factor * (phi' - phi | shaped) as exact Python floats, dones, annealed factors."""
    from overcooked_ai_amd import OvercookedEnv, OvercookedGridworld, OvercookedMultiAgent
    from overcooked_ai_amd.layouts import LayoutSpec

    for name, case, obs, obs_len in _multi_agent_fixture():
        mdp = OvercookedGridworld.from_spec(LayoutSpec(dict(case["layout"])))
        base_env = OvercookedEnv.from_mdp(mdp, horizon=case["horizon"], info_level=0)
        np.random.seed(2024)
        env = OvercookedMultiAgent(base_env, **{k: (list(map(tuple, v)) if isinstance(v, list) else v)
                                                for k, v in case["kwargs"].items()})
        row, total = 0, 0
        for ep in case["episodes"]:
            ob = env.reset()
            agents = list(env.curr_agents)
            assert agents == ep["agents"], name
            for j, a in enumerate(agents):
                assert ob[a].dtype == np.float32 and np.array_equal(ob[a].ravel(), obs[row, j, :obs_len[row, j]])
            row += 1
            for st in ep["steps"]:
                ob, rew, dones, infos = env.step({agents[0]: st["actions"][0], agents[1]: st["actions"][1]})
                total += 1
                assert [rew[a] for a in agents] == st["rewards"], (name, total)
                assert dones["__all__"] == st["done"] and env.reward_shaping_factor == st["factor"] and env.bc_factor == st["bc_factor"]
                assert infos[agents[0]].get("phi_s") == st["phi_s"] and infos[agents[0]].get("phi_s_prime") == st["phi_s_prime"]
                for j, a in enumerate(agents):
                    assert np.array_equal(ob[a].ravel(), obs[row, j, :obs_len[row, j]]), (name, total, a)
                row += 1
                if total % 7 == 0:
                    env.anneal_reward_shaping_factor(total)
                    env.anneal_bc_factor(total)
            assert infos[agents[0]]["episode"]["ep_shaped_r"] == ep["ep_shaped_r"]
    cfg = dict(OvercookedMultiAgent.DEFAULT_CONFIG)
    env = OvercookedMultiAgent.from_config(cfg)
    assert sorted(env.reset()) == ["ppo_0", "ppo_1"] and env.base_env.horizon == 400


def test_urgency_layer_follows_the_callers_horizon_beyond_16_bits():
    """mdp.py:2446-2447: the urgency layer is `horizon - timestep < 40` on the horizon the caller passes.  With the
    reference's default MAX_HORIZON (env.py: horizon = 1e10, "never done") it must stay off at t = 65 500, although the
    packed timestep is a u16 and an earlier version clamped the horizon to 65 535."""
    from overcooked_ai_amd import OvercookedEnv, OvercookedGridworld
    from overcooked_ai_amd.env import MAX_HORIZON

    mdp = OvercookedGridworld.from_layout_name("cramped_room")
    state = mdp.get_standard_start_state()
    state.timestep = 65500
    W, H = mdp.shape
    for horizon, urgent in ((MAX_HORIZON, 0), (10 ** 6, 0), (65540, 0), (65539, 1), (65535, 1), (400, 1)):
        enc = mdp.lossless_state_encoding(state, horizon=horizon)
        for p in range(2):
            assert enc[p].shape == (W, H, 26) and int(enc[p][:, :, 25].sum()) == urgent * W * H, (horizon, p)
    env = OvercookedEnv.from_mdp(mdp, info_level=0)  # the reference's default horizon
    env.state = state
    assert int(np.stack(env.lossless_state_encoding_mdp(state))[:, :, :, 25].sum()) == 0
    env400 = OvercookedEnv.from_mdp(mdp, horizon=400, info_level=0)
    state.timestep = 361
    assert int(np.stack(env400.lossless_state_encoding_mdp(state))[:, :, :, 25].sum()) == 2 * W * H
    state.timestep = 360
    assert int(np.stack(env400.lossless_state_encoding_mdp(state))[:, :, :, 25].sum()) == 0


def test_one_player_env_fixed_plan():
    """The reference's test_one_player_env (overcooked_test.py:1187-1194): a fixed plan on cramped_room_single, horizon 12
    (a FixedPlanAgent stays once its plan is used up), ends at ((2, 1), NORTH)."""
    from overcooked_ai_amd import Action, Direction, OvercookedEnv, OvercookedGridworld

    stay, interact = Action.STAY, Action.INTERACT
    n, s, e, w = Direction.NORTH, Direction.SOUTH, Direction.EAST, Direction.WEST
    mdp = OvercookedGridworld.from_layout_name("cramped_room_single")
    assert mdp.num_players == 1
    env = OvercookedEnv.from_mdp(mdp, horizon=12, info_level=0)
    plan = [stay, w, w, e, e, n, e, interact, w, n, interact]
    done, t = False, 0
    while not done:
        _, _, done, _ = env.step((plan[t] if t < len(plan) else stay,))
        t += 1
    assert t == 12 and env.state.players_pos_and_or == (((2, 1), (0, -1)),)
    env4 = OvercookedEnv.from_mdp(OvercookedGridworld.from_layout_name("multiplayer_schelling"), horizon=4, info_level=0)
    with pytest.raises(ValueError, match="1- and 2-player"):  # four players: outside the packed format, refused loudly
        env4.step((stay,) * 4)


def test_lazy_states_become_plain_states_the_moment_somebody_looks(episodes):
    """OvercookedEnv.step hands back state._LazyState objects (the kernel's packed bytes; players / objects are built on the
    first look).  An env whose states nobody looks at and an env whose states are read — and changed — every step give the
    same trajectory as the reference's; a timestep written into a state nobody has looked at is honoured as well."""
    import copy

    from overcooked_ai_amd import Action, OvercookedEnv, OvercookedGridworld
    from overcooked_ai_amd import state as S
    from overcooked_ai_amd.layouts import LayoutSpec

    name, ep = sorted(episodes.items())[0]
    mdp = OvercookedGridworld.from_spec(LayoutSpec(ep["layout"]))
    blind = OvercookedEnv.from_mdp(mdp, horizon=ep["horizon"], info_level=0)
    seeing = OvercookedEnv.from_mdp(mdp, horizon=ep["horizon"], info_level=0)
    n_lazy = 0
    for t, step in enumerate(ep["steps"]):
        ja = [Action.INDEX_TO_ACTION[a] for a in step["actions"]]
        s_b, r_b, d_b, _ = blind.step(ja)
        s_s, r_s, d_s, _ = seeing.step(ja)
        n_lazy += type(s_b) is S._LazyState
        assert isinstance(s_b, S.OvercookedState) and s_b.timestep == t + 1
        assert s_s.players is not None and type(s_s) is S.OvercookedState  # the look makes it a plain state
        assert (r_b, d_b) == (r_s, d_s) == (step["reward"], step["done"]), (name, t)
        if t % 5 == 0:  # rebuild the seen state from its dict: the next step then packs objects nobody got from the kernel
            seeing.state = S.OvercookedState.from_dict(copy.deepcopy(seeing.state.to_dict()))
    assert n_lazy >= len(ep["steps"]) - 3  # (the first calls go through plain launches before the mailbox opens: lazy as well)
    assert S.canonical_state_dict(blind.state) == S.canonical_state_dict(ep["final_state"])
    assert S.canonical_state_dict(seeing.state) == S.canonical_state_dict(ep["final_state"])
    # a state nobody looked at, with its clock changed: the transition starts from that clock
    mdp2 = OvercookedGridworld.from_spec(LayoutSpec(ep["layout"]))
    env = OvercookedEnv.from_mdp(mdp2, horizon=400, info_level=0)
    for _ in range(4):
        s, *_ = env.step((Action.STAY, Action.STAY))
    assert type(s) is S._LazyState
    s.timestep = 100
    nxt, _ = mdp2.get_state_transition(s, (Action.STAY, Action.INTERACT))
    assert nxt.timestep == 101 and type(s) is S._LazyState
    # ... and one somebody changed after looking: the change is what the kernel steps
    p0 = s.players[0]
    assert type(s) is S.OvercookedState
    free = [c for c in mdp2.get_valid_player_positions() if c not in s.player_positions][0]
    p0.position = free
    nxt2, _ = mdp2.get_state_transition(s, (Action.STAY, Action.STAY))
    assert nxt2.players[0].position == free and nxt2.timestep == 101
    assert copy.deepcopy(nxt2) == nxt2 and hash(copy.deepcopy(nxt2)) == hash(nxt2)
