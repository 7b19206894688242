"""Generate tests/golden/* by running the REAL reference (imported from /root/reference).

TEST INFRASTRUCTURE ONLY; run in the build container:   python -m oracle.gen_golden
The outputs are committed.  They pin both the C oracle (CPU tests) and the HIP kernels (GPU tests)
against the reference's own behaviour, because /root/reference cannot travel to the GPU box.

Fixtures (all states in the packed wire format of include/oc_amd.h, converted from the reference's
`OvercookedState.to_dict()` by overcooked_ai_amd.state.pack_state_dict; a sample of raw dicts is kept
so that the converter itself is pinned):

  ref_mdp_dynamics.npz      the reference's own 1500-step trajectory, data/testing/test_mdp_dynamics/expected.json
                            (overcooked_test.py:516-525), replayed through get_state_transition here first
  ref_small_fixtures.json   test_transitions_and_environments/expected.json, test_start_positions/expected.json,
                            the scripted bonus-order episode of overcooked_test.py:607-998 (final sparse reward 50),
                            old-dynamics cases (overcooked_test.py:527-563), K1..K11 of SURVEY.md §8c
  transitions_<cfg>.npz     randomized (state, joint action) -> (next state, sparse, shaped, events) through
                            OvercookedGridworld.get_state_transition (mdp.py:1375), plus lossless encodings
                            (mdp.py:2385) of a subset of the states
  rollouts_<cfg>.npz        400-step episodes from the standard start state under Philox-drawn actions through
                            OvercookedEnv.step (env.py:244)
"""
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

from oracle import ref_harness  # noqa: E402
from oracle import oracle as oracle_mod  # noqa: E402
from overcooked_ai_amd import layouts as L  # noqa: E402
from overcooked_ai_amd import state as S  # noqa: E402

GOLDEN = os.path.join(ROOT, "tests", "golden")
REF_TESTING = "/root/reference/src/overcooked_ai_py/data/testing"

R = ref_harness.load()
Action, Direction = R.Action, R.Direction

# (fixture name, layout name, overrides)
CONFIGS = [
    ("cramped_room", "cramped_room", {}),
    ("asymmetric_advantages", "asymmetric_advantages", {}),
    ("coordination_ring", "coordination_ring", {}),
    ("forced_coordination", "forced_coordination", {}),
    ("counter_circuit", "counter_circuit", {}),
    ("mdp_test", "mdp_test", {}),
    ("cramped_room_old_dynamics", "cramped_room", {"old_dynamics": True}),
    ("bonus_order_test", "bonus_order_test", {}),
    ("cramped_room_tomato", "cramped_room_tomato", {}),
    ("cramped_room_single", "cramped_room_single", {}),
    ("cramped_room_padded_9x5", "cramped_room", {"pad_to": (9, 5)}),
]

EVENT_TYPES = list(R.EVENT_TYPES)


def make_ref_mdp(layout_name, overrides):
    overrides = dict(overrides)
    pad_to = overrides.pop("pad_to", None)
    spec = L.spec_from_name(layout_name, **overrides)
    if pad_to:
        spec = spec.padded(*pad_to)
        d = L.read_layout_dict(layout_name)
        d.pop("grid")
        mdp = R.OvercookedGridworld.from_grid(spec.grid_rows(), base_layout_params=d, params_to_overwrite=overrides)
    else:
        mdp = R.OvercookedGridworld.from_layout_name(layout_name, **overrides)
    return spec, mdp


def activate(mdp):
    """Recipe is process-global in the reference (mdp.py:221-336): re-configure on every MDP switch."""
    R.Recipe.configure(mdp.recipe_config)


def events_mask(infos, n_players):
    m = 0
    for k, name in enumerate(EVENT_TYPES):
        for p in range(n_players):
            if infos["event_infos"][name][p]:
                m |= 1 << (2 * k + p)
    return m


def random_ref_state(mdp, spec, rng):
    """A random VALID state of `mdp` built from reference objects, inside the packed domain."""
    floor = mdp.get_valid_player_positions()
    idx = rng.choice(len(floor), size=mdp.num_players, replace=False)
    players = []

    def rnd_ings():
        n = int(rng.integers(1, 4))
        names = ["onion", "tomato"] if mdp.get_tomato_dispenser_locations() or rng.random() < 0.2 else ["onion"]
        return [str(rng.choice(names)) for _ in range(n)]

    def ready_soup(pos):
        ings = rnd_ings()
        s = R.SoupState(pos, [R.ObjectState(n, pos) for n in ings], cooking_tick=0)
        s._cooking_tick = s.cook_time
        return s

    def rnd_obj(pos, allow_soup=True):
        r = rng.random()
        if r < 0.3:
            return R.ObjectState("onion", pos)
        if r < 0.45:
            return R.ObjectState("tomato", pos)
        if r < 0.75 or not allow_soup:
            return R.ObjectState("dish", pos)
        return ready_soup(pos)

    # poses that face a non-floor cell; feature cells (dispensers, pots, serving) weighted 4x over counters
    facing = []
    for pos in floor:
        for o in Direction.ALL_DIRECTIONS:
            t = mdp.get_terrain_type_at_pos(Action.move_in_direction(pos, o))
            if t != " ":
                facing.extend([(pos, o)] * (1 if t == "X" else 4))
    taken = set()
    for i in idx:
        pos = floor[int(i)]
        o = Direction.ALL_DIRECTIONS[int(rng.integers(0, 4))]
        if rng.random() < 0.75:
            cand = [f for f in facing if f[0] not in taken]
            pos, o = cand[int(rng.integers(0, len(cand)))]
        elif pos in taken:
            pos = [f for f in floor if f not in taken][0]
        taken.add(pos)
        held = rnd_obj(pos) if rng.random() < 0.65 else None
        players.append(R.PlayerState(pos, o, held))
    objects = {}
    p_counter = rng.choice([0.0, 0.1, 0.35])
    for pos in mdp.get_counter_locations():
        if rng.random() < p_counter:
            objects[pos] = rnd_obj(pos)
    for pos in mdp.get_pot_locations():
        r = rng.random()
        if r < 0.25:
            continue
        ings = rnd_ings()
        soup = R.SoupState(pos, [R.ObjectState(n, pos) for n in ings], cooking_tick=0)
        ct = soup.cook_time
        r2 = rng.random()
        if r2 < 0.4:
            soup._cooking_tick = -1
        elif r2 < 0.6:
            soup._cooking_tick = ct
        elif r2 < 0.75:
            soup._cooking_tick = max(0, ct - 1)
        else:
            soup._cooking_tick = int(rng.integers(0, ct + 1))
        objects[pos] = soup
    timestep = int(rng.integers(0, 400))
    return R.OvercookedState(players, objects, bonus_orders=mdp.start_bonus_orders, all_orders=mdp.start_all_orders,
                             timestep=timestep)


def gen_transitions(name, layout_name, overrides, n_random, n_walk_eps, seed):
    spec, mdp = make_ref_mdp(layout_name, overrides)
    activate(mdp)
    rng = np.random.default_rng(seed)
    n_planes = 1 + (spec.width * spec.height + 15) // 16
    states, actions, nexts, rewards, events, enc_idx, encs = [], [], [], [], [], [], []
    sample_dicts = []

    def record(state, ja_idx):
        ja = [Action.INDEX_TO_ACTION[a] for a in ja_idx]
        ns, infos = mdp.get_state_transition(state, ja)
        states.append(state.to_dict())
        nexts.append(ns.to_dict())
        actions.append(list(ja_idx))
        rewards.append(list(infos["sparse_reward_by_agent"]) + [0] * (2 - mdp.num_players)
                       + list(infos["shaped_reward_by_agent"]) + [0] * (2 - mdp.num_players))
        events.append(events_mask(infos, mdp.num_players))
        return ns

    # (a) independent random states, interact-heavy action distribution
    for i in range(n_random):
        st = random_ref_state(mdp, spec, rng)
        ja = [int(rng.choice(6, p=[0.1, 0.1, 0.1, 0.1, 0.1, 0.5])) for _ in range(mdp.num_players)]
        ja += [4] * (2 - mdp.num_players)
        record(st, ja)
        if mdp.num_players == 2 and i % 8 == 0:
            horizon = 400
            enc = mdp.lossless_state_encoding(st, horizon=horizon)
            enc_idx.append(len(states) - 1)
            encs.append(np.stack(enc).astype(np.int16))
    # (b) random walks from random states (uniform actions) — correlated multi-step coverage
    for ep in range(n_walk_eps):
        st = random_ref_state(mdp, spec, rng) if ep % 2 else mdp.get_standard_start_state()
        for t in range(60):
            ja = [int(rng.integers(0, 6)) for _ in range(mdp.num_players)] + [4] * (2 - mdp.num_players)
            st = record(st, ja)
    n = len(states)
    packed_in = np.zeros((n_planes, n, 16), np.uint8)
    packed_out = np.zeros((n_planes, n, 16), np.uint8)
    for e in range(n):
        S.pack_state_dict(spec, states[e], packed_in, e)
        S.pack_state_dict(spec, nexts[e], packed_out, e)
        # converter round trip against the raw reference dict
        assert S.canonical_state_dict(S.unpack_state_dict(spec, packed_out, e)) == S.canonical_state_dict(nexts[e]), \
            (nexts[e], S.unpack_state_dict(spec, packed_out, e))
    for e in list(range(0, n, max(1, n // 12)))[:12]:
        sample_dicts.append({"index": e, "state": states[e], "next_state": nexts[e]})
    np.savez_compressed(
        os.path.join(GOLDEN, "transitions_%s.npz" % name),
        state_in=packed_in, state_out=packed_out, actions=np.array(actions, np.uint8),
        rewards=np.array(rewards, np.float64), events=np.array(events, np.uint64),
        enc_index=np.array(enc_idx, np.int64),
        enc=(np.stack(encs) if encs else np.zeros((0, 2, spec.width, spec.height, 26), np.int16)),
        enc_horizon=np.array(400),
    )
    return {"layout": spec.to_layout_dict(), "n": n, "samples": sample_dicts}


def gen_rollouts(name, layout_name, overrides, n_envs, seed):
    """Full 400-step episodes through the reference's OvercookedEnv.step with Philox-drawn actions."""
    spec, mdp = make_ref_mdp(layout_name, overrides)
    activate(mdp)
    horizon = 400
    n_planes = 1 + (spec.width * spec.height + 15) // 16
    rewards = np.zeros((horizon, n_envs, 4), np.float64)
    finals = []
    checkpoints = np.zeros((4, n_planes, n_envs, 16), np.uint8)  # states after 100, 200, 300, 400 steps
    for e in range(n_envs):
        env = R.OvercookedEnv.from_mdp(mdp, horizon=horizon, info_level=0)
        env._mp = object()  # never compute/pickle a MotionPlanner into /root/reference (env.py:102-115)
        for t in range(horizon):
            a = oracle_mod.random_actions(seed, e, t, 1)[0]
            ja = [Action.INDEX_TO_ACTION[int(a[p])] for p in range(mdp.num_players)]
            ns, r, done, info = env.step(ja)
            rewards[t, e, 0:mdp.num_players] = info["sparse_r_by_agent"]
            rewards[t, e, 2:2 + mdp.num_players] = info["shaped_r_by_agent"]
            assert done == (t == horizon - 1)
            if (t + 1) % 100 == 0:
                S.pack_state_dict(spec, ns.to_dict(), checkpoints[(t + 1) // 100 - 1], e)
        finals.append(env.state.to_dict())
    np.savez_compressed(os.path.join(GOLDEN, "rollouts_%s.npz" % name), rewards=rewards, checkpoints=checkpoints,
                        seed=np.array(seed), horizon=np.array(horizon))
    return {"layout": spec.to_layout_dict(), "n_envs": n_envs, "seed": seed}


def gen_ref_trajectory():
    """The reference's own golden trajectory (overcooked_test.py:516-525)."""
    d = json.load(open(os.path.join(REF_TESTING, "test_mdp_dynamics", "expected.json")))
    spec, mdp = make_ref_mdp("mdp_test", {})
    activate(mdp)
    ep_states = d["ep_states"][0]
    ep_actions = d["ep_actions"][0]
    ep_rewards = d["ep_rewards"][0]
    n = len(ep_states)
    n_planes = 1 + (spec.width * spec.height + 15) // 16
    packed = np.zeros((n_planes, n, 16), np.uint8)
    acts = np.zeros((n, 2), np.uint8)
    shaped = np.zeros((n, 2), np.float64)
    sparse = np.zeros((n, 2), np.float64)
    n_ok = 0
    for t in range(n):
        S.pack_state_dict(spec, ep_states[t], packed, t)
        ja = [a if isinstance(a, str) else tuple(a) for a in ep_actions[t]]
        acts[t] = [Action.ACTION_TO_INDEX[a] for a in ja]
        st = R.OvercookedState.from_dict(ep_states[t])
        ns, infos = mdp.get_state_transition(st, ja)
        sparse[t] = infos["sparse_reward_by_agent"]
        shaped[t] = infos["shaped_reward_by_agent"]
        assert sum(infos["sparse_reward_by_agent"]) == ep_rewards[t]
        if t + 1 < n:
            assert ns == R.OvercookedState.from_dict(ep_states[t + 1]), t
            n_ok += 1
    print("reference golden trajectory replayed through the live reference: %d/%d transitions match" % (n_ok, n - 1))
    np.savez_compressed(os.path.join(GOLDEN, "ref_mdp_dynamics.npz"), states=packed, actions=acts,
                        ep_rewards=np.array(ep_rewards, np.float64), sparse=sparse, shaped=shaped)
    samples = [{"index": t, "state": ep_states[t]} for t in (0, 100, 500, 1000, 1499)]
    return {"layout": spec.to_layout_dict(), "n": n, "samples": samples}


def gen_small_fixtures():
    out = {}
    # --- test_transitions_and_environments (overcooked_test.py:468-514) and test_start_positions (398) ---
    spec, mdp = make_ref_mdp("mdp_test", {})
    activate(mdp)
    exp = json.load(open(os.path.join(REF_TESTING, "test_transitions_and_environments", "expected.json")))
    start = mdp.get_standard_start_state()
    ns, infos = mdp.get_state_transition(start, [Direction.NORTH, Direction.EAST])
    assert ns.time_independent_equal(R.OvercookedState.from_dict(exp["state"]))
    exp_start = json.load(open(os.path.join(REF_TESTING, "test_start_positions", "expected.json")))
    assert R.OvercookedState.from_dict(exp_start) == start
    out["mdp_test_one_transition"] = {
        "layout": spec.to_layout_dict(), "state": start.to_dict(), "actions": [0, 2],
        "expected_state": exp["state"], "expected_reward": exp["reward"],
    }
    out["mdp_test_start_state"] = {"layout": spec.to_layout_dict(), "expected_state": exp_start}

    # --- scripted episode of test_potential_function ending in a bonus delivery worth 50 (overcooked_test.py:994-998) ---
    n, s, e, w = Direction.NORTH, Direction.SOUTH, Direction.EAST, Direction.WEST
    stay, interact = Action.STAY, Action.INTERACT
    # Re-derived plan: P0 fetches onion/tomato/onion for the pot at (2,0), cooks, plates and serves.
    plan = _bonus_plan(mdp, n, s, e, w, stay, interact)
    st = mdp.get_standard_start_state()
    traj = []
    total = 0
    for ja in plan:
        nxt, infos = mdp.get_state_transition(st, ja)
        traj.append({"actions": [Action.ACTION_TO_INDEX[a] for a in ja],
                     "sparse": list(infos["sparse_reward_by_agent"]), "shaped": list(infos["shaped_reward_by_agent"]),
                     "next_state": nxt.to_dict()})
        total += sum(infos["sparse_reward_by_agent"])
        st = nxt
    assert total == 50, total
    out["mdp_test_bonus_episode"] = {"layout": spec.to_layout_dict(), "start_state": mdp.get_standard_start_state().to_dict(),
                                     "steps": traj, "total_sparse": total}

    # --- old dynamics (overcooked_test.py:527-563); 1-player layouts with a start_state ---
    for lname in ("old_dynamics_cook_test", "old_dynamics_put_test"):
        for old in (False, True):
            spec_o, mdp_o = make_ref_mdp(lname, {"old_dynamics": old})
            activate(mdp_o)
            st0 = mdp_o.start_state
            ns, infos = mdp_o.get_state_transition(st0, [interact])
            out["%s_old%d" % (lname, int(old))] = {
                "layout": spec_o.to_layout_dict(), "state": st0.to_dict(), "actions": [5, 4],
                "expected_state": ns.to_dict(), "is_cooking": bool(ns.get_object((2, 0)).is_cooking),
            }

    # --- K1..K11 micro cases, expected values re-derived from the live reference ---
    out["micro"] = _micro_cases()
    return out


def _bonus_plan(mdp, n, s, e, w, stay, interact):
    """mdp_test grid:  XXPXX / O  2O / T1  T / XDPSX ; P0 starts (1,2), P1 stays parked at (3,1).
    P0 cooks onion+onion+tomato in the pot at (2,0) (cook time 2+2+2) and serves it: the bonus order
    pays (10+10+5)*2 = 50, the figure asserted at overcooked_test.py:994-998."""
    p0 = [
        n, w, interact,        # (1,1) facing the onion dispenser (0,1): take onion
        e, n, interact,        # (2,1) facing pot (2,0): pot it
        w, interact,           # back to (1,1), still facing W: second onion
        e, n, interact,        # pot it
        s, w, interact,        # (2,2) -> (1,2) facing the tomato dispenser (0,2): take tomato
        e, n, interact,        # (2,2) -> (2,1) facing the pot: pot it
        interact,              # start cooking
        s, w, s, interact,     # (2,2) -> (1,2), face the dish dispenser (1,3): take dish
        e, n, interact,        # (2,2) -> (2,1): soup is ready, plate it
        s, e, s, interact,     # (2,2) -> (3,2), face the serving location (3,3): deliver
    ]
    return [(a, stay) for a in p0]


def _micro_cases():
    cases = []
    n, s, e, w = 0, 1, 2, 3
    STAY, INT = 4, 5

    def soup(pos, ings, tick):
        return R.SoupState(pos, [R.ObjectState(i, pos) for i in ings], cooking_tick=tick)

    def ready(pos, ings):
        sp = soup(pos, ings, 0)
        sp._cooking_tick = sp.cook_time
        return sp

    def run(label, lname, overrides, players, objects, ja, timestep=0, encode=False, n_more_stay=0):
        spec, mdp = make_ref_mdp(lname, overrides)
        activate(mdp)
        ps = []
        for (pos, o, held) in players:
            h = None
            if held is not None:
                h = held(pos)
            ps.append(R.PlayerState(pos, Direction.ALL_DIRECTIONS[o], h))
        objs = {}
        for pos, mk in objects.items():
            objs[pos] = mk(pos)
        st = R.OvercookedState(ps, objs, bonus_orders=mdp.start_bonus_orders, all_orders=mdp.start_all_orders,
                               timestep=timestep)
        rec = {"label": label, "layout": spec.to_layout_dict(), "state": st.to_dict(), "actions": list(ja)}
        if encode:
            enc = mdp.lossless_state_encoding(st, horizon=400)
            rec["encoding_layer_sums_p0"] = [int(v) for v in enc[0].sum(axis=(0, 1))]
            rec["encoding_layer_sums_p1"] = [int(v) for v in enc[1].sum(axis=(0, 1))]
            rec["encoding_nonzero_p0"] = [[int(i) for i in idx] + [int(enc[0][tuple(idx)])] for idx in np.argwhere(enc[0])]
        ns, infos = mdp.get_state_transition(st, [Action.INDEX_TO_ACTION[a] for a in ja])
        rec["expected_state"] = ns.to_dict()
        rec["sparse"] = list(infos["sparse_reward_by_agent"])
        rec["shaped"] = list(infos["shaped_reward_by_agent"])
        more = []
        cur = ns
        for _ in range(n_more_stay):
            cur, _i = mdp.get_state_transition(cur, [Action.STAY, Action.STAY])
            more.append(cur.to_dict())
        if more:
            rec["after_more_stay"] = more[-1]
            rec["n_more_stay"] = n_more_stay
        cases.append(rec)

    onion = lambda pos: R.ObjectState("onion", pos)
    dish = lambda pos: R.ObjectState("dish", pos)
    cr = "cramped_room"
    # K1: interact starts cooking; tick 1 after the step; ready after 19 more
    run("K1", cr, {}, [((2, 1), n, None), ((3, 2), s, None)], {(2, 0): lambda p: soup(p, ["onion"] * 3, -1)}, (INT, STAY),
        n_more_stay=19)
    # K2 / K3: collisions
    run("K2", cr, {}, [((1, 1), n, None), ((3, 1), n, None)], {}, (e, w))
    run("K3a", cr, {}, [((1, 1), n, None), ((2, 1), n, None)], {}, (e, w))
    run("K3b", cr, {}, [((1, 1), n, None), ((2, 1), n, None)], {}, (e, STAY))
    run("K3c", cr, {}, [((1, 1), n, None), ((2, 1), n, None)], {}, (e, e))
    run("K3d", cr, {}, [((1, 1), n, None), ((2, 1), n, None)], {}, (w, n))
    # K4: both take onions
    run("K4", cr, {}, [((1, 1), w, None), ((3, 1), e, None)], {}, (INT, INT))
    # K5: deliveries
    run("K5a", cr, {}, [((1, 1), n, None), ((3, 2), s, lambda p: ready(p, ["onion"] * 3))], {}, (STAY, INT))
    run("K5b", cr, {}, [((1, 1), n, None), ((3, 2), s, lambda p: ready(p, ["onion"] * 2))], {}, (STAY, INT))
    # K6: shared counter, sequential interact order
    fc = "forced_coordination"
    run("K6a", fc, {}, [((1, 2), e, onion), ((3, 2), w, None)], {}, (INT, INT))
    run("K6b", fc, {}, [((1, 2), e, None), ((3, 2), w, onion)], {}, (INT, INT))
    # K7: stale pot_states
    run("K7a", cr, {}, [((2, 1), n, dish), ((1, 2), s, None)], {(2, 0): lambda p: ready(p, ["onion"] * 3)}, (INT, INT))
    run("K7b", cr, {}, [((2, 1), n, dish), ((1, 2), s, None)], {}, (INT, INT))
    # K8: 3-idle pot not counted for dish usefulness
    run("K8a", cr, {}, [((2, 1), n, None), ((1, 2), s, None)], {(2, 0): lambda p: soup(p, ["onion"] * 3, -1)}, (STAY, INT))
    run("K8b", cr, {}, [((2, 1), n, None), ((1, 2), s, None)], {(2, 0): lambda p: soup(p, ["onion"] * 2, -1)}, (STAY, INT))
    # K9: no-ops at pots
    run("K9a", cr, {}, [((2, 1), n, dish), ((3, 2), s, None)], {(2, 0): lambda p: soup(p, ["onion"] * 3, 3)}, (INT, STAY))
    run("K9b", cr, {}, [((2, 1), n, onion), ((3, 2), s, None)], {(2, 0): lambda p: soup(p, ["onion"] * 3, -1)}, (INT, STAY))
    # K10: old dynamics
    run("K10a", cr, {"old_dynamics": True}, [((2, 1), n, onion), ((3, 2), s, None)],
        {(2, 0): lambda p: soup(p, ["onion"] * 2, -1)}, (INT, STAY))
    run("K10b", cr, {"old_dynamics": True}, [((2, 1), n, None), ((3, 2), s, None)],
        {(2, 0): lambda p: soup(p, ["onion"] * 2, -1)}, (INT, STAY))
    # K11: encoding
    run("K11", cr, {}, [((2, 1), n, dish), ((3, 2), s, None)],
        {(2, 0): lambda p: soup(p, ["onion"] * 2, 5), (0, 2): onion}, (STAY, STAY), timestep=365, encode=True)
    return cases


def gen_layout_luts():
    """For every 1-/2-player layout of the reference: the quantities overcooked_ai_amd.layouts.compile_layout
    flattens, taken from the live reference objects (Recipe.time mdp.py:163, get_recipe_value mdp.py:1595,
    terrain_mtx, start_player_positions, get_pot_locations)."""
    import itertools
    out = {}
    for name in L.layout_names():
        spec = L.spec_from_name(name)
        if spec.num_players not in (1, 2):
            continue
        mdp = R.OvercookedGridworld.from_layout_name(name)
        activate(mdp)
        st = mdp.get_standard_start_state() if not mdp.start_state else mdp.start_state
        ct, val = [0] * 16, [0.0] * 16
        for n in (1, 2, 3):
            for ings in itertools.combinations_with_replacement(["onion", "tomato"], n):
                r = R.Recipe(list(ings))
                n_t = ings.count("tomato")
                idx = (n - n_t) + 4 * n_t
                ct[idx] = r.time
                val[idx] = mdp.get_recipe_value(st, r)
        out[name] = {
            "terrain": ["".join(row) for row in mdp.terrain_mtx],
            "start_player_positions": [list(p) for p in mdp.start_player_positions],
            "pot_locations": [list(p) for p in mdp.get_pot_locations()],
            "cook_time": ct, "delivery_value": val,
            "rew": [mdp.reward_shaping_params[k] for k in ("PLACEMENT_IN_POT_REW", "DISH_PICKUP_REWARD", "SOUP_PICKUP_REWARD")],
        }
    with open(os.path.join(GOLDEN, "layout_luts.json"), "w") as f:
        json.dump(out, f, indent=0, sort_keys=True)
    print("layout_luts", len(out))


def gen_env_episodes():
    """Whole episodes through the reference's OvercookedEnv.step (env.py:244): per-step returns, the final
    info["episode"] dict (game_stats with event timesteps, env.py:363-401) and the gym wrapper's observations."""
    out = {}
    for name, horizon in (("cramped_room", 150), ("mdp_test", 250), ("counter_circuit", 200)):
        spec, mdp = make_ref_mdp(name, {})
        activate(mdp)
        rng = np.random.default_rng(31337)
        env = R.OvercookedEnv.from_mdp(mdp, horizon=horizon, info_level=0)
        env._mp = object()
        steps = []
        done = False
        while not done:
            ja = [int(rng.choice(6, p=[0.14, 0.14, 0.14, 0.14, 0.04, 0.4])) for _ in range(2)]
            ns, r, done, info = env.step([Action.INDEX_TO_ACTION[a] for a in ja])
            steps.append({"actions": ja, "reward": r, "done": bool(done), "sparse_r_by_agent": list(info["sparse_r_by_agent"]),
                          "shaped_r_by_agent": list(info["shaped_r_by_agent"])})
        ep = info["episode"]
        gs = {k: (v.tolist() if hasattr(v, "tolist") else v) for k, v in ep["ep_game_stats"].items()}
        enc = env.lossless_state_encoding_mdp(env.state)
        out[name] = {
            "layout": spec.to_layout_dict(), "horizon": horizon, "steps": steps, "final_state": env.state.to_dict(),
            "episode": {"ep_game_stats": gs, "ep_sparse_r": int(ep["ep_sparse_r"]), "ep_shaped_r": int(ep["ep_shaped_r"]),
                        "ep_sparse_r_by_agent": ep["ep_sparse_r_by_agent"].tolist(),
                        "ep_shaped_r_by_agent": ep["ep_shaped_r_by_agent"].tolist(), "ep_length": ep["ep_length"]},
            "final_encoding_nonzero": [[int(i) for i in idx] + [int(np.stack(enc)[tuple(idx)])] for idx in np.argwhere(np.stack(enc))],
            "encoding_shape": list(np.stack(enc).shape),
        }
    with open(os.path.join(GOLDEN, "env_episodes.json"), "w") as f:
        json.dump(out, f, default=lambda o: o.tolist() if hasattr(o, "tolist") else str(o))
    print("env_episodes", list(out))


def gen_random_starts():
    """get_random_start_state_fn (mdp.py:1307-1369) under fixed numpy seeds."""
    out = {}
    for name in ("cramped_room", "asymmetric_advantages", "counter_circuit", "mdp_test"):
        spec, mdp = make_ref_mdp(name, {})
        activate(mdp)
        cases = []
        for seed, (rsp, thresh) in enumerate([(False, 0.0), (True, 0.0), (True, 0.5), (False, 0.9), (True, 0.9), (True, 0.3)]):
            fn = mdp.get_random_start_state_fn(random_start_pos=rsp, rnd_obj_prob_thresh=thresh)
            np.random.seed(100 + seed)
            states = [fn().to_dict() for _ in range(12)]
            cases.append({"seed": 100 + seed, "random_start_pos": rsp, "rnd_obj_prob_thresh": thresh, "states": states})
        out[name] = {"layout": spec.to_layout_dict(), "cases": cases}
    with open(os.path.join(GOLDEN, "random_starts.json"), "w") as f:
        json.dump(out, f, default=lambda o: o.tolist() if hasattr(o, "tolist") else str(o))
    print("random_starts", list(out))


def _quiet(fn, *a, **k):
    import contextlib, io
    with contextlib.redirect_stdout(io.StringIO()):
        return fn(*a, **k)


def _planner_modules():
    from overcooked_ai_py.planning import planners as P
    # never pickle planners into /root/reference (planners.py:70-73, 126-137, 1127-1130)
    P.MotionPlanner.save_to_file = lambda self, fn: None
    P.MediumLevelActionManager.save_to_file = lambda self, fn: None
    return P


def gen_featurize():
    """featurize_state (mdp.py:2579-2898) fixtures.
    (1) The reference's own golden: GreedyHumanModel rollouts of overcooked_test.py:1069-1093 reproduced here with
        np.random.seed(0); the features are asserted equal to data/testing/test_state_featurization/expected_{0,1,2}.pickle
        (num_pots = 0, 1, 2) before being stored, together with the states, actions and rewards of those 5 x 400 steps (a second,
        delivery-rich pin for the transition itself).
    (2) Randomized states (objects in row-major dict order) on several layouts, for counter_goals = [] (the
        reference's default NO_COUNTERS_PARAMS) and counter_goals = all counters."""
    import pickle
    import types
    P = _planner_modules()
    from overcooked_ai_py.agents.agent import AgentPair, GreedyHumanModel

    spec, mdp = make_ref_mdp("cramped_room", {})
    activate(mdp)
    mlam = _quiet(P.MediumLevelActionManager.from_pickle_or_compute, mdp, P.NO_COUNTERS_PARAMS, force_compute=True)
    env = R.OvercookedEnv.from_mdp(mdp, horizon=400, info_level=0)
    pair = AgentPair(GreedyHumanModel(mlam), GreedyHumanModel(mlam))
    np.random.seed(0)
    trajs = _quiet(env.get_rollouts, pair, num_games=5, info=False)
    feats = np.array([[mdp.featurize_state(s, mlam, num_pots=2) for s in ep] for ep in trajs["ep_states"]])
    exp = np.array(pickle.load(open(os.path.join(REF_TESTING, "test_state_featurization", "expected_2.pickle"), "rb")))
    assert feats.shape == exp.shape == (5, 400, 2, 96) and np.array_equal(feats, exp), "reference golden pickle not reproduced"
    # the reference's test loops num_pots in range(3) (overcooked_test.py:1069-1093): expected_0 / expected_1 as well
    feats_np = {}
    for num_pots in (0, 1):
        f = np.array([[mdp.featurize_state(s, mlam, num_pots=num_pots) for s in ep] for ep in trajs["ep_states"]])
        e = np.array(pickle.load(open(os.path.join(REF_TESTING, "test_state_featurization", "expected_%d.pickle" % num_pots), "rb")))
        width = mdp.get_featurize_state_shape(num_pots)[0]
        assert f.shape == e.shape == (5, 400, 2, width) and np.array_equal(f, e), "expected_%d.pickle not reproduced" % num_pots
        feats_np[num_pots] = f.reshape(5 * 400, 2, width).astype(np.float32)
    n_planes = 1 + (spec.width * spec.height + 15) // 16
    n_ep, T = 5, 400
    packed = np.zeros((n_planes, n_ep * T, 16), np.uint8)
    acts = np.zeros((n_ep * T, 2), np.uint8)
    rews = np.zeros((n_ep * T,), np.float64)
    for e in range(n_ep):
        for t in range(T):
            S.pack_state_dict(spec, trajs["ep_states"][e][t].to_dict(), packed, e * T + t)
            ja = trajs["ep_actions"][e][t]
            acts[e * T + t] = [Action.ACTION_TO_INDEX[a if isinstance(a, str) else tuple(a)] for a in ja]
            rews[e * T + t] = trajs["ep_rewards"][e][t]
    np.savez_compressed(os.path.join(GOLDEN, "ref_greedy_rollouts.npz"), states=packed, actions=acts, rewards=rews,
                        features=feats.reshape(n_ep * T, 2, 96).astype(np.float32), features_num_pots_0=feats_np[0],
                        features_num_pots_1=feats_np[1], n_episodes=np.array(n_ep), horizon=np.array(T))
    print("greedy rollouts: reference pickle reproduced; sparse return per episode", rews.reshape(n_ep, T).sum(1))

    out = {}
    for name in ("cramped_room", "asymmetric_advantages", "forced_coordination", "counter_circuit", "mdp_test"):
        spec, mdp = make_ref_mdp(name, {})
        activate(mdp)
        rng = np.random.default_rng(555)
        for label, cg in (("none", []), ("all", mdp.get_counter_locations())):
            mp = _quiet(P.MotionPlanner, mdp, cg)
            fake = types.SimpleNamespace(motion_planner=mp)
            n = 300
            n_planes = 1 + (spec.width * spec.height + 15) // 16
            packed = np.zeros((n_planes, n, 16), np.uint8)
            fz = np.zeros((n, 2, 96), np.float32)
            for e in range(n):
                st = random_ref_state(mdp, spec, rng)
                d = st.to_dict()
                d["objects"] = sorted(d["objects"], key=lambda o: (o["position"][1], o["position"][0]))  # row-major dict order
                st = R.OvercookedState.from_dict(d)
                S.pack_state_dict(spec, d, packed, e)
                fz[e] = np.array(mdp.featurize_state(st, fake, num_pots=2))
            np.savez_compressed(os.path.join(GOLDEN, "featurize_%s_%s.npz" % (name, label)), states=packed, features=fz)
            out["%s_%s" % (name, label)] = {"layout": spec.to_layout_dict(), "counter_goals": label, "n": n}
    with open(os.path.join(GOLDEN, "featurize_manifest.json"), "w") as f:
        json.dump(out, f)
    print("featurize fixtures", list(out))


POTENTIAL_LAYOUTS = [  # (fixture name, layout name or None, custom layout dict)
    ("cramped_room", "cramped_room", None),
    ("asymmetric_advantages", "asymmetric_advantages", None),
    ("forced_coordination", "forced_coordination", None),
    ("counter_circuit", "counter_circuit", None),
    ("mdp_test", "mdp_test", None),
    ("cramped_room_tomato", "cramped_room_tomato", None),
    ("marshmallow_experiment", "marshmallow_experiment", None),
    ("seven_pots", None, {
        "grid": "XPPPPPX\nO 1 2 O\nX     X\nXDPSPTX",
        "start_all_orders": [{"ingredients": ["onion", "onion", "onion"]}, {"ingredients": ["onion", "onion", "tomato"]},
                             {"ingredients": ["onion", "tomato", "tomato"]}, {"ingredients": ["tomato", "tomato", "tomato"]},
                             {"ingredients": ["onion", "tomato"]}],
        "start_bonus_orders": [{"ingredients": ["onion", "tomato", "tomato"]}],
        "onion_value": 21, "tomato_value": 13, "onion_time": 7, "tomato_time": 4}),
]


def gen_potential():
    """potential_function (mdp.py:2920-3238) on randomized states, gamma 0.99 (the default) and 0.9, with a
    MotionPlanner without counter goals (what OvercookedEnv.mp builds, env.py:102-115)."""
    P = _planner_modules()
    out = {}
    for name, lname, custom in POTENTIAL_LAYOUTS:
        if custom is None:
            spec, mdp = make_ref_mdp(lname, {})
        else:
            spec = L.LayoutSpec(dict(custom))
            base = {k: v for k, v in custom.items() if k != "grid"}
            mdp = R.OvercookedGridworld.from_grid(custom["grid"].split("\n"), base_layout_params=base)
        activate(mdp)
        mp = _quiet(P.MotionPlanner, mdp, [])
        rng = np.random.default_rng(4242)
        n = 400
        n_planes = 1 + (spec.width * spec.height + 15) // 16
        packed = np.zeros((n_planes, n, 16), np.uint8)
        phi = np.zeros((2, n), np.float64)
        states = []
        for e in range(n):
            st = random_ref_state(mdp, spec, rng)
            if e % 5 == 0:  # make several pots partially full so that the set-ordered tie-breaking is exercised
                for pos in mdp.get_pot_locations():
                    k = int(rng.integers(0, 3))
                    if k and pos in st.objects:
                        st.objects[pos] = R.SoupState(pos, [R.ObjectState("onion", pos) for _ in range(k)])
            states.append(st)
            S.pack_state_dict(spec, st.to_dict(), packed, e)
        for gi, g in enumerate((0.99, 0.9)):
            for e, st in enumerate(states):
                phi[gi, e] = mdp.potential_function(st, mp, gamma=g)
        np.savez_compressed(os.path.join(GOLDEN, "potential_%s.npz" % name), states=packed, phi=phi, gammas=np.array([0.99, 0.9]))
        out[name] = {"layout": spec.to_layout_dict(), "n": n, "gammas": [0.99, 0.9]}
        print("potential", name, phi[0, :4], float(phi.min()), float(phi.max()))
    with open(os.path.join(GOLDEN, "potential_manifest.json"), "w") as f:
        json.dump(out, f)


MULTI_AGENT_CASES = [  # (fixture name, layout, horizon, OvercookedMultiAgent kwargs)
    ("counter_circuit_phi", "counter_circuit", 60, dict(reward_shaping_factor=1.0, reward_shaping_horizon=0, use_phi=True)),
    ("cramped_room_shaped_bc", "cramped_room", 120,
     dict(reward_shaping_factor=0.7, reward_shaping_horizon=400, use_phi=False, bc_schedule=[(0, 0.6), (300, 0.2)])),
    ("asymmetric_advantages_phi_anneal", "asymmetric_advantages", 50,
     dict(reward_shaping_factor=0.5, reward_shaping_horizon=120, use_phi=True, bc_schedule=[(0, 0.3), (1000, 0.3)])),
]


def gen_multi_agent():
    """OvercookedMultiAgent (human_aware_rl/rllib/rllib.py:112-438), the RLlib training environment: agent-role
    assignment, observations per agent type, rewards sparse + factor * (phi' - phi | shaped), annealing.  Five
    seeded episodes per case, actions from an independent generator."""
    import types
    P = _planner_modules()
    rl = ref_harness.load_rllib()
    planners = {}

    def planner(env):
        key = env.mdp.layout_name
        if key not in planners:
            planners[key] = _quiet(P.MotionPlanner, env.mdp, [])
        return planners[key]

    # the env's lazily built planners (env.py:92-115) would be pickled into the reference tree and, for the MLAM, take
    # minutes; featurize_state / potential_function only ever use the motion planner
    R.env_module.OvercookedEnv.mp = property(lambda self: planner(self))
    R.env_module.OvercookedEnv.mlam = property(lambda self: types.SimpleNamespace(motion_planner=planner(self)))
    out = {}
    for name, lname, horizon, kw in MULTI_AGENT_CASES:
        spec, mdp = make_ref_mdp(lname, {})
        activate(mdp)
        base_env = R.OvercookedEnv.from_mdp(mdp, horizon=horizon, info_level=0)
        np.random.seed(2024)
        kw = {k: (list(v) if isinstance(v, list) else v) for k, v in kw.items()}
        env = rl.OvercookedMultiAgent(base_env, **kw)
        arng = np.random.default_rng(99)
        eps, total = [], 0
        for ep in range(5):
            obs = env.reset()
            agents = list(env.curr_agents)
            rec = {"agents": agents, "obs0": [obs[a].tolist() for a in agents], "steps": []}
            done = False
            while not done:
                a = [int(x) for x in arng.choice(6, size=2, p=[0.15, 0.15, 0.15, 0.15, 0.05, 0.35])]
                obs, rew, dones, infos = env.step({agents[0]: a[0], agents[1]: a[1]})
                done = dones["__all__"]
                info = infos[agents[0]]
                total += 1
                rec["steps"].append({"actions": a, "rewards": [float(rew[x]) for x in agents], "done": bool(done),
                                     "obs": [obs[x].tolist() for x in agents],
                                     "phi_s": info.get("phi_s"), "phi_s_prime": info.get("phi_s_prime"),
                                     "factor": float(env.reward_shaping_factor), "bc_factor": float(env.bc_factor)})
                if total % 7 == 0:
                    env.anneal_reward_shaping_factor(total)
                    env.anneal_bc_factor(total)
            rec["ep_sparse_r"] = float(info["episode"]["ep_sparse_r"])
            rec["ep_shaped_r"] = float(info["episode"]["ep_shaped_r"])
            eps.append(rec)
        # observations go to a compressed npz (small integers, mostly zero); everything else to the JSON manifest
        rows = []
        for rec in eps:
            rows.append(rec.pop("obs0"))
            for st_ in rec["steps"]:
                rows.append(st_.pop("obs"))
        width = max(np.asarray(o).size for r in rows for o in r)
        arr = np.zeros((len(rows), 2, width), np.float32)
        lens = np.zeros((len(rows), 2), np.int32)
        for i, r in enumerate(rows):
            for j, o in enumerate(r):
                flat = np.asarray(o, np.float32).ravel()
                arr[i, j, :flat.size] = flat
                lens[i, j] = flat.size
        assert np.array_equal(arr, arr.astype(np.int16))
        np.savez_compressed(os.path.join(GOLDEN, "multi_agent_%s.npz" % name), obs=arr.astype(np.int16), obs_len=lens)
        out[name] = {"layout": spec.to_layout_dict(), "horizon": horizon, "kwargs": kw, "episodes": eps}
        print("multi agent", name, [e["agents"] for e in eps], [e["ep_shaped_r"] for e in eps])
    with open(os.path.join(GOLDEN, "multi_agent.json"), "w") as f:
        json.dump(out, f)


def gen_generated_layouts(k=4096):
    """BASELINE configs[4]: the terrains the reference's own LayoutGenerator produces for
    mdp_gen_fn_from_dict({inner_shape (9, 5), prop_empty 0.9, prop_feats 0.1, one 3-onion order worth 20 cooking for
    20}, outer_shape=(9, 5)) after np.random.seed(0) / random.seed(0) (layout_generator.py:110-142) — shipped as
    package data so that the benchmark runs on the reference's layouts, not on look-alikes."""
    import gzip
    import random

    from overcooked_ai_py.mdp.layout_generator import LayoutGenerator

    params = {"inner_shape": (9, 5), "prop_empty": 0.9, "prop_feats": 0.1,
              "start_all_orders": [{"ingredients": ["onion", "onion", "onion"]}], "recipe_values": [20],
              "recipe_times": [20], "display": False}
    np.random.seed(0)
    random.seed(0)
    fn = LayoutGenerator.mdp_gen_fn_from_dict(params, outer_shape=(9, 5))
    grids = []
    for _ in range(k):
        mdp = fn()
        rows = [list(r) for r in mdp.terrain_mtx]
        for i, (x, y) in enumerate(mdp.start_player_positions):
            rows[y][x] = str(i + 1)
        grids.append(["".join(r) for r in rows])
    out = {"generator": "LayoutGenerator.mdp_gen_fn_from_dict, np.random.seed(0), random.seed(0)",
           "mdp_params": {k_: v for k_, v in params.items() if k_ not in ("display",)}, "outer_shape": [9, 5], "grids": grids}
    path = os.path.join(ROOT, "overcooked_ai_amd", "data", "ref_generated_9x5_seed0.json.gz")
    with gzip.open(path, "wt", compresslevel=9) as f:
        json.dump(out, f)
    print("generated layouts", k, "distinct", len({"|".join(g) for g in grids}), os.path.getsize(path), "bytes")


def gen_state_queries():
    """The reference's host-side state queries agents and planners call on the mdp (get_pot_states,
    get_counter_objects_dict, get_empty_counter_locations, the get_*_pots family, soup_ready_at_location,
    soup_to_be_cooked_at_location; mdp.py:1809-1907) on states sampled from the transition fixtures."""
    out = {}
    for name, lname, ov in CONFIGS:
        if name not in ("cramped_room", "asymmetric_advantages", "counter_circuit", "mdp_test", "cramped_room_tomato"):
            continue
        spec, mdp = make_ref_mdp(lname, ov)
        activate(mdp)
        d = np.load(os.path.join(GOLDEN, "transitions_%s.npz" % name))
        from overcooked_ai_amd.state import unpack_states
        idx = list(range(0, d["state_in"].shape[1], d["state_in"].shape[1] // 60))[:60]
        dicts = unpack_states(spec, np.ascontiguousarray(d["state_in"][:, idx]), as_dict=True)
        cases = []
        for sd in dicts:
            st = R.OvercookedState.from_dict(sd)
            ps = mdp.get_pot_states(st)
            rec = {"state": sd, "pot_states": {k: [list(p) for p in v] for k, v in dict(ps).items() if v},
                   "counter_objects": {k: [list(p) for p in v] for k, v in mdp.get_counter_objects_dict(st).items()},
                   "empty_counters": [list(p) for p in mdp.get_empty_counter_locations(st)],
                   "empty_pots": [list(p) for p in mdp.get_empty_pots(ps)], "ready_pots": [list(p) for p in mdp.get_ready_pots(ps)],
                   "cooking_pots": [list(p) for p in mdp.get_cooking_pots(ps)],
                   "full_not_cooking": [list(p) for p in mdp.get_full_but_not_cooking_pots(ps)],
                   "full_pots": [list(p) for p in mdp.get_full_pots(ps)],
                   "partially_full": sorted(list(p) for p in mdp.get_partially_full_pots(ps)),
                   "soup_ready": [bool(mdp.soup_ready_at_location(st, p)) for p in mdp.get_pot_locations()],
                   "soup_to_cook": [bool(mdp.soup_to_be_cooked_at_location(st, p)) for p in mdp.get_pot_locations()],
                   "adjacent": [[[list(pos), t] for pos, t in mdp.get_adjacent_features(pl)] for pl in st.players]}
            cases.append(rec)
        out[name] = {"layout": spec.to_layout_dict(), "cases": cases}
    with open(os.path.join(GOLDEN, "state_queries.json"), "w") as f:
        json.dump(out, f, separators=(",", ":"))
    print("state queries", {k: len(v["cases"]) for k, v in out.items()})


def gen_agent_pair_rollouts():
    """Episodes the REFERENCE's own agents play in the reference's own env: AgentPair(RandomAgent(all_actions=True),
    RandomAgent(all_actions=True)).joint_action + OvercookedEnv.run_agents (agents/agent.py:137, 223;
    overcooked_env.py:425-470), np.random seeded — the trajectory (states, joint actions with their action_probs infos,
    rewards, dones) and the episode summary that the mirror env must reproduce when it replays the same actions."""
    from overcooked_ai_py.agents.agent import AgentPair, RandomAgent
    out = {}
    for lname, horizon, seed in (("cramped_room", 120, 5), ("counter_circuit", 90, 6)):
        spec, mdp = make_ref_mdp(lname, {})
        activate(mdp)
        env = R.OvercookedEnv.from_mdp(mdp, horizon=horizon, info_level=0)
        env._mp = object()
        np.random.seed(seed)
        pair = AgentPair(RandomAgent(all_actions=True), RandomAgent(all_actions=True))
        traj, t_elapsed, total_sparse, total_shaped = env.run_agents(pair, include_final_state=True)
        steps = []
        for (s_t, a_t, r_t, done, info) in traj[:-1]:
            steps.append({"state": s_t.to_dict(), "action": [R.Action.ACTION_TO_INDEX[a] for a in a_t], "reward": float(r_t),
                          "done": bool(done), "agent_infos": [{k: np.asarray(v).tolist() for k, v in ai.items()} for ai in info["agent_infos"]],
                          "sparse_r_by_agent": [float(x) for x in info["sparse_r_by_agent"]],
                          "shaped_r_by_agent": [float(x) for x in info["shaped_r_by_agent"]]})
        ep = traj[-2][4]["episode"]
        out[lname] = {"layout": spec.to_layout_dict(), "horizon": horizon, "seed": seed, "steps": steps,
                      "final_state": traj[-1][0].to_dict(), "t_elapsed": int(t_elapsed), "total_sparse": float(total_sparse),
                      "total_shaped": float(total_shaped), "ep_length": int(ep["ep_length"]),
                      "ep_game_stats": {k: ([list(map(int, x)) for x in v] if isinstance(v, list) else np.asarray(v).tolist())
                                        for k, v in ep["ep_game_stats"].items()}}
    with open(os.path.join(GOLDEN, "agent_pair_rollouts.json"), "w") as f:
        json.dump(out, f, separators=(",", ":"))
    print("agent pair rollouts", {k: len(v["steps"]) for k, v in out.items()})


def main():
    if "--rollouts-only" in sys.argv:  # (only the episodes driven by the Philox action stream)
        for i, (name, lname, ov) in enumerate(CONFIGS):
            if name in ("cramped_room", "asymmetric_advantages", "counter_circuit", "mdp_test", "cramped_room_old_dynamics"):
                gen_rollouts(name, lname, ov, n_envs=24, seed=77 + i)
                print("rollouts", name)
        return
    if "--queries-only" in sys.argv:
        gen_state_queries()
        gen_agent_pair_rollouts()
        return
    if "--generated-layouts-only" in sys.argv:
        gen_generated_layouts()
        return
    if "--multi-agent-only" in sys.argv:
        gen_multi_agent()
        return
    if "--potential-only" in sys.argv:
        gen_potential()
        return
    if "--featurize-only" in sys.argv:
        gen_featurize()
        return
    if "--random-starts-only" in sys.argv:
        gen_random_starts()
        return
    if "--env-episodes-only" in sys.argv:
        gen_env_episodes()
        return
    if "--luts-only" in sys.argv:
        os.makedirs(GOLDEN, exist_ok=True)
        gen_layout_luts()
        return
    os.makedirs(GOLDEN, exist_ok=True)
    manifest = {"generator": "oracle/gen_golden.py", "reference": "HumanCompatibleAI/overcooked_ai @ /root/reference",
                "event_types": EVENT_TYPES, "configs": {}}
    manifest["ref_mdp_dynamics"] = gen_ref_trajectory()
    small = gen_small_fixtures()
    with open(os.path.join(GOLDEN, "ref_small_fixtures.json"), "w") as f:
        json.dump(small, f, indent=1, default=lambda o: o.tolist() if hasattr(o, "tolist") else str(o))
    for i, (name, lname, ov) in enumerate(CONFIGS):
        single = name.endswith("_single")
        info = gen_transitions(name, lname, ov, n_random=6000 if not single else 1500,
                               n_walk_eps=40 if not single else 10, seed=1000 + i)
        manifest["configs"][name] = {"layout_name": lname, "overrides": {k: list(v) if isinstance(v, tuple) else v for k, v in ov.items()},
                                     "transitions": info}
        print("transitions", name, info["n"])
    for i, (name, lname, ov) in enumerate(CONFIGS):
        if name in ("cramped_room", "asymmetric_advantages", "counter_circuit", "mdp_test", "cramped_room_old_dynamics"):
            manifest["configs"][name]["rollouts"] = gen_rollouts(name, lname, ov, n_envs=24, seed=77 + i)
            print("rollouts", name)
    gen_layout_luts()
    gen_env_episodes()
    gen_random_starts()
    gen_featurize()
    gen_potential()
    gen_multi_agent()
    gen_generated_layouts()
    gen_state_queries()
    gen_agent_pair_rollouts()
    with open(os.path.join(GOLDEN, "manifest.json"), "w") as f:
        json.dump(manifest, f, indent=1, default=lambda o: o.tolist() if hasattr(o, "tolist") else str(o))
    print("done")


if __name__ == "__main__":
    main()
