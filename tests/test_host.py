"""Host-side logic: layout loading/compilation pinned against the live reference's values
(tests/golden/layout_luts.json), state types and the packed wire format."""
import json
import os
import struct

import numpy as np
import pytest

from conftest import GOLDEN
from helpers import random_packed_states
from overcooked_ai_amd import layouts as L
from overcooked_ai_amd import state as S
from overcooked_ai_amd.actions import Action, Direction


@pytest.fixture(scope="module")
def luts():
    with open(os.path.join(GOLDEN, "layout_luts.json")) as f:
        return json.load(f)


def test_action_tables():
    assert Action.INDEX_TO_ACTION == [(0, -1), (0, 1), (1, 0), (-1, 0), (0, 0), "interact"]
    assert [Action.to_index(a) for a in Action.ALL_ACTIONS] == list(range(6))
    assert Action.to_index([1, 0]) == 2
    with pytest.raises(ValueError):
        Action.to_index("jump")
    assert Direction.OPPOSITE_DIRECTIONS[Direction.NORTH] == Direction.SOUTH


def test_compiled_layouts_match_reference(luts):
    assert len(luts) == 48
    for name, ref in luts.items():
        spec = L.spec_from_name(name)
        rec = L.compile_layout(spec).tobytes()
        W, H = spec.width, spec.height
        assert rec[0] == W and rec[1] == H and rec[2] == W * H
        assert ["".join(r) for r in spec.terrain_mtx] == ref["terrain"], name
        assert [list(p) for p in spec.start_player_positions] == ref["start_player_positions"], name
        pots = [tuple(p) for p in ref["pot_locations"]]
        assert rec[3] == len(pots)
        for k, (x, y) in enumerate(pots):
            assert rec[16 + k] == y * W + x
            assert rec[128 + y * W + x] == 4 | (k << 3)
        for y in range(H):
            for x in range(W):
                assert rec[128 + y * W + x] & 7 == L.TERRAIN_CODE[ref["terrain"][y][x]]
        assert list(rec[48:64]) == [int(v) for v in ref["cook_time"]], name
        vals = struct.unpack("<16f", rec[64:128])
        for a, b in zip(vals, ref["delivery_value"]):
            assert a == np.float32(b), (name, vals, ref["delivery_value"])
        assert list(struct.unpack("<3f", rec[32:44])) == [float(v) for v in ref["rew"]]
        for i, (x, y) in enumerate(ref["start_player_positions"]):
            assert rec[8 + i] == y * W + x
        if len(ref["start_player_positions"]) == 1:
            assert rec[9] == 0xFF


def test_layout_validation_and_overrides():
    with pytest.raises(AssertionError):
        L.LayoutSpec({"grid": "XXX\nX1 \nXXX"})  # right border free
    with pytest.raises(AssertionError):
        L.LayoutSpec({"grid": "XXPXX\nO  2O\nX   X\nXDXSX"})  # player 1 missing
    with pytest.raises(AssertionError):
        L.spec_from_name("mdp_test", old_dynamics=True)  # overcooked_test.py:528-532
    with pytest.raises(ValueError):
        L.LayoutSpec({"grid": "XXPXX\nO1 2O\nXDXSX", "onion_time": 3})
    with pytest.raises(ValueError):
        L.compile_layout(L.spec_from_name("multiplayer_schelling"))
    spec = L.spec_from_name("cramped_room", old_dynamics=True)
    assert L.compile_layout(spec)[5] == 1
    # Recipe.value truthiness (mdp.py:136-161): a zero ingredient value falls back to 20
    spec = L.LayoutSpec({"grid": "XXPXX\nO1 2O\nXDXSX", "onion_value": 0, "tomato_value": 7})
    assert spec.recipe_value((3, 0)) == 20


def test_padding_keeps_dynamics_relevant_fields():
    spec = L.spec_from_name("cramped_room")
    pad = spec.padded(9, 5)
    assert pad.shape == (9, 5) and pad.start_player_positions == spec.start_player_positions
    assert pad.cells_of("P") == spec.cells_of("P") and len(pad.cells_of(" ")) == len(spec.cells_of(" "))
    table = L.LayoutTable([L.spec_from_name(n) for n in ("cramped_room", "asymmetric_advantages")])
    assert (table.width, table.height, table.n_planes) == (9, 5, 4) and table.records.shape == (2, 256)


def test_load_layout_file_roundtrip(tmp_path):
    p = tmp_path / "x.layout"
    p.write_text('{\n "grid": """XXPXX\n            O1 2O\n            XDXSX""",\n "start_all_orders": [{"ingredients": ["onion"]}],\n'
                 ' "order_bonus": float(\'inf\'), "rew_shaping_params": None}\n')
    d = L.load_layout_file(str(p))
    spec = L.LayoutSpec(d)
    assert spec.shape == (5, 3) and spec.order_bonus == float("inf")


def test_pack_unpack_identity_on_random_states():
    rng = np.random.default_rng(0)
    for name in ("cramped_room", "counter_circuit", "corridor", "cramped_room_single", "mdp_test"):
        spec = L.spec_from_name(name)
        packed = random_packed_states(spec, 200, rng)
        dicts = S.unpack_states(spec, packed, as_dict=True)
        again = S.pack_states(spec, dicts)
        assert np.array_equal(again, packed), name
        objs = S.unpack_states(spec, packed)  # through our OvercookedState objects and their to_dict()
        assert np.array_equal(S.pack_states(spec, objs), packed), name
        for o, d in zip(objs[:20], dicts[:20]):
            assert S.OvercookedState.from_dict(o.to_dict()) == o
            assert S.canonical_state_dict(o) == S.canonical_state_dict(d)


def test_state_dict_schema_matches_reference_samples(manifest):
    """to_dict() of our types reproduces the reference's JSON schema key for key."""
    cfg = manifest["configs"]["mdp_test"]["transitions"]
    for smp in cfg["samples"]:
        ref = smp["next_state"]
        ours = S.OvercookedState.from_dict(ref).to_dict()
        assert set(ours) == set(ref)
        assert S.canonical_state_dict(ours) == S.canonical_state_dict(ref)
        for a, b in zip(ours["players"], ref["players"]):
            assert set(a) == set(b)
            if b["held_object"] is not None:
                assert set(a["held_object"]) == set(b["held_object"])


def test_pack_rejects_states_outside_the_domain():
    spec = L.spec_from_name("cramped_room")
    base = {"players": [{"position": (1, 2), "orientation": (0, -1), "held_object": None},
                        {"position": (3, 1), "orientation": (0, -1), "held_object": None}],
            "objects": [], "timestep": 0}

    def bad(**kw):
        d = json.loads(json.dumps(base))
        d.update(kw)
        with pytest.raises(ValueError):
            S.pack_states(spec, [d])

    bad(objects=[{"name": "onion", "position": (1, 1)}])                      # object on a floor cell
    bad(objects=[{"name": "onion", "position": (2, 0)}])                      # non-soup in a pot
    bad(objects=[{"name": "soup", "position": (0, 0), "_ingredients": [{"name": "onion", "position": (0, 0)}],
                  "cooking_tick": 3}])                                        # half-cooked soup on a counter
    bad(objects=[{"name": "soup", "position": (2, 0), "_ingredients": [{"name": "onion", "position": (2, 0)}],
                  "cooking_tick": 21}])                                       # tick beyond cook time
    bad(players=[base["players"][0], dict(base["players"][0])])               # overlapping players
    bad(players=[base["players"][0], {"position": (0, 0), "orientation": (0, 1), "held_object": None}])
    bad(timestep=70000)
    ok = json.loads(json.dumps(base))
    ok["objects"] = [{"name": "soup", "position": (2, 0), "_ingredients": [{"name": "onion", "position": (2, 0)}],
                      "cooking_tick": 20}]
    assert S.pack_states(spec, [ok])[0, 0, 8] == 21


def test_random_start_states_reproduce_reference_draws():
    """get_random_start_state_fn consumes np.random exactly like mdp.py:1307-1369: same seed -> same states."""
    from overcooked_ai_amd.mdp import OvercookedGridworld

    with open(os.path.join(GOLDEN, "random_starts.json")) as f:
        fx = json.load(f)
    n = 0
    for name, item in fx.items():
        spec = L.LayoutSpec(item["layout"])
        mdp = OvercookedGridworld.from_spec(spec)
        for case in item["cases"]:
            fn = mdp.get_random_start_state_fn(random_start_pos=case["random_start_pos"],
                                               rnd_obj_prob_thresh=case["rnd_obj_prob_thresh"])
            np.random.seed(case["seed"])
            for ref in case["states"]:
                got = fn()
                assert S.canonical_state_dict(got) == S.canonical_state_dict(ref), (name, case["seed"])
                S.pack_states(spec, [got])  # every random start lies inside the packed domain
                n += 1
    assert n == 4 * 6 * 12


def test_generated_layouts_are_valid_and_diverse():
    from overcooked_ai_amd.layout_gen import generate_layouts

    specs = generate_layouts(200, seed=3, inner_shape=(9, 5), prop_empty=0.9, prop_feats=0.1)
    grids = set()
    for s in specs:
        assert s.shape == (9, 5) and s.num_players == 2
        flat = "".join("".join(r) for r in s.terrain_mtx)
        assert 1 <= flat.count("P") <= 2 and flat.count("O") >= 1 and flat.count("D") >= 1 and flat.count("S") >= 1
        assert 15 <= flat.count(" ") <= 21
        # the free region is connected
        free = set(s.cells_of(" "))
        seen, todo = set(), [next(iter(free))]
        while todo:
            x, y = todo.pop()
            if (x, y) in seen:
                continue
            seen.add((x, y))
            todo += [p for p in ((x + 1, y), (x - 1, y), (x, y + 1), (x, y - 1)) if p in free]
        assert seen == free
        L.compile_layout(s)
        grids.add(flat)
    assert len(grids) > 190
    small = generate_layouts(5, seed=1, inner_shape=(5, 4), outer_shape=(9, 5))
    assert all(s.shape == (9, 5) for s in small)


def test_potential_tables_host_side():
    """Host-built records for k_potential (potential.py): sizes, the DFS of _get_optimal_possible_recipe
    (mdp.py:1976-2016) and the steady-state value (mdp.py:2985-3001) on known configurations."""
    import struct

    from overcooked_ai_amd.layouts import spec_from_name
    from overcooked_ai_amd.potential import (PHI_BYTES, optimal_possible_recipe, pack_phi_tables, phi_record,
                                               potential_params)

    spec = spec_from_name("cramped_room")  # only order: 3 onions worth 20, cook time 20
    pp = potential_params(spec, 0.99)
    assert (pp["onion_value"], pp["tomato_value"], pp["max_delivery_steps"]) == (21, 13, 10)
    key, value = optimal_possible_recipe(spec, None, pp)
    assert key == (3, 0) and value == 0.99 ** 20 * 0.99 ** 30 * 0.99 ** 0 * 20
    assert optimal_possible_recipe(spec, (1, 0), pp) == ((3, 0), 0.99 ** 20 * 0.99 ** 20 * 0.99 ** 0 * 20)
    assert optimal_possible_recipe(spec, (0, 2), pp) == ((0, 2), 0)  # nothing valuable reachable: stays put
    rec = phi_record(spec, 0.99)
    assert len(rec) == PHI_BYTES
    steady, onion_value, tomato_value = struct.unpack_from("<3d", rec, 0)
    d = value / 20
    assert steady == (d / (1 - d)) * 20 and (onion_value, tomato_value) == (21.0, 13.0)
    assert struct.unpack_from("<4i", rec, 24) == (10, 10, 10, 10)
    assert rec[456] == 1 and rec[457] == 3 * 5 + 3  # cramped_room: one serving cell at (3, 3)
    pw = struct.unpack_from("<512d", rec, 472)
    assert pw[0] == 1.0 and pw[1] == 0.99 and pw[37] == 0.99 ** 37
    assert rec[424 + 1] == 3 and rec[440 + 1] == 20  # one onion -> completes to ooo, cook time 20
    cc = spec_from_name("counter_circuit")  # bonus order onion+tomato worth 2 * 34
    assert optimal_possible_recipe(cc, None, potential_params(cc, 0.99))[0] == (1, 1)
    assert pack_phi_tables([spec, cc]).shape == (2, PHI_BYTES)


def test_py_set_order_port_matches_interpreter():
    """potential.py's port of CPython's set ordering (it fills the two-pot order bits of the phi record)."""
    import random
    import sys

    from overcooked_ai_amd.potential import py_set_order

    if not (3, 8) <= sys.version_info[:2] <= (3, 12):
        pytest.skip("set/tuple-hash internals restated for CPython 3.8-3.12")
    rnd = random.Random(5)
    for _ in range(3000):
        W, H = rnd.randint(3, 14), rnd.randint(3, 9)
        cells = rnd.sample(range(W * H), rnd.randint(0, 9))
        pos = [(c % W, c // W) for c in cells]
        assert py_set_order(pos) == list(set().union(pos))


def test_reference_generated_layouts_package_data():
    """BASELINE configs[4] terrains recorded from the reference's LayoutGenerator: all valid, distinct, one pot, 9x5."""
    from overcooked_ai_amd.layout_gen import reference_generated_layouts
    from overcooked_ai_amd.layouts import LayoutTable

    specs = reference_generated_layouts()
    assert len(specs) == 4096 and len({(s.layout_name, tuple(s.start_player_positions)) for s in specs}) == 4096
    assert all(s.shape == (9, 5) and s.num_players == 2 and len(s.cells_of("P")) >= 1 for s in specs)
    assert specs[0].layout_name == "XXXDPXXOX|X    X  X|S       X|X  X    X|XXXXXXXXX"
    assert specs[0].start_player_positions == [(4, 1), (6, 3)]
    table = LayoutTable(specs[:100])
    assert table.records.shape == (100, 256) and table.max_pots >= 1
    assert all(s.delivery_value((3, 0)) == 20 and s.recipe_time((3, 0)) == 20 for s in specs[:50])


# ------------------------------------------------------------------ drop-in query surface (host side, no GPU)
def _state_queries():
    import json

    with open(os.path.join(GOLDEN, "state_queries.json")) as f:
        return json.load(f)


def test_state_queries_match_the_reference():
    """get_pot_states / get_counter_objects_dict / get_empty_counter_locations / the get_*_pots family /
    soup_ready_at_location / soup_to_be_cooked_at_location / get_adjacent_features (mdp.py:1773-1907) on 300 states
    recorded from the reference (oracle/gen_golden.py gen_state_queries)."""
    from overcooked_ai_amd.layouts import LayoutSpec
    from overcooked_ai_amd.mdp import OvercookedGridworld
    from overcooked_ai_amd.state import OvercookedState

    tup = lambda ps: [tuple(p) for p in ps]
    n = 0
    for name, rec in _state_queries().items():
        mdp = OvercookedGridworld.from_spec(LayoutSpec(rec["layout"]))
        for case in rec["cases"]:
            st = OvercookedState.from_dict(case["state"])
            ps = mdp.get_pot_states(st)
            assert {k: v for k, v in dict(ps).items() if v} == {k: tup(v) for k, v in case["pot_states"].items()}, name
            assert dict(mdp.get_counter_objects_dict(st)) == {k: tup(v) for k, v in case["counter_objects"].items()}
            assert mdp.get_empty_counter_locations(st) == tup(case["empty_counters"])
            assert mdp.get_empty_pots(ps) == tup(case["empty_pots"]) and mdp.get_ready_pots(ps) == tup(case["ready_pots"])
            assert mdp.get_cooking_pots(ps) == tup(case["cooking_pots"])
            assert mdp.get_full_but_not_cooking_pots(ps) == tup(case["full_not_cooking"])
            assert mdp.get_full_pots(ps) == tup(case["full_pots"])
            assert sorted(mdp.get_partially_full_pots(ps)) == sorted(tup(case["partially_full"]))
            assert sorted(mdp.get_non_empty_pots(ps)) == sorted(tup(case["full_pots"]) + tup(case["partially_full"]))
            pots = mdp.get_pot_locations()
            assert [mdp.soup_ready_at_location(st, p) for p in pots] == case["soup_ready"]
            assert [mdp.soup_to_be_cooked_at_location(st, p) for p in pots] == case["soup_to_cook"]
            assert [[[list(pos), t] for pos, t in mdp.get_adjacent_features(pl)] for pl in st.players] == case["adjacent"]
            txt = mdp.state_string(st)
            assert txt.count("\n") >= mdp.height and all(str(i) in txt for i in range(len(st.players)))
            n += 1
    assert n == 300


def test_motion_planner_facade_and_mlam_error():
    """planner.MotionPlanner (what OvercookedEnv.mp returns): its costs equal the tables the kernels use, plans have the
    length the cost says and end on the goal; OvercookedEnv.mlam's stand-in names what is unsupported."""
    from overcooked_ai_amd.actions import Action, Direction
    from overcooked_ai_amd.env import _UnsupportedPlanner
    from overcooked_ai_amd.layouts import spec_from_name
    from overcooked_ai_amd.planner import UNREACHABLE, MotionPlanner, feature_costs

    for name in ("cramped_room", "forced_coordination", "counter_circuit"):
        spec = spec_from_name(name)
        mp = MotionPlanner(spec)
        fi, cost = feature_costs(spec, "none")
        W = spec.width
        feats = [p for t in "OTPDS" for p in spec.cells_of(t)]
        for (x, y) in spec.cells_of(" "):
            for o, d in enumerate(Direction.ALL_DIRECTIONS):
                for f in feats:
                    c = int(cost[4 * fi[y * W + x] + o, f[1] * W + f[0]])
                    got = mp.min_cost_to_feature(((x, y), d), [f])
                    assert got == (np.inf if c == UNREACHABLE else c + 1), (name, x, y, o, f)
        start = (spec.cells_of(" ")[0], Direction.NORTH)
        for f in feats:
            for goal in mp.motion_goals_for_pos[tuple(f)]:
                if mp.is_valid_motion_start_goal_pair(start, goal):
                    plan, path, n = mp.get_plan(start, goal)
                    assert n == len(plan) == len(path) == mp.get_gridworld_distance(start, goal) + 1
                    assert plan[-1] == Action.INTERACT and path[-1] == goal
        assert mp.min_cost_between_features(spec.cells_of("P"), spec.cells_of("S")) >= 1
    m = _UnsupportedPlanner(mp, {"counter_goals": []})
    assert m.motion_planner is mp
    with pytest.raises(NotImplementedError, match="MediumLevelActionManager"):
        m.get_medium_level_actions


def test_gridworld_copies_and_pickles_without_device_state():
    import copy
    import pickle

    from overcooked_ai_amd.mdp import OvercookedGridworld

    mdp = OvercookedGridworld.from_layout_name("asymmetric_advantages")
    mdp._envs[3], mdp._single = object(), object()  # stand-ins for the per-process device plumbing
    for other in (copy.deepcopy(mdp), pickle.loads(pickle.dumps(mdp))):
        assert other == mdp and other._envs == {} and other._single is None and other.terrain_mtx == mdp.terrain_mtx
    assert mdp._single is not None  # the original keeps its own


def test_reference_layout_generator_draws_are_reproduced():
    """layout_gen.generate_reference_layouts restates the reference's LayoutGenerator draw for draw (same numpy legacy-stream
    calls, layout_generator.py:277-420, 493-520): RandomState(0) here yields exactly the 4 096 grids recorded from the
    reference after np.random.seed(0) (package data), so any number of further — unique — terrains can be generated without
    the reference.  (Other shapes / feature sets were checked against the live reference in the build container.)"""
    import gzip

    from overcooked_ai_amd.layout_gen import generate_reference_layouts, reference_generated_layouts

    path = os.path.join(os.path.dirname(L.__file__), "data", "ref_generated_9x5_seed0.json.gz")
    with gzip.open(path, "rt") as f:
        recorded = json.load(f)["grids"]
    mine = generate_reference_layouts(len(recorded) + 50, seed=0)
    assert [s.to_layout_dict()["grid"].split("\n") for s in mine[:len(recorded)]] == [list(g) for g in recorded]
    shipped = reference_generated_layouts(8)
    assert [s.to_layout_dict()["grid"] for s in shipped] == [s.to_layout_dict()["grid"] for s in mine[:8]]
    assert shipped[0].recipe_value((3, 0)) == mine[0].recipe_value((3, 0)) == 20
    # beyond the recorded set: valid, new grids; another seed: another sequence
    tail = {s.to_layout_dict()["grid"] for s in mine[len(recorded):]}
    assert len(tail) >= 49 and all(s.width == 9 and s.height == 5 and s.num_players == 2 for s in mine[-50:])
    assert generate_reference_layouts(3, seed=1)[0].to_layout_dict()["grid"] != mine[0].to_layout_dict()["grid"]


def test_walk_records_equal_a_walk_over_the_grid():
    """planner.walk_records (what k_featurize reads instead of walking the grid): for every (free cell, orientation) state the
    static arg-min keys, the pot order and the goal-counter order equal a brute-force walk over all cells in row-major order
    with the reference's tie rule — the lexicographic minimum of (cost, cell) (min_cost_to_feature, planners.py:391-423)."""
    from overcooked_ai_amd import planner as P
    from overcooked_ai_amd.layouts import spec_from_name

    for name in ("cramped_room", "asymmetric_advantages", "counter_circuit", "forced_coordination", "mdp_test"):
        spec = spec_from_name(name)
        W = spec.width
        for cg in ("none", "all"):
            fi, cost = P.feature_costs(spec, cg)
            rec, stride = P.walk_records(spec, cost)
            assert stride % 16 == 0 and rec.shape == (cost.shape[0], stride)
            blob, offs = P.pack_plan_tables([spec], cg)
            assert len(offs) == 2 and int(blob[offs[1]:offs[1] + 4].view(np.uint32)[0]) == stride
            assert np.array_equal(blob[offs[1] + 16:offs[1] + 16 + rec.size], rec.reshape(-1))
            for s in range(cost.shape[0]):
                words = rec[s, :32].view(np.uint32)
                for k, kind in enumerate("OTDS"):
                    best = 0xFFFFFFFF
                    for (x, y) in spec.cells_of(kind):
                        c = y * W + x
                        if cost[s, c] < 255:
                            best = min(best, (int(cost[s, c]) << 9) | c)
                    assert int(words[k]) == best, (name, cg, s, kind)
                pots = sorted((int(cost[s, y * W + x]) << 9) | (y * W + x) for (x, y) in spec.cells_of("P") if cost[s, y * W + x] < 255)
                assert [int(v) for v in words[4:8]] == (pots + [0xFFFFFFFF] * 4)[:4], (name, cg, s)
                goal = sorted((int(cost[s, y * W + x]), y * W + x) for (x, y) in spec.cells_of("X") if cost[s, y * W + x] < 255)
                assert int(rec[s, 32]) == len(goal) and [int(v) for v in rec[s, 33:33 + len(goal)]] == [c for _, c in goal]
                assert (cg == "all") == (len(goal) > 0) or not spec.cells_of("X")


def test_lazy_infos_dict_behaves_like_the_reference_infos():
    """mdp._Infos builds `event_infos` (25 lists of booleans, mdp.py:1416-1419) from the kernel's bit mask only when somebody
    asks: every way of looking at the dict must still show the reference's three keys."""
    import copy
    import json

    from overcooked_ai_amd.mdp import EVENT_TYPES, _Infos, events_from_mask

    mask = (1 << (2 * 3 + 1)) | (1 << (2 * 10))  # event 3 for agent 1, event 10 for agent 0
    def make():
        d = _Infos(sparse_reward_by_agent=[0, 20], shaped_reward_by_agent=[3, 0])
        d.event_mask, d.num_players = mask, 2
        return d
    full = {"event_infos": events_from_mask(mask, 2), "sparse_reward_by_agent": [0, 20], "shaped_reward_by_agent": [3, 0]}
    assert make()["event_infos"][EVENT_TYPES[3]] == [False, True] and make()["event_infos"][EVENT_TYPES[10]] == [True, False]
    assert "event_infos" in make() and make().get("event_infos") == full["event_infos"] and make().get("phi_s") is None
    assert set(make().keys()) == set(full) and len(make()) == 3 and dict(make().items()) == full
    assert make() == full and dict(make()) == full and {**make()} == full and sorted(make()) == sorted(full)
    assert json.loads(json.dumps(make())) == full and copy.deepcopy(make()) == full
    with pytest.raises(KeyError):
        make()["nope"]
    one = _Infos(sparse_reward_by_agent=[0], shaped_reward_by_agent=[0])
    one.event_mask, one.num_players = 0, 1
    assert all(v == [False] for v in one["event_infos"].values()) and len(one["event_infos"]) == len(EVENT_TYPES)


def test_untile_flags_is_the_inverse_of_the_tiled_layout():
    """OC_OPT_FLAGS_TILED8 stores the flags of env e after step k at [k // 8, e, k % 8]; untile_flags gives [k, e]."""
    import torch

    from overcooked_ai_amd.vec_env import VecOvercookedEnv

    steps, n = 24, 5
    ref = torch.arange(steps * n, dtype=torch.int64).reshape(steps, n).to(torch.uint8)
    tiled = torch.zeros((steps // 8, n, 8), dtype=torch.uint8)
    for k in range(steps):
        tiled[k // 8, :, k % 8] = ref[k]
    assert torch.equal(VecOvercookedEnv.untile_flags(tiled), ref)



def test_lazy_single_state_is_a_plain_state_after_the_first_look():
    """state._LazyState (what the single-state port hands back): packs back from its bytes while nobody has looked, takes its
    timestep from the attribute, and is an ordinary OvercookedState — class included — after the first look, a copy or a pickle."""
    import copy
    import pickle

    from overcooked_ai_amd import OvercookedGridworld
    from overcooked_ai_amd.layouts import spec_from_name
    from overcooked_ai_amd.state import OvercookedState, SingleStateCodec, _LazyState

    spec = spec_from_name("cramped_room")
    codec = SingleStateCodec(spec, 3)
    s0 = OvercookedGridworld.from_layout_name("cramped_room", device="cpu").get_standard_start_state()
    buf = bytearray(48)
    assert codec.pack(s0, memoryview(buf))
    lazy = codec.unpack_lazy(buf)
    assert type(lazy) is _LazyState and isinstance(lazy, OvercookedState) and lazy.timestep == 0
    again = bytearray(48)
    assert codec.pack(lazy, memoryview(again)) and again == buf and type(lazy) is _LazyState
    lazy.timestep = 7
    assert codec.pack(lazy, memoryview(again)) and again[6] == 7 and again[:6] == buf[:6]
    other = SingleStateCodec(spec, 3)  # another codec (another mdp of the same layout): packs it as a plain state
    assert other.pack(codec.unpack_lazy(buf), memoryview(again)) and again == buf
    dup = copy.deepcopy(lazy)
    assert type(dup) is OvercookedState and dup.timestep == 7 and type(lazy) is OvercookedState and "_packed" not in lazy.__dict__
    assert codec.unpack_lazy(buf) == s0 and hash(codec.unpack_lazy(buf)) == hash(s0)
    assert pickle.loads(pickle.dumps(codec.unpack_lazy(buf))) == s0
    assert str(codec.unpack_lazy(buf)) == str(s0) and codec.unpack_lazy(buf).to_dict() == s0.to_dict()
    # an attribute assigned WITHOUT a look stands: the bytes are no longer the state
    lazy = codec.unpack_lazy(buf)
    swapped = tuple(reversed(s0.deepcopy().players))
    lazy.players = swapped
    assert codec.pack(lazy, memoryview(again)) and again != buf and type(lazy) is OvercookedState
    assert lazy.players is swapped and lazy.objects == s0.objects and again[0] == buf[3] and again[3] == buf[0]
    with pytest.raises(AttributeError):
        codec.unpack_lazy(buf).no_such_attribute
    with pytest.raises(TypeError):
        _LazyState()


def test_soup_state_mutators_follow_the_reference():
    """The mutating SoupState API (mdp.py:565-612): same transitions, same errors; cook times resolve against the layout of the
    mdp constructed last (the reference's process-global Recipe.configure).  With /root/reference present the same script runs
    on the reference's SoupState and every observable must agree."""
    from overcooked_ai_amd.mdp import OvercookedGridworld

    def script(Soup, Obj, make_mdp):
        out = []
        make_mdp()  # configures the recipes (cook time 20 for every soup on cramped_room)
        s = Soup.get_soup((2, 0), num_onions=0, num_tomatoes=0)
        out.append((s.ingredients, s.is_idle, s.is_full, s.is_cooking, s.is_ready))
        for call in (lambda: s.begin_cooking(), lambda: s.cook(), lambda: s.pop_ingredient(), lambda: s.auto_finish(),
                     lambda: s.add_ingredient(Obj("dish", (2, 0)))):
            try:
                call()
                out.append("ok")
            except ValueError as e:
                out.append(str(e))
        s.add_ingredient(Obj("onion", (0, 0)))
        s.add_ingredient_from_str("tomato")
        out.append((s.ingredients, s.is_full))
        popped = s.pop_ingredient()
        out.append((popped.name, tuple(popped.position), s.ingredients))
        s.add_ingredient_from_str("onion")
        s.add_ingredient_from_str("onion")
        out.append((s.ingredients, s.is_full))
        try:
            s.add_ingredient_from_str("onion")
        except ValueError as e:
            out.append(str(e))
        s.begin_cooking()
        out.append((s.is_idle, s.is_cooking, s.is_ready, s.cook_time, s.cook_time_remaining, s.is_full))
        for call in (lambda: s.begin_cooking(), lambda: s.pop_ingredient(), lambda: s.add_ingredient_from_str("onion")):
            try:
                call()
            except ValueError as e:
                out.append(str(e))
        for _ in range(20):
            s.cook()
        out.append((s.is_cooking, s.is_ready, s.cook_time_remaining, s.to_dict()["cooking_tick"], s.to_dict()["cook_time"]))
        try:
            s.cook()
        except ValueError as e:
            out.append(str(e))
        f = Soup.get_soup((2, 0), num_onions=2, num_tomatoes=1, finished=True)
        out.append((f.is_ready, f.to_dict()["cooking_tick"], f.ingredients))
        g = Soup.get_soup((2, 0), num_onions=1)
        g.auto_finish()
        out.append((g.is_ready, g.to_dict()["cooking_tick"]))
        return out

    mine = script(S.SoupState, S.ObjectState, lambda: OvercookedGridworld.from_layout_name("cramped_room"))
    assert mine[0] == ([], True, False, False, False)
    assert mine[1:6] == ["Must add at least one ingredient to soup before you can begin cooking",
                         "Must begin cooking before advancing cook tick", "No ingredient to remove",
                         "Cannot finish soup with no ingredients", "Invalid ingredient"]
    assert mine[-2] == (True, 20, ["onion", "onion", "tomato"]) and mine[-1] == (True, 20)
    if os.path.isdir("/root/reference/src/overcooked_ai_py"):
        from oracle import ref_harness

        R = ref_harness.load()
        theirs = script(R.SoupState, R.ObjectState, lambda: R.OvercookedGridworld.from_layout_name("cramped_room"))
        assert mine == theirs

    # a finished soup built by hand goes through the packed format like the reference's
    mdp = OvercookedGridworld.from_layout_name("cramped_room")
    soup = S.SoupState.get_soup((2, 0), num_onions=3, finished=True)
    spec = mdp.spec
    st = mdp.get_standard_start_state()
    st.objects[(2, 0)] = soup
    packed = S.pack_states(spec, [st])
    back = S.unpack_states(spec, packed)[0]
    assert back.objects[(2, 0)].is_ready and back.objects[(2, 0)].ingredients == ["onion"] * 3
