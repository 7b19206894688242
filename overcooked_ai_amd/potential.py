"""Host-side tables for `potential_function` (the phi(s) of potential-based reward shaping).

The reference's `OvercookedGridworld.potential_function(state, mp, gamma)` (mdp.py:2920-3238) mixes three kinds of
quantities:

* per-layout constants — the steady-state value of cooking optimal soups forever, and, for every ingredient
  multiset a pot can hold, the best recipe it can still be completed to (`_get_optimal_possible_recipe`, a DFS
  over `Recipe.neighbors()`, mdp.py:1976-2016, valued by the discounted branch of `get_recipe_value`,
  mdp.py:1603-1628);
* powers `gamma ** k` with integer k (numbers of steps);
* per-state motion costs (`MotionPlanner.min_cost_to_feature`, planners.py:391-423) — the COST tables of
  `overcooked_ai_amd.planner`, plus 1 for the interact action.

Everything of the first two kinds is computed here, in Python floats with the reference's own operand order, and
shipped to the GPU as one record per layout (`PHI_BYTES` bytes, layout below); the kernel `k_potential` then only
multiplies and adds table entries in the reference's order, so phi comes out bit-identical in float64.

Record layout (little endian, 8-byte aligned):
    0    f64  steady_state_value
    8    f64  onion_value    (Recipe._onion_value or 21, mdp.py:2975-2978)
    16   f64  tomato_value   (Recipe._tomato_value or 13)
    24   i32  max_delivery_steps, max_pickup_steps, pot_onion_steps, pot_tomato_steps   (POTENTIAL_CONSTANTS)
    40   f64  sort_value[16]     discounted value of the best completion of multiset k = n_onion + 4 * n_tomato
    168  f64  opt_value_max1[16] max(get_recipe_value(best completion), 1)
    296  f64  value_max1[16]     max(get_recipe_value(multiset itself), 1)
    424  u8   opt_key[16]        best completion as n_onion | n_tomato << 2; entry 0 (no such multiset) carries the set
                             order of the layout's first two pots A, B: bit 0 = B comes first out of
                             list(set().union([A], [B])), bit 1 = B comes first out of list(set().union([B], [A]))
    440  u8   opt_time[16]       Recipe.time of the best completion
    456  u8   serve[16]          serve[0] = number of serving cells (255: more than 15, scan the terrain), then the cells
    472  f64  pow[POW_N]         gamma ** k
"""
import struct

import numpy as np

from .layouts import MAX_NUM_INGREDIENTS

POW_N = 512
PHI_BYTES = 472 + 8 * POW_N

POTENTIAL_CONSTANTS = {  # mdp.py:1060-1073
    "default": {"max_delivery_steps": 10, "max_pickup_steps": 10, "pot_onion_steps": 10, "pot_tomato_steps": 10},
    "mdp_test_tomato": {"max_delivery_steps": 4, "max_pickup_steps": 4, "pot_onion_steps": 5, "pot_tomato_steps": 6},
}


def potential_params(spec, gamma):
    """The `potential_params` dict of mdp.py:2972-2982."""
    c = spec.recipe_config
    p = {"gamma": gamma,
         "tomato_value": c.get("tomato_value") if c.get("tomato_value") else 13,
         "onion_value": c.get("onion_value") if c.get("onion_value") else 21}
    p.update(POTENTIAL_CONSTANTS.get(spec.layout_name, POTENTIAL_CONSTANTS["default"]))
    return p


def discounted_value(spec, key, base_key, params):
    """get_recipe_value(..., discounted=True, base_recipe=base), mdp.py:1603-1628; operand order kept."""
    n_onions = key[0] - (base_key[0] if base_key else 0)
    n_tomatoes = key[1] - (base_key[1] if base_key else 0)
    gamma = params["gamma"]
    return (gamma ** spec.recipe_time(key)
            * gamma ** (params["pot_onion_steps"] * n_onions)
            * gamma ** (params["pot_tomato_steps"] * n_tomatoes)
            * spec.delivery_value(key))


def optimal_possible_recipe(spec, start_key, params):
    """_get_optimal_possible_recipe with discounting, mdp.py:1976-2016: depth-first over the recipes reachable by
    adding ingredients, first strictly-better value wins.  start_key None = empty pot.  -> (best key, best value)."""
    visited, stack = set(), []
    best_key, best_value = start_key, 0
    if start_key is None:
        stack += [(1, 0), (0, 1)]  # Recipe.ALL_INGREDIENTS order (mdp.py:23)
    else:
        stack.append(start_key)
    while stack:
        cur = stack.pop()
        if cur in visited:
            continue
        visited.add(cur)
        value = discounted_value(spec, cur, start_key, params)
        if value > best_value:
            best_value, best_key = value, cur
        if sum(cur) < MAX_NUM_INGREDIENTS:
            for nb in ((cur[0] + 1, cur[1]), (cur[0], cur[1] + 1)):
                if nb not in visited:
                    stack.append(nb)
    return best_key, best_value


def _tuple2_hash(x, y):
    """CPython's tuplehash for a 2-tuple of small non-negative ints (Objects/tupleobject.c, 3.8+)."""
    m = (1 << 64) - 1
    p1, p2, p5 = 11400714785074694791, 14029467366897019727, 2870177450012600261
    acc = p5
    for lane in (x, y):
        acc = (acc + lane * p2) & m
        acc = ((acc << 31) | (acc >> 33)) & m
        acc = (acc * p1) & m
    acc = (acc + (2 ^ (p5 ^ 3527539))) & m
    return 1546275796 if acc == m else acc


def py_set_order(positions):
    """Iteration order of `set().union(positions)` for distinct (x, y) tuples, restating CPython 3.8-3.12's
    open-addressing set (Objects/setobject.c: set_add_entry, set_table_resize, set_insert_clean) — the order
    `get_partially_full_pots` (mdp.py:1882-1890) hands to potential_function.  Independent of the interpreter
    running this module, and identical to what k_potential computes on the GPU."""
    table, mask, fill = [None] * 8, 7, 0

    def insert_clean(tab, msk, h, v):
        perturb, i = h, h & msk
        while True:
            if tab[i] is None:
                tab[i] = (h, v)
                return
            if i + 9 <= msk:
                for j in range(1, 10):
                    if tab[i + j] is None:
                        tab[i + j] = (h, v)
                        return
            perturb >>= 5
            i = (i * 5 + 1 + perturb) & msk

    for pos in positions:
        h = _tuple2_hash(int(pos[0]), int(pos[1]))
        perturb, i = h, h & mask
        while True:
            probes = 9 if i + 9 <= mask else 0
            slot = next((i + j for j in range(probes + 1) if table[i + j] is None), None)
            if slot is not None:
                table[slot] = (h, pos)
                break
            perturb >>= 5
            i = (i * 5 + 1 + perturb) & mask
        fill += 1
        if fill * 5 >= mask * 3:
            newsize = 8
            while newsize <= fill * 4:
                newsize <<= 1
            new = [None] * newsize
            for ent in table:
                if ent is not None:
                    insert_clean(new, newsize - 1, ent[0], ent[1])
            table, mask = new, newsize - 1
    return [ent[1] for ent in table if ent is not None]


def phi_record(spec, gamma=0.99):
    """The PHI_BYTES record of one layout for one discount factor."""
    gamma = float(gamma)
    params = potential_params(spec, gamma)
    opt_key, disc_value = optimal_possible_recipe(spec, None, params)
    if opt_key is None or not spec.delivery_value(opt_key):
        raise ValueError("potential_function needs at least one order with a positive value")
    opt_value = spec.delivery_value(opt_key)
    discount = disc_value / opt_value
    steady = (discount / (1 - discount)) * opt_value  # mdp.py:2998-3000
    sort_value, opt_value_max1, value_max1 = [0.0] * 16, [0.0] * 16, [0.0] * 16
    opt_keys, opt_times = [0] * 16, [0] * 16
    for n_t in range(4):
        for n_o in range(4 - n_t):
            if n_o + n_t == 0:
                continue
            key, k = (n_o, n_t), n_o + 4 * n_t
            best, value = optimal_possible_recipe(spec, key, params)
            t = spec.recipe_time(best)
            if not 0 <= t < 255:
                raise ValueError("cook time out of range")
            sort_value[k] = float(value)
            opt_value_max1[k] = float(max(spec.delivery_value(best), 1))
            value_max1[k] = float(max(spec.delivery_value(key), 1))
            opt_keys[k] = best[0] | (best[1] << 2)
            opt_times[k] = t
    pots = spec.cells_of("P")
    if len(pots) >= 2:  # set order of the first two pots, for k_potential2
        a, b = pots[0], pots[1]
        opt_keys[0] = (1 if py_set_order([a, b])[0] == b else 0) | (2 if py_set_order([b, a])[0] == b else 0)
    steps = (params["max_delivery_steps"], params["max_pickup_steps"], params["pot_onion_steps"], params["pot_tomato_steps"])
    if 3 * max(steps) + 256 > POW_N:
        raise ValueError("potential constants too large for the power table")
    rec = struct.pack("<3d4i", steady, float(params["onion_value"]), float(params["tomato_value"]), *steps)
    rec += struct.pack("<16d", *sort_value) + struct.pack("<16d", *opt_value_max1) + struct.pack("<16d", *value_max1)
    rec += bytes(opt_keys) + bytes(opt_times)
    serve = [y * spec.width + x for (x, y) in spec.cells_of("S")]
    rec += bytes([len(serve)] + serve + [0] * (15 - len(serve))) if len(serve) <= 15 else bytes([255] + [0] * 15)
    rec += struct.pack("<%dd" % POW_N, *[gamma ** k for k in range(POW_N)])
    assert len(rec) == PHI_BYTES
    return rec


def pack_phi_tables(specs, gamma=0.99):
    """uint8 [n_layouts, PHI_BYTES]: the records of a layout table."""
    return np.frombuffer(b"".join(phi_record(s, gamma) for s in specs), dtype=np.uint8).reshape(len(specs), PHI_BYTES).copy()
