"""Random layout generation (host side) for batches with per-env terrain (BASELINE configs[4]).

Plays the role of the reference's `LayoutGenerator` (layout_generator.py:99-610: dig a connected free region into
a grid of counters, turn some counters that touch it into pots / dispensers / serving locations, embed the grid in
an outer shape padded with counters, drop the players on free cells).  Two generators:

* `reference_draw_grid` / `generate_reference_layouts`: the reference's procedure restated draw for draw — the same
  `numpy.random` legacy-stream calls with the same arguments in the same order (`make_disjoint_sets_layout`,
  layout_generator.py:277-420, 493-520) — so that `RandomState(seed)` here yields the grids `np.random.seed(seed)` yields
  there: a fully unique set of any size (e.g. one terrain per env of the 1 M-env configuration) can be produced wherever the
  GPU is, without the reference.  Pinned by the 4 096 grids recorded from the reference itself (package data, below).
* `generate_grid` / `generate_layouts`: an independent generator (randomized region growth, its own
  `numpy.random.Generator`; tomatoes, other shapes, at most two pots) used by the soak test.

Every grid passes the reference's validity rules (`LayoutSpec._assert_valid_grid` = mdp.py:2064-2115) and can equally be fed to
the reference via `OvercookedGridworld.from_grid`.  This is tooling around the hot path, not part of it.
"""
import numpy as np

from .layouts import LayoutSpec

DEFAULT_FEATURES = ("P", "O", "D", "S")  # like DEFAULT_FEATURE_TYPES (layout_generator.py:90-96): no tomato dispenser


def generate_grid(rng, inner_shape=(9, 5), outer_shape=None, prop_empty=0.9, prop_feats=0.1,
                  feature_types=DEFAULT_FEATURES, n_players=2):
    """One random grid as a list of row strings (with player digits)."""
    W, H = inner_shape
    OW, OH = outer_shape or inner_shape
    assert W >= 4 and H >= 4 and OW >= W and OH >= H
    interior = [(x, y) for y in range(1, H - 1) for x in range(1, W - 1)]
    target = int(round(prop_empty * len(interior)))
    target = min(len(interior), max(n_players + 1, target))
    grid = [["X"] * W for _ in range(H)]
    # grow a connected free region from a random interior cell
    start = interior[int(rng.integers(len(interior)))]
    free = {start}
    frontier = [start]
    while len(free) < target and frontier:
        cx, cy = frontier[int(rng.integers(len(frontier)))]
        nbrs = [(cx + dx, cy + dy) for dx, dy in ((1, 0), (-1, 0), (0, 1), (0, -1))]
        nbrs = [p for p in nbrs if 1 <= p[0] < W - 1 and 1 <= p[1] < H - 1 and p not in free]
        if not nbrs:
            frontier.remove((cx, cy))
            continue
        p = nbrs[int(rng.integers(len(nbrs)))]
        free.add(p)
        frontier.append(p)
    for x, y in free:
        grid[y][x] = " "
    # counters that touch the free region can carry a feature
    touching = [(x, y) for y in range(H) for x in range(W) if grid[y][x] == "X" and any(
        (x + dx, y + dy) in free for dx, dy in ((1, 0), (-1, 0), (0, 1), (0, -1)))]
    order = list(rng.permutation(len(touching)))
    n_feats = max(len(feature_types), int(round(prop_feats * len(touching))))
    n_feats = min(n_feats, len(touching))
    for k in range(n_feats):
        x, y = touching[order[k]]
        # one of every required type first, then random extras (at most 2 pots keep the fast kernels applicable)
        if k < len(feature_types):
            f = feature_types[k]
        else:
            f = feature_types[int(rng.integers(len(feature_types)))]
            if f == "P" and sum(row.count("P") for row in grid) >= 2:
                f = "O"
        grid[y][x] = f
    cells = sorted(free)
    picks = rng.choice(len(cells), size=n_players, replace=False)
    for i, k in enumerate(picks):
        x, y = cells[int(k)]
        grid[y][x] = str(i + 1)
    # embed in the outer shape, padded with counters (layout_generator.py:309-329)
    ox = int(rng.integers(0, OW - W + 1))
    oy = int(rng.integers(0, OH - H + 1))
    outer = [["X"] * OW for _ in range(OH)]
    for y in range(H):
        for x in range(W):
            outer[y + oy][x + ox] = grid[y][x]
    return ["".join(r) for r in outer]


def generate_layouts(n, seed=0, inner_shape=(9, 5), outer_shape=None, prop_empty=0.9, prop_feats=0.1,
                     feature_types=DEFAULT_FEATURES, base_params=None):
    """n LayoutSpecs with random terrains; `base_params` are the recipe / order parameters shared by all of them
    (default: the reference's DEFAULT_MDP_GEN_PARAMS — one 3-onion order worth 20 cooking for 20 steps,
    layout_generator.py:40-48)."""
    rng = np.random.default_rng(seed)
    params = dict(base_params or {"start_all_orders": [{"ingredients": ["onion", "onion", "onion"]}],
                                  "recipe_values": [20], "recipe_times": [20]})
    specs = []
    for _ in range(n):
        rows = generate_grid(rng, inner_shape, outer_shape, prop_empty, prop_feats, feature_types)
        d = dict(params)
        d["grid"] = "\n".join(rows)
        specs.append(LayoutSpec(d))
    return specs


_NBRS = ((0, -1), (0, 1), (1, 0), (-1, 0))  # Direction.ALL_DIRECTIONS order (actions.py:12-16)


def reference_draw_grid(rs, inner_shape=(9, 5), outer_shape=(9, 5), prop_empty=0.9, prop_feats=0.1,
                        feature_types=DEFAULT_FEATURES):
    """One grid (list of row strings with the player digits) from the legacy stream `rs` (a `numpy.random.RandomState`, or
    the `numpy.random` module for the global one), consuming exactly the draws of the reference's
    `LayoutGenerator.make_disjoint_sets_layout` + `embed_grid` + `get_random_starting_positions`."""
    W, H = inner_shape
    OW, OH = outer_shape
    assert W <= OW and H <= OH, "inner_shape cannot fit into the outer shape"
    free, comp = set(), {}  # dug cells; cell -> component label (what the reference's DisjointSets tracks)
    eligible = W * H - 2 * (W + H) + 4
    n_comp = 0

    def interior(w, h):
        return int(rs.randint(low=1, high=w - 1)), int(rs.randint(low=1, high=h - 1))

    while not (len(free) / float(eligible) > prop_empty and n_comp == 1):  # dig until enough is free AND it is one region
        loc = interior(W, H)
        while loc in free:
            loc = interior(W, H)
        free.add(loc)
        comp[loc] = loc
        n_comp += 1
        for dx, dy in _NBRS:
            nb = (loc[0] + dx, loc[1] + dy)
            if nb in comp and comp[nb] != comp[loc]:
                old, new = comp[nb], comp[loc]
                for c in comp:
                    if comp[c] == old:
                        comp[c] = new
                n_comp -= 1
    cell = {(x, y): (" " if (x, y) in free else "X") for x in range(W) for y in range(H)}

    def can_hold_feature(p):  # a counter with a free neighbour
        return cell[p] == "X" and any(cell.get((p[0] + dx, p[1] + dy)) == " " for dx, dy in _NBRS)

    spots = np.array([(x, y) for x in range(W) for y in range(H) if can_hold_feature((x, y))])
    rs.shuffle(spots)  # (rows of an [n, 2] array, like the reference)
    assert len(spots) > len(feature_types)
    placed = 0
    for x, y in spots:
        if placed < len(feature_types):
            cell[(int(x), int(y))] = feature_types[placed]  # one of every type first
        elif placed / len(spots) >= prop_feats:
            break
        else:
            cell[(int(x), int(y))] = str(rs.choice(feature_types))
        placed += 1
    sx = int(rs.randint(0, OW - W)) if OW - W else 0
    sy = int(rs.randint(0, OH - H)) if OH - H else 0
    outer = {(x, y): "X" for x in range(OW) for y in range(OH)}
    for (x, y), ch in cell.items():
        outer[(x + sx, y + sy)] = ch

    def empty_cell():
        p = interior(OW, OH)
        while outer[p] != " ":
            p = interior(OW, OH)
        return p

    p0 = empty_cell()
    p1 = empty_cell()
    while p0 == p1:
        p0 = empty_cell()
    outer[p0], outer[p1] = "1", "2"
    return ["".join(outer[(x, y)] for x in range(OW)) for y in range(OH)]


def generate_reference_layouts(n, seed=0, inner_shape=(9, 5), outer_shape=(9, 5), prop_empty=0.9, prop_feats=0.1,
                               feature_types=DEFAULT_FEATURES, base_params=None):
    """n LayoutSpecs: what the reference's `LayoutGenerator.mdp_gen_fn_from_dict({inner_shape, prop_empty, prop_feats, ...},
    outer_shape)` yields, call after call, after `np.random.seed(seed)` — from a private RandomState (the global stream is not
    touched).  The first 4 096 grids of the default arguments are the package data `reference_generated_layouts` ships."""
    rs = np.random.RandomState(seed)
    params = dict(base_params or {"start_all_orders": [{"ingredients": ["onion", "onion", "onion"]}],
                                  "recipe_values": [20], "recipe_times": [20]})
    specs = []
    for _ in range(n):
        d = dict(params)
        d["grid"] = "\n".join(reference_draw_grid(rs, inner_shape, outer_shape, prop_empty, prop_feats, feature_types))
        specs.append(LayoutSpec(d))
    return specs


def reference_generated_layouts(n=None):
    """The terrains BASELINE configs[4] names: what the reference's own `LayoutGenerator.mdp_gen_fn_from_dict(
    {inner_shape (9, 5), prop_empty 0.9, prop_feats 0.1, one 3-onion order worth 20 cooking for 20}, outer_shape=(9, 5))`
    yields after `np.random.seed(0); random.seed(0)` — 4096 grids recorded from the reference (oracle/gen_golden.py
    --generated-layouts-only) and shipped as package data, since the generator itself is host-side tooling of the
    reference and draws from numpy's global stream."""
    import gzip
    import json
    import os

    path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "data", "ref_generated_9x5_seed0.json.gz")
    with gzip.open(path, "rt") as f:
        d = json.load(f)
    params = {k: v for k, v in d["mdp_params"].items() if k in ("start_all_orders", "recipe_values", "recipe_times")}
    grids = d["grids"] if n is None else d["grids"][:n]
    specs = []
    for rows in grids:
        cfg = dict(params)
        cfg["grid"] = "\n".join(rows)
        specs.append(LayoutSpec(cfg))
    return specs
