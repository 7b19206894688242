"""Layout loading and the host-side layout compiler.

A layout is what the reference stores in a `.layout` file (a dict literal with a `grid` string and the
recipe / order / reward-shaping parameters, src/overcooked_ai_py/data/layouts/*.layout, loaded by
`read_layout_dict`, utils.py:223-226) plus the keyword overrides of
`OvercookedGridworld.from_layout_name(name, **params)` (mdp.py:1150-1172).  `compile_layout` flattens one
such configuration into the 256-byte `OcLayout` record of include/oc_amd.h: terrain codes, pot slots,
start positions and the two 16-entry look-up tables (cook time, delivery value) the kernels index by
(n_onion, n_tomato).  The process-global `Recipe` configuration of the reference (mdp.py:221-336) becomes
per-layout data here, so layouts with different recipe settings can live in one batch.
"""
import ast
import json
import math
import os
import struct

import numpy as np

_DATA = os.path.join(os.path.dirname(os.path.abspath(__file__)), "data", "layouts.json")

MAX_CELLS = 128
MAX_POTS = 8
LAYOUT_BYTES = 256
MAX_NUM_INGREDIENTS = 3

# the reference's own code table, layout_generator.py:18-27
TERRAIN_CODE = {" ": 0, "X": 1, "O": 2, "T": 3, "P": 4, "D": 5, "S": 6}
CODE_TERRAIN = {v: k for k, v in TERRAIN_CODE.items()}

BASE_REW_SHAPING_PARAMS = {  # mdp.py:1018-1025
    "PLACEMENT_IN_POT_REW": 3,
    "DISH_PICKUP_REWARD": 3,
    "SOUP_PICKUP_REWARD": 5,
    "DISH_DISP_DISTANCE_REW": 0,
    "POT_DISTANCE_REW": 0,
    "SOUP_DISTANCE_REW": 0,
}

_registry = None


def layout_registry():
    global _registry
    if _registry is None:
        with open(_DATA) as f:
            _registry = json.load(f)
    return _registry


def layout_names():
    return sorted(layout_registry())


def read_layout_dict(layout_name):
    """Same role as the reference's utils.read_layout_dict: name -> dict with a `grid` string."""
    reg = layout_registry()
    if layout_name not in reg:
        raise FileNotFoundError("unknown layout %r" % (layout_name,))
    d = json.loads(json.dumps(reg[layout_name]))
    d["grid"] = "\n".join(d["grid"])
    return d


def load_layout_file(path):
    """Parse a reference-format `.layout` file (a Python dict literal) without eval()."""
    text = open(path).read().replace("float('inf')", "1e999").replace('float("inf")', "1e999")
    return ast.literal_eval(text)


def recipe_key(ingredients):
    """A recipe is its ingredient multiset (Recipe.ingredients is sorted, mdp.py:126-128)."""
    n_o = sum(1 for i in ingredients if i == "onion")
    n_t = sum(1 for i in ingredients if i == "tomato")
    if n_o + n_t != len(ingredients) or not 1 <= len(ingredients) <= MAX_NUM_INGREDIENTS:
        raise ValueError("Invalid recipe %r" % (ingredients,))
    return (n_o, n_t)


ALL_RECIPE_KEYS = [(n_o, n - n_o) for n in (1, 2, 3) for n_o in range(n, -1, -1)]


class LayoutSpec:
    """One fully-resolved OvercookedGridworld configuration (the kwargs of mdp.py:1090-1103)."""

    def __init__(self, layout_dict, **params_to_overwrite):
        cfg = dict(layout_dict)
        cfg.update(params_to_overwrite)
        grid = cfg.pop("grid")
        rows = grid.split("\n") if isinstance(grid, str) else list(grid)
        rows = [list(r.strip()) if isinstance(r, str) else list(r) for r in rows]
        self._assert_valid_grid(rows)
        self.height, self.width = len(rows), len(rows[0])
        starts = [None] * 9
        for y, row in enumerate(rows):
            for x, c in enumerate(row):
                if c in "123456789":
                    assert starts[int(c) - 1] is None, "Duplicate player in grid"
                    starts[int(c) - 1] = (x, y)
                    row[x] = " "
        self.start_player_positions = [p for p in starts if p is not None]
        self.num_players = len(self.start_player_positions)
        self.terrain_mtx = rows
        self.layout_name = cfg.pop("layout_name", None) or "|".join("".join(r) for r in rows)
        self.start_all_orders = cfg.pop("start_all_orders", None) or []
        self.start_bonus_orders = cfg.pop("start_bonus_orders", None) or []
        self.rew_shaping_params = cfg.pop("rew_shaping_params", None) or dict(BASE_REW_SHAPING_PARAMS)
        self.order_bonus = cfg.pop("order_bonus", 2)
        self.old_dynamics = bool(cfg.pop("old_dynamics", False))
        self.num_items_for_soup = cfg.pop("num_items_for_soup", 3)
        self.start_state = cfg.pop("start_state", None)
        # what is left is the Recipe configuration (mdp.py:1224-1232 passes **kwargs through)
        self.recipe_config = cfg
        self._check_recipe_config()
        if self.old_dynamics:
            orders = self.start_all_orders or [{"ingredients": ["onion"] * a + ["tomato"] * b} for a, b in ALL_RECIPE_KEYS]
            assert all(len(o["ingredients"]) == 3 for o in orders), \
                "Only accept orders with 3 items when using the old_dynamics"  # mdp.py:1121-1127

    # -- validation: same acceptance rules and messages as OvercookedGridworld._assert_valid_grid (mdp.py:2064-2115) --
    @staticmethod
    def _assert_valid_grid(rows):
        H, W = len(rows), len(rows[0])
        assert all(len(r) == W for r in rows), "Ragged grid"
        assert W * H <= MAX_CELLS, "grid larger than %d cells" % MAX_CELLS
        solid = set("XOPDST")
        border = {"Left": [r[0] for r in rows], "Right": [r[-1] for r in rows], "Top": rows[0], "Bottom": rows[-1]}
        for side, cells in border.items():
            assert all(c in solid for c in cells), "%s border must not be free" % side
        flat = "".join("".join(r) for r in rows)
        players = sorted(int(c) for c in flat if c.isdigit() and c != "0")
        assert players, "No players (digits) in grid"
        assert players == list(range(1, len(players) + 1)), "Some players were missing"
        assert set(flat) <= set("XOPDST123456789 "), "Invalid character in grid"
        for ch in "DSP":
            assert ch in flat, "'%s' must be present at least once" % ch
        assert "O" in flat or "T" in flat, "'O' or 'T' must be present at least once"

    def _check_recipe_config(self):
        c = self.recipe_config
        # the mutual-exclusion rules of Recipe.configure, mdp.py:238-300
        if ("tomato_time" in c) != ("onion_time" in c):
            raise ValueError("Must specify both 'onion_time' and 'tomato_time'")
        if ("tomato_value" in c) != ("onion_value" in c):
            raise ValueError("Must specify both 'onion_value' and 'tomato_value'")
        for a, b in (("tomato_value", "delivery_reward"), ("tomato_value", "recipe_values"),
                     ("recipe_values", "delivery_reward"), ("tomato_time", "cook_time"),
                     ("tomato_time", "recipe_times"), ("recipe_times", "cook_time")):
            if a in c and b in c:
                raise ValueError("%r incompatible with %r" % (a, b))
        for k in ("recipe_values", "recipe_times"):
            if k in c:
                if not self.start_all_orders:
                    raise ValueError("Must specify 'all_orders' if %r specified" % k)
                if len(self.start_all_orders) != len(c[k]):
                    raise ValueError("Number of recipes in 'all_orders' must be the same as number in %r" % k)
        if c.get("max_num_ingredients", 3) != 3:
            raise ValueError("only max_num_ingredients == 3 is supported")

    # -- Recipe.value / Recipe.time with the reference's truthiness rules, mdp.py:136-188 --
    def recipe_value(self, key):
        c = self.recipe_config
        if c.get("delivery_reward"):
            return c["delivery_reward"]
        if c.get("recipe_values"):
            for o, v in zip(self.start_all_orders, c["recipe_values"]):
                if recipe_key(o["ingredients"]) == key:
                    return v
        if c.get("onion_value") and c.get("tomato_value"):
            return c["tomato_value"] * key[1] + c["onion_value"] * key[0]
        return 20

    def recipe_time(self, key):
        c = self.recipe_config
        if c.get("cook_time"):
            return c["cook_time"]
        if c.get("recipe_times"):
            for o, t in zip(self.start_all_orders, c["recipe_times"]):
                if recipe_key(o["ingredients"]) == key:
                    return t
        if c.get("onion_time") and c.get("tomato_time"):
            return c["onion_time"] * key[0] + c["tomato_time"] * key[1]
        return 20

    def delivery_value(self, key):
        """get_recipe_value, non-discounted branch, mdp.py:1595-1602."""
        all_keys = [recipe_key(o["ingredients"]) for o in self.start_all_orders] or ALL_RECIPE_KEYS
        if key not in all_keys:
            return 0
        if key in [recipe_key(o["ingredients"]) for o in self.start_bonus_orders]:
            return self.order_bonus * self.recipe_value(key)
        return self.recipe_value(key)

    def optimal_possible_value(self, key):
        """Value of the best recipe reachable from `key` (n_onion, n_tomato; (0, 0) = empty pot) by adding
        ingredients: _get_optimal_possible_recipe's DFS (mdp.py:1976-2016) + get_recipe_value of its result."""
        best = self.delivery_value(key) if sum(key) >= 1 else 0
        if sum(key) < MAX_NUM_INGREDIENTS:
            best = max(best, self.optimal_possible_value((key[0] + 1, key[1])),
                       self.optimal_possible_value((key[0], key[1] + 1)))
        return best

    def potting_class(self, key, ingredient):
        """Bit mask of the potting events logged when `ingredient` is added to a soup with multiset `key`
        (is_potting_optimal / viable / catastrophic / useless, mdp.py:2256-2308): 1, 2, 4, 8."""
        new = (key[0] + 1, key[1]) if ingredient == "onion" else (key[0], key[1] + 1)
        old_val, new_val = self.optimal_possible_value(key), self.optimal_possible_value(new)
        return ((1 if old_val == new_val else 0) | (2 if new_val > 0 else 0)
                | (4 if (old_val > 0 and new_val == 0) else 0) | (8 if old_val == 0 else 0))

    @property
    def shape(self):
        return (self.width, self.height)

    def cells_of(self, char):
        """Row-major (y then x) like _get_terrain_type_pos_dict, mdp.py:1711-1716."""
        return [(x, y) for y, row in enumerate(self.terrain_mtx) for x, c in enumerate(row) if c == char]

    def padded(self, width, height):
        """Embed at the top-left of a width x height grid of counters (LayoutGenerator.embed_grid pads
        with counters too, layout_generator.py:309-329; dynamics are unchanged by unreachable counters)."""
        assert width >= self.width and height >= self.height
        rows = [["X"] * width for _ in range(height)]
        for y, row in enumerate(self.terrain_mtx):
            for x, c in enumerate(row):
                rows[y][x] = c
        for i, (x, y) in enumerate(self.start_player_positions):
            rows[y][x] = str(i + 1)
        d = self.to_layout_dict()
        d["grid"] = "\n".join("".join(r) for r in rows)
        return LayoutSpec(d)

    def grid_rows(self):
        rows = [list(r) for r in self.terrain_mtx]
        for i, (x, y) in enumerate(self.start_player_positions):
            rows[y][x] = str(i + 1)
        return ["".join(r) for r in rows]

    def to_layout_dict(self):
        """Back to the reference's `.layout` dict form (+ overrides) — the oracle consumes this."""
        d = dict(self.recipe_config)
        d.update(
            grid="\n".join(self.grid_rows()),
            layout_name=self.layout_name,
            start_all_orders=self.start_all_orders,
            start_bonus_orders=self.start_bonus_orders,
            rew_shaping_params=dict(self.rew_shaping_params),
            order_bonus=self.order_bonus,
            old_dynamics=self.old_dynamics,
        )
        return d


def spec_from_name(layout_name, **params_to_overwrite):
    d = read_layout_dict(layout_name)
    d["layout_name"] = layout_name
    return LayoutSpec(d, **params_to_overwrite)


def compile_layout(spec):
    """LayoutSpec -> np.uint8[256], the OcLayout record (include/oc_amd.h)."""
    if spec.num_players not in (1, 2):
        raise ValueError("the MI355X path covers 1- and 2-player layouts (got %d players)" % spec.num_players)
    W, H = spec.width, spec.height
    n_cells = W * H
    pots = spec.cells_of("P")
    if len(pots) > MAX_POTS:
        raise ValueError("more than %d pots" % MAX_POTS)
    buf = bytearray(LAYOUT_BYTES)
    n_obj_planes = (n_cells + 15) // 16
    struct.pack_into("<8B", buf, 0, W, H, n_cells, len(pots), spec.num_players, int(spec.old_dynamics),
                     n_obj_planes, 0)
    for i in range(2):
        if i < spec.num_players:
            x, y = spec.start_player_positions[i]
            buf[8 + i] = y * W + x
        else:
            buf[8 + i] = 0xFF
        buf[10 + i] = 0  # NORTH, mdp.py:947
    for k in range(MAX_POTS):
        buf[16 + k] = (pots[k][1] * W + pots[k][0]) if k < len(pots) else 0xFF
    for n_o in range(3):
        for n_t in range(3 - n_o):
            buf[24 + n_o + 3 * n_t] = spec.potting_class((n_o, n_t), "onion") | (spec.potting_class((n_o, n_t), "tomato") << 4)
    rew = spec.rew_shaping_params
    struct.pack_into("<4f", buf, 32, float(rew["PLACEMENT_IN_POT_REW"]), float(rew["DISH_PICKUP_REWARD"]),
                     float(rew["SOUP_PICKUP_REWARD"]), 0.0)
    for n_o, n_t in ALL_RECIPE_KEYS:
        idx = n_o + 4 * n_t
        t = spec.recipe_time((n_o, n_t))
        if t != int(t) or not 0 < t <= 254:
            raise ValueError("cook time %r of recipe %r is not an integer in 1..254" % (t, (n_o, n_t)))
        buf[48 + idx] = int(t)
        v = float(spec.delivery_value((n_o, n_t)))
        if math.isinf(v) or math.isnan(v):
            # tutorial_3.layout sets order_bonus = inf; f32 carries inf fine, nan (inf*0) never arises here
            pass
        struct.pack_into("<f", buf, 64 + 4 * idx, v)
    pot_slot = {p: k for k, p in enumerate(pots)}
    for y in range(H):
        for x in range(W):
            c = spec.terrain_mtx[y][x]
            code = TERRAIN_CODE[c]
            if c == "P":
                code |= pot_slot[(x, y)] << 3
            buf[128 + y * W + x] = code
    return np.frombuffer(bytes(buf), dtype=np.uint8).copy()


class LayoutTable:
    """A batch-wide table of compiled layouts sharing one grid shape (padded on request)."""

    def __init__(self, specs, pad_to=None):
        specs = list(specs)
        assert len(specs) >= 1
        if pad_to is None and len({s.shape for s in specs}) > 1:
            pad_to = (max(s.width for s in specs), max(s.height for s in specs))
        if pad_to is not None:
            specs = [s if s.shape == tuple(pad_to) else s.padded(*pad_to) for s in specs]
        self.specs = specs
        self.width, self.height = specs[0].shape
        self.n_cells = self.width * self.height
        self.n_obj_planes = (self.n_cells + 15) // 16
        self.n_planes = 1 + self.n_obj_planes
        self.records = np.stack([compile_layout(s) for s in specs])  # [L, 256] u8
        self.max_pots = max(len(s.cells_of("P")) for s in specs)

    def __len__(self):
        return len(self.specs)
