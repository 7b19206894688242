"""ctypes front-end of the C oracle (oracle/overcooked_oracle.c).  TEST INFRASTRUCTURE ONLY.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module; the
shipped package `overcooked_ai_amd` never does.  The oracle is a CPU restatement of the
reference's hot path and is pinned against fixtures generated from the real reference
(tests/golden/, produced by oracle/gen_golden.py).

A layout is described here in the reference's own terms: the dict literal stored in a
`.layout` file (src/overcooked_ai_py/data/layouts/*.layout, read by utils.py:223-226) plus
keyword overrides such as old_dynamics (OvercookedGridworld.from_layout_name, mdp.py:1150-1172).
This module parses that dict independently of overcooked_ai_amd.layouts.
"""
import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SRC = os.path.join(_HERE, "overcooked_oracle.c")
_BUILD = os.path.join(_HERE, "_build")
_LIB = os.path.join(_BUILD, "liboracle.so")

MAX_CELLS = 128
MAX_RECIPES = 9


class OracleMdp(ctypes.Structure):
    _fields_ = [
        ("width", ctypes.c_int32),
        ("height", ctypes.c_int32),
        ("n_players", ctypes.c_int32),
        ("start_x", ctypes.c_int32 * 2),
        ("start_y", ctypes.c_int32 * 2),
        ("old_dynamics", ctypes.c_int32),
        ("max_num_ingredients", ctypes.c_int32),
        ("n_all_orders", ctypes.c_int32),
        ("all_orders", (ctypes.c_int32 * 2) * MAX_RECIPES),
        ("n_bonus_orders", ctypes.c_int32),
        ("bonus_orders", (ctypes.c_int32 * 2) * MAX_RECIPES),
        ("has_cook_time", ctypes.c_int32),
        ("has_delivery_reward", ctypes.c_int32),
        ("has_recipe_values", ctypes.c_int32),
        ("has_recipe_times", ctypes.c_int32),
        ("has_ingredient_value", ctypes.c_int32),
        ("has_ingredient_time", ctypes.c_int32),
        ("cook_time", ctypes.c_double),
        ("delivery_reward", ctypes.c_double),
        ("recipe_values", ctypes.c_double * MAX_RECIPES),
        ("recipe_times", ctypes.c_double * MAX_RECIPES),
        ("onion_value", ctypes.c_double),
        ("tomato_value", ctypes.c_double),
        ("onion_time", ctypes.c_double),
        ("tomato_time", ctypes.c_double),
        ("order_bonus", ctypes.c_double),
        ("rew_placement_in_pot", ctypes.c_double),
        ("rew_dish_pickup", ctypes.c_double),
        ("rew_soup_pickup", ctypes.c_double),
        ("terrain", ctypes.c_char * MAX_CELLS),
    ]


def build(force=False):
    """gcc -O2 -shared the C restatement into oracle/_build/liboracle.so."""
    if not force and os.path.exists(_LIB) and os.path.getmtime(_LIB) >= os.path.getmtime(_SRC):
        return _LIB
    os.makedirs(_BUILD, exist_ok=True)
    tmp = _LIB + ".%d.tmp" % os.getpid()
    base = ["gcc", "-O2", "-std=c99", "-ffp-contract=off", "-fPIC", "-shared", "-Wall"]
    try:  # OpenMP only parallelises the env loop of oracle_rollout_random (bench.py's all-cores CPU baseline)
        subprocess.check_call(base + ["-fopenmp", "-o", tmp, _SRC, "-lm"], stderr=subprocess.DEVNULL)
    except (subprocess.CalledProcessError, OSError):
        subprocess.check_call(base + ["-o", tmp, _SRC, "-lm"])
    os.replace(tmp, _LIB)
    return _LIB


_lib = None


def lib():
    global _lib
    if _lib is None:
        path = _LIB if os.path.exists(_LIB) and not os.path.exists(_SRC) else build()
        L = ctypes.CDLL(path)
        assert L.oracle_mdp_size() == ctypes.sizeof(OracleMdp), "OracleMdp struct mismatch"
        _lib = L
    return _lib


def _ptr(a, typ):
    if a is None:
        return None
    assert a.flags["C_CONTIGUOUS"]
    return a.ctypes.data_as(ctypes.POINTER(typ))


def _recipe_key(ingredients):
    n_o = sum(1 for i in ingredients if i == "onion")
    n_t = sum(1 for i in ingredients if i == "tomato")
    assert n_o + n_t == len(ingredients)
    return n_o, n_t


def mdp_from_layout_dict(layout, **overrides):
    """layout: dict as stored in a reference `.layout` file; overrides as in from_layout_name(**params)."""
    cfg = dict(layout)
    cfg.update(overrides)
    rows = [r.strip() for r in cfg.pop("grid").split("\n")]  # mdp.py:1169
    rows = [list(r) for r in rows]
    H, W = len(rows), len(rows[0])
    assert all(len(r) == W for r in rows) and W * H <= MAX_CELLS
    m = OracleMdp()
    m.width, m.height = W, H
    starts = {}
    for y, row in enumerate(rows):
        for x, c in enumerate(row):
            if c in "123456789":  # mdp.py:1193-1203
                starts[int(c) - 1] = (x, y)
                rows[y][x] = " "
    n_players = len(starts)
    assert n_players in (1, 2), "oracle covers 1- and 2-player layouts"
    m.n_players = n_players
    for i in range(n_players):
        m.start_x[i], m.start_y[i] = starts[i]
    m.terrain = "".join("".join(r) for r in rows).encode()
    m.old_dynamics = int(bool(cfg.get("old_dynamics", False)))
    m.max_num_ingredients = int(cfg.get("max_num_ingredients", 3))  # Recipe.configure, mdp.py:225
    all_orders = cfg.get("start_all_orders") or []
    m.n_all_orders = len(all_orders)
    for i, o in enumerate(all_orders):
        m.all_orders[i][0], m.all_orders[i][1] = _recipe_key(o["ingredients"])
    bonus = cfg.get("start_bonus_orders") or []
    m.n_bonus_orders = len(bonus)
    for i, o in enumerate(bonus):
        m.bonus_orders[i][0], m.bonus_orders[i][1] = _recipe_key(o["ingredients"])
    if "cook_time" in cfg:
        m.has_cook_time, m.cook_time = 1, float(cfg["cook_time"])
    if "delivery_reward" in cfg:
        m.has_delivery_reward, m.delivery_reward = 1, float(cfg["delivery_reward"])
    if "recipe_values" in cfg:
        m.has_recipe_values = 1
        for i, v in enumerate(cfg["recipe_values"]):
            m.recipe_values[i] = float(v)
    if "recipe_times" in cfg:
        m.has_recipe_times = 1
        for i, v in enumerate(cfg["recipe_times"]):
            m.recipe_times[i] = float(v)
    if "onion_value" in cfg:
        m.has_ingredient_value = 1
        m.onion_value, m.tomato_value = float(cfg["onion_value"]), float(cfg["tomato_value"])
    if "onion_time" in cfg:
        m.has_ingredient_time = 1
        m.onion_time, m.tomato_time = float(cfg["onion_time"]), float(cfg["tomato_time"])
    m.order_bonus = float(cfg.get("order_bonus", 2))  # mdp.py:1099
    rew = cfg.get("rew_shaping_params") or {  # BASE_REW_SHAPING_PARAMS, mdp.py:1018-1025
        "PLACEMENT_IN_POT_REW": 3,
        "DISH_PICKUP_REWARD": 3,
        "SOUP_PICKUP_REWARD": 5,
    }
    m.rew_placement_in_pot = float(rew["PLACEMENT_IN_POT_REW"])
    m.rew_dish_pickup = float(rew["DISH_PICKUP_REWARD"])
    m.rew_soup_pickup = float(rew["SOUP_PICKUP_REWARD"])
    return m


class OracleStartSpec(ctypes.Structure):
    _fields_ = [("seed", ctypes.c_uint64), ("env_offset", ctypes.c_int64), ("epoch", ctypes.c_uint32),
                ("random_start_pos", ctypes.c_int32), ("rnd_obj_prob_thresh", ctypes.c_double),
                ("regen_first", ctypes.c_uint32), ("regen_count", ctypes.c_uint32)]


def start_spec(seed=0, env_offset=0, epoch=0, random_start_pos=False, rnd_obj_prob_thresh=0.0, regen=None):
    """start_state_fn of a batch for Oracle.step / rollout_random (restarts at the horizon draw from it).  regen =
    (first, count): a restarting env first moves to a layout drawn from that range (regen_mdp=True, env.py:288-302) — the
    layout_id array handed to step / rollout_random must then be a uint16 numpy array: it is updated in place."""
    first, count = regen if regen else (0, 0)
    return OracleStartSpec(int(seed), int(env_offset), int(epoch) & 0xFFFFFFFF, int(bool(random_start_pos)),
                           float(rnd_obj_prob_thresh), int(first), int(count))


def regen_layouts(layout_id, spec, mask=None, mask_bits=0xFF):
    """oc_regen_layouts on the host: new ids (in place, uint16 array) for the selected envs."""
    assert layout_id.dtype == np.uint16 and layout_id.flags["C_CONTIGUOUS"]
    m = None if mask is None else np.ascontiguousarray(mask, dtype=np.uint8)
    rc = lib().oracle_regen_layouts(_ptr(layout_id, ctypes.c_uint16), _ptr(m, ctypes.c_uint8), ctypes.c_uint8(mask_bits),
                                    ctypes.c_int64(layout_id.shape[0]), ctypes.byref(spec))
    assert rc == 0
    return layout_id


class Oracle:
    """A table of oracle MDPs sharing one grid shape, operating on wire-format state arrays.

    state arrays: np.uint8 [n_planes, n_envs, 16] (plane-major SoA, include/oc_amd.h).
    """

    def __init__(self, mdps):
        if isinstance(mdps, OracleMdp):
            mdps = [mdps]
        self.n = len(mdps)
        self.arr = (OracleMdp * self.n)(*mdps)
        self.W, self.H = mdps[0].width, mdps[0].height
        assert all(m.width == self.W and m.height == self.H for m in mdps)
        self.n_planes = 1 + (self.W * self.H + 15) // 16

    def _lid(self, layout_id, n_envs):
        if layout_id is None:
            assert self.n == 1
            return None
        lid = np.ascontiguousarray(layout_id, dtype=np.uint16)
        assert lid.shape == (n_envs,) and int(lid.max(initial=0)) < self.n
        return lid

    def new_state(self, n_envs):
        return np.zeros((self.n_planes, n_envs, 16), dtype=np.uint8)

    def reset(self, state, layout_id=None, mask=None, ep_returns=None):
        n = state.shape[1]
        lid = self._lid(layout_id, n)
        m = None if mask is None else np.ascontiguousarray(mask, dtype=np.uint8)
        lib().oracle_reset(self.arr, self.n, _ptr(lid, ctypes.c_uint16), _ptr(state, ctypes.c_uint8),
                           _ptr(m, ctypes.c_uint8), _ptr(ep_returns, ctypes.c_float), ctypes.c_int64(n))
        return state

    def reset_random(self, state, seed=0, env_offset=0, epoch=0, random_start_pos=False, rnd_obj_prob_thresh=0.0,
                     layout_id=None, mask=None):
        """Randomized start states (get_random_start_state_fn, mdp.py:1307-1369) from the stream of oc_reset_random."""
        n = state.shape[1]
        lid = self._lid(layout_id, n)
        rc = lib().oracle_reset_random(self.arr, self.n, _ptr(lid, ctypes.c_uint16), _ptr(state, ctypes.c_uint8),
                                       _ptr(mask, ctypes.c_uint8), ctypes.c_int64(n), ctypes.c_uint64(seed),
                                       ctypes.c_int64(env_offset), ctypes.c_uint32(epoch), int(bool(random_start_pos)),
                                       ctypes.c_double(rnd_obj_prob_thresh))
        assert rc == 0
        return state

    def step(self, state, actions, horizon=400, options=0, layout_id=None, ep_returns=None, start=None):
        """Returns (next_state, rewards[n,4] f32, flags[n] u8). `state` is not modified."""
        n = state.shape[1]
        assert state.shape == (self.n_planes, n, 16) and state.dtype == np.uint8
        actions = np.ascontiguousarray(actions, dtype=np.uint8)
        assert actions.shape == (n, 2)
        lid = self._lid(layout_id, n)
        out = np.empty_like(state)
        rewards = np.zeros((n, 4), dtype=np.float32)
        flags = np.zeros((n,), dtype=np.uint8)
        self.last_events = np.zeros((n,), dtype=np.uint64)  # EVENT_TYPES bit mask, bit 2*k + player
        rc = lib().oracle_step(self.arr, self.n, _ptr(lid, ctypes.c_uint16), _ptr(np.ascontiguousarray(state), ctypes.c_uint8),
                               _ptr(out, ctypes.c_uint8), _ptr(actions, ctypes.c_uint8), _ptr(rewards, ctypes.c_float),
                               _ptr(flags, ctypes.c_uint8), _ptr(ep_returns, ctypes.c_float),
                               _ptr(self.last_events, ctypes.c_uint64), ctypes.c_int64(n),
                               int(horizon), ctypes.c_uint32(options), ctypes.byref(start) if start is not None else None)
        assert rc == 0
        return out, rewards, flags

    def rollout_random(self, state, n_steps, horizon=400, options=0, seed=0, env_offset=0, t0=0, layout_id=None,
                       ep_returns=None, want_outputs=True, start=None):
        """In-place n_steps random-policy steps. Returns (rewards[T,n,4], flags[T,n]) or (None, None)."""
        n = state.shape[1]
        lid = self._lid(layout_id, n)
        rewards = np.zeros((n_steps, n, 4), dtype=np.float32) if want_outputs else None
        flags = np.zeros((n_steps, n), dtype=np.uint8) if want_outputs else None
        rc = lib().oracle_rollout_random(self.arr, self.n, _ptr(lid, ctypes.c_uint16), _ptr(state, ctypes.c_uint8),
                                         _ptr(rewards, ctypes.c_float), _ptr(flags, ctypes.c_uint8),
                                         _ptr(ep_returns, ctypes.c_float), ctypes.c_int64(n), int(horizon),
                                         ctypes.c_uint32(options), ctypes.c_uint64(seed), ctypes.c_int64(env_offset),
                                         ctypes.c_int64(t0), int(n_steps), ctypes.byref(start) if start is not None else None)
        assert rc == 0
        return rewards, flags

    def encode_lossless(self, state, horizon=400, layout_id=None):
        """Returns int32 [n_envs, 2, W, H, 26]."""
        n = state.shape[1]
        lid = self._lid(layout_id, n)
        obs = np.zeros((n, 2, self.W, self.H, 26), dtype=np.int32)
        rc = lib().oracle_encode_lossless(self.arr, self.n, _ptr(lid, ctypes.c_uint16),
                                          _ptr(np.ascontiguousarray(state), ctypes.c_uint8), _ptr(obs, ctypes.c_int32),
                                          ctypes.c_int64(n), int(horizon))
        assert rc == 0, "lossless encoding requires 2 players (mdp.py:2389)"
        return obs


def encode_lossless_u8(orc, state, horizon=400, layout_id=None):
    """The same encoding as uint8 [n_envs, 2, W, H, 26], threaded over oracle_set_threads cores (bench.py's check of
    every observation of a BASELINE configs[2] launch)."""
    n = state.shape[1]
    lid = orc._lid(layout_id, n)
    obs = np.empty((n, 2, orc.W, orc.H, 26), dtype=np.uint8)
    rc = lib().oracle_encode_lossless_u8(orc.arr, orc.n, _ptr(lid, ctypes.c_uint16),
                                         _ptr(np.ascontiguousarray(state), ctypes.c_uint8), _ptr(obs, ctypes.c_uint8),
                                         ctypes.c_int64(n), int(horizon))
    assert rc == 0, "lossless encoding: 2 players required (mdp.py:2389) / a layer value above 255"
    return obs


def featurize(orc, state, counter_goals="none", num_pots=2, layout_id=None):
    """featurize_state of every env: float32 [n_envs, 2, 2*(num_pots*10+26)+4]. counter_goals: "none" (the reference's
    NO_COUNTERS_PARAMS) or "all" (every counter is a motion goal)."""
    n = state.shape[1]
    lid = orc._lid(layout_id, n)
    mask = None
    if counter_goals == "all":
        mask = np.zeros((orc.n, MAX_CELLS), dtype=np.int32)
        for l in range(orc.n):
            t = orc.arr[l].terrain
            for c in range(orc.arr[l].width * orc.arr[l].height):
                mask[l, c] = 1 if t[c:c + 1] == b"X" else 0
    out = np.zeros((n, 2, 2 * (num_pots * 10 + 26) + 4), dtype=np.float32)
    rc = lib().oracle_featurize(orc.arr, orc.n, _ptr(lid, ctypes.c_uint16), _ptr(mask, ctypes.c_int32),
                                _ptr(np.ascontiguousarray(state), ctypes.c_uint8), _ptr(out, ctypes.c_float),
                                ctypes.c_int64(n), int(num_pots))
    assert rc == 0
    return out


class PotentialParams(ctypes.Structure):
    _fields_ = [("gamma", ctypes.c_double), ("tomato_value", ctypes.c_double), ("onion_value", ctypes.c_double),
                ("max_delivery_steps", ctypes.c_int32), ("max_pickup_steps", ctypes.c_int32),
                ("pot_onion_steps", ctypes.c_int32), ("pot_tomato_steps", ctypes.c_int32)]


def potential(orc, state, params, layout_id=None):
    """potential_function of every env: float64 [n_envs].  params: one dict per mdp with the keys of the reference's
    `potential_params` (mdp.py:2972-2982): gamma, tomato_value, onion_value, max_delivery_steps, max_pickup_steps,
    pot_onion_steps, pot_tomato_steps."""
    n = state.shape[1]
    lid = orc._lid(layout_id, n)
    arr = (PotentialParams * orc.n)()
    for l, p in enumerate(params):
        for k, _ in PotentialParams._fields_:
            setattr(arr[l], k, p[k])
    out = np.zeros((n,), dtype=np.float64)
    rc = lib().oracle_potential(orc.arr, orc.n, _ptr(lid, ctypes.c_uint16), arr,
                                _ptr(np.ascontiguousarray(state), ctypes.c_uint8), _ptr(out, ctypes.c_double), ctypes.c_int64(n))
    assert rc == 0
    return out


def py_set_order(width, cells):
    """Iteration order of `set().union(cells)` for (x, y) = (c % width, c // width), per the oracle's restatement."""
    a = np.asarray(cells, dtype=np.int32)
    out = np.zeros_like(a)
    lib().oracle_py_set_order(int(width), _ptr(a, ctypes.c_int32), len(a), _ptr(out, ctypes.c_int32))
    return [int(c) for c in out]


def set_threads(n):
    """Host threads for oracle_rollout_random's env loop (OpenMP build only); returns the number in effect."""
    return int(lib().oracle_set_threads(int(n)))


def random_actions(seed, env_offset, t, n_envs):
    a = np.zeros((n_envs, 2), dtype=np.uint8)
    lib().oracle_random_actions(ctypes.c_uint64(seed), ctypes.c_int64(env_offset), ctypes.c_int64(t),
                                ctypes.c_int64(n_envs), _ptr(a, ctypes.c_uint8))
    return a


def philox4x32_10(ctr, key):
    c = (ctypes.c_uint32 * 4)(*ctr)
    k = (ctypes.c_uint32 * 2)(*key)
    o = (ctypes.c_uint32 * 4)()
    lib().oracle_philox4x32_10(c, k, o)
    return tuple(int(x) for x in o)
