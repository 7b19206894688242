"""OvercookedGridworld — drop-in mirror of the reference's MDP class for the accelerated path.

Same constructor conventions and method names as `overcooked_ai_py.mdp.overcooked_mdp.OvercookedGridworld`
(mdp.py:1076-1430, 2382-2561) for the hot path: `from_layout_name`, `from_grid`, `get_standard_start_state`,
`get_state_transition`, `lossless_state_encoding`, the terrain look-ups.  Every transition / encoding is computed
by the HIP kernels (a batch of one env for the single-state calls; `get_state_transitions` /
`lossless_state_encodings` take whole batches).  For throughput use `VecOvercookedEnv` directly: the single-state
calls pay a host<->device round trip each.
"""
import copy
import math
import os

import numpy as np

from .actions import Action, Direction
from .layouts import BASE_REW_SHAPING_PARAMS, LayoutSpec, read_layout_dict
from .state import (ObjectState, OvercookedState, PlayerState, SoupState, configure_recipes, pack_states,  # noqa: F401
                    unpack_states)

EVENT_TYPES = [  # mdp.py:1027-1058
    "tomato_pickup", "useful_tomato_pickup", "tomato_drop", "useful_tomato_drop", "potting_tomato",
    "onion_pickup", "useful_onion_pickup", "onion_drop", "useful_onion_drop", "potting_onion",
    "dish_pickup", "useful_dish_pickup", "dish_drop", "useful_dish_drop",
    "soup_pickup", "soup_delivery", "soup_drop",
    "optimal_onion_potting", "optimal_tomato_potting", "viable_onion_potting", "viable_tomato_potting",
    "catastrophic_onion_potting", "catastrophic_tomato_potting", "useless_onion_potting", "useless_tomato_potting",
]


_ACTION_INDEX = {a: i for i, a in enumerate(Action.INDEX_TO_ACTION)}


def events_from_mask(mask, num_players=2):
    """u64 event mask (bit 2*k + p) -> the reference's event_infos dict {event: [bool] * num_players}."""
    mask = int(mask)
    if mask == 0:  # most steps: nothing happened
        if num_players == 2:
            return {name: [False, False] for name in EVENT_TYPES}
        return {name: [False] * num_players for name in EVENT_TYPES}
    return {name: [bool((mask >> (2 * k + p)) & 1) for p in range(num_players)] for k, name in enumerate(EVENT_TYPES)}


def default_device():
    return os.environ.get("OC_AMD_DEVICE", "cuda:0")


class _Infos(dict):
    """The infos dict of get_state_transition (same keys as the reference's) that also remembers the kernel's event bit
    mask, so that OvercookedEnv._update_game_stats need not walk 25 x 2 flags.  `event_infos` — 25 lists of booleans — is
    built from the mask the first time somebody asks for it (OvercookedEnv.step never does)."""
    __slots__ = ("event_mask", "num_players")

    def __missing__(self, key):
        if key != "event_infos":
            raise KeyError(key)
        v = self["event_infos"] = events_from_mask(self.event_mask, self.num_players)
        return v

    def _all(self):
        self["event_infos"]
        return self

    def keys(self):
        return dict.keys(self._all())

    def items(self):
        return dict.items(self._all())

    def values(self):
        return dict.values(self._all())

    def __iter__(self):
        return dict.__iter__(self._all())

    def __len__(self):
        return dict.__len__(self._all())

    def __contains__(self, key):
        return key == "event_infos" or dict.__contains__(self, key)

    def get(self, key, default=None):
        return self[key] if key in self else default

    def __eq__(self, other):
        return dict.__eq__(self._all(), other._all() if isinstance(other, _Infos) else other)

    __hash__ = None


class _SingleEnvPort:
    """One env stepped by the GPU with no Python objects and no kernel launch in between: the single-state calls of the
    drop-in API (`get_state_transition`, hence `OvercookedEnv.step`) write the packed state and the joint action into the
    MAILBOX of a resident kernel (oc_mailbox_*, include/oc_amd.h: 4 KiB of pinned, GPU-mapped host memory), post the request
    and spin until the kernel has written next state, rewards, flags and the event mask back — two PCIe round trips per
    step instead of a launch and a stream wait.  Grids above 64 cells (which the mailbox kernel does not serve) take the
    round-3 form: `oc_step` launched on pointers into a pinned buffer, one stream wait per call."""

    MAILBOX_AFTER = 2  # single-state calls served by plain launches before the resident kernel is brought up

    def __init__(self, mdp):
        import ctypes

        import torch

        from .state import SingleStateCodec

        self.env = mdp._env(1)  # owns the device copy of the layout table and the OcBatch
        env = self.env
        self.lib, self.bref = env.lib, env._bref
        self.n_state = env.n_planes * 16
        self.codec = SingleStateCodec(mdp.spec, env.n_planes)
        self.device, self.dev_index, self.torch, self.num_players = env.device, env._dev_index, torch, mdp.num_players
        self.mailbox, self._finalizer, self.calls = None, None, 0
        # The mailbox (pinned mapped memory, a HIP stream, a resident polling kernel) is opened LAZILY, by the third
        # single-state call: an mdp that is constructed and asked for one or two transitions — a layout generator's
        # regen_mdp, a test fixture — never holds one.  OC_AMD_NO_MAILBOX keeps every call on the launch path.
        self.mailbox_ok = env.n_planes <= 5 and not os.environ.get("OC_AMD_NO_MAILBOX")
        off = lambda x: (x + 15) & ~15
        o_act = off(self.n_state)
        o_out = o_act + 16
        o_rew = o_out + off(self.n_state)
        self.pinned = torch.zeros((o_rew + 48,), dtype=torch.uint8).pin_memory()
        base = self.pinned.data_ptr()
        self.ptrs = tuple(ctypes.c_void_p(base + o) for o in (0, o_out, o_act, o_rew, o_rew + 16, o_rew + 32))
        self.stream = torch.cuda.Stream(device=env.device)
        self.stream_ptr = ctypes.c_void_p(self.stream.cuda_stream)
        self._bind(self.pinned.numpy(), 0, o_act, o_out, o_rew, o_rew + 16, o_rew + 32)

    def _bind(self, arr, o_in, o_act, o_out, o_rew, o_flag, o_ev):
        """Point the pack / unpack views at a request / response buffer (the pinned launch buffer, or the mailbox)."""
        self.np = arr
        self.o_in, self.o_act, self.o_out, self.o_rew, self.o_flag, self.o_ev = o_in, o_act, o_out, o_rew, o_flag, o_ev
        self.mv = memoryview(self.np)
        self.mv_in = self.mv[self.o_in:self.o_in + self.n_state]
        self.mv_out = self.mv[self.o_out:self.o_out + self.n_state]
        self.rew = self.np[self.o_rew:self.o_rew + 16].view(np.float32)
        self.ev = self.np[self.o_ev:self.o_ev + 8].view(np.uint64)
        import struct
        # (sparse0, sparse1, shaped0, shaped1, event mask) in one call: four floats at o_rew, the u64 mask at o_ev
        self._outputs = struct.Struct("<4f%dxQ" % (self.o_ev - self.o_rew - 16)).unpack_from

    def _open_mailbox(self):
        import ctypes
        import weakref

        from . import _lib

        mb = ctypes.c_void_p()
        with self.torch.cuda.device(self.device):
            _lib.check(self.lib.oc_mailbox_open(self.bref, 65535, ctypes.byref(mb)), "oc_mailbox_open")
        buf = (ctypes.c_uint8 * _lib.MB_BYTES).from_address(self.lib.oc_mailbox_buffer(mb))
        request = bytes(self.mv[self.o_in:self.o_in + self.n_state]), self.mv[self.o_act], self.mv[self.o_act + 1]
        self._bind(np.frombuffer(buf, dtype=np.uint8), _lib.MB_STATE_IN, _lib.MB_ACTIONS, _lib.MB_STATE_OUT, _lib.MB_REWARDS,
                   _lib.MB_FLAGS, _lib.MB_EVENTS)
        self.mv_in[:] = request[0]  # (the call that opens the mailbox has already packed its request)
        self.mv[self.o_act], self.mv[self.o_act + 1] = request[1], request[2]
        self._step = self.lib.oc_mailbox_step
        self.mailbox = mb
        # (the finalizer also runs at interpreter exit: the kernel is told to leave)
        self._finalizer = weakref.finalize(self, self.lib.oc_mailbox_close, mb)

    def close(self):
        """Give the mailbox back now (the resident kernel leaves, the stream and the pinned pages are freed) instead of at
        garbage collection; the port keeps working on the launch path and re-opens a mailbox after MAILBOX_AFTER further calls."""
        if self.mailbox is not None:
            self._finalizer.detach()
            self.lib.oc_mailbox_close(self.mailbox)
            self.mailbox, self._finalizer, self.calls = None, None, 0
            o = [p.value - self.pinned.data_ptr() for p in self.ptrs]
            self._bind(self.pinned.numpy(), o[0], o[2], o[1], o[3], o[4], o[5])

    def transition(self, state, a0, a1):
        """(next_state, (sparse0, sparse1, shaped0, shaped1, event mask)) or None when the state needs the general path.
        The state comes back as a state._LazyState: its players / objects are built when somebody looks at them."""
        if not self.codec.pack(state, self.mv_in):
            return None
        mv = self.mv
        mv[self.o_act] = a0
        mv[self.o_act + 1] = a1
        if self.mailbox is None and self.mailbox_ok:
            self.calls += 1
            if self.calls > self.MAILBOX_AFTER:
                try:
                    self._open_mailbox()
                except RuntimeError as exc:  # (pinned / mapped memory or streams exhausted with many mdps alive)
                    import warnings

                    self.mailbox_ok = False  # this port stays on the launch path instead of retrying at every call
                    warnings.warn("oc_mailbox_open failed (%s): single-state calls keep launching oc_step" % exc)
        if self.mailbox is not None:
            rc = self._step(self.mailbox)
            if rc:
                from . import _lib
                _lib.check(rc, "oc_mailbox_step")
            return self.codec.unpack_lazy(self.mv_out), self._outputs(self.mv, self.o_rew)
        p = self.ptrs
        if self.torch.cuda.current_device() == self.dev_index:
            rc = self.lib.oc_step(self.bref, p[0], p[1], p[2], p[3], p[4], None, p[5], 65535, 0, None, None, self.stream_ptr)
        else:  # the launch must happen with this env's device current (its stream and buffers live there)
            with self.torch.cuda.device(self.device):
                rc = self.lib.oc_step(self.bref, p[0], p[1], p[2], p[3], p[4], None, p[5], 65535, 0, None, None, self.stream_ptr)
        if rc:
            from . import _lib
            _lib.check(rc, "oc_step")
        self.stream.synchronize()
        return self.codec.unpack_lazy(self.mv_out), self._outputs(self.mv, self.o_rew)


class OvercookedGridworld:
    def __init__(self, terrain, start_player_positions, start_bonus_orders=[], rew_shaping_params=None,
                 layout_name="unnamed_layout", start_all_orders=[], num_items_for_soup=3, order_bonus=2,
                 start_state=None, old_dynamics=False, device=None, **kwargs):
        rows = [list(r) for r in terrain]
        for i, (x, y) in enumerate(start_player_positions):
            rows[y][x] = str(i + 1)
        cfg = dict(kwargs)
        cfg.update(grid="\n".join("".join(r) for r in rows), layout_name=layout_name,
                   start_all_orders=start_all_orders, start_bonus_orders=start_bonus_orders,
                   rew_shaping_params=rew_shaping_params, order_bonus=order_bonus, old_dynamics=old_dynamics,
                   num_items_for_soup=num_items_for_soup)
        self._init_from_spec(LayoutSpec(cfg), start_state, device)

    def _init_from_spec(self, spec, start_state=None, device=None):
        self.spec = spec
        configure_recipes(spec)  # (Recipe.configure of the reference's constructor, mdp.py:1014: hand-built SoupStates resolve here)
        self.terrain_mtx = spec.terrain_mtx
        self.height, self.width = spec.height, spec.width
        self.shape = (self.width, self.height)
        self.start_player_positions = [tuple(p) for p in spec.start_player_positions]
        self.num_players = spec.num_players
        self.start_bonus_orders = spec.start_bonus_orders
        self.start_all_orders = spec.start_all_orders or [
            {"ingredients": ["onion"] * a + ["tomato"] * (n - a)} for n in (1, 2, 3) for a in range(n, -1, -1)]
        self.reward_shaping_params = dict(BASE_REW_SHAPING_PARAMS)
        self.reward_shaping_params.update(spec.rew_shaping_params)
        self.layout_name = spec.layout_name
        self.order_bonus = spec.order_bonus
        self.old_dynamics = spec.old_dynamics
        self.recipe_config = dict(spec.recipe_config, num_items_for_soup=spec.num_items_for_soup,
                                  all_orders=spec.start_all_orders)
        if isinstance(start_state, dict):
            start_state = OvercookedState.from_dict(start_state)
        self.start_state = start_state
        self.terrain_pos_dict = {c: spec.cells_of(c) for c in " XOTPDS"}
        self._device = device
        self._envs = {}
        self._single = None

    # ---------------------------------------------------------------- construction (mdp.py:1150-1222)
    @staticmethod
    def from_layout_name(layout_name, device=None, **params_to_overwrite):
        d = read_layout_dict(layout_name)
        d["layout_name"] = layout_name
        start_state = d.pop("start_state", None)
        start_state = params_to_overwrite.pop("start_state", start_state)
        mdp = OvercookedGridworld.__new__(OvercookedGridworld)
        mdp._init_from_spec(LayoutSpec(d, **params_to_overwrite), start_state, device)
        return mdp

    @staticmethod
    def from_grid(layout_grid, base_layout_params={}, params_to_overwrite={}, debug=False, device=None):
        cfg = copy.deepcopy(dict(base_layout_params))
        cfg.update(params_to_overwrite)
        cfg["grid"] = "\n".join("".join(row) for row in layout_grid)
        start_state = cfg.pop("start_state", None)
        mdp = OvercookedGridworld.__new__(OvercookedGridworld)
        mdp._init_from_spec(LayoutSpec(cfg), start_state, device)
        return mdp

    @staticmethod
    def from_spec(spec, device=None):
        mdp = OvercookedGridworld.__new__(OvercookedGridworld)
        mdp._init_from_spec(spec, spec.start_state, device)
        return mdp

    def __eq__(self, other):
        return (isinstance(other, OvercookedGridworld) and self.terrain_mtx == other.terrain_mtx
                and self.start_player_positions == other.start_player_positions
                and self.start_bonus_orders == other.start_bonus_orders
                and self.start_all_orders == other.start_all_orders
                and self.reward_shaping_params == other.reward_shaping_params and self.layout_name == other.layout_name)

    @property
    def mdp_params(self):
        return {"layout_name": self.layout_name, "terrain": self.terrain_mtx,
                "start_player_positions": self.start_player_positions, "start_bonus_orders": self.start_bonus_orders,
                "rew_shaping_params": copy.deepcopy(self.reward_shaping_params),
                "start_all_orders": self.start_all_orders}

    # ---------------------------------------------------------------- device plumbing
    def vec_env(self, n_envs, horizon=65535, **kw):
        """A fresh batched env over this MDP (the fast path)."""
        from .vec_env import VecOvercookedEnv

        return VecOvercookedEnv(self.spec, n_envs, horizon=horizon, device=self._device or default_device(), **kw)

    def _env(self, n):
        if n not in self._envs:
            if len(self._envs) > 8:
                self._envs.clear()
            self._envs[n] = self.vec_env(n, track_returns=False)
        return self._envs[n]

    # ---------------------------------------------------------------- game logic
    def get_actions(self, state):
        self._check_valid_state(state)
        return [Action.ALL_ACTIONS for _ in state.players]

    def _check_valid_state(self, state):
        """AssertionError for states the reference's _check_valid_state rejects (mdp.py:1910-1949)."""
        try:
            pack_states(self.spec, [state])
        except (ValueError, KeyError) as e:
            raise AssertionError(str(e))

    def get_standard_start_state(self):
        if self.start_state:
            return self.start_state
        return OvercookedState.from_player_positions(self.start_player_positions, bonus_orders=self.start_bonus_orders,
                                                     all_orders=self.start_all_orders)

    def is_terminal(self, state):
        return False

    def get_valid_joint_player_positions(self):
        """All joint positions on free cells without overlap, in the reference's order (mdp.py:1736-1747)."""
        import itertools

        valid = self.get_valid_player_positions()
        return [jp for jp in itertools.product(valid, repeat=self.num_players) if len(set(jp)) == len(jp)]

    def get_random_start_state_fn(self, random_start_pos=False, rnd_obj_prob_thresh=0.0):
        """Same start-state distribution AND the same numpy draw sequence as mdp.py:1307-1369, so that with an
        identical np.random seed this returns exactly the state the reference would."""
        def start_state_fn():
            if random_start_pos:
                valid_positions = self.get_valid_joint_player_positions()
                start_pos = valid_positions[np.random.choice(len(valid_positions))]
            else:
                start_pos = self.start_player_positions
            start_state = OvercookedState.from_player_positions(start_pos, bonus_orders=self.start_bonus_orders,
                                                                all_orders=self.start_all_orders)
            if rnd_obj_prob_thresh == 0:
                return start_state
            for pot_loc in self.get_pot_locations():
                p = np.random.rand()
                if p < rnd_obj_prob_thresh:
                    n = int(np.random.randint(low=1, high=4))
                    m = int(np.random.randint(low=0, high=4 - n))
                    q = np.random.rand()
                    cooking_tick = 0 if q < rnd_obj_prob_thresh else -1
                    ings = ["onion"] * n + ["tomato"] * m
                    start_state.objects[pot_loc] = SoupState(pot_loc, ings, cooking_tick,
                                                             self.spec.recipe_time((n, m)))
            for player in start_state.players:
                p = np.random.rand()
                if p < rnd_obj_prob_thresh:
                    obj = np.random.choice(["dish", "onion", "soup"], p=[0.2, 0.6, 0.2])
                    n = int(np.random.randint(low=1, high=4))
                    m = int(np.random.randint(low=0, high=4 - n))
                    if obj == "soup":
                        ct = self.spec.recipe_time((n, m))
                        player.set_object(SoupState(player.position, ["onion"] * n + ["tomato"] * m, ct, ct))
                    else:
                        player.set_object(ObjectState(str(obj), player.position))
            return start_state

        return start_state_fn

    def get_state_transitions(self, states, joint_actions):
        """Batched get_state_transition: lists of states / joint actions -> (next_states, infos list)."""
        import torch

        n = len(states)
        idx = np.zeros((n, 2), np.uint8)
        for e, ja in enumerate(joint_actions):
            if len(ja) != self.num_players:
                raise ValueError("Illegal joint action %r" % (ja,))
            for p, a in enumerate(ja):
                try:
                    idx[e, p] = Action.to_index(a)
                except ValueError:
                    raise ValueError("Illegal action %s in state %s" % (a, states[e]))  # mdp.py:1394-1398
            if self.num_players == 1:
                idx[e, 1] = 4
        try:
            packed = pack_states(self.spec, states)
        except (ValueError, KeyError) as e:
            raise AssertionError(str(e))
        env = self._env(n)
        env.horizon = 65535
        env.set_packed_state(packed)
        ev = torch.zeros((n,), dtype=torch.int64, device=env.device)
        rew, flags = env.step(torch.from_numpy(idx).to(env.device), events_out=ev)
        rew = rew.cpu().numpy()
        masks = ev.cpu().numpy().view(np.uint64)
        nxt = unpack_states(self.spec, env.get_packed_state())
        infos = []
        for e in range(n):
            sp = [_num(v) for v in rew[e, 0:self.num_players]]
            sh = [_num(v) for v in rew[e, 2:2 + self.num_players]]
            infos.append({"event_infos": events_from_mask(masks[e], self.num_players),
                          "sparse_reward_by_agent": sp, "shaped_reward_by_agent": sh})
        return nxt, infos

    def _port(self):
        if self._single is None:
            self._single = _SingleEnvPort(self)
        return self._single

    def close(self):
        """Release the single-state port's mailbox (resident polling kernel, HIP stream, pinned pages) now rather than at
        garbage collection — for code that builds many mdps (layout generators, regen_mdp) or interleaves env.step with
        device-wide synchronisations.  The mdp stays usable: the next single-state calls go through plain launches and a
        mailbox is opened again after a few of them.  Also the exit of `with OvercookedGridworld...(...) as mdp:`."""
        if self._single is not None:
            self._single.close()

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()
        return False

    def _fast_step(self, state, joint_action):
        """One env through the single-state port: (next_state, sparse rewards, shaped rewards, event mask), or None -> the
        general batched path (which also raises the reference's errors for illegal actions / invalid states)."""
        if len(joint_action) != self.num_players:
            return None
        try:
            a0 = _ACTION_INDEX[joint_action[0]]
            a1 = _ACTION_INDEX[joint_action[1]] if self.num_players == 2 else 4
        except (KeyError, TypeError):
            return None
        out = self._port().transition(state, a0, a1)
        if out is None:
            return None
        nxt, (r0, r1, r2, r3, mask) = out
        n = self.num_players
        # (rewards are Python floats; all-zero on most steps; the reference returns ints for integer-valued configs)
        sparse = [0] * n if not (r0 or r1) else [_num(v) for v in (r0, r1)[:n]]
        shaped = [0] * n if not (r2 or r3) else [_num(v) for v in (r2, r3)[:n]]
        return nxt, sparse, shaped, mask

    def _fast_transition(self, state, joint_action):
        """get_state_transition through the port: (next_state, infos) or None."""
        out = self._fast_step(state, joint_action)
        if out is None:
            return None
        nxt, sparse, shaped, mask = out
        infos = _Infos(sparse_reward_by_agent=sparse, shaped_reward_by_agent=shaped)
        infos.event_mask, infos.num_players = mask, self.num_players
        return nxt, infos

    def get_state_transition(self, state, joint_action, display_phi=False, motion_planner=None):
        """(new_state, infos) exactly like mdp.py:1375-1430; the input state is not modified."""
        fast = self._fast_transition(state, joint_action)
        if fast is not None and not display_phi:
            return fast
        nxt, infos = self.get_state_transitions([state], [joint_action])
        if display_phi:  # mdp.py:1421-1429; the motion planner's distances are built in (planner.py)
            phi = self.potential_functions([state, nxt[0]])
            infos[0]["phi_s"], infos[0]["phi_s_prime"] = float(phi[0]), float(phi[1])
        return nxt[0], infos[0]

    # ---------------------------------------------------------------- potential (mdp.py:2920-3238)
    def potential_functions(self, states, gamma=0.99):
        """Batched potential_function: float64 [n]."""
        env = self._env(len(states))
        env.set_packed_state(pack_states(self.spec, states))
        return env.potential(gamma).cpu().numpy()

    def potential_function(self, state, mp=None, gamma=0.99):
        """phi(state) for potential-based reward shaping.  `mp` (the reference's MotionPlanner argument) is accepted
        for signature compatibility; the distances it would supply are precomputed per layout."""
        return float(self.potential_functions([state], gamma)[0])

    # ---------------------------------------------------------------- encodings (mdp.py:2382-2561)
    def get_lossless_state_encoding_shape(self):
        return np.array(list(self.shape) + [26])

    @property
    def lossless_state_encoding_shape(self):
        return self.get_lossless_state_encoding_shape()

    def lossless_state_encodings(self, states, horizon=400):
        import torch

        assert self.num_players == 2, "Functionality has to be added to support encondings for > 2 players"
        env = self._env(len(states))
        env.set_packed_state(pack_states(self.spec, states))
        # the urgency layer is `horizon - timestep < 40` (mdp.py:2446-2447) on the caller's horizon — NOT clamped to the
        # packed timestep's 16 bits: with the reference's default MAX_HORIZON the layer never turns on (int32 range is
        # all oc_encode_lossless needs: timesteps stop at 65 535)
        saved, env.horizon = env.horizon, int(min(max(horizon, 1), 2 ** 31 - 1))
        try:
            return env.encode_lossless(torch.uint8).cpu().numpy().astype(np.int64)
        finally:
            env.horizon = saved

    def lossless_state_encoding(self, overcooked_state, horizon=400, debug=False):
        assert type(debug) is bool
        enc = self.lossless_state_encodings([overcooked_state], horizon)
        return tuple(enc[0, i] for i in range(2))

    def get_featurize_state_shape(self, num_pots=2):
        return (self.num_players * (num_pots * 10 + 28),)

    def featurize_states(self, states, num_pots=2, counter_goals="none"):
        """Batched featurize_state: float32 [n, 2, 2*(num_pots*10+26)+4]."""
        assert self.num_players == 2
        env = self._env(len(states))
        env.set_packed_state(pack_states(self.spec, states))
        return env.featurize(num_pots=num_pots, counter_goals=counter_goals).cpu().numpy()

    def featurize_state(self, overcooked_state, mlam=None, num_pots=2, **kwargs):
        """Hand-crafted features of both players (mdp.py:2579-2898) -> [features_p0, features_p1].  `mlam` is only
        consulted for its motion planner's counter_goals (the reference's default NO_COUNTERS_PARAMS has none)."""
        counter_goals = kwargs.get("counter_goals", "none")
        mp = getattr(mlam, "motion_planner", None)
        if mp is not None and getattr(mp, "counter_goals", None):
            counter_goals = list(mp.counter_goals)
        f = self.featurize_states([overcooked_state], num_pots, counter_goals)[0].astype(np.float64)
        return [f[0], f[1]]

    # ---------------------------------------------------------------- layout info (mdp.py:1733-1807)
    def get_valid_player_positions(self):
        return self.terrain_pos_dict[" "]

    def get_terrain_type_at_pos(self, pos):
        x, y = pos
        return self.terrain_mtx[y][x]

    def get_dish_dispenser_locations(self):
        return list(self.terrain_pos_dict["D"])

    def get_onion_dispenser_locations(self):
        return list(self.terrain_pos_dict["O"])

    def get_tomato_dispenser_locations(self):
        return list(self.terrain_pos_dict["T"])

    def get_serving_locations(self):
        return list(self.terrain_pos_dict["S"])

    def get_pot_locations(self):
        return list(self.terrain_pos_dict["P"])

    def get_counter_locations(self):
        return list(self.terrain_pos_dict["X"])

    @property
    def num_pots(self):
        return len(self.get_pot_locations())

    def get_valid_player_positions_and_orientations(self):
        return [(pos, d) for pos in self.get_valid_player_positions() for d in Direction.ALL_DIRECTIONS]

    def get_adjacent_features(self, player):
        """[(cell next to the player, its terrain character)] for the four directions (mdp.py:1773-1781)."""
        x, y = player.position
        return [((x + dx, y + dy), self.get_terrain_type_at_pos((x + dx, y + dy))) for dx, dy in Direction.ALL_DIRECTIONS]

    # ---------------------------------------------------------------- state queries agents use (mdp.py:1809-1907)
    # Host-side views of one OvercookedState: what the reference's planners and scripted agents (agents/agent.py) ask the mdp.
    def get_pot_states(self, state):
        """{"empty": [...], "1_items" / "2_items" / "3_items": idle pots by fill, "cooking": [...], "ready": [...]} — pot
        positions in get_pot_locations() order; a missing key reads as [] (mdp.py:1809-1838)."""
        from collections import defaultdict

        out = defaultdict(list)
        for pos in self.get_pot_locations():
            soup = state.objects.get(pos)
            if soup is None:
                key = "empty"
            else:
                assert soup.name == "soup", "soup at %s is not a soup but a %s" % (pos, soup.name)
                key = "ready" if soup.is_ready else "cooking" if soup.is_cooking else "%d_items" % len(soup.ingredients)
            out[key].append(pos)
        return out

    def get_counter_objects_dict(self, state, counter_subset=None):
        """{object name: [positions]} of what lies on counters (all of them, or `counter_subset`), in the order of the
        state's objects dict (mdp.py:1840-1852)."""
        from collections import defaultdict

        where = self.terrain_pos_dict["X"] if counter_subset is None else counter_subset
        out = defaultdict(list)
        for obj in state.objects.values():
            if obj.position in where:
                out[obj.name].append(obj.position)
        return out

    def get_empty_counter_locations(self, state):
        return [pos for pos in self.get_counter_locations() if not state.has_object(pos)]

    def get_empty_pots(self, pot_states):
        return pot_states["empty"]

    def get_ready_pots(self, pot_states):
        return pot_states["ready"]

    def get_cooking_pots(self, pot_states):
        return pot_states["cooking"]

    def get_full_but_not_cooking_pots(self, pot_states):
        return pot_states["%d_items" % self.spec.num_items_for_soup]

    def get_full_pots(self, pot_states):
        return self.get_cooking_pots(pot_states) + self.get_ready_pots(pot_states) + self.get_full_but_not_cooking_pots(pot_states)

    def get_partially_full_pots(self, pot_states):
        # (a set union in the reference, mdp.py:1882-1890: the order of its result is the interpreter's set order)
        return list(set().union(*[pot_states["%d_items" % i] for i in range(1, self.spec.num_items_for_soup)]))

    def get_non_empty_pots(self, pot_states):
        return self.get_full_pots(pot_states) + self.get_partially_full_pots(pot_states)

    def soup_ready_at_location(self, state, pos):
        if not state.has_object(pos):
            return False
        obj = state.get_object(pos)
        assert obj.name == "soup", "Object in pot was not soup"
        return obj.is_ready

    def soup_to_be_cooked_at_location(self, state, pos):
        if not state.has_object(pos):
            return False
        obj = state.get_object(pos)
        return obj.name == "soup" and not obj.is_cooking and not obj.is_ready and len(obj.ingredients) > 0

    def state_string(self, state):
        """Text picture of a state, one row of the grid per line: the terrain character of every cell; a player as its
        orientation arrow + player index (+ a letter for what it holds: o / t / d, or the soup's ingredients); a loose
        object by its letter; a pot as its content, e.g. "ooo" idle, "oo5" cooking at tick 5, "ooo+" ready.  (Same
        information as the reference's state_string, mdp.py:3253-3312; the exact glyphs are this package's.)"""
        arrows = {Direction.NORTH: "^", Direction.SOUTH: "v", Direction.EAST: ">", Direction.WEST: "<"}

        def obj_txt(obj):
            if obj.name != "soup":
                return obj.name[0]
            txt = "".join(i[0] for i in obj.ingredients)
            return txt + ("+" if obj.is_ready else str(obj.cooking_tick) if obj.is_cooking else "")

        players = {p.position: (i, p) for i, p in enumerate(state.players)}
        lines = []
        for y, row in enumerate(self.terrain_mtx):
            cells = []
            for x, ch in enumerate(row):
                if (x, y) in players:
                    i, p = players[(x, y)]
                    txt = arrows[tuple(p.orientation)] + str(i) + (obj_txt(p.held_object) if p.has_object() else "")
                elif state.has_object((x, y)):
                    txt = ch + "{" + obj_txt(state.get_object((x, y))) + "}"
                else:
                    txt = ch
                cells.append(txt.ljust(6))
            lines.append("".join(cells).rstrip())
        orders = ", ".join("+".join(o["ingredients"]) if isinstance(o, dict) else str(o) for o in (state.bonus_orders or []))
        return "\n".join(lines) + ("\nbonus orders: " + orders if orders else "") + "\n"

    # ---------------------------------------------------------------- copying / pickling
    # The device plumbing (cached batched envs, the pinned-buffer port with its stream and ctypes pointers) is per process
    # and rebuilt on first use: copies and pickles carry the layout only.
    def __getstate__(self):
        d = dict(self.__dict__)
        d["_envs"], d["_single"] = {}, None
        return d

    def __deepcopy__(self, memo):
        new = OvercookedGridworld.__new__(OvercookedGridworld)
        memo[id(self)] = new
        for k, v in self.__dict__.items():
            new.__dict__[k] = {} if k == "_envs" else None if k == "_single" else copy.deepcopy(v, memo)
        return new


def _num(v):
    """float32 reward -> int when integral (the reference returns Python ints for integer-valued configs)."""
    v = float(v)
    return int(v) if math.isfinite(v) and abs(v) < 2 ** 31 and v == int(v) else v  # inf: tutorial_3's order_bonus
