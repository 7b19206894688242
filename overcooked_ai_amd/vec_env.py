"""VecOvercookedEnv — N independent Overcooked envs resident in HBM, stepped by the HIP kernels.

This is the batched form of the reference's `OvercookedEnv` (overcooked_env.py:33-325): `reset` ~ env.py:288,
`step` ~ env.py:244 (get_state_transition mdp.py:1375 + done/`game_stats` bookkeeping), `encode_lossless` ~
`lossless_state_encoding_mdp` env.py:276.  PyTorch is used only for device memory and streams; all game logic
is in liboc_amd.so, called through its C-ABI (include/oc_amd.h).  There is no CPU fallback.
"""
import ctypes

import numpy as np
import torch

from . import _lib
from .layouts import LayoutSpec, LayoutTable, spec_from_name
from .state import pack_states, unpack_states

_raw_stream = getattr(torch._C, "_cuda_getCurrentRawStream", None)  # (device index) -> hipStream_t as an int
_get_device = getattr(torch._C, "_cuda_getDevice", None) or torch.cuda.current_device


def as_layout_table(layouts, pad_to=None):
    if isinstance(layouts, LayoutTable):
        return layouts
    if isinstance(layouts, (str, LayoutSpec)):
        layouts = [layouts]
    specs = [spec_from_name(l) if isinstance(l, str) else l for l in layouts]
    return LayoutTable(specs, pad_to=pad_to)


class VecOvercookedEnv:
    def __init__(self, layouts, n_envs, horizon=400, device="cuda", layout_id=None, auto_reset=False, seed=0,
                 env_offset=0, pad_to=None, track_returns=True, random_start_pos=False, rnd_obj_prob_thresh=0.0,
                 track_events=False, regen_layout=False):
        self.lib = _lib.load()
        self.table = as_layout_table(layouts, pad_to)
        self.n_envs = int(n_envs)
        self.horizon = int(horizon)
        if not 1 <= self.horizon <= 65535:
            raise ValueError("horizon must be in 1..65535 (timestep is a u16 in the packed state)")
        self.device = torch.device(device)
        if self.device.type != "cuda":
            raise _lib.OcAmdError("VecOvercookedEnv needs a ROCm GPU device (got %r); there is no CPU fallback" % device)
        self.auto_reset = bool(auto_reset)
        self.lane_pair = False     # rollout_random: force the lane-pair kernel where the table allows it
        self.one_wavefront = False # rollout_random: keep every env-step in one wavefront (OC_OPT_ONE_WAVEFRONT: no mover / interact split)
        self.predicate_interact = False  # rollout_random: lane-per-env kernel with the predicate-network interact
        self.one_kernel = False          # step_encode / rollout_encode: the single-kernel path whatever the batch size
        self.seed = int(seed)
        self.env_offset = int(env_offset)
        self.t_global = 0  # global step counter feeding the Philox counter of rollout_random
        # ONE epoch counter for every drawn start state of this env (key word of the Philox stream documented at
        # oc_reset_random): an explicit randomized reset consumes one epoch, a launch of K steps consumes K (a restart at
        # step k of the launch draws from epoch + k) — explicit resets and in-kernel restarts never share a draw
        self._epoch = 0
        # start_state_fn = get_random_start_state_fn(random_start_pos, rnd_obj_prob_thresh) (mdp.py:1307-1369): when set,
        # reset() and every restart inside the step kernels (auto_reset) draw the start state instead of the standard one
        self.random_start_pos, self.rnd_obj_prob_thresh = bool(random_start_pos), float(rnd_obj_prob_thresh)
        self.steps_done = 0  # batched steps executed
        # regen_layout: every new episode of an env runs on a layout drawn from the table (True: any of its layouts; (first,
        # count): that range) — OvercookedEnv.reset(regen_mdp=True) over a layout generator (env.py:288-302), inside the
        # fused auto-reset; the current ids are in self.layout_id (device) / layout_ids()
        self.regen = None
        if regen_layout:
            self.regen = (0, len(self.table)) if regen_layout is True else (int(regen_layout[0]), int(regen_layout[1]))
            if not (0 <= self.regen[0] and self.regen[1] >= 1 and self.regen[0] + self.regen[1] <= len(self.table)):
                raise ValueError("regen_layout range %r outside the table of %d layouts" % (self.regen, len(self.table)))
            if len(self.table) > 1 and layout_id is None:
                raise ValueError("regen_layout needs layout_id (the layouts of the first episodes)")
        self._start = _lib.OcStartSpec()
        self.width, self.height = self.table.width, self.table.height
        self.n_planes = self.table.n_planes
        assert self.lib.oc_state_planes(self.width, self.height) == self.n_planes
        dev = self.device
        self.d_layouts = torch.from_numpy(self.table.records.copy()).to(dev)
        if layout_id is None:
            if len(self.table) != 1:
                raise ValueError("layout_id is required when the table holds more than one layout")
            self.layout_id = None
        else:
            lid = np.ascontiguousarray(np.asarray(layout_id), dtype=np.int64)
            if lid.shape != (self.n_envs,) or lid.min() < 0 or lid.max() >= len(self.table):
                raise ValueError("layout_id must be [n_envs] with values in [0, %d)" % len(self.table))
            self.layout_id_host = lid.astype(np.uint16)
            # torch has no uint16 arithmetic, but we only need the bytes on the device
            self.layout_id = torch.from_numpy(self.layout_id_host.view(np.int16).copy()).to(dev)
        self.state = torch.zeros((self.n_planes, self.n_envs, 16), dtype=torch.uint8, device=dev)
        self.rewards = torch.zeros((self.n_envs, 4), dtype=torch.float32, device=dev)
        self.flags = torch.zeros((self.n_envs,), dtype=torch.uint8, device=dev)
        self.ep_returns = torch.zeros((self.n_envs, 4), dtype=torch.float32, device=dev) if track_returns else None
        self._batch = _lib.OcBatch(
            d_layouts=self.d_layouts.data_ptr(),
            d_layout_id=self.layout_id.data_ptr() if self.layout_id is not None else None,
            n_envs=self.n_envs, n_layouts=len(self.table), width=self.width, height=self.height)
        self._bref = ctypes.byref(self._batch)
        # per-episode event counters (game_stats lengths, env.py:382-401): [n_envs, 25] int32, player 0 in the low half-word
        self.event_counts = self.event_counts_done = None
        self._sink = _lib.OcEventSink()
        if track_events:
            self.event_counts = torch.zeros((self.n_envs, 25), dtype=torch.int32, device=dev)
            self.event_counts_done = torch.zeros((self.n_envs, 25), dtype=torch.int32, device=dev)
            self._sink.d_counts, self._sink.d_counts_done = self.event_counts.data_ptr(), self.event_counts_done.data_ptr()
        # kernel-variant hints (max pots, two players everywhere, max free cells) from the host copy of the table
        host_table = np.ascontiguousarray(self.table.records)
        _lib.check(self.lib.oc_batch_hints(host_table.ctypes.data, len(self.table), self._bref), "oc_batch_hints")
        assert self._batch.max_pots == self.table.max_pots
        self._plans = {}
        self._phi_tables = {}
        self._dev_index = self.device.index if self.device.index is not None else torch.cuda.current_device()
        self._state_ptr, self._rewards_ptr, self._flags_ptr = self.state.data_ptr(), self.rewards.data_ptr(), self.flags.data_ptr()
        self._ep_ptr = self.ep_returns.data_ptr() if self.ep_returns is not None else None
        self._act_shape, self._tdev, self._oc_step = torch.Size((self.n_envs, 2)), self.state.device, self.lib.oc_step
        self._constructed = False  # (the first reset keeps the layouts the caller assigned)
        self.reset()
        self._constructed = True
        self._epoch = max(self._epoch, 1)  # epoch 0 belongs to the first states, drawn or not: launches count from 1

    # ------------------------------------------------------------------ helpers
    def _check(self, t, dtype, numel, what):
        """Caller-owned output / input buffers must be contiguous tensors of the right type ON THIS GPU: the kernels
        write through the raw pointer."""
        if t.dtype != dtype or t.device != self.state.device or not t.is_contiguous() or t.numel() != numel:
            raise ValueError("%s must be a contiguous %s tensor with %d elements on %s" % (what, dtype, numel, self.state.device))

    def _stream(self):
        """Raw handle of torch's current stream on this env's device (torch.cuda.current_stream() builds a Stream object
        per call: 2.7 us of the 8 us a one-step call used to cost on the host; the raw getter is 0.06 us)."""
        if _raw_stream is not None:
            return _raw_stream(self._dev_index)
        return torch.cuda.current_stream(self.device).cuda_stream

    @property
    def options(self):
        return ((_lib.OPT_AUTO_RESET if self.auto_reset else 0)
                | (_lib.OPT_LANE_PAIR if self.lane_pair else 0) | (_lib.OPT_PREDICATE_INTERACT if self.predicate_interact else 0)
                | (_lib.OPT_ONE_WAVEFRONT if self.one_wavefront else 0))

    @property
    def reset_epoch(self):
        """Epoch the next drawn start state (explicit reset or in-kernel restart) uses."""
        return self._epoch

    def _advance(self, k):
        self.steps_done += k
        self._epoch += k

    @property
    def random_starts(self):
        return self.random_start_pos or self.rnd_obj_prob_thresh > 0.0

    def layout_ids(self):
        """Current layout index of every env (numpy uint16 [n_envs]); changes at restarts when regen_layout is on."""
        if self.layout_id is None:
            return np.zeros((self.n_envs,), np.uint16)
        return self.layout_id.cpu().numpy().view(np.uint16).copy()

    def _start_spec(self):
        """OcStartSpec* for the next launch (None = restarts from the standard start state)."""
        if not self.random_starts and self.regen is None:
            return None
        sp = self._start
        sp.regen_first, sp.regen_count = self.regen if self.regen is not None else (0, 0)
        sp.seed, sp.env_offset, sp.epoch = self.seed, self.env_offset, self._epoch & 0xFFFFFFFF
        sp.random_start_pos, sp.rnd_obj_prob_thresh = int(self.random_start_pos), self.rnd_obj_prob_thresh
        return ctypes.byref(sp)

    def _event_sink(self, events_out=None):
        """OcEventSink* for the next launch: the per-episode counters when tracking is on, plus an optional per-step mask
        buffer (int64 [n_steps, n_envs]); None = no event logging."""
        if self.event_counts is None and events_out is None:
            return None
        self._sink.d_events = events_out.data_ptr() if events_out is not None else None
        return ctypes.byref(self._sink)

    def event_stats(self, finished=False):
        """{event name: int32 tensor [n_envs, 2]}: how often each event happened per player in the running episode, or
        (finished=True) in each env's last finished episode — the lengths of the reference's game_stats lists."""
        from .mdp import EVENT_TYPES

        c = self.event_counts_done if finished else self.event_counts
        if c is None:
            raise ValueError("construct the env with track_events=True")
        both = torch.stack([c & 0xFFFF, (c >> 16) & 0xFFFF], dim=-1)
        return {name: both[:, k] for k, name in enumerate(EVENT_TYPES)}

    def spec_of(self, e):
        return self.table.specs[0 if self.layout_id is None else int(self.layout_id_host[e])]

    # ------------------------------------------------------------------ env API
    def reset(self, mask=None, random_start_pos=None, rnd_obj_prob_thresh=None):
        """Start states for all envs, or those with mask != 0 (u8/bool tensor [n_envs]): the standard start state
        (mdp.py:1297), or — with random_start_pos / rnd_obj_prob_thresh, which default to the env's own — the randomized
        start states of get_random_start_state_fn (mdp.py:1307-1369), drawn on the GPU from (seed, global env index,
        reset epoch)."""
        random_start_pos = self.random_start_pos if random_start_pos is None else random_start_pos
        rnd_obj_prob_thresh = self.rnd_obj_prob_thresh if rnd_obj_prob_thresh is None else rnd_obj_prob_thresh
        d_mask = None
        if mask is not None:
            mask = mask.to(device=self.device, dtype=torch.uint8).contiguous()
            d_mask = mask.data_ptr()
        d_ep = self.ep_returns.data_ptr() if self.ep_returns is not None else None
        with torch.cuda.device(self.device):
            if self.regen is not None and self.layout_id is not None and getattr(self, "_constructed", False):
                # regen_mdp=True semantics for an explicit reset: the selected envs move to freshly drawn layouts first
                sp = self._start
                sp.seed, sp.env_offset, sp.epoch = self.seed, self.env_offset, self._epoch & 0xFFFFFFFF
                sp.regen_first, sp.regen_count = self.regen
                _lib.check(self.lib.oc_regen_layouts(self._bref, self.layout_id.data_ptr(), d_mask, 0xFF, ctypes.byref(sp),
                                                     self._stream()), "oc_regen_layouts")
                if not (random_start_pos or rnd_obj_prob_thresh):
                    self._epoch += 1  # (the layout draw used this epoch; a drawn start state below shares it)
            if random_start_pos or rnd_obj_prob_thresh:
                rc = self.lib.oc_reset_random(self._bref, self.state.data_ptr(), d_mask, d_ep, self.seed, self.env_offset,
                                              self._epoch & 0xFFFFFFFF, int(bool(random_start_pos)),
                                              float(rnd_obj_prob_thresh), self._stream())
                self._epoch += 1
            else:
                rc = self.lib.oc_reset(self._bref, self.state.data_ptr(), d_mask, d_ep, self._stream())
        _lib.check(rc, "oc_reset")

    def _launch(self, fn, *args):
        """Call a C-ABI entry point with this env's device current (skips the context switch when it already is)."""
        if _get_device() == self._dev_index:
            return fn(*args, self._stream())
        with torch.cuda.device(self.device):
            return fn(*args, self._stream())

    def step(self, actions, state_out=None, events_out=None):
        """actions: uint8 tensor [n_envs, 2] of action indices (Action.INDEX_TO_ACTION order).
        Returns (rewards [n_envs,4] = sparse0, sparse1, shaped0, shaped1; flags [n_envs] OC_F_* bits).
        The returned tensors are reused by the next call.  events_out: optional int64 tensor [n_envs] that
        receives the event_infos bit mask of the step (bit 2*k + p = EVENT_TYPES[k] for player p)."""
        if events_out is not None:
            self._check(events_out, torch.int64, self.n_envs, "events_out")
        if state_out is not None:
            self._check(state_out, torch.uint8, self.state.numel(), "state_out")
        if actions.dtype is not torch.uint8 or actions.shape != self._act_shape or not actions.is_contiguous() \
                or actions.device != self._tdev:
            raise ValueError("actions must be a contiguous uint8 [n_envs, 2] tensor on %s" % self.device)
        if state_out is None and events_out is None and self.event_counts is None and _raw_stream is not None \
                and _get_device() == self._dev_index:
            # the call a policy loop makes every step: in place, no event logging — straight to the C entry point (the host
            # side of a one-step call is what bounds it: ~1 us of ctypes + ~2.7 us of hipLaunchKernel + this wrapper)
            sp = self._state_ptr
            rc = self._oc_step(self._bref, sp, sp, actions.data_ptr(), self._rewards_ptr, self._flags_ptr, self._ep_ptr, None,
                               self.horizon, self.options, self._start_spec() if self.auto_reset else None, None,
                               _raw_stream(self._dev_index))
            if rc:
                _lib.check(rc, "oc_step")
            self.steps_done += 1
            self._epoch += 1
            return self.rewards, self.flags
        out = self._state_ptr if state_out is None else state_out.data_ptr()
        rc = self._launch(self.lib.oc_step, self._bref, self._state_ptr, out, actions.data_ptr(), self._rewards_ptr,
                          self._flags_ptr, self._ep_ptr, events_out.data_ptr() if events_out is not None else None,
                          self.horizon, self.options, self._start_spec() if self.auto_reset else None,
                          self._event_sink() if self.event_counts is not None else None)
        if rc:
            _lib.check(rc, "oc_step")
        self._advance(1)
        return self.rewards, self.flags

    def step_encode(self, actions, dtype=torch.uint8, out=None):
        """step(actions) followed by encode_lossless of the resulting states, as one C call (oc_step_encode: one kernel
        for a single layout with at most two pots, u8 observations and a batch that fills the GPU — or `one_kernel` —
        else the two kernels back to back).  Returns (rewards, flags, obs)."""
        code = {torch.uint8: _lib.OBS_U8, torch.float32: _lib.OBS_F32}[dtype]
        if out is None:
            out = torch.empty((self.n_envs, 2, self.width, self.height, 26), dtype=dtype, device=self.device)
        else:
            self._check(out, dtype, self.n_envs * 2 * self.width * self.height * 26, "out")
        if actions.dtype != torch.uint8 or actions.shape != (self.n_envs, 2) or not actions.is_contiguous() \
                or actions.device != self.state.device:
            raise ValueError("actions must be a contiguous uint8 [n_envs, 2] tensor on %s" % self.device)
        if self.event_counts is not None:  # per-episode event counters ride on oc_step's OcEventSink
            r, f = self.step(actions)
            return r, f, self.encode_lossless(dtype, out=out)
        rc = self._launch(self.lib.oc_step_encode, self._bref, self._state_ptr, actions.data_ptr(), self._rewards_ptr,
                          self._flags_ptr, self._ep_ptr, out.data_ptr(), code, self.horizon,
                          self.options | (_lib.OPT_ONE_KERNEL if self.one_kernel else 0),
                          self._start_spec() if self.auto_reset else None)
        _lib.check(rc, "oc_step_encode")
        self._advance(1)
        return self.rewards, self.flags, out

    def step_many(self, actions, rewards_out, flags_out, events_out=None):
        """K consecutive steps enqueued from C: actions uint8 [K, n_envs, 2] -> rewards_out float32 [K, n_envs, 4],
        flags_out uint8 [K, n_envs]; events_out: optional int64 [K, n_envs] event masks."""
        K = actions.shape[0]
        if events_out is not None:
            self._check(events_out, torch.int64, int(K) * self.n_envs, "events_out")
        self._check(actions, torch.uint8, K * self.n_envs * 2, "actions")
        self._check(rewards_out, torch.float32, K * self.n_envs * 4, "rewards_out")
        self._check(flags_out, torch.uint8, K * self.n_envs, "flags_out")
        rc = self._launch(self.lib.oc_step_many, self._bref, self._state_ptr, actions.data_ptr(), rewards_out.data_ptr(),
                          flags_out.data_ptr(), self._ep_ptr, int(K), self.horizon, self.options,
                          self._start_spec() if self.auto_reset else None, self._event_sink(events_out))
        _lib.check(rc, "oc_step_many")
        self._advance(int(K))
        return rewards_out, flags_out

    def step_server(self, idle_ms=0.0, life_s=0.0):
        """A StepServer for this env: the resident batched step (no launch per step; include/oc_amd.h oc_step_server_*).  Event
        tracking is not served."""
        if self.event_counts is not None:
            raise ValueError("step_server: event tracking (track_events) is not served by the resident kernel")
        return StepServer(self, idle_ms, life_s)

    def rollout_random(self, n_steps, rewards_out=None, flags_out=None, events_out=None, flags_tiled8=False):
        """n_steps fused random-policy transitions in one launch (Philox actions, see include/oc_amd.h).  A launch
        costs ~16 us outside its step loop (tables, state load / store, dispatch): 12 % of a 400-step launch at 65 536
        envs, 3 % of a 2 000-step one — prefer few long launches.
        rewards_out: float32 [n_steps, n_envs, 4] or None; flags_out: uint8 [n_steps, n_envs] or None; events_out:
        int64 [n_steps, n_envs] event masks or None.
        flags_tiled8 (OC_OPT_FLAGS_TILED8): flags_out is [n_steps // 8, n_envs, 8] — byte [k // 8, e, k % 8] = step k of env e
        (`untile_flags` gives the [n_steps, n_envs] view's copy) —, which the kernel writes as full lines: worth ~6 % on the
        joint-table kernel (one cramped_room-like layout; whole 256-env workgroups up to 524 288 envs, else up to ~98 000; n_steps and the step counter multiples of 8;
        ValueError via OC_EINVAL otherwise)."""
        if events_out is not None:
            self._check(events_out, torch.int64, int(n_steps) * self.n_envs, "events_out")
        if rewards_out is not None:
            self._check(rewards_out, torch.float32, int(n_steps) * self.n_envs * 4, "rewards_out")
        if flags_out is not None:
            self._check(flags_out, torch.uint8, int(n_steps) * self.n_envs, "flags_out")
        with torch.cuda.device(self.device):
            rc = self.lib.oc_rollout_random(
                self._bref, self.state.data_ptr(),
                rewards_out.data_ptr() if rewards_out is not None else None,
                flags_out.data_ptr() if flags_out is not None else None,
                self.ep_returns.data_ptr() if self.ep_returns is not None else None,
                self.horizon, self.options | (_lib.OPT_FLAGS_TILED8 if flags_tiled8 else 0), self.seed, self.env_offset,
                self.t_global, int(n_steps), self._start_spec() if self.auto_reset else None, self._event_sink(events_out),
                self._stream())
        _lib.check(rc, "oc_rollout_random")
        self.t_global += int(n_steps)
        self._advance(int(n_steps))
        return rewards_out, flags_out

    @staticmethod
    def untile_flags(flags_tiled):
        """[n_steps // 8, n_envs, 8] (OC_OPT_FLAGS_TILED8) -> a [n_steps, n_envs] copy."""
        t, n, _ = flags_tiled.shape
        return flags_tiled.permute(0, 2, 1).reshape(t * 8, n)

    def rollout_encode(self, n_steps, obs_out, rewards_out=None, flags_out=None, actions=None, dtype=torch.uint8):
        """n_steps transitions with the lossless observation after every step, one C call (oc_rollout_encode; a single
        kernel for one layout / u8 / at most two pots).  actions: None = the random policy of rollout_random (same Philox
        stream), or uint8 [n_steps, n_envs, 2].  obs_out: [n_steps, n_envs, 2, W, H, 26] (the whole trajectory) or
        [n_envs, 2, W, H, 26] (every step overwrites it: only the last observation survives).  rewards_out float32
        [n_steps, n_envs, 4] / flags_out uint8 [n_steps, n_envs] (required with caller actions).  With track_events the
        one-step calls run step by step (the event counters ride on their OcEventSink)."""
        K = int(n_steps)
        code = {torch.uint8: _lib.OBS_U8, torch.float32: _lib.OBS_F32}[dtype]
        per_step = self.n_envs * 2 * self.width * self.height * 26
        if obs_out.dim() == 6:
            self._check(obs_out, dtype, K * per_step, "obs_out")
            stride = per_step * obs_out.element_size()
        else:
            self._check(obs_out, dtype, per_step, "obs_out")
            stride = 0
        if actions is not None:
            self._check(actions, torch.uint8, K * self.n_envs * 2, "actions")
        if rewards_out is not None:
            self._check(rewards_out, torch.float32, K * self.n_envs * 4, "rewards_out")
        if flags_out is not None:
            self._check(flags_out, torch.uint8, K * self.n_envs, "flags_out")
        if actions is not None and (rewards_out is None or flags_out is None):
            raise ValueError("caller actions need rewards_out and flags_out")
        if stride % 16 != 0 or self.event_counts is not None:  # rows of odd sizes / event counters: the one-step calls, step by step
            tmp = torch.empty_like(obs_out[0]) if stride and stride % 16 != 0 else None  # rows of odd sizes: encode into an aligned buffer
            for k in range(K):
                obs_k = tmp if tmp is not None else (obs_out[k] if stride else obs_out)
                if actions is None:
                    self.rollout_random(1, None if rewards_out is None else rewards_out[k:k + 1],
                                        None if flags_out is None else flags_out[k:k + 1])
                else:
                    r, f = self.step(actions[k])
                    rewards_out[k].copy_(r)
                    flags_out[k].copy_(f)
                self.encode_lossless(dtype, out=obs_k)
                if tmp is not None:
                    obs_out[k].copy_(tmp)
            return obs_out, rewards_out, flags_out
        rc = self._launch(self.lib.oc_rollout_encode, self._bref, self._state_ptr,
                          actions.data_ptr() if actions is not None else None,
                          rewards_out.data_ptr() if rewards_out is not None else None,
                          flags_out.data_ptr() if flags_out is not None else None, self._ep_ptr, obs_out.data_ptr(), code,
                          stride, self.horizon,
                          (_lib.OPT_AUTO_RESET if self.auto_reset else 0) | (_lib.OPT_ONE_KERNEL if self.one_kernel else 0),
                          self.seed, self.env_offset, self.t_global, K, self._start_spec() if self.auto_reset else None)
        _lib.check(rc, "oc_rollout_encode")
        if actions is None:
            self.t_global += K
        self._advance(K)
        return obs_out, rewards_out, flags_out

    def encode_lossless(self, dtype=torch.uint8, out=None, state=None):
        """[n_envs, 2, W, H, 26] observation (mdp.py:2385); out[:, i] is the encoding for player i."""
        code = {torch.uint8: _lib.OBS_U8, torch.float32: _lib.OBS_F32}[dtype]
        if out is None:
            out = torch.empty((self.n_envs, 2, self.width, self.height, 26), dtype=dtype, device=self.device)
        else:
            self._check(out, dtype, self.n_envs * 2 * self.width * self.height * 26, "out")
        st = self.state if state is None else state
        with torch.cuda.device(self.device):
            rc = self.lib.oc_encode_lossless(self._bref, st.data_ptr(), out.data_ptr(), code, self.horizon,
                                             self._stream())
        _lib.check(rc, "oc_encode_lossless")
        return out

    def featurize(self, num_pots=2, counter_goals="none", out=None, state=None):
        """[n_envs, 2, 2*(num_pots*10+26)+4] float32 hand-crafted features (mdp.py:2579); out[:, i] is for player i.
        counter_goals: "none" (the reference's NO_COUNTERS_PARAMS default), "all", or a list of (x, y) counters."""
        blob, offs = self._plan(counter_goals)
        total = 2 * (num_pots * 10 + 26) + 4
        if out is None:
            out = torch.empty((self.n_envs, 2, total), dtype=torch.float32, device=self.device)
        else:
            self._check(out, torch.float32, self.n_envs * 2 * total, "out")
        st = self.state if state is None else state
        rc = self._launch(self.lib.oc_featurize, self._bref, blob.data_ptr(), offs.data_ptr(), st.data_ptr(), out.data_ptr(),
                          int(num_pots))
        _lib.check(rc, "oc_featurize")
        return out

    def _plan(self, counter_goals):
        key = counter_goals if isinstance(counter_goals, str) else tuple(sorted(map(tuple, counter_goals)))
        if key not in self._plans:
            from .planner import pack_plan_tables
            blob, offs = pack_plan_tables(self.table.specs, counter_goals)
            self._plans[key] = (torch.from_numpy(blob).to(self.device), torch.from_numpy(offs.view(np.int32).copy()).to(self.device))
        return self._plans[key]

    def potential(self, gamma=0.99, out=None, state=None):
        """[n_envs] float64 phi(s) of potential-based reward shaping (potential_function, mdp.py:2920) for the
        discount factor `gamma`; the per-layout tables are built once per gamma on the host (potential.py)."""
        gamma = float(gamma)
        if gamma not in self._phi_tables:
            from .potential import pack_phi_tables
            self._phi_tables[gamma] = torch.from_numpy(pack_phi_tables(self.table.specs, gamma)).to(self.device)
        blob, offs = self._plan("none")
        if out is None:
            out = torch.empty((self.n_envs,), dtype=torch.float64, device=self.device)
        else:
            self._check(out, torch.float64, self.n_envs, "out")
        st = self.state if state is None else state
        rc = self._launch(self.lib.oc_potential, self._bref, blob.data_ptr(), offs.data_ptr(),
                          self._phi_tables[gamma].data_ptr(), st.data_ptr(), out.data_ptr())
        _lib.check(rc, "oc_potential")
        return out

    # ------------------------------------------------------------------ host <-> device state
    def set_packed_state(self, packed):
        packed = np.ascontiguousarray(packed, dtype=np.uint8)
        assert packed.shape == (self.n_planes, self.n_envs, 16)
        self.state.copy_(torch.from_numpy(packed))

    def get_packed_state(self):
        return self.state.cpu().numpy()

    def _refresh_layout_ids(self):
        if self.regen is not None and self.layout_id is not None:  # restarts may have moved envs to other layouts
            self.layout_id_host = self.layout_ids()

    def set_states(self, states):
        assert len(states) == self.n_envs
        self._refresh_layout_ids()
        out = np.zeros((self.n_planes, self.n_envs, 16), np.uint8)
        for e, s in enumerate(states):
            out[:, e:e + 1] = pack_states(self.spec_of(e), [s], self.n_planes)
        self.set_packed_state(out)

    def get_states(self, as_dict=False):
        self._refresh_layout_ids()
        packed = self.get_packed_state()
        return [unpack_states(self.spec_of(e), packed[:, e:e + 1], as_dict=as_dict)[0] for e in range(self.n_envs)]


class StepServer:
    """The resident batched step of a VecOvercookedEnv (include/oc_amd.h, oc_step_server_*): a kernel that keeps the envs on chip
    and serves every step through per-env mailboxes in device memory, for callers that live on the GPU themselves.

        with env.step_server() as sv:
            sv.play(actions, rewards_out, flags_out)   # K steps == env.step_many(actions, ...), bit for bit
            r, f = sv.step(actions_1)                  # one step == env.step(actions_1)
        # here env.state / env.ep_returns are current again

    While it is open, `env.state` is stale (the states live on chip) until sync() / close().  Device-wide synchronisation
    (torch.cuda.synchronize()) waits until the kernel leaves by itself (idle_ms without a request): synchronise streams or events.
    `requests` / `responses` are the raw mailboxes for device-side callers (formats in include/oc_amd.h)."""

    def __init__(self, env, idle_ms=0.0, life_s=0.0):
        self.env = env
        self._h = ctypes.c_void_p()
        with torch.cuda.device(env.device):
            torch.cuda.current_stream(env.device).synchronize()  # (the resident kernel reads env.state on a stream of its own)
            rc = env.lib.oc_step_server_open(env._bref, env._state_ptr, env._ep_ptr, env.horizon, _lib.OPT_AUTO_RESET if env.auto_reset else 0,
                                             env._start_spec() if env.auto_reset else None, float(idle_ms), float(life_s), ctypes.byref(self._h))
        _lib.check(rc, "oc_step_server_open")
        self.requests = env.lib.oc_step_server_requests(self._h)    # device pointer: uint64 [n_envs]
        self.responses = env.lib.oc_step_server_responses(self._h)  # device pointer: uint32 [n_envs][8]
        self.last_play_ms = 0.0
        self._ms = ctypes.c_float()

    def play(self, actions, rewards_out, flags_out):
        """K steps with the caller's actions uint8 [K, n_envs, 2] -> rewards_out float32 [K, n_envs, 4], flags_out uint8 [K, n_envs]
        (the client kernel on torch's current stream; returns when the outputs are there)."""
        env, K = self.env, int(actions.shape[0])
        env._check(actions, torch.uint8, K * env.n_envs * 2, "actions")
        env._check(rewards_out, torch.float32, K * env.n_envs * 4, "rewards_out")
        env._check(flags_out, torch.uint8, K * env.n_envs, "flags_out")
        with torch.cuda.device(env.device):
            rc = env.lib.oc_step_server_play(self._h, actions.data_ptr(), rewards_out.data_ptr(), flags_out.data_ptr(), K,
                                             env._stream(), ctypes.byref(self._ms))
        _lib.check(rc, "oc_step_server_play")
        self.last_play_ms = float(self._ms.value)
        env._advance(K)
        return rewards_out, flags_out

    def step(self, actions):
        """One step: actions uint8 [n_envs, 2] -> (env.rewards, env.flags), as VecOvercookedEnv.step."""
        env = self.env
        self.play(actions.view(1, env.n_envs, 2), env.rewards.view(1, env.n_envs, 4), env.flags.view(1, env.n_envs))
        return env.rewards, env.flags

    def resume(self):
        """(Re)launch the resident kernel if it has left — for device-side callers, before a burst of requests."""
        torch.cuda.current_stream(self.env.device).synchronize()  # (a relaunch reads env.state on the server's own stream)
        _lib.check(self.env.lib.oc_step_server_resume(self._h), "oc_step_server_resume")

    def sync(self):
        """The resident kernel leaves and writes the states back: env.state / env.ep_returns are current (the next play relaunches)."""
        _lib.check(self.env.lib.oc_step_server_sync(self._h), "oc_step_server_sync")

    @property
    def steps(self):
        return int(self.env.lib.oc_step_server_steps(self._h))

    def close(self):
        if self._h:
            h, self._h = self._h, ctypes.c_void_p()
            _lib.check(self.env.lib.oc_step_server_close(h), "oc_step_server_close")

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()
        return False

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass
