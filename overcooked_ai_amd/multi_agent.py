"""The batched form of the reference's RLlib training environment (`OvercookedMultiAgent`,
human_aware_rl/rllib/rllib.py:112-438) on top of the accelerated path.

`VecOvercookedMultiAgent` is that environment for N envs resident in HBM: per batched step one
`oc_multi_agent_step` call (step -> phi(s') -> per-agent training reward sparse + factor * (phi(s') - phi(s) | shaped)
-> restart of finished envs -> observation), no host synchronisation.  Every env has two learning ("ppo") agents;
`bc` partners need a behaviour-cloning model, which is outside this package — their observation (`featurize_state`)
is available through `observations("bc")`.  The reference's per-env dict-API class itself (agent-role sampling, RLlib
spaces) belongs to its RLlib stack and is out of scope; only its reward / restart / observation
arithmetic is reproduced, pinned by episodes recorded from the reference class (tests/golden/multi_agent_*.npz).
"""
import numpy as np


def linear_anneal(start_value, t, t_end, end_value=0.0, t_start=0):
    """Value of the reward-shaping schedule at timestep t: a straight line from (t_start, start_value) to
    (t_end, end_value), constant afterwards; t_end == 0 switches annealing off (rllib.py:280-291)."""
    if t_end == 0:
        return start_value
    w = min(max((t - t_start) / float(t_end - t_start), 0.0), 1.0)
    return (1.0 - w) * start_value + w * end_value


class VecOvercookedMultiAgent:
    """N two-agent training envs in HBM.  `step(actions)` takes uint8 [n_envs, 2] action indices (player order) and
    returns (obs, rewards, dones, infos) as device tensors; finished episodes restart inside the call, and `obs` of
    those envs is the first observation of the new episode (the usual vector-env convention)."""

    def __init__(self, layouts, n_envs, horizon=400, reward_shaping_factor=0.0, reward_shaping_horizon=0, use_phi=True,
                 gamma=0.99, obs="ppo", obs_dtype=None, device="cuda", random_start_pos=False, rnd_obj_prob_thresh=0.0,
                 **venv_kwargs):
        import torch

        from . import _lib
        from .vec_env import VecOvercookedEnv

        self._torch, self._lib = torch, _lib
        venv_kwargs.pop("auto_reset", None)
        self.venv = VecOvercookedEnv(layouts, n_envs, horizon=horizon, device=device, auto_reset=False,
                                     random_start_pos=random_start_pos, rnd_obj_prob_thresh=rnd_obj_prob_thresh,
                                     **venv_kwargs)
        v = self.venv
        self.n_envs, self.horizon, self.use_phi, self.gamma = v.n_envs, v.horizon, bool(use_phi), float(gamma)
        self._initial_reward_shaping_factor = self.reward_shaping_factor = reward_shaping_factor
        self.reward_shaping_horizon = reward_shaping_horizon
        self.obs_kind = obs
        # start_state_fn = mdp.get_random_start_state_fn(random_start_pos, rnd_obj_prob_thresh) of the reference's
        # env_params (mdp.py:1307-1369): every episode, including the restarts inside step(), begins from a random state
        self.random_start_pos, self.rnd_obj_prob_thresh = bool(random_start_pos), float(rnd_obj_prob_thresh)
        self._random_starts = self.random_start_pos or self.rnd_obj_prob_thresh > 0.0
        self.obs_dtype = obs_dtype or torch.float32  # the reference casts observations to float32 (rllib.py:257)
        dev = v.device
        self.shaped = torch.zeros((self.n_envs, 2), dtype=torch.float64, device=dev)
        self.done = torch.zeros((self.n_envs,), dtype=torch.uint8, device=dev)
        self.phi_next = torch.zeros((self.n_envs,), dtype=torch.float64, device=dev)
        self.phi_cur = torch.zeros((self.n_envs,), dtype=torch.float64, device=dev)
        self.phi_start = torch.zeros((len(v.table),), dtype=torch.float64, device=dev)
        self.ep_returns = torch.zeros((self.n_envs, 4), dtype=torch.float32, device=dev)
        self._obs = None
        self._phi_args = None
        if self.use_phi:
            # phi of each layout's STANDARD start state (what a standard reset carries into phi_cur), from a probe batch of
            # one env per layout — the training batch itself may already hold drawn start states (epoch 0)
            L = len(v.table)
            probe = VecOvercookedEnv(v.table, L, horizon=horizon, device=dev, layout_id=np.arange(L) if L > 1 else None)
            self.phi_start.copy_(probe.potential(self.gamma))
            v.potential(self.gamma, out=self.phi_cur)

    def _obs_buffer(self):
        if self._obs is None or self._obs.dtype != self.obs_dtype:
            self._obs = self._torch.empty((self.n_envs, 2, self.venv.width, self.venv.height, 26),
                                          dtype=self.obs_dtype, device=self.venv.device)
        return self._obs

    def observations(self, kind=None):
        """Both agents' observations of the current states: "ppo" -> [n_envs, 2, W, H, 26] (lossless encoding),
        "bc" -> [n_envs, 2, 96] (featurize_state)."""
        kind = kind or self.obs_kind
        if kind == "ppo":
            return self.venv.encode_lossless(self.obs_dtype, out=self._obs_buffer())
        if kind == "bc":
            return self.venv.featurize()
        raise ValueError("Unsupported agent type {0}".format(kind))

    def reset(self):
        if self._random_starts:
            self.venv.reset(random_start_pos=self.random_start_pos, rnd_obj_prob_thresh=self.rnd_obj_prob_thresh)
            if self.use_phi:
                self.venv.potential(self.gamma, out=self.phi_cur)
            return self.observations()
        self.venv.reset()
        if self.use_phi:
            lid = self.venv.layout_id
            self.phi_cur.copy_(self.phi_start[lid.long() & 0xFFFF] if lid is not None else self.phi_start.expand(self.n_envs))
        return self.observations()

    def step(self, actions):
        """One batched training step, enqueued by a single C call (oc_multi_agent_step)."""
        v, torch = self.venv, self._torch
        if actions.dtype != torch.uint8 or actions.shape != (self.n_envs, 2) or not actions.is_contiguous() \
                or actions.device != v.state.device:
            raise ValueError("actions must be a contiguous uint8 [n_envs, 2] tensor on %s" % v.device)
        obs = self._obs_buffer() if self.obs_kind == "ppo" else None
        code = {torch.uint8: self._lib.OBS_U8, torch.float32: self._lib.OBS_F32}[self.obs_dtype]
        if self.use_phi:
            if self._phi_args is None:
                v.potential(self.gamma, out=self.phi_next)  # builds and caches the tables
                blob, offs = v._plan("none")
                self._phi_args = (blob.data_ptr(), offs.data_ptr(), v._phi_tables[self.gamma].data_ptr())
            plan, off, tables = self._phi_args
        else:
            plan = off = tables = None
        rc = v._launch(v.lib.oc_multi_agent_step, v._bref, v._state_ptr, actions.data_ptr(), v._rewards_ptr, v._flags_ptr,
                       v._ep_ptr, self.ep_returns.data_ptr(), plan, off, tables, self.phi_next.data_ptr(),
                       self.phi_cur.data_ptr(), self.phi_start.data_ptr(), float(self.reward_shaping_factor),
                       self.shaped.data_ptr(), self.done.data_ptr(), obs.data_ptr() if obs is not None else None, code,
                       self.horizon, v._start_spec(),  # random starts: finished envs restart from drawn states in the same call
                       v._event_sink() if v.event_counts is not None else None)
        self._lib.check(rc, "oc_multi_agent_step")
        v._advance(1)
        infos = {"sparse_r_by_agent": v.rewards[:, 0:2], "shaped_r_by_agent": v.rewards[:, 2:4], "flags": v.flags,
                 "ep_returns": self.ep_returns}  # episode totals so far; final where done (the reset cleared the live ones)
        if self.use_phi:
            infos["phi_s_prime"] = self.phi_next
        return (obs if obs is not None else self.observations()), self.shaped, self.done, infos

    def anneal_reward_shaping_factor(self, timesteps):
        self.reward_shaping_factor = linear_anneal(self._initial_reward_shaping_factor, timesteps,
                                                   self.reward_shaping_horizon)

    def set_reward_shaping_factor(self, factor):
        self.reward_shaping_factor = factor
