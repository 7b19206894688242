"""The batched form of the reference's RLlib training environment (`OvercookedMultiAgent`,
human_aware_rl/rllib/rllib.py:112-438) on top of the accelerated path.

`VecOvercookedMultiAgent` is that environment for N envs resident in HBM: per batched step one
`oc_multi_agent_step` call (step -> phi(s') -> per-agent training reward sparse + factor * (phi(s') - phi(s) | shaped)
-> restart of finished envs -> observation), no host synchronisation.  Every env has two learning ("ppo") agents;
`bc` partners need a behaviour-cloning model, which is outside this package — their observation (`featurize_state`)
is available through `observations("bc")`.

`OvercookedMultiAgent` is the per-env, dict-keyed form RLlib-style training code holds (one `OvercookedEnv` underneath;
agent ids "ppo_0" / "bc_1" redrawn every reset from the bc schedule, observations by agent kind, the same reward
arithmetic) — the surface of the reference class without its RLlib base class or registry (those belong to the
out-of-scope RLlib stack).  Both are pinned by episodes recorded from the reference class
(tests/golden/multi_agent_*.npz).
"""
import numpy as np


def linear_anneal(start_value, t, t_end, end_value=0.0, t_start=0):
    """Value of a linear schedule at timestep t: from (t_start, start_value) to (t_end, end_value), constant afterwards;
    t_end == 0 switches annealing off.  Same floating-point expression as rllib.py:283-291 (weight of the start value
    first), so that annealed factors compare equal to the reference's."""
    if t_end == 0:
        return start_value
    keep = max(1 - float(t - t_start) / (t_end - t_start), 0)
    return keep * start_value + (1 - keep) * end_value


def schedule_value(points, t):
    """Piecewise-linear schedule through (timestep, value) points, flat after the last one (the bc_schedule of
    rllib.py:369-384): the value at timestep t."""
    lo, hi = points[0], points[1]
    for nxt in points[2:]:
        if t <= hi[0]:
            break
        lo, hi = hi, nxt
    return linear_anneal(lo[1], t, hi[0], hi[1], lo[0])


def _checked_schedule(points):
    """A bc schedule as the reference accepts it (rllib.py:187-211): >= 2 points, starting at timestep 0, increasing
    timesteps, values in [0, 1]; a closing point at infinity keeps the last value forever."""
    pts = [tuple(p) for p in points]
    ts, vs = [p[0] for p in pts], [p[1] for p in pts]
    assert len(pts) >= 2, "Need at least 2 points to linearly interpolate schedule"
    assert ts[0] == 0, "Schedule must start at timestep 0"
    assert all(t >= 0 for t in ts), "All timesteps in schedule must be non-negative"
    assert all(0 <= v <= 1 for v in vs), "All values in schedule must be between 0 and 1"
    assert sorted(ts) == ts, "Timesteps must be in increasing order in schedule"
    if ts[-1] < float("inf"):
        pts.append((float("inf"), vs[-1]))
    return pts


class OvercookedMultiAgent:
    """One two-agent training env with per-agent dict observations / rewards / dones, as RLlib-style code drives it
    (API of human_aware_rl/rllib/rllib.py:112-438).  Each reset seats a "ppo" learner and, with probability `bc_factor`
    (read off `bc_schedule`), a "bc" partner instead of a second learner, in random player order; agent ids are
    `<kind>_<player index>`.  ppo agents observe the lossless encoding of their player, bc agents featurize_state.
    reward_i = sparse + reward_shaping_factor * (phi(s') - phi(s)  if use_phi  else  shaped_i)."""

    supported_agents = ["ppo", "bc"]
    bc_schedule = self_play_bc_schedule = [(0, 0), (float("inf"), 0)]  # no bc partner at any time
    DEFAULT_CONFIG = {
        "mdp_params": {"layout_name": "cramped_room", "rew_shaping_params": {}},
        "env_params": {"horizon": 400},
        "multi_agent_params": {"reward_shaping_factor": 0.0, "reward_shaping_horizon": 0,
                               "bc_schedule": self_play_bc_schedule, "use_phi": True},
    }

    def __init__(self, base_env, reward_shaping_factor=0.0, reward_shaping_horizon=0, bc_schedule=None, use_phi=True):
        self.bc_schedule = _checked_schedule(bc_schedule or self.bc_schedule)
        self.base_env, self.use_phi = base_env, use_phi
        self._initial_reward_shaping_factor = self.reward_shaping_factor = reward_shaping_factor
        self.reward_shaping_horizon = reward_shaping_horizon
        self.anneal_bc_factor(0)
        self._agent_ids = set(self.reset())
        self._spaces_in_preferred_format = True

    # ---- observations
    def _features(self, kind, state):
        if kind == "ppo":
            return self.base_env.lossless_state_encoding_mdp(state)
        if kind == "bc":
            return self.base_env.featurize_state_mdp(state)
        raise ValueError("Unsupported agent type {0}".format(kind))

    def _observe(self, state):
        """{agent id: float32 observation of that agent's player}."""
        return {name: np.asarray(self._features(name.split("_")[0], state)[i], dtype=np.float32)
                for i, name in enumerate(self.curr_agents)}

    def _spaces(self, names):
        from .env import _mk_box, _mk_discrete

        n_act = 6
        probe = self.base_env.mdp.get_standard_start_state()
        box = {}
        for kind, hi, lo in (("ppo", float("inf"), 0.0), ("bc", 100.0, -100.0)):
            shape = np.asarray(self._features(kind, probe)[0]).shape
            box[kind] = _mk_box(np.full(shape, lo, np.float32), np.full(shape, hi, np.float32), np.float32)
        self.ppo_observation_space, self.bc_observation_space = box["ppo"], box["bc"]
        self.shared_action_space = _mk_discrete(n_act)
        self.action_space = {a: _mk_discrete(n_act) for a in names}
        self.observation_space = {a: box[a.split("_")[0]] for a in names}

    def _seat_agents(self):
        """One learner plus (bc_factor coin) a bc partner or a second learner, shuffled over the two player slots — the
        same two np.random draws, in the same order, as the reference (rllib.py:262-281)."""
        kinds = ["ppo", "bc" if np.random.uniform() < self.bc_factor else "ppo"]
        np.random.shuffle(kinds)
        names = ["%s_%d" % (k, i) for i, k in enumerate(kinds)]
        self._spaces(names)
        return names

    # ---- env API
    def reset(self, regen_mdp=True):
        self.base_env.reset(regen_mdp)
        self.curr_agents = self._seat_agents()
        return self._observe(self.base_env.state)

    def step(self, action_dict):
        from .actions import Action

        idx = [action_dict[a] for a in self.curr_agents]
        assert all(self.action_space[a].contains(action_dict[a]) for a in action_dict), "%r (%s) invalid" % (idx, type(idx))
        joint = [Action.INDEX_TO_ACTION[i] for i in idx]
        next_state, sparse, done, info = self.base_env.step(joint, display_phi=self.use_phi)
        if self.use_phi:
            dense = [info["phi_s_prime"] - info["phi_s"]] * 2
        else:
            dense = info["shaped_r_by_agent"]
        a0, a1 = self.curr_agents
        rewards = {a0: sparse + self.reward_shaping_factor * dense[0], a1: sparse + self.reward_shaping_factor * dense[1]}
        return self._observe(next_state), rewards, {a0: done, a1: done, "__all__": done}, {a0: info, a1: info}

    # ---- schedules
    def anneal_reward_shaping_factor(self, timesteps):
        self.set_reward_shaping_factor(linear_anneal(self._initial_reward_shaping_factor, timesteps,
                                                     self.reward_shaping_horizon))

    def anneal_bc_factor(self, timesteps):
        self.set_bc_factor(schedule_value(self.bc_schedule, timesteps))

    def set_reward_shaping_factor(self, factor):
        self.reward_shaping_factor = factor

    def set_bc_factor(self, factor):
        self.bc_factor = factor

    def seed(self, seed):
        """(the env itself draws nothing but the seating, from np.random)"""

    @classmethod
    def from_config(cls, env_config):
        """RLlib-style factory: {"mdp_params": kwargs of OvercookedGridworld.from_layout_name, "env_params": kwargs of
        OvercookedEnv.from_mdp, "multi_agent_params": kwargs of this class}.  The reference's alternative
        "mdp_params_schedule_fn" (a LayoutGenerator curriculum, rllib.py:415-424) is not supported here."""
        from .env import OvercookedEnv
        from .mdp import OvercookedGridworld

        assert env_config and "env_params" in env_config and "multi_agent_params" in env_config
        if "mdp_params" not in env_config:
            raise NotImplementedError("OvercookedMultiAgent.from_config: only fixed 'mdp_params' are supported "
                                      "('mdp_params_schedule_fn' needs the reference's LayoutGenerator curriculum)")
        params = dict(env_config["mdp_params"])
        mdp = OvercookedGridworld.from_layout_name(params.pop("layout_name"), **{k: v for k, v in params.items() if v not in ({}, None)})
        base_env = OvercookedEnv.from_mdp(mdp, info_level=0, **env_config["env_params"])
        return cls(base_env, **env_config["multi_agent_params"])


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
