"""The RLlib training environment of the reference, `OvercookedMultiAgent`
(human_aware_rl/rllib/rllib.py:112-438), on top of the accelerated path — without the `ray` dependency.

* `OvercookedMultiAgent` keeps the reference's per-env dict API (`reset() -> {agent: obs}`,
  `step({agent: action}) -> obs, rewards, dones, infos`, the annealing setters, `from_config`), its agent-role
  assignment (same `np.random` calls, so seeded runs agree) and its reward: sparse + factor * (phi(s') - phi(s)) when
  `use_phi`, else sparse + factor * shaped_r_by_agent[i].  RLlib only needs the duck-typed methods of
  `MultiAgentEnv`, so the class can be registered with `ray.tune.registry.register_env` unchanged where ray exists.
* `VecOvercookedMultiAgent` is the same environment for N envs resident in HBM: one `oc_step`, one `oc_potential`,
  one `oc_shape_rewards`, a masked `oc_reset` and one `oc_encode_lossless` / `oc_featurize` per batched step, no
  host synchronisation.  Every env has two learning ("ppo") agents; `bc` partners need a behaviour-cloning model, which
  is outside this package — their observation (`featurize_state`) is available through `observations("bc")`.
"""
import numpy as np

from .actions import Action
from .env import OvercookedEnv, _Box, _Discrete


class _DictSpace(dict):
    def contains(self, d):
        return all(k in self and self[k].contains(v) for k, v in d.items())


class OvercookedMultiAgent:
    """Drop-in mirror of rllib.py:112-438 for one env."""

    supported_agents = ["ppo", "bc"]
    bc_schedule = self_play_bc_schedule = [(0, 0), (float("inf"), 0)]
    DEFAULT_CONFIG = {
        "mdp_params": {"layout_name": "cramped_room", "rew_shaping_params": {}},
        "env_params": {"horizon": 400},
        "multi_agent_params": {"reward_shaping_factor": 0.0, "reward_shaping_horizon": 0,
                               "bc_schedule": self_play_bc_schedule, "use_phi": True},
    }

    def __init__(self, base_env, reward_shaping_factor=0.0, reward_shaping_horizon=0, bc_schedule=None, use_phi=True):
        if bc_schedule:
            self.bc_schedule = bc_schedule
        self._validate_schedule(self.bc_schedule)
        self.base_env = base_env
        self.featurize_fn_map = {"ppo": lambda state: self.base_env.lossless_state_encoding_mdp(state),
                                 "bc": lambda state: self.base_env.featurize_state_mdp(state)}
        self._initial_reward_shaping_factor = reward_shaping_factor
        self.reward_shaping_factor = reward_shaping_factor
        self.reward_shaping_horizon = reward_shaping_horizon
        self.use_phi = use_phi
        self.anneal_bc_factor(0)
        self._agent_ids = set(self.reset().keys())
        self._spaces_in_preferred_format = True

    @staticmethod
    def _validate_schedule(schedule):
        timesteps, values = [p[0] for p in schedule], [p[1] for p in schedule]
        assert len(schedule) >= 2, "Need at least 2 points to linearly interpolate schedule"
        assert schedule[0][0] == 0, "Schedule must start at timestep 0"
        assert all(t >= 0 for t in timesteps), "All timesteps in schedule must be non-negative"
        assert all(0 <= v <= 1 for v in values), "All values in schedule must be between 0 and 1"
        assert sorted(timesteps) == timesteps, "Timesteps must be in increasing order in schedule"
        if schedule[-1][0] < float("inf"):  # flatline after the last point (rllib.py:203-205)
            schedule.append((float("inf"), schedule[-1][1]))

    def _setup_action_space(self, agents):
        self.action_space = _DictSpace({a: _Discrete(len(Action.ALL_ACTIONS)) for a in agents})
        self.shared_action_space = _Discrete(len(Action.ALL_ACTIONS))

    def _setup_observation_space(self, agents):
        dummy_state = self.base_env.mdp.get_standard_start_state()
        shape = self.base_env.lossless_state_encoding_mdp(dummy_state)[0].shape
        self.ppo_observation_space = _Box(np.zeros(shape, np.float32), np.full(shape, np.inf, np.float32), dtype=np.float32)
        shape = self.base_env.featurize_state_mdp(dummy_state)[0].shape
        self.bc_observation_space = _Box(np.full(shape, -100, np.float32), np.full(shape, 100, np.float32), dtype=np.float32)
        self.observation_space = _DictSpace({a: self.ppo_observation_space if a.startswith("ppo") else self.bc_observation_space
                                             for a in agents})

    def _get_featurize_fn(self, agent_id):
        if agent_id.startswith("ppo"):
            return self.featurize_fn_map["ppo"]
        if agent_id.startswith("bc"):
            return self.featurize_fn_map["bc"]
        raise ValueError("Unsupported agent type {0}".format(agent_id))

    def _get_obs(self, state):
        ob_p0 = self._get_featurize_fn(self.curr_agents[0])(state)[0]
        ob_p1 = self._get_featurize_fn(self.curr_agents[1])(state)[1]
        return ob_p0.astype(np.float32), ob_p1.astype(np.float32)

    def _populate_agents(self):
        agents = ["ppo"]  # always at least one learning agent (rllib.py:259-278)
        agents.append("bc" if np.random.uniform() < self.bc_factor else "ppo")
        np.random.shuffle(agents)
        agents[0], agents[1] = agents[0] + "_0", agents[1] + "_1"
        self._setup_action_space(agents)
        self._setup_observation_space(agents)
        return agents

    @staticmethod
    def _anneal(start_v, curr_t, end_t, end_v=0, start_t=0):
        if end_t == 0:
            return start_v
        fraction = max(1 - float(curr_t - start_t) / (end_t - start_t), 0)
        return fraction * start_v + (1 - fraction) * end_v

    def step(self, action_dict):
        action = [action_dict[self.curr_agents[0]], action_dict[self.curr_agents[1]]]
        assert all(self.action_space[agent].contains(action_dict[agent]) for agent in action_dict), \
            "%r (%s) invalid" % (action, type(action))
        joint_action = [Action.INDEX_TO_ACTION[a] for a in action]
        if self.use_phi:
            next_state, sparse_reward, done, info = self.base_env.step(joint_action, display_phi=True)
            potential = info["phi_s_prime"] - info["phi_s"]
            dense_reward = (potential, potential)
        else:
            next_state, sparse_reward, done, info = self.base_env.step(joint_action, display_phi=False)
            dense_reward = info["shaped_r_by_agent"]
        ob_p0, ob_p1 = self._get_obs(next_state)
        a0, a1 = self.curr_agents
        rewards = {a0: sparse_reward + self.reward_shaping_factor * dense_reward[0],
                   a1: sparse_reward + self.reward_shaping_factor * dense_reward[1]}
        return {a0: ob_p0, a1: ob_p1}, rewards, {a0: done, a1: done, "__all__": done}, {a0: info, a1: info}

    def reset(self, regen_mdp=True):
        self.base_env.reset(regen_mdp)
        self.curr_agents = self._populate_agents()
        ob_p0, ob_p1 = self._get_obs(self.base_env.state)
        return {self.curr_agents[0]: ob_p0, self.curr_agents[1]: ob_p1}

    def anneal_reward_shaping_factor(self, timesteps):
        self.set_reward_shaping_factor(self._anneal(self._initial_reward_shaping_factor, timesteps,
                                                    self.reward_shaping_horizon))

    def anneal_bc_factor(self, timesteps):
        p_0, p_1, i = self.bc_schedule[0], self.bc_schedule[1], 2
        while timesteps > p_1[0] and i < len(self.bc_schedule):
            p_0, p_1 = p_1, self.bc_schedule[i]
            i += 1
        (start_t, start_v), (end_t, end_v) = p_0, p_1
        self.set_bc_factor(self._anneal(start_v, timesteps, end_t, end_v, start_t))

    def set_reward_shaping_factor(self, factor):
        self.reward_shaping_factor = factor

    def set_bc_factor(self, factor):
        self.bc_factor = factor

    def seed(self, seed):
        pass  # the environment is deterministic (rllib.py:391-396)

    @classmethod
    def from_config(cls, env_config):
        """rllib.py:398-438 for a fixed layout (`mdp_params`); layout schedules are not supported."""
        from .mdp import OvercookedGridworld

        assert env_config and "env_params" in env_config and "multi_agent_params" in env_config
        if "mdp_params" not in env_config:
            raise NotImplementedError("mdp_params_schedule_fn is not supported; pass fixed mdp_params")
        mdp_params = dict(env_config["mdp_params"])
        mdp = OvercookedGridworld.from_layout_name(mdp_params.pop("layout_name"), **mdp_params)
        base_env = OvercookedEnv.from_mdp(mdp, **env_config["env_params"])
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
        self.venv = VecOvercookedEnv(layouts, n_envs, horizon=horizon, device=device, auto_reset=False, **venv_kwargs)
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
        if self.use_phi:  # potential of each layout's standard start state: the fresh batch holds exactly those
            self.phi_cur.copy_(v.potential(self.gamma))
            lid = v.layout_id_host if v.layout_id is not None else np.zeros(self.n_envs, np.int64)
            first = [int(np.nonzero(lid == l)[0][0]) if (lid == l).any() else 0 for l in range(len(v.table))]
            self.phi_start.copy_(self.phi_cur[torch.as_tensor(first, device=dev)])
        if self._random_starts:
            self.reset()

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
        obs = self._obs_buffer() if (self.obs_kind == "ppo" and not self._random_starts) else None
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
                       self.horizon)
        self._lib.check(rc, "oc_multi_agent_step")
        if self._random_starts:  # finished envs were restarted from the standard state: redraw them, then observe
            v.reset(mask=self.done, random_start_pos=self.random_start_pos, rnd_obj_prob_thresh=self.rnd_obj_prob_thresh)
            if self.use_phi:
                v.potential(self.gamma, out=self.phi_cur)  # == phi(s') where the episode goes on, phi(new start) elsewhere
        infos = {"sparse_r_by_agent": v.rewards[:, 0:2], "shaped_r_by_agent": v.rewards[:, 2:4], "flags": v.flags,
                 "ep_returns": self.ep_returns}  # episode totals so far; final where done (the reset cleared the live ones)
        if self.use_phi:
            infos["phi_s_prime"] = self.phi_next
        return (obs if obs is not None else self.observations()), self.shaped, self.done, infos

    def anneal_reward_shaping_factor(self, timesteps):
        self.reward_shaping_factor = OvercookedMultiAgent._anneal(self._initial_reward_shaping_factor, timesteps,
                                                                  self.reward_shaping_horizon)

    def set_reward_shaping_factor(self, factor):
        self.reward_shaping_factor = factor
