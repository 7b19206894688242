"""OvercookedEnv and the gym-style Overcooked wrapper — drop-in mirrors of overcooked_env.py:33-325, 782-909.

One env per object, like the reference; every transition goes through the HIP kernels via `OvercookedGridworld`.
The bookkeeping (`game_stats`, episode info) is restated on the host exactly as env.py:308-401 does it.
For throughput, use `VecOvercookedEnv` — this class exists so that agents and evaluation code written against the
reference's `OvercookedEnv.step()/reset()` surface run unchanged.
"""
import numpy as np

from .actions import Action
from .mdp import EVENT_TYPES, OvercookedGridworld

DEFAULT_ENV_PARAMS = {"horizon": 400}
MAX_HORIZON = 1e10


class OvercookedEnv:
    def __init__(self, mdp_generator_fn, start_state_fn=None, horizon=MAX_HORIZON, mlam_params=None, info_level=0,
                 num_mdp=1, initial_info={}):
        assert callable(mdp_generator_fn), (
            "OvercookedEnv takes in a OvercookedGridworld generator function. If trying to instantiate directly "
            "from a OvercookedGridworld instance, use the OvercookedEnv.from_mdp method")
        self.num_mdp = num_mdp
        self.variable_mdp = num_mdp > 1
        self.mdp_generator_fn = mdp_generator_fn
        self.horizon = horizon
        self.mlam_params = mlam_params
        self.start_state_fn = start_state_fn
        self.info_level = info_level
        self.reset(outside_info=initial_info)

    @staticmethod
    def from_mdp(mdp, start_state_fn=None, horizon=MAX_HORIZON, mlam_params=None, info_level=1, num_mdp=None):
        assert isinstance(mdp, OvercookedGridworld)
        if num_mdp is not None:
            assert num_mdp == 1
        return OvercookedEnv(mdp_generator_fn=lambda _ignored: mdp, start_state_fn=start_state_fn, horizon=horizon,
                             mlam_params=mlam_params, info_level=info_level, num_mdp=1)

    @property
    def env_params(self):
        return {"start_state_fn": self.start_state_fn, "horizon": self.horizon, "info_level": self.info_level,
                "num_mdp": self.num_mdp}

    def copy(self):
        return OvercookedEnv(mdp_generator_fn=self.mdp_generator_fn, start_state_fn=self.start_state_fn,
                             horizon=self.horizon, info_level=self.info_level, num_mdp=self.num_mdp)

    # ---------------------------------------------------------------- stepping (API of env.py:244-325)
    def step(self, joint_action, joint_agent_action_info=None, display_phi=False):
        """One joint action -> (next_state, summed sparse reward, done, env_info); refuses to step a finished env."""
        assert not self.is_done()
        agent_infos = joint_agent_action_info if joint_agent_action_info is not None else [{}, {}]
        next_state, mdp_infos = self.mdp.get_state_transition(self.state, joint_action, display_phi)
        self._update_game_stats(mdp_infos)  # events are stamped with the pre-step timestep (env.py:385)
        self.state = next_state
        done = self.is_done()
        env_info = self._prepare_info_dict(agent_infos, mdp_infos)
        if done:
            self._add_episode_info(env_info)
        return next_state, sum(mdp_infos["sparse_reward_by_agent"]), done, env_info

    def lossless_state_encoding_mdp(self, state):
        return self.mdp.lossless_state_encoding(state, self.horizon)

    def featurize_state_mdp(self, state, num_pots=2):
        """env.py:282-286; the reference passes env.mlam built from mlam_params (default: no counter goals)."""
        cg = (self.mlam_params or {}).get("counter_goals") or "none"
        return self.mdp.featurize_state(state, None, num_pots=num_pots, counter_goals=cg)

    def reset(self, regen_mdp=True, outside_info={}):
        if regen_mdp:
            self.mdp = self.mdp_generator_fn(outside_info)
        self.state = self.start_state_fn() if self.start_state_fn is not None else self.mdp.get_standard_start_state()
        n = self.mdp.num_players
        self.game_stats = {name: [[] for _ in range(n)] for name in EVENT_TYPES}
        for key in ("cumulative_sparse_rewards_by_agent", "cumulative_shaped_rewards_by_agent"):
            self.game_stats[key] = np.zeros(n, dtype=np.int64)

    def is_done(self):
        return self.state.timestep >= self.horizon or self.mdp.is_terminal(self.state)

    def _prepare_info_dict(self, joint_agent_action_info, mdp_infos):
        """Per-step info: the keys of env.py:339-361."""
        n = self.mdp.num_players
        return {
            "agent_infos": list(joint_agent_action_info[:n]),
            "sparse_r_by_agent": mdp_infos["sparse_reward_by_agent"],
            "shaped_r_by_agent": mdp_infos["shaped_reward_by_agent"],
            "phi_s": mdp_infos.get("phi_s"),
            "phi_s_prime": mdp_infos.get("phi_s_prime"),
        }

    def _add_episode_info(self, env_info):
        """Episode summary attached to the last step's info: the keys of env.py:363-380."""
        sparse = self.game_stats["cumulative_sparse_rewards_by_agent"]
        shaped = self.game_stats["cumulative_shaped_rewards_by_agent"]
        env_info["episode"] = dict(ep_game_stats=self.game_stats, ep_sparse_r=sum(sparse), ep_shaped_r=sum(shaped),
                                   ep_sparse_r_by_agent=sparse, ep_shaped_r_by_agent=shaped,
                                   ep_length=self.state.timestep)
        return env_info

    def _update_game_stats(self, infos):
        t = self.state.timestep
        for key, src in (("cumulative_sparse_rewards_by_agent", "sparse_reward_by_agent"),
                         ("cumulative_shaped_rewards_by_agent", "shaped_reward_by_agent")):
            self.game_stats[key] = self.game_stats[key] + np.asarray(infos[src])
        for name, flags in infos["event_infos"].items():
            for agent, happened in enumerate(flags):
                if happened:
                    self.game_stats[name][agent].append(t)

    def execute_plan(self, start_state, joint_action_plan, display=False):
        """Run a list of joint actions from start_state; returns (end_state, done) like env.py:407-424."""
        self.state = start_state
        done = False
        for joint_action in joint_action_plan:
            self.step(joint_action)
            done = self.is_done()
            if done:
                break
        return self.state, done


class _Discrete:
    def __init__(self, n):
        self.n = n

    def contains(self, a):
        return 0 <= int(a) < self.n


class _Box:
    def __init__(self, low, high, dtype=None):
        self.low, self.high, self.dtype, self.shape = low, high, dtype, low.shape


class Overcooked:
    """Gym-style single-agent view (env.py:782-909): old 4-tuple step API, obs dict with both agents'
    observations, the main agent's index drawn at every reset."""

    env_name = "Overcooked-v0"

    def __init__(self, base_env, featurize_fn, baselines_reproducible=False):
        if baselines_reproducible:
            np.random.seed(0)
        self.base_env = base_env
        self.featurize_fn = featurize_fn
        self.observation_space = self._setup_observation_space()
        self.action_space = _Discrete(len(Action.ALL_ACTIONS))
        self.reset()

    def _setup_observation_space(self):
        dummy_state = self.base_env.mdp.get_standard_start_state()
        obs_shape = self.featurize_fn(dummy_state)[0].shape
        high = np.ones(obs_shape, dtype=np.float32) * float("inf")
        low = np.zeros(obs_shape, dtype=np.float32)
        return _Box(low, high, dtype=np.float32)

    def step(self, action):
        assert all(self.action_space.contains(a) for a in action), "%r (%s) invalid" % (action, type(action))
        agent_action, other_agent_action = [Action.INDEX_TO_ACTION[a] for a in action]
        joint_action = (agent_action, other_agent_action) if self.agent_idx == 0 else (other_agent_action, agent_action)
        next_state, reward, done, env_info = self.base_env.step(joint_action)
        ob_p0, ob_p1 = self.featurize_fn(next_state)
        both_agents_ob = (ob_p0, ob_p1) if self.agent_idx == 0 else (ob_p1, ob_p0)
        env_info["policy_agent_idx"] = self.agent_idx
        if "episode" in env_info.keys():
            env_info["episode"]["policy_agent_idx"] = self.agent_idx
        obs = {"both_agent_obs": both_agents_ob, "overcooked_state": next_state,
               "other_agent_env_idx": 1 - self.agent_idx}
        return obs, reward, done, env_info

    def reset(self):
        self.base_env.reset()
        self.mdp = self.base_env.mdp
        self.agent_idx = np.random.choice([0, 1])
        ob_p0, ob_p1 = self.featurize_fn(self.base_env.state)
        both_agents_ob = (ob_p0, ob_p1) if self.agent_idx == 0 else (ob_p1, ob_p0)
        return {"both_agent_obs": both_agents_ob, "overcooked_state": self.base_env.state,
                "other_agent_env_idx": 1 - self.agent_idx}
