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
MAX_TIMESTEP = 65535  # the packed state's timestep is a u16 (include/oc_amd.h)
DEFAULT_TRAJ_KEYS = ["ep_states", "ep_actions", "ep_rewards", "ep_dones", "ep_infos", "ep_returns", "ep_lengths",
                     "mdp_params", "env_params", "metadatas"]  # get_rollouts' trajectory dict (env.py:27-38)


class _UnsupportedPlanner:
    """What `OvercookedEnv.mlam` returns: the reference builds a MediumLevelActionManager there (joint motion planning
    and medium-level action enumeration, planning/planners.py:453-1400), which lies outside the accelerated path.  The
    object carries the parts that ARE available — `motion_planner` (planner.MotionPlanner) and `params` — so that
    `featurize_state(state, env.mlam)`-style calls work, and names what is missing when anything else is asked of it."""

    def __init__(self, motion_planner, params):
        self.motion_planner, self.params = motion_planner, params

    def __getattr__(self, name):
        raise NotImplementedError(
            "OvercookedEnv.mlam.%s: the MediumLevelActionManager / JointMotionPlanner of the reference "
            "(overcooked_ai_py/planning/planners.py:453-1400) is not part of overcooked_ai_amd; available here: "
            ".motion_planner (single-agent MotionPlanner: get_plan, min_cost_to_feature, ...) and .params.  Agents that "
            "need medium-level actions (GreedyHumanModel, agents/agent.py:298) must be given the reference's planner." % name)


class OvercookedEnv:
    """The reference's OvercookedEnv surface (overcooked_env.py:33-405) over the HIP transition.

    One difference in range: the packed state carries the timestep as a u16, so an episode can run for at most
    65 535 steps.  The reference's default horizon (MAX_HORIZON = 1e10: "never done") is accepted, but step() raises
    ValueError once the state's timestep has reached 65 535 instead of silently freezing it there (the kernels saturate
    the stored timestep); give a horizon <= 65 535 for episodes that end."""

    _CTOR_KEYS = ("start_state_fn", "horizon", "info_level", "num_mdp")  # what env_params / copy() carry over

    def __init__(self, mdp_generator_fn, start_state_fn=None, horizon=MAX_HORIZON, mlam_params=None, info_level=0,
                 num_mdp=1, initial_info={}):
        if not callable(mdp_generator_fn):
            raise AssertionError("OvercookedEnv is built from a function that returns an OvercookedGridworld; to wrap "
                                 "an existing OvercookedGridworld use OvercookedEnv.from_mdp")
        self.mdp_generator_fn, self.mlam_params = mdp_generator_fn, mlam_params
        self.start_state_fn, self.horizon, self.info_level, self.num_mdp = start_state_fn, horizon, info_level, num_mdp
        self.variable_mdp = num_mdp > 1
        self._mp = self._mlam = None
        # step() inlines is_done / _update_game_stats / _prepare_info_dict on its fast path: a subclass that overrides one of
        # these hooks (as users of the reference env do) is stepped through the hooks instead
        cls = type(self)
        self._plain_hooks = all(getattr(cls, h) is getattr(OvercookedEnv, h)
                                for h in ("is_done", "_update_game_stats", "_prepare_info_dict", "_add_episode_info"))
        self.reset(outside_info=initial_info)

    @staticmethod
    def from_mdp(mdp, start_state_fn=None, horizon=MAX_HORIZON, mlam_params=None, info_level=1, num_mdp=None):
        assert isinstance(mdp, OvercookedGridworld) and num_mdp in (None, 1)
        return OvercookedEnv(lambda _outside_info: mdp, start_state_fn, horizon, mlam_params, info_level, 1)

    @property
    def env_params(self):
        return {k: getattr(self, k) for k in self._CTOR_KEYS}

    def copy(self):
        """A fresh env (reset to a start state) over the same generator and parameters (env.py:236-242)."""
        return OvercookedEnv(self.mdp_generator_fn, **self.env_params)

    # ---------------------------------------------------------------- planners (env.py:92-115)
    @property
    def mp(self):
        """Single-agent motion planner of the current mdp (lazily built, dropped when reset() regenerates the mdp)."""
        if self._mp is None:
            from .planner import MotionPlanner

            self._mp = MotionPlanner(self.mdp, (self.mlam_params or {}).get("counter_goals") or ())
        return self._mp

    @property
    def mlam(self):
        if self._mlam is None:
            self._mlam = _UnsupportedPlanner(self.mp, self.mlam_params)
        return self._mlam

    def __repr__(self):
        return self.mdp.state_string(self.state)

    # ---------------------------------------------------------------- stepping (API of env.py:244-325)
    def step(self, joint_action, joint_agent_action_info=None, display_phi=False):
        """One joint action -> (next_state, summed sparse reward, done, env_info); refuses to step a finished env."""
        state, mdp = self.state, self.mdp
        t = state.timestep
        if not self._plain_hooks:
            return self._step_through_hooks(joint_action, joint_agent_action_info, display_phi)
        assert t < self.horizon and not mdp.is_terminal(state)  # = not self.is_done()
        if t >= MAX_TIMESTEP:
            raise ValueError("timestep %d: the packed state counts steps in 16 bits; use a horizon <= %d" % (t, MAX_TIMESTEP))
        agent_infos = joint_agent_action_info if joint_agent_action_info is not None else [{}, {}]
        fast = None if display_phi else mdp._fast_step(state, joint_action)
        if fast is not None:
            # the single-state port: rewards and the kernel's event mask as plain values, no infos dict in between
            next_state, sparse, shaped, mask = fast
            phi_s = phi_s_prime = None
            if mask or sparse[0] or shaped[0] or (len(sparse) > 1 and (sparse[1] or shaped[1])):  # (most steps: nothing happened)
                gs = self.game_stats
                if any(sparse):
                    gs["cumulative_sparse_rewards_by_agent"] = gs["cumulative_sparse_rewards_by_agent"] + np.asarray(sparse)
                if any(shaped):
                    gs["cumulative_shaped_rewards_by_agent"] = gs["cumulative_shaped_rewards_by_agent"] + np.asarray(shaped)
                while mask:  # events are stamped with the pre-step timestep (env.py:385); bit 2 * event + agent
                    low = mask & -mask
                    k = low.bit_length() - 1
                    gs[EVENT_TYPES[k >> 1]][k & 1].append(t)
                    mask ^= low
        else:
            next_state, mdp_infos = mdp.get_state_transition(state, joint_action, display_phi)
            self._update_game_stats(mdp_infos)  # events are stamped with the pre-step timestep (env.py:385)
            sparse, shaped = mdp_infos["sparse_reward_by_agent"], mdp_infos["shaped_reward_by_agent"]
            phi_s, phi_s_prime = mdp_infos.get("phi_s"), mdp_infos.get("phi_s_prime")
        self.state = next_state
        done = next_state.timestep >= self.horizon or mdp.is_terminal(next_state)  # = self.is_done()
        env_info = {"agent_infos": list(agent_infos[:mdp.num_players]), "sparse_r_by_agent": sparse,
                    "shaped_r_by_agent": shaped, "phi_s": phi_s, "phi_s_prime": phi_s_prime}  # the keys of env.py:339-361
        if done:
            self._add_episode_info(env_info)
        return next_state, sum(sparse), done, env_info

    def _step_through_hooks(self, joint_action, joint_agent_action_info=None, display_phi=False):
        """step() in the reference's own order of hook calls (env.py:244-274), for subclasses that override them."""
        assert not self.is_done()
        if self.state.timestep >= MAX_TIMESTEP:
            raise ValueError("timestep %d: the packed state counts steps in 16 bits; use a horizon <= %d"
                             % (self.state.timestep, MAX_TIMESTEP))
        if joint_agent_action_info is None:
            joint_agent_action_info = [{}, {}]
        next_state, mdp_infos = self.mdp.get_state_transition(self.state, joint_action, display_phi)
        self._update_game_stats(mdp_infos)
        self.state = next_state
        done = self.is_done()
        env_info = self._prepare_info_dict(joint_agent_action_info, mdp_infos)
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
        """New episode (env.py:288-319): optionally a new mdp from the generator, a start state, empty game_stats."""
        if regen_mdp:
            self.mdp = self.mdp_generator_fn(outside_info)
            self._mp = self._mlam = None
        start = self.start_state_fn or self.mdp.get_standard_start_state
        self.state = start()
        n = self.mdp.num_players
        self.game_stats = {name: [[] for _ in range(n)] for name in EVENT_TYPES}
        for key in ("cumulative_sparse_rewards_by_agent", "cumulative_shaped_rewards_by_agent"):
            self.game_stats[key] = np.zeros(n, dtype=np.int64)

    def is_done(self):
        out_of_time = self.state.timestep >= self.horizon
        return out_of_time or self.mdp.is_terminal(self.state)

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
            r = infos[src]
            if any(r):  # (most steps pay nothing)
                self.game_stats[key] = self.game_stats[key] + np.asarray(r)
        mask = getattr(infos, "event_mask", None)
        if mask is not None:  # the kernel's event bit mask: visit only what happened (bit 2 * event + agent)
            while mask:
                low = mask & -mask
                k = low.bit_length() - 1
                self.game_stats[EVENT_TYPES[k >> 1]][k & 1].append(t)
                mask ^= low
            return
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
        successor_state = self.state
        self.reset(False)  # like the reference: the env is left at a start state of the same mdp
        return successor_state, done

    def run_agents(self, agent_pair, include_final_state=False, display=False, dir=None, display_phi=False,
                   display_until=np.inf):
        """One episode driven by an agent pair (anything with `joint_action(state) -> ((a0, info0), (a1, info1))`, e.g. the
        reference's AgentPair, agents/agent.py:137): -> (trajectory, timesteps, total sparse, total shaped) with
        trajectory an object array of (state, joint action, reward, done, info) rows (+ a closing row holding the final
        state when include_final_state), exactly like env.py:425-483."""
        assert self.state.timestep == 0, "Did not reset environment before running agents"
        rows, done = [], False
        while not done:
            s_t = self.state
            a_t, a_info_t = zip(*agent_pair.joint_action(s_t))
            assert all(a in Action.ALL_ACTIONS for a in a_t) and all(type(i) is dict for i in a_info_t)
            s_next, r_t, done, info = self.step(a_t, a_info_t, display_phi)
            rows.append((s_t, a_t, r_t, done, info))
            if display and self.state.timestep < display_until:
                print(self)
        assert len(rows) == self.state.timestep, "%d vs %d" % (len(rows), self.state.timestep)
        if include_final_state:
            rows.append((s_next, (None, None), 0, True, None))
        trajectory = np.empty((len(rows), 5), dtype=object)
        for i, row in enumerate(rows):
            for j, item in enumerate(row):
                trajectory[i, j] = item
        return (trajectory, self.state.timestep, sum(self.game_stats["cumulative_sparse_rewards_by_agent"]),
                sum(self.game_stats["cumulative_shaped_rewards_by_agent"]))

    def get_rollouts(self, agent_pair, num_games, display=False, dir=None, final_state=False, display_phi=False,
                     display_until=np.inf, metadata_fn=None, metadata_info_fn=None, info=True):
        """`num_games` episodes of `agent_pair` in the codebase's standard trajectories format (env.py:485-580): a dict of
        per-episode lists / arrays under DEFAULT_TRAJ_KEYS."""
        out = {k: [] for k in DEFAULT_TRAJ_KEYS}
        for _ in range(num_games):
            if hasattr(agent_pair, "set_mdp"):
                agent_pair.set_mdp(self.mdp)
            rollout = self.run_agents(agent_pair, include_final_state=final_state, display=display, dir=dir,
                                      display_phi=display_phi, display_until=display_until)
            traj, length, sparse_total, _ = rollout
            for key, col in zip(DEFAULT_TRAJ_KEYS[:5], traj.T):
                out[key].append(col)
            out["ep_returns"].append(sparse_total)
            out["ep_lengths"].append(length)
            out["mdp_params"].append(self.mdp.mdp_params)
            out["env_params"].append(self.env_params)
            out["metadatas"].append(metadata_fn(rollout) if metadata_fn else {})
            self.reset(regen_mdp=False)
            if hasattr(agent_pair, "reset"):
                agent_pair.reset()
        # every key becomes an ndarray, mdp_params / env_params (object arrays of dicts) included: env.py:574
        out = {k: np.array(v, dtype=object) if k in DEFAULT_TRAJ_KEYS[:5] else v if k == "metadatas" else np.array(v)
               for k, v in out.items()}
        if out["metadatas"]:  # list of dicts -> dict of lists, like the reference's merge
            keys = out["metadatas"][0].keys()
            out["metadatas"] = {k: [m[k] for m in out["metadatas"]] for k in keys}
        else:
            out["metadatas"] = {}
        return out


class _Discrete:
    """Minimal stand-in for gymnasium.spaces.Discrete when gymnasium is not installed."""

    def __init__(self, n):
        self.n = n

    def contains(self, a):
        return 0 <= int(a) < self.n


class _Box:
    """Minimal stand-in for gymnasium.spaces.Box."""

    def __init__(self, low, high, dtype=None):
        self.low, self.high, self.dtype, self.shape = low, high, dtype, low.shape


try:  # the reference derives from gymnasium.Env and registers "Overcooked-v0" (overcooked_ai_py/__init__.py:1-6)
    import gymnasium as _gym
    from gymnasium import spaces as _spaces

    _EnvBase, _mk_discrete = _gym.Env, _spaces.Discrete
    _mk_box = lambda low, high, dtype: _spaces.Box(low, high, dtype=dtype)  # noqa: E731
except ImportError:  # no gymnasium in this image: same surface, duck-typed spaces
    _gym, _EnvBase, _mk_discrete, _mk_box = None, object, _Discrete, _Box


class Overcooked(_EnvBase):
    """Gym-style view of one OvercookedEnv for a single learning agent (API of env.py:782-909): the old 4-tuple
    `step`, an observation dict carrying both agents' featurizations, and a seat (`agent_idx`) redrawn every reset.

    Everything here is seat bookkeeping: the learning agent always speaks first in `step((mine, partner))` and reads
    its own observation first in `both_agent_obs`, whichever player index it currently occupies."""

    env_name = "Overcooked-v0"

    def __init__(self, base_env, featurize_fn, baselines_reproducible=False):
        if baselines_reproducible:
            np.random.seed(0)  # fixed seat sequence, as the reference's flag of the same name does
        self.base_env, self.featurize_fn = base_env, featurize_fn
        probe = featurize_fn(base_env.mdp.get_standard_start_state())[0]
        self.observation_space = _mk_box(np.zeros(probe.shape, np.float32), np.full(probe.shape, np.inf, np.float32),
                                         np.float32)
        self.action_space = _mk_discrete(Action.NUM_ACTIONS)
        self.reset()

    def _seated(self, pair):
        """(player 0's item, player 1's item) -> (learning agent's, partner's)."""
        return tuple(pair) if self.agent_idx == 0 else (pair[1], pair[0])

    def _observe(self, state):
        return {"both_agent_obs": self._seated(self.featurize_fn(state)), "overcooked_state": state,
                "other_agent_env_idx": 1 - self.agent_idx}

    def step(self, action):
        mine, partner = action
        if not (self.action_space.contains(mine) and self.action_space.contains(partner)):
            raise AssertionError("%r (%s) invalid" % (action, type(action)))
        by_player = self._seated((Action.INDEX_TO_ACTION[mine], Action.INDEX_TO_ACTION[partner]))  # a swap is its own inverse
        state, reward, done, info = self.base_env.step(by_player)
        info["policy_agent_idx"] = self.agent_idx
        if done and "episode" in info:
            info["episode"]["policy_agent_idx"] = self.agent_idx
        return self._observe(state), reward, done, info

    def reset(self, **_gym_kwargs):
        self.base_env.reset()
        self.mdp = self.base_env.mdp
        self.agent_idx = np.random.choice([0, 1])
        return self._observe(self.base_env.state)

    def render(self, mode="human", close=False):
        raise NotImplementedError("rendering is outside the accelerated path (SURVEY §2: visualisation is out of scope)")


if _gym is not None:
    try:
        _gym.envs.registration.register(id="Overcooked-v0", entry_point="overcooked_ai_amd.env:Overcooked")
    except Exception:  # already registered (e.g. the reference package is importable too)
        pass
