"""Import the upstream Python reference (read-only at /root/reference) as a live oracle.

TEST INFRASTRUCTURE ONLY.  This module exists solely to (a) generate the golden
fixtures committed under tests/golden/ and (b) validate the C restatement in
oracle/overcooked_oracle.c inside the build container.  /root/reference does not
exist on the GPU box, so nothing at run time may depend on this file; it is never
imported by the product package `overcooked_ai_amd`.

The stub set follows SURVEY.md Appendix A: the reference's package __init__ imports
gymnasium (absent here) and overcooked_env imports cv2/pygame and a visualizer that
loads sprites at import time.  None of those are touched by the hot path
(OvercookedGridworld.get_state_transition, mdp.py:1375; lossless_state_encoding,
mdp.py:2385).
"""
import os
import sys
import types

REFERENCE_SRC = os.environ.get("OVERCOOKED_REFERENCE_SRC", "/root/reference/src")


def available():
    return os.path.isdir(os.path.join(REFERENCE_SRC, "overcooked_ai_py"))


def load():
    """Returns a namespace with the reference classes. Never writes into /root/reference."""
    if not available():
        raise RuntimeError("reference sources not present at %s" % REFERENCE_SRC)
    sys.dont_write_bytecode = True

    def stub(name, **attrs):
        m = types.ModuleType(name)
        m.__dict__.update(attrs)
        sys.modules[name] = m
        return m

    if "overcooked_ai_py" not in sys.modules:
        g = stub("gymnasium")
        ge = stub("gymnasium.envs")
        gr = stub("gymnasium.envs.registration", register=lambda **k: None)
        g.envs, ge.registration = ge, gr
        g.Env = type("Env", (), {})

        class Discrete:
            def __init__(self, n):
                self.n = n

            def contains(self, a):
                return 0 <= int(a) < self.n

        class Box:
            def __init__(self, low, high, dtype=None):
                self.low, self.high, self.dtype, self.shape = low, high, dtype, low.shape

        g.spaces = stub("gymnasium.spaces", Discrete=Discrete, Box=Box)
        stub("cv2")
        stub("pygame")
        stub(
            "overcooked_ai_py.visualization.state_visualizer",
            StateVisualizer=type("StateVisualizer", (), {}),
        )
        sys.path.insert(0, REFERENCE_SRC)

    from overcooked_ai_py.mdp import overcooked_mdp as m
    from overcooked_ai_py.mdp import overcooked_env as e
    from overcooked_ai_py.mdp.actions import Action, Direction

    ns = types.SimpleNamespace(
        mdp_module=m,
        env_module=e,
        OvercookedGridworld=m.OvercookedGridworld,
        OvercookedState=m.OvercookedState,
        PlayerState=m.PlayerState,
        ObjectState=m.ObjectState,
        SoupState=m.SoupState,
        Recipe=m.Recipe,
        OvercookedEnv=e.OvercookedEnv,
        Overcooked=e.Overcooked,
        Action=Action,
        Direction=Direction,
        EVENT_TYPES=m.EVENT_TYPES,
    )
    return ns


def load_rllib():
    """The reference's RLlib environment class (human_aware_rl/rllib/rllib.py) with `ray` and `gym` stubbed out —
    only OvercookedMultiAgent (a plain class once MultiAgentEnv is `object`) is used, for fixture generation."""
    import numpy as np

    load()

    def stub(name, **attrs):
        m = types.ModuleType(name)
        m.__dict__.update(attrs)
        sys.modules[name] = m
        return m

    if "human_aware_rl.rllib.rllib" not in sys.modules:
        class Discrete:
            def __init__(self, n):
                self.n = n

            def contains(self, a):
                return isinstance(a, (int, np.integer)) and 0 <= int(a) < self.n

        class Box:
            def __init__(self, low, high, dtype=None):
                self.low, self.high, self.dtype, self.shape = low, high, dtype, low.shape

        class Dict(dict):
            def __init__(self, d):
                super().__init__(d)

        gym = stub("gym")
        gym.spaces = stub("gym.spaces", Discrete=Discrete, Box=Box, Dict=Dict)
        stub("ray")
        for name, attrs in (("ray.rllib", {}), ("ray.rllib.agents", {}), ("ray.rllib.agents.ppo", {"PPOTrainer": object}),
                            ("ray.rllib.algorithms", {}), ("ray.rllib.algorithms.callbacks", {"DefaultCallbacks": object}),
                            ("ray.rllib.env", {}), ("ray.rllib.env.multi_agent_env", {"MultiAgentEnv": object}),
                            ("ray.rllib.models", {"ModelCatalog": object}), ("ray.tune", {}),
                            ("ray.tune.logger", {"UnifiedLogger": object}),
                            ("ray.tune.registry", {"register_env": lambda *a, **k: None}),
                            ("ray.tune.result", {"DEFAULT_RESULTS_DIR": "/tmp"})):
            stub(name, **attrs)
    from human_aware_rl.rllib import rllib

    return rllib
