"""overcooked_ai_amd — MI355X-native batched Overcooked simulator (hot path of HumanCompatibleAI/overcooked_ai).

    from overcooked_ai_amd import VecOvercookedEnv            # N envs in HBM, HIP kernels (the fast path)
    from overcooked_ai_amd import ShardedVecOvercookedEnv     # the same batch partitioned over the GPUs of a node
    from overcooked_ai_amd import OvercookedGridworld, OvercookedEnv, Overcooked   # reference-shaped API

Importing the package does not touch the GPU; the HIP library (liboc_amd.so) is loaded on first use and there is
no CPU fallback.
"""
from .actions import Action, Direction  # noqa: F401
from .state import ObjectState, OvercookedState, PlayerState, SoupState  # noqa: F401


def __getattr__(name):  # lazy: these import torch
    if name == "VecOvercookedEnv":
        from .vec_env import VecOvercookedEnv
        return VecOvercookedEnv
    if name in ("ShardedVecOvercookedEnv", "ShardedVecOvercookedMultiAgent"):
        from . import sharded_env
        return getattr(sharded_env, name)
    if name in ("OvercookedGridworld", "EVENT_TYPES"):
        from . import mdp
        return getattr(mdp, name)
    if name in ("OvercookedEnv", "Overcooked", "DEFAULT_ENV_PARAMS", "MAX_HORIZON"):
        from . import env
        return getattr(env, name)
    if name in ("VecOvercookedMultiAgent", "OvercookedMultiAgent"):
        from . import multi_agent
        return getattr(multi_agent, name)
    raise AttributeError(name)
