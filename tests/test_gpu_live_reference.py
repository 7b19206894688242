"""The HIP path against the LIVE reference on the GPU box.

/root/reference cannot travel, but oracle/build_ref.py byte-compiles the reference's own hot-path modules
(overcooked_mdp.py, overcooked_env.py, actions.py ...) into oracle/_ref/src in the build container, and that build
output travels with the snapshot like liboc_amd.so.  When it is there, these tests run the reference's
`OvercookedEnv.step` / `lossless_state_encoding` and the mirror classes (HIP kernels underneath) side by side on the
same random joint actions and compare every state, reward, info and observation.  Skipped when the byte code is absent
or was written by another CPython (the committed fixtures under tests/golden/ pin the same behaviour either way)."""
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF_SRC = os.path.join(ROOT, "oracle", "_ref", "src")

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def R():
    if not os.path.exists(os.path.join(REF_SRC, "overcooked_ai_py", "mdp", "overcooked_env.pyc")):
        pytest.skip("oracle/_ref/src not built (python -m oracle.build_ref in the build container)")
    os.environ["OVERCOOKED_REFERENCE_SRC"] = REF_SRC
    sys.path.insert(0, ROOT)
    import importlib

    from oracle import ref_harness

    importlib.reload(ref_harness)
    try:
        return ref_harness.load()
    except ImportError as exc:  # bad magic number: byte code of another CPython
        pytest.skip("byte-compiled reference does not load here: %r" % (exc,))


def _ref_mdp(R, name):
    from overcooked_ai_amd import layouts as L

    d = dict(L.read_layout_dict(name))
    d.pop("grid")
    mdp = R.OvercookedGridworld.from_grid(L.spec_from_name(name).grid_rows(), base_layout_params=d)
    R.Recipe.configure(mdp.recipe_config)  # process-global in the reference (mdp.py:221-336)
    return mdp


@pytest.mark.parametrize("name", ["cramped_room", "asymmetric_advantages", "counter_circuit", "forced_coordination"])
def test_env_step_and_encoding_against_the_live_reference(name, R):
    from overcooked_ai_amd import Action, OvercookedEnv, OvercookedGridworld
    from overcooked_ai_amd import state as S

    ref_mdp = _ref_mdp(R, name)
    ref_env = R.OvercookedEnv.from_mdp(ref_mdp, horizon=120, info_level=0)
    mdp = OvercookedGridworld.from_layout_name(name)
    env = OvercookedEnv.from_mdp(mdp, horizon=120, info_level=0)
    rng = np.random.RandomState(7)
    for episode in range(2):
        ref_env.reset(regen_mdp=False)
        ref_env._mp = object()  # never compute / pickle a MotionPlanner (overcooked_env.py:102-115)
        env.reset(regen_mdp=False)
        done, t = False, 0
        while not done:
            # biased towards INTERACT so that soups get cooked and served within 120 steps
            a = [int(x) if rng.rand() > 0.25 else 5 for x in rng.randint(0, 6, 2)]
            ja_ref = tuple(R.Action.INDEX_TO_ACTION[i] for i in a)
            ja = tuple(Action.INDEX_TO_ACTION[i] for i in a)
            s_ref, r_ref, done_ref, info_ref = ref_env.step(ja_ref)
            s, r, done, info = env.step(ja)
            assert S.canonical_state_dict(s) == S.canonical_state_dict(s_ref.to_dict()), (name, episode, t)
            assert r == r_ref and done == done_ref, (name, episode, t)
            assert list(info["sparse_r_by_agent"]) == list(info_ref["sparse_r_by_agent"])
            assert list(info["shaped_r_by_agent"]) == list(info_ref["shaped_r_by_agent"])
            if t % 7 == 0 or done:
                enc_ref = np.stack(ref_env.lossless_state_encoding_mdp(s_ref))
                enc = np.stack(env.lossless_state_encoding_mdp(s))
                assert enc.shape == enc_ref.shape and np.array_equal(enc, enc_ref), (name, episode, t)
            t += 1
        assert info["episode"]["ep_sparse_r"] == info_ref["episode"]["ep_sparse_r"]
        assert info["episode"]["ep_shaped_r"] == info_ref["episode"]["ep_shaped_r"]
        for k, v in info_ref["episode"]["ep_game_stats"].items():
            got = info["episode"]["ep_game_stats"][k]
            assert [list(x) if not np.isscalar(x) else x for x in np.asarray(got, dtype=object)] == \
                [list(x) if not np.isscalar(x) else x for x in np.asarray(v, dtype=object)], (name, k)


def test_batched_transitions_against_the_live_reference(R):
    """4 000 random (state, joint action) pairs through VecOvercookedEnv.step vs the reference's get_state_transition."""
    import torch

    from overcooked_ai_amd import VecOvercookedEnv
    from overcooked_ai_amd import state as S
    from overcooked_ai_amd.layouts import spec_from_name

    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from helpers import random_packed_states

    name, n = "asymmetric_advantages", 4000
    spec = spec_from_name(name)
    ref_mdp = _ref_mdp(R, name)
    rng = np.random.default_rng(3)
    st = random_packed_states(spec, n, rng)
    acts = rng.integers(0, 6, size=(n, 2)).astype(np.uint8)
    env = VecOvercookedEnv(spec, n, horizon=65535, device="cuda:0")
    env.set_packed_state(st)
    rew, _ = env.step(torch.from_numpy(acts).to("cuda:0"))
    rew = rew.cpu().numpy()
    got = env.get_states(as_dict=True)
    states = S.unpack_states(spec, st, as_dict=True)
    for e in range(n):
        s_ref = R.OvercookedState.from_dict(states[e])
        ja = tuple(R.Action.INDEX_TO_ACTION[i] for i in acts[e])
        nxt, infos = ref_mdp.get_state_transition(s_ref, ja)
        assert S.canonical_state_dict(got[e]) == S.canonical_state_dict(nxt.to_dict()), e
        assert list(rew[e, 0:2]) == list(infos["sparse_reward_by_agent"]) and list(rew[e, 2:4]) == list(infos["shaped_reward_by_agent"]), e
