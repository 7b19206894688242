"""oc_mailbox_* — the resident single-env kernel (include/oc_amd.h): bit-exact against the C oracle on random states of
several layouts (events and illegal actions included), across the kernel's idle exits and relaunches, and underneath the
drop-in OvercookedEnv."""
import ctypes
import time

import numpy as np
import pytest

from helpers import random_packed_states

pytestmark = pytest.mark.gpu

torch = pytest.importorskip("torch")


@pytest.fixture(scope="module")
def gpu():
    if not torch.cuda.is_available():
        pytest.fail("gpu-marked test run without a GPU")
    from overcooked_ai_amd import _lib

    _lib.load()
    return torch.device("cuda:0")


class Mailbox:
    def __init__(self, spec, horizon, gpu):
        from overcooked_ai_amd import _lib
        from overcooked_ai_amd.vec_env import VecOvercookedEnv

        self.lib = _lib.load()
        self.env = VecOvercookedEnv(spec, 1, device=gpu)  # owns the device layout record and the OcBatch
        self.mb = ctypes.c_void_p()
        _lib.check(self.lib.oc_mailbox_open(self.env._bref, horizon, ctypes.byref(self.mb)), "oc_mailbox_open")
        buf = (ctypes.c_uint8 * _lib.MB_BYTES).from_address(self.lib.oc_mailbox_buffer(self.mb))
        self.np = np.frombuffer(buf, dtype=np.uint8)
        self.n_state = self.env.n_planes * 16
        self.L = _lib

    def step(self, planes, a0, a1):
        L = self.L
        self.np[L.MB_STATE_IN:L.MB_STATE_IN + self.n_state] = planes.reshape(-1)
        self.np[L.MB_ACTIONS], self.np[L.MB_ACTIONS + 1] = a0, a1
        L.check(self.lib.oc_mailbox_step(self.mb), "oc_mailbox_step")
        return (self.np[L.MB_STATE_OUT:L.MB_STATE_OUT + self.n_state].copy(),
                self.np[L.MB_REWARDS:L.MB_REWARDS + 16].view(np.float32).copy(),
                int(self.np[L.MB_FLAGS:L.MB_FLAGS + 4].view(np.uint32)[0]),
                int(self.np[L.MB_EVENTS:L.MB_EVENTS + 8].view(np.uint64)[0]))

    def close(self):
        self.lib.oc_mailbox_close(self.mb)


@pytest.mark.parametrize("name", ["cramped_room", "asymmetric_advantages", "counter_circuit", "mdp_test",
                                  "cramped_room_single", "bonus_order_test"])
def test_mailbox_steps_equal_the_oracle(name, gpu):
    from oracle import oracle as O
    from overcooked_ai_amd.layouts import spec_from_name

    spec = spec_from_name(name)
    orc = O.Oracle([O.mdp_from_layout_dict(spec.to_layout_dict())])
    n, horizon = 1500, 200
    rng = np.random.default_rng(17)
    st = random_packed_states(spec, n, rng, timestep_max=210)
    acts = rng.integers(0, 6, size=(n, 2)).astype(np.uint8)
    if spec.num_players == 1:
        acts[:, 1] = 4
    acts[rng.random(n) < 0.02, 0] = 77  # illegal: the state comes back untouched, BAD_ACTION
    out_o, rew_o, fl_o = orc.step(st, acts, horizon=horizon, options=0)
    ev_o = orc.last_events
    mb = Mailbox(spec, horizon, gpu)
    try:
        for e in range(n):
            if e in (500, 1000):
                time.sleep(0.02)  # longer than the kernel's idle limit: it has left, the next step relaunches it
            s, r, fl, ev = mb.step(np.ascontiguousarray(st[:, e]), int(acts[e, 0]), int(acts[e, 1]))
            assert np.array_equal(s.reshape(-1, 16), out_o[:, e]), (name, e)
            assert np.array_equal(r, rew_o[e]) and fl == int(fl_o[e]), (name, e, r, rew_o[e], fl, fl_o[e])
            assert ev == (0 if fl & 2 else int(ev_o[e])), (name, e)
        assert (fl_o & 1).any() and (fl_o & 2).any() and ev_o.any()
    finally:
        mb.close()


def test_mailbox_rejects_what_it_does_not_serve(gpu):
    from overcooked_ai_amd import _lib
    from overcooked_ai_amd.layouts import LayoutTable, spec_from_name
    from overcooked_ai_amd.vec_env import VecOvercookedEnv

    lib = _lib.load()
    mb = ctypes.c_void_p()
    table = LayoutTable([spec_from_name("cramped_room"), spec_from_name("cramped_room_tomato")])
    env = VecOvercookedEnv(table, 1, device=gpu, layout_id=np.zeros((1,), np.uint16))
    assert lib.oc_mailbox_open(env._bref, 400, ctypes.byref(mb)) != 0 and b"one layout" in lib.oc_last_error()
    big = VecOvercookedEnv("corridor", 1, device=gpu)  # 14 x 9 cells: 8 object planes
    assert lib.oc_mailbox_open(big._bref, 400, ctypes.byref(mb)) != 0
    assert lib.oc_mailbox_open(env._bref, 400, None) != 0


def test_drop_in_env_runs_on_the_mailbox_and_matches_the_launch_path(gpu, monkeypatch):
    """OvercookedEnv.step through the mailbox == through one oc_step launch per call (OC_AMD_NO_MAILBOX), whole episodes."""
    from overcooked_ai_amd import Action, OvercookedEnv, OvercookedGridworld
    from overcooked_ai_amd import state as S

    rng = np.random.RandomState(5)
    plan = rng.randint(0, 6, (150, 2))
    runs = []
    for no_mailbox in (False, True):
        if no_mailbox:
            monkeypatch.setenv("OC_AMD_NO_MAILBOX", "1")
        mdp = OvercookedGridworld.from_layout_name("asymmetric_advantages")
        env = OvercookedEnv.from_mdp(mdp, horizon=150, info_level=0)
        assert mdp._port().mailbox is None  # opened lazily: the first calls are plain launches
        trace = []
        for k, (a0, a1) in enumerate(plan):
            s, r, done, info = env.step((Action.INDEX_TO_ACTION[a0], Action.INDEX_TO_ACTION[a1]))
            trace.append((S.canonical_state_dict(s), r, done, list(info["shaped_r_by_agent"])))
            if k == 5:
                assert (mdp._port().mailbox is None) == no_mailbox  # the resident kernel is up by the third call
            if k == 60 and not no_mailbox:  # an explicit close in the middle of an episode: launches again, then a new mailbox
                mdp.close()
                assert mdp._port().mailbox is None
            if k == 70:
                assert (mdp._port().mailbox is None) == no_mailbox
        trace.append(info["episode"]["ep_game_stats"])
        runs.append(trace)
    assert runs[0][:-1] == runs[1][:-1]
    for k, v in runs[0][-1].items():
        assert [list(x) if not np.isscalar(x) else x for x in np.asarray(v, dtype=object)] == \
            [list(x) if not np.isscalar(x) else x for x in np.asarray(runs[1][-1][k], dtype=object)], k
