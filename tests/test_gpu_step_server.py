"""The resident batched step (oc_step_server_*, csrc/step_server.hpp): k_step_server + its device-side client against
oc_step_many / oc_step and the oracle — OvercookedEnv.step (overcooked_env.py:244-274) without a launch per step."""
import numpy as np
import pytest
import torch

from test_gpu_parity import CANONICAL_5, make_env, oracle_for

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def gpu():
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    return torch.device("cuda:0")


def test_step_server_equals_single_steps_with_illegal_actions_and_resets(gpu):
    """K steps through the resident kernel == K oc_step launches == oc_step_many == the oracle, across episode ends (auto-reset)
    and with illegal actions sprinkled in, on a single layout, a mixed table, a 7-pot layout and ragged batches; the states come
    back at sync() and the server resumes where it stopped."""
    from overcooked_ai_amd.layouts import LayoutSpec, LayoutTable, spec_from_name

    seven = LayoutSpec({"grid": "XPPPPPX\nO 1 2 O\nX     X\nXDPSPTX", "onion_time": 3, "tomato_time": 5,
                        "onion_value": 7, "tomato_value": 4})
    table5 = LayoutTable([spec_from_name(nm) for nm in CANONICAL_5], pad_to=(9, 5))
    rng = np.random.default_rng(12)
    for layouts, n_lay, n in (("cramped_room", 0, 2500), (table5, 5, 2560), (seven, 0, 700), ("asymmetric_advantages", 0, 65)):
        K, horizon = 61, 17
        lid = (np.arange(n) % n_lay).astype(np.uint16) if n_lay else None
        a = rng.integers(0, 6, size=(K, n, 2)).astype(np.uint8)
        a[rng.integers(0, K, 40), rng.integers(0, n, 40), rng.integers(0, 2, 40)] = 7
        acts = torch.from_numpy(a).to(gpu)
        res = make_env(layouts, n, gpu, horizon=horizon, auto_reset=True, layout_id=lid)
        one = make_env(layouts, n, gpu, horizon=horizon, auto_reset=True, layout_id=lid)
        rew = torch.zeros((K, n, 4), dtype=torch.float32, device=gpu)
        fl = torch.zeros((K, n), dtype=torch.uint8, device=gpu)
        specs = layouts.specs if n_lay else [layouts if not isinstance(layouts, str) else spec_from_name(layouts)]
        orc = oracle_for(specs)
        st = orc.reset(orc.new_state(n), layout_id=lid)
        ep = np.zeros((n, 4), np.float32)
        with res.step_server(idle_ms=5.0, life_s=5.0) as sv:
            half = 30
            sv.play(acts[:half], rew[:half], fl[:half])  # one launch of the client, 30 round trips
            sv.sync()                                    # the states come back ...
            for k in range(half):
                r1, f1 = one.step(acts[k])
                assert torch.equal(r1, rew[k]) and torch.equal(f1, fl[k]), k
            assert torch.equal(res.state, one.state) and torch.equal(res.ep_returns, one.ep_returns)
            assert sv.steps == half
            for k in range(half, K):                     # ... and the server resumes: one step per call from here
                r2, f2 = sv.step(acts[k])
                rew[k], fl[k] = r2, f2
                r1, f1 = one.step(acts[k])
                assert torch.equal(r1, r2) and torch.equal(f1, f2), k
        for k in range(K):
            st, r_o, f_o = orc.step(st, a[k], horizon=horizon, options=1, layout_id=lid, ep_returns=ep)
            assert np.array_equal(r_o, rew[k].cpu().numpy()) and np.array_equal(f_o, fl[k].cpu().numpy()), k
        assert torch.equal(res.state, one.state) and torch.equal(res.ep_returns, one.ep_returns)
        assert np.array_equal(res.get_packed_state(), st) and np.array_equal(res.ep_returns.cpu().numpy(), ep)
        assert (fl & 2).any() and (fl & 4).any()


def test_step_server_drawn_starts_and_idle_exit(gpu):
    """Restarts inside the resident kernel draw their start states (OcStartSpec) with the epochs consecutive oc_step calls use;
    a server that has left for idleness (its states written back) is relaunched by the next play."""
    import time

    n, horizon, K = 3000, 9, 40
    kw = dict(random_start_pos=True, rnd_obj_prob_thresh=0.35, seed=11, env_offset=1000)
    res = make_env("asymmetric_advantages", n, gpu, horizon=horizon, auto_reset=True, **kw)
    one = make_env("asymmetric_advantages", n, gpu, horizon=horizon, auto_reset=True, **kw)
    assert torch.equal(res.state, one.state)
    acts = torch.randint(0, 6, (K, n, 2), dtype=torch.uint8, device=gpu, generator=torch.Generator(device=gpu).manual_seed(3))
    rew = torch.zeros((K, n, 4), dtype=torch.float32, device=gpu)
    fl = torch.zeros((K, n), dtype=torch.uint8, device=gpu)
    with res.step_server(idle_ms=4.0, life_s=5.0) as sv:
        sv.play(acts[:25], rew[:25], fl[:25])
        time.sleep(0.1)  # 25 x the idle window: the kernel has left by itself
        torch.cuda.synchronize()  # (a device-wide wait returns once it has)
        for k in range(25):
            r1, f1 = one.step(acts[k])
            assert torch.equal(r1, rew[k]) and torch.equal(f1, fl[k]), k
        assert torch.equal(res.state, one.state)  # written back at the idle exit
        sv.play(acts[25:], rew[25:], fl[25:])     # relaunched
        for k in range(25, K):
            r1, f1 = one.step(acts[k])
            assert torch.equal(r1, rew[k]) and torch.equal(f1, fl[k]), k
    assert torch.equal(res.state, one.state) and torch.equal(res.ep_returns, one.ep_returns)
    assert (fl & 4).any()


def test_step_server_refuses_what_it_does_not_serve(gpu):
    from overcooked_ai_amd import _lib

    ev = make_env("cramped_room", 256, gpu, horizon=20, auto_reset=True, track_events=True)
    with pytest.raises(ValueError, match="event tracking"):
        ev.step_server()
    big = make_env("cramped_room", 256, gpu, horizon=20, auto_reset=True)
    with pytest.raises(_lib.OcAmdError, match="idle_ms"):
        big.step_server(idle_ms=-1.0)


def test_step_server_leaves_and_returns_around_its_idle_and_life_windows(gpu):
    """Bursts separated by pauses around the idle window (1 ms) and a lifetime (20 ms) that expires many times during the test: the
    kernel leaves between bursts — never inside one — and every relaunch resumes from the states it wrote back."""
    import time

    n, horizon = 1500, 11
    res = make_env("coordination_ring", n, gpu, horizon=horizon, auto_reset=True)
    one = make_env("coordination_ring", n, gpu, horizon=horizon, auto_reset=True)
    rng = np.random.default_rng(5)
    g = torch.Generator(device=gpu).manual_seed(9)
    with res.step_server(idle_ms=1.0, life_s=0.02) as sv:
        total = 0
        for burst in range(120):
            K = int(rng.integers(1, 9))
            acts = torch.randint(0, 6, (K, n, 2), dtype=torch.uint8, device=gpu, generator=g)
            rew = torch.zeros((K, n, 4), dtype=torch.float32, device=gpu)
            fl = torch.zeros((K, n), dtype=torch.uint8, device=gpu)
            sv.play(acts, rew, fl)
            for k in range(K):
                r1, f1 = one.step(acts[k])
                assert torch.equal(r1, rew[k]) and torch.equal(f1, fl[k]), (burst, k)
            total += K
            time.sleep(float(rng.choice([0.0, 0.0005, 0.001, 0.002, 0.03])))
        assert sv.steps == total
    assert torch.equal(res.state, one.state) and torch.equal(res.ep_returns, one.ep_returns)


def test_step_server_relaunch_sees_the_callers_pending_work_on_the_states(gpu):
    """Between sync() and the next play the caller may change the states with asynchronous calls on its own stream (here: a reset
    and a burst of ordinary steps): the relaunch inside play waits for that stream before the resident kernel loads them."""
    n, horizon = 4096, 50
    res = make_env("cramped_room", n, gpu, horizon=horizon, auto_reset=True)
    one = make_env("cramped_room", n, gpu, horizon=horizon, auto_reset=True)
    g = torch.Generator(device=gpu).manual_seed(4)
    acts = torch.randint(0, 6, (60, n, 2), dtype=torch.uint8, device=gpu, generator=g)
    rew = torch.zeros((60, n, 4), dtype=torch.float32, device=gpu)
    fl = torch.zeros((60, n), dtype=torch.uint8, device=gpu)
    with res.step_server(idle_ms=5.0) as sv:
        sv.play(acts[:20], rew[:20], fl[:20])
        sv.sync()
        for env in (res, one):
            if env is one:
                for k in range(20):
                    env.step(acts[k])
            env.reset()
            for k in range(20, 40):  # (no synchronisation between these launches and the play below)
                env.step(acts[k])
        sv.play(acts[40:], rew[40:], fl[40:])
        for k in range(40, 60):
            r1, f1 = one.step(acts[k])
            assert torch.equal(r1, rew[k]) and torch.equal(f1, fl[k]), k
    assert torch.equal(res.state, one.state) and torch.equal(res.ep_returns, one.ep_returns)


def test_step_server_on_a_table_read_through_l2_with_layout_redraws(gpu):
    """More layouts than LDS stages (the general instance reads the records through L2), every restart on a re-drawn layout
    (regen_layout): the resident kernel against consecutive oc_step calls, layout ids included."""
    from overcooked_ai_amd.layout_gen import reference_generated_layouts
    from overcooked_ai_amd.layouts import LayoutTable

    table = LayoutTable(reference_generated_layouts(40))
    n, horizon, K = 1000, 7, 30
    lid = (np.arange(n) * 7 % 40).astype(np.uint16)
    kw = dict(horizon=horizon, auto_reset=True, layout_id=lid, regen_layout=True, seed=3)
    res = make_env(table, n, gpu, **kw)
    one = make_env(table, n, gpu, **kw)
    acts = torch.randint(0, 6, (K, n, 2), dtype=torch.uint8, device=gpu, generator=torch.Generator(device=gpu).manual_seed(8))
    rew = torch.zeros((K, n, 4), dtype=torch.float32, device=gpu)
    fl = torch.zeros((K, n), dtype=torch.uint8, device=gpu)
    with res.step_server(idle_ms=5.0) as sv:
        sv.play(acts, rew, fl)
    for k in range(K):
        r1, f1 = one.step(acts[k])
        assert torch.equal(r1, rew[k]) and torch.equal(f1, fl[k]), k
    assert torch.equal(res.state, one.state) and np.array_equal(res.layout_ids(), one.layout_ids())
    assert (fl & 4).any() and not np.array_equal(res.layout_ids(), lid)
