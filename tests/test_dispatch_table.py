"""docs/DISPATCH.md (which rollout kernel instance serves which registry layout) must be what the library's own dispatch answers
through oc_rollout_plan — no GPU needed: the dispatch is walked with stand-in pointers and nothing is launched."""
import os

import pytest

from overcooked_ai_amd import _lib, dispatch, layouts

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_dispatch_table_is_current():
    with open(os.path.join(ROOT, "docs", "DISPATCH.md")) as f:
        committed = f.read()
    assert committed == dispatch.render(dispatch.table()), "docs/DISPATCH.md is stale: python tools/gen_dispatch_table.py"


def test_baseline_configs_take_the_mover_interact_kernel():
    tiled = _lib.OPT_AUTO_RESET | _lib.OPT_FLAGS_TILED8
    one = layouts.LayoutTable([layouts.spec_from_name("cramped_room")])
    assert dispatch.rollout_plan(one, 65536, options=tiled).startswith("k_rollout5<LAY_LDS=true, FT8=true, OLD=false, BIG=false, EV=false>")
    five = layouts.LayoutTable([layouts.spec_from_name(n) for n in ("cramped_room", "asymmetric_advantages", "coordination_ring",
                                                                      "forced_coordination", "counter_circuit_o_1order")])
    p = dispatch.rollout_plan(five, 65536, options=tiled)
    assert p.startswith("k_rollout5<LAY_LDS=true, FT8=true") and "1 round(s)" in p
    assert "2 round(s)" in dispatch.rollout_plan(five, 131072, options=tiled)
    # a ragged batch (not whole 256-env workgroups) stays in one wavefront per env group
    assert dispatch.rollout_plan(five, 65536 + 64).startswith("k_rollout4<UNIFORM=false")
    # the cross-check families
    assert dispatch.rollout_plan(one, 4096, options=_lib.OPT_LANE_PAIR).startswith("k_rollout_pair")
    assert dispatch.rollout_plan(one, 4096, options=_lib.OPT_PREDICATE_INTERACT).startswith("k_rollout ")
    assert dispatch.rollout_plan(one, 65536, options=_lib.OPT_ONE_WAVEFRONT).startswith("k_rollout4<UNIFORM=true, MAXP=1, LAY_LDS=true, MODE=1")


def test_plan_applies_the_argument_checks_of_the_real_call():
    one = layouts.LayoutTable([layouts.spec_from_name("cramped_room")])
    with pytest.raises(_lib.OcAmdError, match="multiples of 8"):
        dispatch.rollout_plan(one, 65536, n_steps=12, options=_lib.OPT_FLAGS_TILED8)
    with pytest.raises(_lib.OcAmdError, match="horizon"):
        dispatch.rollout_plan(one, 65536, horizon=0)
    assert dispatch.rollout_plan(one, 0) == "nothing to launch (no envs or no steps)"


def test_generated_terrain_tables_take_the_l2_table_instance_in_rounds():
    """BASELINE configs[4]'s rank shape: a table of generated 9 x 5 terrains (more layouts than LDS stages), 131 072 envs."""
    from overcooked_ai_amd.layout_gen import reference_generated_layouts

    tab = layouts.LayoutTable(reference_generated_layouts(64))
    p = dispatch.rollout_plan(tab, 131072, options=_lib.OPT_AUTO_RESET | _lib.OPT_FLAGS_TILED8)
    assert p.startswith("k_rollout5<LAY_LDS=false, FT8=true, OLD=false, BIG=false, EV=false>") and "2 round(s)" in p
    # 1 M envs: beyond eight rounds the one-wavefront instances serve the batch (one-pot tables keep their tiled-flags instance)
    assert dispatch.rollout_plan(tab, 1 << 20).startswith("k_rollout4<UNIFORM=false")
