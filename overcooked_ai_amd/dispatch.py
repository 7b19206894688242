"""Which rollout kernel instance serves which batch: the host side of oc_rollout_plan (include/oc_amd.h, ABI 6).

The answers come from oc_rollout_random's own dispatch walked with stand-in pointers (nothing is launched, no device memory is
touched), so this works on a host without a GPU.  `table()` is what tools/gen_dispatch_table.py writes to docs/DISPATCH.md and
what tests/test_dispatch_table.py compares that file with."""
import ctypes

import numpy as np

from . import _lib, layouts


def batch_for(table, n_envs):
    """OcBatch of `n_envs` envs over a LayoutTable with the kernel-variant hints filled in; stand-in device pointers."""
    L = _lib.load()
    b = _lib.OcBatch(d_layouts=4096, d_layout_id=4096 if len(table) > 1 else None, n_envs=int(n_envs), n_layouts=len(table),
                     width=table.width, height=table.height)
    rec = np.ascontiguousarray(table.records)
    _lib.check(L.oc_batch_hints(rec.ctypes.data, len(table), ctypes.byref(b)), "oc_batch_hints")
    return b


def rollout_plan(table, n_envs, n_steps=4000, t0=0, horizon=400, options=_lib.OPT_AUTO_RESET, with_outputs=True, event_sink=0,
                 start=None):
    """The kernel instance `oc_rollout_random` launches for this table and launch shape (text), or the library's refusal
    (OcAmdError) for shapes it does not serve."""
    L = _lib.load()
    b = batch_for(table, n_envs)
    out = ctypes.create_string_buffer(320)
    rc = L.oc_rollout_plan(ctypes.byref(b), int(horizon), int(options), int(t0), int(n_steps), int(bool(with_outputs)),
                           int(event_sink), ctypes.byref(start) if start is not None else None, out, len(out))
    _lib.check(rc, "oc_rollout_plan")
    return out.value.decode()


# the launch shapes docs/DISPATCH.md lists per layout: (column title, kwargs)
SHAPES = (
    ("65 536 envs x 4 000 steps, tiled flags (bench.py's shape)", dict(n_envs=65536, options=_lib.OPT_AUTO_RESET | _lib.OPT_FLAGS_TILED8)),
    ("1 048 576 envs x 4 000 steps", dict(n_envs=1 << 20)),
    ("65 536 envs, per-episode event counters", dict(n_envs=65536, event_sink=1)),
    ("100 envs x 5 steps", dict(n_envs=100, n_steps=5)),
    ("65 536 envs x 4 000 steps, no output arrays", dict(n_envs=65536, with_outputs=False)),
)


def _short(text):
    """'k_rollout5<LAY_LDS=true, ...> mover + ..., 130864 B LDS' -> instance name + LDS bytes"""
    head = text.split(">")[0] + ">" if "<" in text else text
    lds = text.rsplit(",", 1)[-1].strip() if "B LDS" in text else ""
    rounds = [p.strip() for p in text.split(",") if "round(s)" in p]
    return head + (" " + rounds[0] if rounds else "") + (" " + lds if lds else "")


def table(names=None):
    """Rows (layout, cells, pots, old_dynamics, [instance per SHAPES entry]) for every registry layout the library serves (two
    players), plus its old-dynamics form for the layouts the paper-reproduction runs use."""
    rows = []
    for name in (names or layouts.layout_names()):
        for old in (False, True):
            try:
                spec = layouts.spec_from_name(name, old_dynamics=True) if old else layouts.spec_from_name(name)
            except (AssertionError, ValueError):
                continue  # (old dynamics: three-item orders only, mdp.py:1121-1127)
            if old and name not in ("cramped_room", "asymmetric_advantages", "coordination_ring", "forced_coordination", "counter_circuit_o_1order"):
                continue
            if spec.num_players != 2:
                rows.append((name, spec.width * spec.height, len(spec.cells_of("P")), old, ["refused: %d players" % spec.num_players] * len(SHAPES)))
                continue
            tab = layouts.LayoutTable([spec])
            cols = []
            for _, kw in SHAPES:
                try:
                    cols.append(_short(rollout_plan(tab, **kw)))
                except _lib.OcAmdError as e:
                    cols.append("refused: " + str(e).split(": ", 2)[-1][:80])
            rows.append((name, tab.n_cells, tab.max_pots, old, cols))
    return rows


def render(rows):
    out = ["# Rollout dispatch per layout (generated: `python tools/gen_dispatch_table.py`; checked by tests/test_dispatch_table.py)", "",
           "What `oc_rollout_random` launches for a batch of ONE registry layout, as `oc_rollout_plan` (include/oc_amd.h) reports it — the",
           "library's own dispatch walked with stand-in pointers.  `k_rollout5` = mover + interact wavefronts (csrc/step_duo5.hpp);",
           "`k_rollout4` = one wavefront per 64 envs (csrc/step_lut4.hpp: MODE 0 arithmetic movement, 1 joint move table, 2 floor mask).", "",
           "| layout | cells | pots | dynamics | " + " | ".join(t for t, _ in SHAPES) + " |", "|---|---|---|---|" + "---|" * len(SHAPES)]
    for name, cells, pots, old, cols in rows:
        out.append("| %s | %d | %d | %s | %s |" % (name, cells, pots, "old" if old else "new", " | ".join("`%s`" % c for c in cols)))
    return "\n".join(out) + "\n"
