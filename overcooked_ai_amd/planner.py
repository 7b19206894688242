"""Single-agent motion costs for `featurize_state` (host-side precompute).

The reference's `featurize_state` (mdp.py:2579-2898) asks its `MotionPlanner` (planning/planners.py:46-450) for
the feature that is cheapest to reach and interact with from a player's (position, orientation):
`min_cost_to_feature(pos_and_or, locations, with_argmin=True)`.  The planner's graph has one node per
(free cell, orientation); a motion action moves the player and turns it, or only turns it when the target cell
is not free (`_move_if_direction`, mdp.py:1718-1727); every action costs 1; a feature is reached by standing on a
free neighbour cell facing it; counters are goals only when listed in `counter_goals`.

`feature_costs(spec, counter_goals)` flattens that into COST[state][cell] = fewest actions from the state to any
valid goal of the feature at `cell` (255 = unreachable / not a goal), which is all the GPU kernel needs: the
argmin over candidate cells, ties broken by the reference's list order.
"""
import numpy as np

DIRS = [(0, -1), (0, 1), (1, 0), (-1, 0)]  # N, S, E, W (actions.py:12-16)
OPPOSITE = [1, 0, 3, 2]
UNREACHABLE = 255


def _floor_cells(spec):
    W = spec.width
    return [y * W + x for (x, y) in spec.cells_of(" ")]


def state_distances(spec):
    """All-pairs action counts between (free cell, orientation) states: int16 [n_states, n_states], -1 if unreachable."""
    W, H = spec.width, spec.height
    floor = _floor_cells(spec)
    fidx = {c: i for i, c in enumerate(floor)}
    n_states = 4 * len(floor)
    succ = np.zeros((n_states, 4), dtype=np.int32)
    for i, c in enumerate(floor):
        x, y = c % W, c // W
        for o in range(4):
            for a, (dx, dy) in enumerate(DIRS):
                nc = (y + dy) * W + (x + dx)
                succ[4 * i + o, a] = 4 * fidx[nc] + a if nc in fidx else 4 * i + a
    dist = np.full((n_states, n_states), -1, dtype=np.int16)
    for s0 in range(n_states):
        d = dist[s0]
        d[s0] = 0
        frontier = [s0]
        step = 0
        while frontier:
            step += 1
            nxt = []
            for s in frontier:
                for t in succ[s]:
                    if d[t] < 0:
                        d[t] = step
                        nxt.append(int(t))
            frontier = nxt
    return dist, floor


def feature_costs(spec, counter_goals="none"):
    """(floor_index uint8[128], cost uint8[n_states, n_cells]).  counter_goals: "none" (NO_COUNTERS_PARAMS, the
    reference's default for env.mlam), "all", or an iterable of (x, y) counter positions."""
    W, H = spec.width, spec.height
    n_cells = W * H
    dist, floor = state_distances(spec)
    fidx = {c: i for i, c in enumerate(floor)}
    if counter_goals == "none":
        goals_ok = set()
    elif counter_goals == "all":
        goals_ok = {y * W + x for (x, y) in spec.cells_of("X")}
    else:
        goals_ok = {y * W + x for (x, y) in counter_goals}
    n_states = 4 * len(floor)
    cost = np.full((n_states, n_cells), UNREACHABLE, dtype=np.uint8)
    for c in range(n_cells):
        t = spec.terrain_mtx[c // W][c % W]
        if t == " " or (t == "X" and c not in goals_ok):
            continue
        x, y = c % W, c // W
        goals = []
        for d, (dx, dy) in enumerate(DIRS):
            ax, ay = x + dx, y + dy
            if 0 <= ax < W and 0 <= ay < H and (ay * W + ax) in fidx:
                goals.append(4 * fidx[ay * W + ax] + OPPOSITE[d])  # stand next to the feature, face it
        if not goals:
            continue
        best = np.full((n_states,), 10 ** 6, dtype=np.int64)
        for g in goals:
            dg = dist[:, g].astype(np.int64)
            best = np.where((dg >= 0) & (dg < best), dg, best)
        ok = best < 10 ** 6
        if best[ok].size and best[ok].max() >= UNREACHABLE:
            raise ValueError("motion cost does not fit a byte")
        cost[ok, c] = best[ok].astype(np.uint8)
    floor_index = np.full((128,), 0xFF, dtype=np.uint8)
    for i, c in enumerate(floor):
        floor_index[c] = i
    return floor_index, cost


def pack_plan_tables(specs, counter_goals="none"):
    """Blob + offsets for a layout table: per layout [floor_index: 128 B][cost: n_states rows of n_cells bytes, each
    row padded to a multiple of 16 bytes], 16-byte aligned."""
    offs, parts, pos = [], [], 0
    for s in specs:
        fi, cost = feature_costs(s, counter_goals)
        stride = (cost.shape[1] + 15) & ~15
        rows = np.full((cost.shape[0], stride), UNREACHABLE, dtype=np.uint8)
        rows[:, :cost.shape[1]] = cost
        raw = fi.tobytes() + rows.tobytes()
        raw += b"\0" * ((-len(raw)) % 16)
        offs.append(pos)
        parts.append(raw)
        pos += len(raw)
    return np.frombuffer(b"".join(parts), dtype=np.uint8).copy(), np.asarray(offs, dtype=np.uint32)
