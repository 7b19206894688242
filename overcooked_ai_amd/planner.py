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


def walk_records(spec, cost):
    """Per (free cell, orientation) state, what `k_featurize` would find by walking the whole grid, as far as it depends
    on the terrain alone (round 4: the kernel walked 45 cells per lane for what is mostly static):
      [4 x u32]  arg-min keys (cost << 9 | cell; 0xFFFFFFFF = none reachable) of the onion / tomato / dish dispensers and the
                 serving cells — min_cost_to_feature's ties go to the first position in list order = the lower cell index;
      [4 x u32]  the (up to) four closest pots, ascending keys;
      [u8 n][n x u8]  the counters that are motion goals, ascending (cost, cell): the closest EMPTY one is the first of the
                 list without an object; a counter that holds something competes through its own cost-row byte.
    Returns (uint8 [n_states, stride], stride) with stride a multiple of 16."""
    W = spec.width
    n_states, n_cells = cost.shape
    cells = {t: [y * W + x for (x, y) in spec.cells_of(t)] for t in "OTDSPX"}
    n_goal_counters = max((int((cost[s, cells["X"]] < UNREACHABLE).sum()) if cells["X"] else 0) for s in range(n_states)) if n_states else 0
    stride = (32 + 1 + n_goal_counters + 15) & ~15
    rec = np.zeros((n_states, stride), dtype=np.uint8)
    for s in range(n_states):
        def keys(kind):
            return sorted((int(cost[s, c]) << 9) | c for c in cells[kind] if cost[s, c] < UNREACHABLE)
        words = [(keys(k) or [0xFFFFFFFF])[0] for k in "OTDS"] + (keys("P") + [0xFFFFFFFF] * 4)[:4]
        rec[s, :32] = np.asarray(words, dtype=np.uint32).view(np.uint8)
        goal = [k & 0x7F for k in keys("X")]
        rec[s, 32] = len(goal)
        rec[s, 33:33 + len(goal)] = goal
    return rec, stride


def pack_plan_tables(specs, counter_goals="none"):
    """Blob + offsets for a layout table.  offsets[l] (l < n): per layout [floor_index: 128 B][cost: n_states rows of
    n_cells bytes, each row padded to a multiple of 16 bytes], 16-byte aligned — what oc_potential and oc_featurize index;
    offsets[n + l]: the layout's walk section for oc_featurize, [u32 stride, 12 B pad][n_states records of walk_records]."""
    offs, walk_offs, parts, pos = [], [], [], 0
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
        rec, rstride = walk_records(s, cost)
        raw = np.asarray([rstride, 0, 0, 0], dtype=np.uint32).tobytes() + rec.tobytes()
        walk_offs.append(pos)
        parts.append(raw)
        pos += len(raw)
    return np.frombuffer(b"".join(parts), dtype=np.uint8).copy(), np.asarray(offs + walk_offs, dtype=np.uint32)


class MotionPlanner:
    """Host-side stand-in for the reference's single-agent `MotionPlanner` (planning/planners.py:46-450) over the BFS
    tables above: what `OvercookedEnv.mp` hands to code written against the reference (featurize_state's
    `min_cost_to_feature`, potential_function's distances, scripted agents that walk a `get_plan`).  Costs and goal rules
    are the reference's: a plan ends standing next to the feature, facing it, and pays one more action for the INTERACT.
    Positions are (x, y) tuples and orientations Direction tuples, as in the reference."""

    def __init__(self, mdp, counter_goals=()):
        from .actions import Action, Direction

        self._A, self._D = Action, Direction
        self.mdp = mdp
        spec = mdp.spec if hasattr(mdp, "spec") else mdp
        self.spec = spec
        self.counter_goals = [tuple(c) for c in counter_goals]
        self._W = spec.width
        self._dist, self._floor = state_distances(spec)
        self._fidx = {c: i for i, c in enumerate(self._floor)}
        self._dirs = [tuple(d) for d in Direction.ALL_DIRECTIONS]
        goal_cells = {t: spec.cells_of(t) for t in "OTPDS"}
        feats = [p for t in "OTPDS" for p in goal_cells[t]] + list(spec.cells_of("X"))
        # every terrain feature has its list of goals (stand on a free neighbour, face the feature) — reference planners.py:425-450
        self.motion_goals_for_pos = {tuple(p): self._goals_of(tuple(p)) for p in feats}
        self._valid_goal_states = set()
        for t in "OTPDS":
            for p in goal_cells[t]:
                self._valid_goal_states.update(self.motion_goals_for_pos[tuple(p)])
        for p in self.counter_goals:
            self._valid_goal_states.update(self.motion_goals_for_pos.get(p, []))

    def _goals_of(self, pos):
        x, y = pos
        out = []
        for d, (dx, dy) in enumerate(DIRS):
            c = (y + dy) * self._W + (x + dx)
            if 0 <= x + dx < self._W and 0 <= y + dy < self.spec.height and c in self._fidx:
                out.append(((x + dx, y + dy), self._dirs[OPPOSITE[d]]))
        return out

    def _state(self, pos_and_or):
        (x, y), o = pos_and_or
        return 4 * self._fidx[y * self._W + x] + self._dirs.index(tuple(o))

    def is_valid_motion_goal(self, goal_pos_and_or):
        return (tuple(goal_pos_and_or[0]), tuple(goal_pos_and_or[1])) in self._valid_goal_states

    def is_valid_motion_start_goal_pair(self, start_pos_and_or, goal_pos_and_or):
        if not self.is_valid_motion_goal(goal_pos_and_or):
            return False
        try:
            return self._dist[self._state(start_pos_and_or), self._state(goal_pos_and_or)] >= 0
        except (KeyError, ValueError):
            return False

    def get_gridworld_distance(self, start_pos_and_or, goal_pos_and_or):
        """Actions from start to goal, the closing INTERACT not counted."""
        assert self.is_valid_motion_start_goal_pair(start_pos_and_or, goal_pos_and_or), \
            "Goal position and orientation were not a valid motion goal"
        return int(self._dist[self._state(start_pos_and_or), self._state(goal_pos_and_or)])

    def get_plan(self, start_pos_and_or, goal_pos_and_or):
        """(action_plan ending in INTERACT, [(pos, orientation) after every action], number of actions): one shortest
        plan, found by walking down the distance table (the reference's planner may pick a different shortest path)."""
        n = self.get_gridworld_distance(start_pos_and_or, goal_pos_and_or)
        goal = self._state(goal_pos_and_or)
        (x, y), o = start_pos_and_or
        o = tuple(o)
        actions, path = [], []
        for left in range(n, 0, -1):
            for d, (dx, dy) in enumerate(DIRS):
                c = (y + dy) * self._W + (x + dx)
                nx, ny = (x + dx, y + dy) if c in self._fidx and 0 <= x + dx < self._W else (x, y)
                s = 4 * self._fidx[ny * self._W + nx] + d
                if self._dist[s, goal] == left - 1:
                    x, y, o = nx, ny, self._dirs[d]
                    actions.append(self._dirs[d])
                    path.append(((x, y), o))
                    break
            else:  # the table says a shorter state exists among the four successors
                raise AssertionError("inconsistent distance table")
        actions.append(self._A.INTERACT)
        path.append(((x, y), o))
        return actions, path, len(actions)

    def min_cost_to_feature(self, start_pos_and_or, feature_pos_list, with_argmin=False, debug=False):
        """Fewest actions (INTERACT included) from a (position, orientation) to any of the features; inf when none is
        reachable.  Ties keep the first feature of the list, like the reference's strict '<'."""
        best, arg = np.inf, None
        (sx, sy), _ = start_pos_and_or
        assert (sy * self._W + sx) in self._fidx, "start position is not a walkable cell"  # planners.py:402-403
        s = self._state(start_pos_and_or)
        for pos in feature_pos_list:
            for g in self.motion_goals_for_pos[tuple(pos)]:  # (KeyError for a position that is no feature, as upstream)
                if g not in self._valid_goal_states:
                    continue
                d = self._dist[s, self._state(g)]
                if 0 <= d < best:
                    best, arg = int(d), tuple(pos)
        cost = best + 1
        return (cost, arg) if with_argmin else cost

    def min_cost_between_features(self, pos_list1, pos_list2, manhattan_if_fail=False):
        """Fewest actions from (any goal of) a feature of the first list to interacting with a feature of the second."""
        best, manhattan = np.inf, np.inf
        for p1 in pos_list1:
            for p2 in pos_list2:
                for g1 in self.motion_goals_for_pos[tuple(p1)]:
                    for g2 in self.motion_goals_for_pos[tuple(p2)]:
                        if self.is_valid_motion_start_goal_pair(g1, g2):
                            best = min(best, self.get_gridworld_distance(g1, g2))
                        elif manhattan_if_fail:
                            manhattan = min(manhattan, abs(g1[0][0] - g2[0][0]) + abs(g1[0][1] - g2[0][1]))
        if manhattan_if_fail and best == np.inf:
            best = manhattan
        return best + 1
