"""Direction / Action constants of the Overcooked gridworld.

Keeps the public names of the reference's `overcooked_ai_py.mdp.actions` (actions.py:7-57) so that agent code
written against the reference keeps working: directions are (dx, dy) tuples, INTERACT is the string "interact",
and the index tables give the 0..5 encoding the HIP kernels consume (include/oc_amd.h: OC_A_*).

Everything is derived from one table, `_MOVES`, whose row order IS the wire encoding.
"""
import itertools

# (name, (dx, dy), glyph, opposite) in wire order: action / orientation index 0..3
_MOVES = (("NORTH", (0, -1), "↑", "SOUTH"), ("SOUTH", (0, 1), "↓", "NORTH"),
          ("EAST", (1, 0), "→", "WEST"), ("WEST", (-1, 0), "←", "EAST"))
_VEC = {name: vec for name, vec, _, _ in _MOVES}


class Direction:
    NORTH, SOUTH, EAST, WEST = (vec for _, vec, _, _ in _MOVES)
    INDEX_TO_DIRECTION = [vec for _, vec, _, _ in _MOVES]
    ALL_DIRECTIONS = INDEX_TO_DIRECTION
    DIRECTION_TO_INDEX = {vec: i for i, (_, vec, _, _) in enumerate(_MOVES)}
    DIRECTION_TO_NAME = {vec: name for name, vec, _, _ in _MOVES}
    OPPOSITE_DIRECTIONS = {vec: _VEC[opp] for _, vec, _, opp in _MOVES}

    @staticmethod
    def get_adjacent_directions(direction):
        """The two directions perpendicular to `direction` (east/west for a vertical one and vice versa)."""
        if direction not in Direction.DIRECTION_TO_INDEX:
            raise ValueError("Invalid direction: %s" % (direction,))
        vertical = direction[0] == 0
        return [d for d in Direction.INDEX_TO_DIRECTION if (d[0] == 0) != vertical]


class Action:
    STAY = (0, 0)          # wire index 4
    INTERACT = "interact"  # wire index 5
    INDEX_TO_ACTION = Direction.INDEX_TO_DIRECTION + [STAY, INTERACT]
    ALL_ACTIONS = INDEX_TO_ACTION
    NUM_ACTIONS = len(INDEX_TO_ACTION)
    ACTION_TO_INDEX = {a: i for i, a in enumerate(INDEX_TO_ACTION)}
    MOTION_ACTIONS = INDEX_TO_ACTION[:5]
    INDEX_TO_ACTION_INDEX_PAIRS = list(itertools.product(range(NUM_ACTIONS), repeat=2))
    ACTION_TO_CHAR = dict([(vec, glyph) for _, vec, glyph, _ in _MOVES] + [(STAY, "stay"), (INTERACT, INTERACT)])

    @staticmethod
    def move_in_direction(point, direction):
        assert direction in Action.MOTION_ACTIONS
        x, y = point
        return (x + direction[0], y + direction[1])

    @staticmethod
    def to_index(action):
        """Action (tuple / "interact" / list form of a tuple) or an int index -> index 0..5."""
        if isinstance(action, int) and not isinstance(action, bool):
            if not 0 <= action < Action.NUM_ACTIONS:
                raise ValueError("Illegal action index %r" % (action,))
            return int(action)
        key = tuple(action) if isinstance(action, list) else action
        try:
            return Action.ACTION_TO_INDEX[key]
        except (KeyError, TypeError):
            raise ValueError("Illegal action %r" % (action,))

    @staticmethod
    def to_char(action):
        assert action in Action.ALL_ACTIONS
        return Action.ACTION_TO_CHAR[action]

    @staticmethod
    def joint_action_to_char(joint_action):
        return tuple(map(Action.to_char, joint_action))
