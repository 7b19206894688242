"""Direction / Action constants of the Overcooked gridworld.

Mirrors the public names of the reference's `overcooked_ai_py.mdp.actions` (actions.py:7-57) so that
agent code written against the reference keeps working: directions are (dx, dy) tuples, INTERACT is the
string "interact", and the index tables give the 0..5 encoding the HIP kernels consume
(include/oc_amd.h: OC_A_*).
"""
import itertools


class Direction:
    NORTH = (0, -1)
    SOUTH = (0, 1)
    EAST = (1, 0)
    WEST = (-1, 0)
    ALL_DIRECTIONS = INDEX_TO_DIRECTION = [NORTH, SOUTH, EAST, WEST]
    DIRECTION_TO_INDEX = {d: i for i, d in enumerate(INDEX_TO_DIRECTION)}
    OPPOSITE_DIRECTIONS = {NORTH: SOUTH, SOUTH: NORTH, EAST: WEST, WEST: EAST}
    DIRECTION_TO_NAME = {NORTH: "NORTH", SOUTH: "SOUTH", EAST: "EAST", WEST: "WEST"}

    @staticmethod
    def get_adjacent_directions(direction):
        if direction in (Direction.NORTH, Direction.SOUTH):
            return [Direction.EAST, Direction.WEST]
        if direction in (Direction.EAST, Direction.WEST):
            return [Direction.NORTH, Direction.SOUTH]
        raise ValueError("Invalid direction: %s" % (direction,))


class Action:
    STAY = (0, 0)
    INTERACT = "interact"
    ALL_ACTIONS = INDEX_TO_ACTION = Direction.INDEX_TO_DIRECTION + [STAY, INTERACT]
    INDEX_TO_ACTION_INDEX_PAIRS = list(itertools.product(range(len(INDEX_TO_ACTION)), repeat=2))
    ACTION_TO_INDEX = {a: i for i, a in enumerate(INDEX_TO_ACTION)}
    MOTION_ACTIONS = Direction.ALL_DIRECTIONS + [STAY]
    ACTION_TO_CHAR = {
        Direction.NORTH: "↑",
        Direction.SOUTH: "↓",
        Direction.EAST: "→",
        Direction.WEST: "←",
        STAY: "stay",
        INTERACT: INTERACT,
    }
    NUM_ACTIONS = len(ALL_ACTIONS)

    @staticmethod
    def move_in_direction(point, direction):
        assert direction in Action.MOTION_ACTIONS
        return (point[0] + direction[0], point[1] + direction[1])

    @staticmethod
    def to_index(action):
        """Action (tuple / "interact" / list form of a tuple) or an int index -> index 0..5."""
        if isinstance(action, (int,)) and not isinstance(action, bool):
            if not 0 <= action < Action.NUM_ACTIONS:
                raise ValueError("Illegal action index %r" % (action,))
            return int(action)
        if isinstance(action, list):
            action = tuple(action)
        try:
            return Action.ACTION_TO_INDEX[action]
        except (KeyError, TypeError):
            raise ValueError("Illegal action %r" % (action,))

    @staticmethod
    def to_char(action):
        assert action in Action.ALL_ACTIONS
        return Action.ACTION_TO_CHAR[action]

    @staticmethod
    def joint_action_to_char(joint_action):
        return tuple(Action.to_char(a) for a in joint_action)
