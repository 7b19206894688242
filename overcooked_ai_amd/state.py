"""Host-side state types and the packed wire format.

`PlayerState`, `ObjectState`, `SoupState` and `OvercookedState` carry the same fields and the same
`to_dict` / `from_dict` JSON schema as the reference's types (mdp.py:384-430, 433-693, 696-781, 784-1015) so
that trajectories, fixtures and agents written against the reference interoperate.  They are thin records:
all game logic runs on the GPU over the packed representation described in include/oc_amd.h, and
`pack_states` / `unpack_states` convert between the two.

Packed form: np.uint8 array [n_planes, n_envs, 16] (plane-major struct-of-arrays of 16-byte words).
"""
import copy

import numpy as np

from .actions import Direction

O_NONE, O_ONION, O_TOMATO, O_DISH, O_SOUP = 0, 1, 2, 3, 0x80
_SIMPLE_CODE = {"onion": O_ONION, "tomato": O_TOMATO, "dish": O_DISH}
_SIMPLE_NAME = {v: k for k, v in _SIMPLE_CODE.items()}


class ObjectState:
    def __init__(self, name, position, **kwargs):
        self.name = name
        self._position = tuple(position)

    @property
    def position(self):
        return self._position

    @position.setter
    def position(self, new_pos):
        self._position = tuple(new_pos)

    def is_valid(self):
        return self.name in ("onion", "tomato", "dish")

    def deepcopy(self):
        return ObjectState(self.name, self.position)

    def __eq__(self, other):
        return isinstance(other, ObjectState) and self.name == other.name and self.position == other.position

    def __hash__(self):
        return hash((self.name, self.position))

    def __repr__(self):
        return "{}@{}".format(self.name, self.position)

    def to_dict(self):
        return {"name": self.name, "position": self.position}

    @classmethod
    def from_dict(cls, obj_dict):
        d = copy.deepcopy(obj_dict)
        return ObjectState(d["name"], d["position"])


# The reference's Recipe class is process-global state configured by the most recently constructed OvercookedGridworld
# (Recipe.configure, mdp.py:221-336).  The kernels carry recipes per layout instead; this one hook keeps the reference's
# behaviour for SoupStates a user builds by hand and cooks / finishes through the mutators below (mdp.py:565-612): their cook
# time resolves against the layout of the mdp constructed last, exactly as upstream.
_RECIPE_SPEC = [None]


def configure_recipes(spec):
    """Called by OvercookedGridworld's constructors (the mirror of Recipe.configure)."""
    _RECIPE_SPEC[0] = spec


class SoupState(ObjectState):
    """Soup record. `cook_time` is the recipe's cook time as resolved by the layout (Recipe.time,
    mdp.py:163-188); -1 / None while the soup is idle, exactly like the reference's to_dict (mdp.py:625)."""

    def __init__(self, position, ingredients=(), cooking_tick=-1, cook_time=None, **kwargs):
        super().__init__("soup", position)
        self._ingredients = [i if isinstance(i, str) else i.name for i in ingredients]
        self._cooking_tick = cooking_tick
        self._cook_time = cook_time

    @property
    def ingredients(self):
        return list(self._ingredients)

    @property
    def cooking_tick(self):
        return self._cooking_tick

    @property
    def cook_time(self):
        """The cook time the soup came with (state dicts, kernel states), else — once it cooks — its recipe's time under the
        layout configured last (mdp.py:520-526)."""
        if self._cook_time is not None:
            return self._cook_time
        if self.is_idle:
            return None
        spec = _RECIPE_SPEC[0]
        if spec is None:
            return None
        n_o = sum(1 for i in self._ingredients if i == "onion")
        return spec.recipe_time((n_o, len(self._ingredients) - n_o))

    @property
    def is_idle(self):
        return self._cooking_tick < 0

    @property
    def is_ready(self):
        if self.is_idle:
            return False
        ct = self.cook_time
        return ct is not None and self._cooking_tick >= ct

    @property
    def is_full(self):
        return not self.is_idle or len(self._ingredients) == 3  # Recipe.MAX_NUM_INGREDIENTS (mdp.py:546-551)

    # ---- the mutating API of mdp.py:565-612, with the reference's checks and messages
    def auto_finish(self):
        if len(self._ingredients) == 0:
            raise ValueError("Cannot finish soup with no ingredients")
        self._cooking_tick = 0
        ct = self.cook_time
        if ct is None:
            raise ValueError("Recipe class must be configured before recipes can be created")
        self._cooking_tick = ct

    def add_ingredient(self, ingredient):
        name = ingredient if isinstance(ingredient, str) else ingredient.name
        if name not in ("onion", "tomato"):
            raise ValueError("Invalid ingredient")
        if self.is_full:
            raise ValueError("Reached maximum number of ingredients in recipe")
        if not isinstance(ingredient, str):
            ingredient.position = self.position
        self._ingredients.append(name)

    def add_ingredient_from_str(self, ingredient_str):
        self.add_ingredient(ObjectState(ingredient_str, self.position))

    def pop_ingredient(self):
        if not self.is_idle:
            raise ValueError("Cannot remove an ingredient from this soup at this time")
        if len(self._ingredients) == 0:
            raise ValueError("No ingredient to remove")
        return ObjectState(self._ingredients.pop(), self.position)

    def begin_cooking(self):
        if not self.is_idle:
            raise ValueError("Cannot begin cooking this soup at this time")
        if len(self._ingredients) == 0:
            raise ValueError("Must add at least one ingredient to soup before you can begin cooking")
        self._cooking_tick = 0

    def cook(self):
        if self.is_idle:
            raise ValueError("Must begin cooking before advancing cook tick")
        if self.is_ready:
            raise ValueError("Cannot cook a soup that is already done")
        self._cooking_tick += 1

    @property
    def is_cooking(self):
        return not self.is_idle and not self.is_ready

    @property
    def cook_time_remaining(self):
        return max(0, self.cook_time - self._cooking_tick)

    def is_valid(self):
        return len(self._ingredients) <= 3

    def deepcopy(self):
        return SoupState(self.position, list(self._ingredients), self._cooking_tick, self._cook_time)

    def __eq__(self, other):
        return (isinstance(other, SoupState) and self.position == other.position
                and self._cooking_tick == other._cooking_tick and self._ingredients == other._ingredients)

    def __hash__(self):
        return hash((self.position, self._cooking_tick, tuple(self._ingredients)))

    def __repr__(self):
        return "soup@{}\nIngredients:\t{}\nCooking Tick:\t{}".format(self.position, self._ingredients, self._cooking_tick)

    def to_dict(self):
        d = super().to_dict()
        d["_ingredients"] = [{"name": n, "position": self.position} for n in self._ingredients]
        d["cooking_tick"] = self._cooking_tick
        d["is_cooking"] = self.is_cooking
        d["is_ready"] = self.is_ready
        d["is_idle"] = self.is_idle
        d["cook_time"] = -1 if self.is_idle else self.cook_time
        d["_cooking_tick"] = self._cooking_tick
        return d

    @classmethod
    def from_dict(cls, obj_dict):
        d = copy.deepcopy(obj_dict)
        if d["name"] != "soup":
            return ObjectState.from_dict(d)
        if "state" in d:  # legacy (ingredient, count, time) triple, mdp.py:638-656
            ingredient, num, time = d["state"]
            tick = -1 if time == 0 else time
            cook_time = None
            if time >= 20:
                cook_time = tick
            return cls(d["position"], [ingredient] * num, tick, cook_time)
        ings = [i["name"] for i in d.get("_ingredients", [])]
        tick = d.get("cooking_tick", d.get("_cooking_tick", -1))
        ct = d.get("cook_time", None)
        if ct is not None and ct < 0:
            ct = None
        return cls(d["position"], ings, tick, ct)

    @classmethod
    def get_soup(cls, position, num_onions=1, num_tomatoes=0, cooking_tick=-1, finished=False, cook_time=None, **kw):
        if num_onions < 0 or num_tomatoes < 0:
            raise ValueError("Number of active ingredients must be positive")
        if num_onions + num_tomatoes > 3:
            raise ValueError("Too many ingredients specified for this soup")
        if cooking_tick >= 0 and num_tomatoes + num_onions == 0:
            raise ValueError("_cooking_tick must be -1 for empty soup")
        if finished and num_tomatoes + num_onions == 0:
            raise ValueError("Empty soup cannot be finished")
        soup = cls(position, ["onion"] * num_onions + ["tomato"] * num_tomatoes, cooking_tick, cook_time)
        if finished:
            soup.auto_finish()  # (mdp.py:693-694: the tick jumps to the recipe's cook time)
        return soup


class PlayerState:
    def __init__(self, position, orientation, held_object=None):
        self.position = tuple(position)
        self.orientation = tuple(orientation)
        self.held_object = held_object
        assert self.orientation in Direction.ALL_DIRECTIONS
        if self.held_object is not None:
            assert isinstance(self.held_object, ObjectState)
            assert self.held_object.position == self.position

    @property
    def pos_and_or(self):
        return (self.position, self.orientation)

    def has_object(self):
        return self.held_object is not None

    def get_object(self):
        assert self.has_object()
        return self.held_object

    def set_object(self, obj):
        assert not self.has_object()
        obj.position = self.position
        self.held_object = obj

    def remove_object(self):
        assert self.has_object()
        obj, self.held_object = self.held_object, None
        return obj

    def deepcopy(self):
        return PlayerState(self.position, self.orientation,
                           None if self.held_object is None else self.held_object.deepcopy())

    def __eq__(self, other):
        return (isinstance(other, PlayerState) and self.position == other.position
                and self.orientation == other.orientation and self.held_object == other.held_object)

    def __hash__(self):
        return hash((self.position, self.orientation, self.held_object))

    def __repr__(self):
        return "{} facing {} holding {}".format(self.position, self.orientation, str(self.held_object))

    def to_dict(self):
        return {
            "position": self.position,
            "orientation": self.orientation,
            "held_object": self.held_object.to_dict() if self.held_object is not None else None,
        }

    @staticmethod
    def from_dict(player_dict):
        d = copy.deepcopy(player_dict)
        held = d.get("held_object", None)
        if held is not None:
            held = SoupState.from_dict(held)
        return PlayerState(d["position"], d["orientation"], held)


def _recipe_sort_key(order):
    """Ordering of Recipe objects (Recipe.__int__/__lt__, mdp.py:71-94) for the sorted order lists."""
    ings = order["ingredients"]
    n_t = sum(1 for i in ings if i == "tomato")
    n_o = sum(1 for i in ings if i == "onion")
    enc = n_o + 4 * n_t
    return (1 if n_t * n_o else 0) * enc * 16 + enc


def _norm_orders(orders):
    out = [{"ingredients": tuple(sorted(o["ingredients"]))} for o in orders]
    return sorted(out, key=_recipe_sort_key)


class OvercookedState:
    def __init__(self, players, objects, bonus_orders=(), all_orders=(), timestep=0, **kwargs):
        for pos, obj in objects.items():
            assert obj.position == pos
        self.players = tuple(players)
        self.objects = objects
        self._bonus_orders = _norm_orders(bonus_orders)
        self._all_orders = _norm_orders(all_orders)
        self.timestep = timestep

    @property
    def player_positions(self):
        return tuple(p.position for p in self.players)

    @property
    def player_orientations(self):
        return tuple(p.orientation for p in self.players)

    @property
    def players_pos_and_or(self):
        return tuple(zip(self.player_positions, self.player_orientations))

    @property
    def all_orders(self):
        return list(self._all_orders)

    @property
    def bonus_orders(self):
        return list(self._bonus_orders)

    @property
    def all_objects_list(self):
        return list(self.objects.values()) + [p.held_object for p in self.players if p.held_object is not None]

    def has_object(self, pos):
        return tuple(pos) in self.objects

    def get_object(self, pos):
        assert self.has_object(pos)
        return self.objects[tuple(pos)]

    def add_object(self, obj, pos=None):
        pos = obj.position if pos is None else tuple(pos)
        assert not self.has_object(pos)
        obj.position = pos
        self.objects[pos] = obj

    def remove_object(self, pos):
        assert self.has_object(pos)
        return self.objects.pop(tuple(pos))

    def reverse_players(self):
        self.players = tuple(reversed(self.players))
        return self

    @classmethod
    def from_players_pos_and_or(cls, players_pos_and_or, bonus_orders=(), all_orders=()):
        return cls([PlayerState(*p) for p in players_pos_and_or], objects={}, bonus_orders=bonus_orders,
                   all_orders=all_orders)

    @classmethod
    def from_player_positions(cls, player_positions, bonus_orders=(), all_orders=()):
        return cls.from_players_pos_and_or([(pos, Direction.NORTH) for pos in player_positions], bonus_orders,
                                           all_orders)

    def deepcopy(self):
        return OvercookedState([p.deepcopy() for p in self.players],
                               {pos: o.deepcopy() for pos, o in self.objects.items()},
                               bonus_orders=self._bonus_orders, all_orders=self._all_orders, timestep=self.timestep)

    def time_independent_equal(self, other):
        return (isinstance(other, OvercookedState) and self.players == other.players
                and set(self.objects.items()) == set(other.objects.items())
                and self._bonus_orders == other._bonus_orders and self._all_orders == other._all_orders)

    def __eq__(self, other):
        return self.time_independent_equal(other) and self.timestep == other.timestep

    def __hash__(self):
        return hash((self.players, tuple(self.objects.values())))

    def __str__(self):
        return "Players: {}, Objects: {}, Bonus orders: {} All orders: {} Timestep: {}".format(
            str(self.players), str(list(self.objects.values())), str(self.bonus_orders), str(self.all_orders),
            str(self.timestep))

    def to_dict(self):
        return {
            "players": [p.to_dict() for p in self.players],
            "objects": [o.to_dict() for o in self.objects.values()],
            "bonus_orders": [dict(o) for o in self._bonus_orders],
            "all_orders": [dict(o) for o in self._all_orders],
            "timestep": self.timestep,
        }

    @staticmethod
    def from_dict(state_dict):
        d = copy.deepcopy(state_dict)
        players = [PlayerState.from_dict(p) for p in d["players"]]
        objs = [SoupState.from_dict(o) for o in d.get("objects", [])]
        return OvercookedState(players, {o.position: o for o in objs}, bonus_orders=d.get("bonus_orders") or [],
                               all_orders=d.get("all_orders") or [], timestep=d.get("timestep", 0))


class _LazyState(OvercookedState):
    """The state the single-state port (mdp._SingleEnvPort) hands back: an OvercookedState that still IS the packed bytes the
    kernel wrote.  `players` and `objects` are built from them the first time anything looks at them — `OvercookedEnv.step`
    itself never does — and from then on the object is a plain OvercookedState (its class is switched back, the bytes are
    dropped: whoever has seen the player / object instances may have changed them).  A state nobody looked at goes back into
    the next `get_state_transition` as a 48-80 byte copy instead of a walk over its objects; `timestep`, a plain attribute
    from the start, is always taken from the object.  (The reference's own timing loop, SURVEY 8d-1, never reads the states.)"""

    def __init__(self, *args, **kwargs):
        raise TypeError("_LazyState is built by SingleStateCodec.unpack_lazy")

    def _materialize(self):
        """Build players / objects from the bytes and become a plain OvercookedState."""
        d = self.__dict__
        codec, packed = d.pop("_codec"), d.pop("_packed")
        self.__class__ = OvercookedState
        keep = {k: d[k] for k in ("players", "objects") if k in d}  # (one of the two assigned without a look: it stands)
        codec.fill(self, packed)
        d.update(keep)

    def __getattr__(self, name):  # (only reached when normal lookup fails: before the first look at players / objects)
        if (name == "players" or name == "objects") and "_packed" in self.__dict__:
            self._materialize()
            return self.__dict__[name]
        raise AttributeError(name)

    def __getstate__(self):  # pickle / copy: as a plain state
        self._materialize()
        return self.__dict__

    def __reduce_ex__(self, protocol):
        self._materialize()
        return object.__reduce_ex__(self, protocol)


# ------------------------------------------------------------------------------------------------
# wire format
# ------------------------------------------------------------------------------------------------

def _soup_code(ingredients):
    n = len(ingredients)
    if n > 3:
        raise ValueError("soup with more than 3 ingredients")
    bits = 0
    for i, name in enumerate(ingredients):
        if name == "tomato":
            bits |= 1 << i
        elif name != "onion":
            raise ValueError("Invalid ingredient %r" % (name,))
    return O_SOUP | (n << 3) | bits


def _ingredient_names(obj_dict):
    if "_ingredients" in obj_dict:
        return [i["name"] if isinstance(i, dict) else i for i in obj_dict["_ingredients"]]
    if "state" in obj_dict and obj_dict["state"] is not None:
        ingredient, num, _ = obj_dict["state"]
        return [ingredient] * num
    return list(obj_dict.get("ingredients", []))


def _obj_code(obj_dict):
    name = obj_dict["name"]
    if name == "soup":
        return _soup_code(_ingredient_names(obj_dict))
    try:
        return _SIMPLE_CODE[name]
    except KeyError:
        raise ValueError("Unrecognized object %r" % (name,))


def _soup_tick(obj_dict):
    if "cooking_tick" in obj_dict:
        return obj_dict["cooking_tick"]
    if "_cooking_tick" in obj_dict:
        return obj_dict["_cooking_tick"]
    if obj_dict.get("state") is not None:
        t = obj_dict["state"][2]
        return -1 if t == 0 else t
    return -1


def _cook_time_of(spec, ingredients):
    n_o = sum(1 for i in ingredients if i == "onion")
    return spec.recipe_time((n_o, len(ingredients) - n_o))


def pack_state_dict(spec, state_dict, out, e):
    """Write one reference-format state dict (OvercookedState.to_dict schema) into env slot e of `out`.

    Raises ValueError for states outside the packed domain: objects on floor cells, soups outside pots that
    are not fully cooked, non-soup objects in pots, ticks beyond the recipe's cook time."""
    W, H = spec.width, spec.height
    n_cells = W * H
    out[:, e, :] = 0
    hdr = out[0, e]
    players = state_dict["players"]
    if len(players) != spec.num_players:
        raise ValueError("state has %d players, layout has %d" % (len(players), spec.num_players))
    pots = spec.cells_of("P")
    pot_slot = {p: k for k, p in enumerate(pots)}
    seen = set()
    for p, pl in enumerate(players):
        x, y = pl["position"]
        if not (0 <= x < W and 0 <= y < H) or spec.terrain_mtx[y][x] != " ":
            raise ValueError("player %d is not on a free cell" % p)  # _check_valid_state, mdp.py:1921-1924
        if (x, y) in seen:
            raise ValueError("Overlapping players or objects")
        seen.add((x, y))
        hdr[3 * p] = y * W + x
        hdr[3 * p + 1] = Direction.DIRECTION_TO_INDEX[tuple(pl["orientation"])]
        held = pl.get("held_object")
        if held is not None:
            if tuple(held["position"]) != (x, y):
                raise ValueError("held object position differs from its holder")
            code = _obj_code(held)
            if code & O_SOUP:
                ings = _ingredient_names(held)
                if not ings or _soup_tick(held) != _cook_time_of(spec, ings):
                    raise ValueError("held soups must be fully cooked (tick == cook time)")
            hdr[3 * p + 2] = code
    if len(players) == 1:
        hdr[3] = 0xFF
    t = int(state_dict.get("timestep", 0))
    if not 0 <= t < 65536:
        raise ValueError("timestep %d does not fit the packed state (u16): episodes are limited to 65 535 steps" % t)
    hdr[6], hdr[7] = t & 0xFF, t >> 8
    objects = state_dict.get("objects", [])
    if isinstance(objects, dict):
        objects = list(objects.values())
    for od in objects:
        x, y = od["position"]
        if (x, y) in seen:
            raise ValueError("Overlapping players or objects")
        seen.add((x, y))
        terrain = spec.terrain_mtx[y][x]
        if terrain == " ":
            raise ValueError("non-held object on a free cell")  # mdp.py:1938
        c = y * W + x
        code = _obj_code(od)
        if terrain == "P":
            if not code & O_SOUP:
                raise ValueError("object in pot is not a soup")  # mdp.py:1825
            ings = _ingredient_names(od)
            tick = _soup_tick(od)
            if tick >= 0:
                if not ings:
                    raise ValueError("_cooking_tick must be -1 for empty soup")
                if tick > _cook_time_of(spec, ings):
                    raise ValueError("cooking tick beyond the recipe's cook time")
            hdr[8 + pot_slot[(x, y)]] = tick + 1
        elif code & O_SOUP:
            ings = _ingredient_names(od)
            if not ings or _soup_tick(od) != _cook_time_of(spec, ings):
                raise ValueError("soups outside pots must be fully cooked (tick == cook time)")
        out[1 + (c >> 4), e, c & 15] = code


def _code_to_obj_dict(spec, code, pos, tick=None):
    pos = tuple(int(v) for v in pos)
    if code & O_SOUP:
        n = (code >> 3) & 3
        ings = ["tomato" if (code >> i) & 1 else "onion" for i in range(n)]
        ct = _cook_time_of(spec, ings) if n else None
        if tick is None:
            tick = ct  # soups outside pots are cooked
        idle = tick < 0
        ready = (not idle) and tick >= ct
        return {
            "name": "soup",
            "position": pos,
            "_ingredients": [{"name": i, "position": pos} for i in ings],
            "cooking_tick": tick,
            "is_cooking": (not idle) and (not ready),
            "is_ready": ready,
            "is_idle": idle,
            "cook_time": -1 if idle else ct,
            "_cooking_tick": tick,
        }
    return {"name": _SIMPLE_NAME[int(code)], "position": pos}


def unpack_state_dict(spec, packed, e):
    """Env slot e of a packed array -> reference-format state dict (objects sorted by cell index)."""
    W, H = spec.width, spec.height
    hdr = packed[0, e]
    players = []
    for p in range(spec.num_players):
        pos = int(hdr[3 * p])
        xy = (pos % W, pos // W)
        code = int(hdr[3 * p + 2])
        players.append({
            "position": xy,
            "orientation": Direction.INDEX_TO_DIRECTION[int(hdr[3 * p + 1])],
            "held_object": _code_to_obj_dict(spec, code, xy) if code else None,
        })
    pot_slot = {p: k for k, p in enumerate(spec.cells_of("P"))}
    objects = []
    for c in range(W * H):
        code = int(packed[1 + (c >> 4), e, c & 15])
        if not code:
            continue
        xy = (c % W, c // W)
        tick = None
        if xy in pot_slot:
            tick = int(hdr[8 + pot_slot[xy]]) - 1
        objects.append(_code_to_obj_dict(spec, code, xy, tick))
    all_orders = spec.start_all_orders or [{"ingredients": ["onion"] * a + ["tomato"] * b}
                                           for n in (1, 2, 3) for a in range(n, -1, -1) for b in [n - a]]
    return {
        "players": players,
        "objects": objects,
        "bonus_orders": [dict(o) for o in _norm_orders(spec.start_bonus_orders)],
        "all_orders": [dict(o) for o in _norm_orders(all_orders)],
        "timestep": int(hdr[6]) | (int(hdr[7]) << 8),
    }


def _as_dict(state):
    return state if isinstance(state, dict) else state.to_dict()


def pack_states(spec, states, n_planes=None):
    """List of states (our OvercookedState, the reference's, or to_dict() dicts) -> packed array."""
    n_planes = n_planes or (1 + (spec.width * spec.height + 15) // 16)
    out = np.zeros((n_planes, len(states), 16), dtype=np.uint8)
    for e, s in enumerate(states):
        pack_state_dict(spec, _as_dict(s), out, e)
    return out


def unpack_states(spec, packed, as_dict=False):
    out = [unpack_state_dict(spec, packed, e) for e in range(packed.shape[1])]
    return out if as_dict else [OvercookedState.from_dict(d) for d in out]


def canonical_state_dict(state):
    """Order-insensitive normal form of a to_dict() state for comparisons: the reference's object dict is
    insertion-ordered and not part of equality (mdp.py:970), so objects are sorted by position; tuples/lists
    are unified."""
    d = _as_dict(state)

    def norm_obj(o):
        if o is None:
            return None
        out = {"name": o["name"], "position": tuple(o["position"])}
        if o["name"] == "soup":
            out["ingredients"] = tuple(_ingredient_names(o))
            out["cooking_tick"] = _soup_tick(o)
            for k in ("is_cooking", "is_ready", "is_idle", "cook_time"):
                if k in o:
                    out[k] = o[k]
        return out

    objs = d["objects"].values() if isinstance(d["objects"], dict) else d["objects"]
    return {
        "players": [{"position": tuple(p["position"]), "orientation": tuple(p["orientation"]),
                     "held_object": norm_obj(p.get("held_object"))} for p in d["players"]],
        "objects": sorted((norm_obj(o) for o in objs), key=lambda o: (o["position"][1], o["position"][0])),
        "bonus_orders": [tuple(sorted(o["ingredients"])) for o in d.get("bonus_orders", [])],
        "all_orders": [tuple(sorted(o["ingredients"])) for o in d.get("all_orders", [])],
        "timestep": d.get("timestep", 0),
    }


# ------------------------------------------------------------------------------------------------
# one env, no dict round trip: the single-state calls of the drop-in API (OvercookedGridworld.get_state_transition,
# OvercookedEnv.step) go through this codec; anything it does not recognise falls back to pack_states / unpack_states,
# which validate like the reference's _check_valid_state and raise the matching errors
# ------------------------------------------------------------------------------------------------
class SingleStateCodec:
    def __init__(self, spec, n_planes):
        self.spec, self.n_planes = spec, n_planes
        self.W, self.H = spec.width, spec.height
        self.terrain = [c for row in spec.terrain_mtx for c in row]
        self.pot_slot = {y * self.W + x: k for k, (x, y) in enumerate(spec.cells_of("P"))}
        self.xy = [(c % self.W, c // self.W) for c in range(self.W * self.H)]
        self.or_index = Direction.DIRECTION_TO_INDEX
        self.or_of = Direction.INDEX_TO_DIRECTION
        self.num_players = spec.num_players
        all_orders = spec.start_all_orders or [{"ingredients": ["onion"] * a + ["tomato"] * b}
                                               for n in (1, 2, 3) for a in range(n, -1, -1) for b in [n - a]]
        self.bonus_orders = _norm_orders(spec.start_bonus_orders)
        self.all_orders = _norm_orders(all_orders)
        self._zeros = bytes(16 * n_planes)
        self._soup = {}  # soup code -> (ingredient names, cook time)
        for code in range(O_SOUP, O_SOUP + 32):
            n = (code >> 3) & 3
            ings = ["tomato" if (code >> i) & 1 else "onion" for i in range(n)]
            self._soup[code] = (ings, _cook_time_of(spec, ings) if n else None)
        self._soup_code = {tuple(v[0]): k for k, v in self._soup.items() if (k & 7) >> ((k >> 3) & 3) == 0}

    # -- state -> bytes -----------------------------------------------------------------------
    def _code(self, obj, in_pot):
        """Object code, or None when the object is outside what the fast path handles."""
        if type(obj) is ObjectState:
            return _SIMPLE_CODE.get(obj.name)
        if type(obj) is SoupState:
            code = self._soup_code.get(tuple(obj._ingredients))
            if code is None:
                return None
            if not in_pot and (not obj._ingredients or obj._cooking_tick != self._soup[code][1]):
                return None  # soups outside pots are fully cooked (the general path raises the proper error)
            return code
        return None

    def pack(self, state, buf):
        """Write `state` (an OvercookedState of this package) into the 16 * n_planes bytes of `buf`; False = use the
        general path (unknown types, anything invalid: it raises the proper error)."""
        if type(state) is _LazyState:  # nobody has looked at it since the kernel wrote it: the bytes are the state
            d = state.__dict__
            if d.get("_codec") is self and "players" not in d and "objects" not in d:  # (an attribute assigned without a look: the slow way)
                t = d["timestep"]
                if type(t) is int and 0 <= t < 65536:
                    buf[:] = d["_packed"]
                    buf[6], buf[7] = t & 0xFF, t >> 8
                    return True
            state._materialize()  # (another layout's codec, an odd timestep, an assigned attribute: as a plain state)
        if type(state) is not OvercookedState or len(state.players) != self.num_players:
            return False
        W, terrain = self.W, self.terrain
        buf[:] = self._zeros  # buf: a writable memoryview / bytearray of exactly 16 * n_planes bytes
        seen = []
        for p, pl in enumerate(state.players):
            x, y = pl.position
            if not (0 <= x < W and 0 <= y < self.H):
                return False
            c = y * W + x
            if terrain[c] != " " or c in seen:
                return False
            seen.append(c)
            buf[3 * p] = c
            buf[3 * p + 1] = self.or_index[pl.orientation]
            held = pl.held_object
            if held is not None:
                code = self._code(held, False)
                if code is None or held.position != (x, y):
                    return False
                buf[3 * p + 2] = code
        if self.num_players == 1:
            buf[3] = 0xFF
        t = state.timestep
        if not 0 <= t < 65536:
            return False
        buf[6], buf[7] = t & 0xFF, t >> 8
        for (x, y), obj in state.objects.items():
            if not (0 <= x < W and 0 <= y < self.H) or obj.position != (x, y):
                return False
            c = y * W + x
            tc = terrain[c]
            if tc == " " or c in seen:
                return False
            seen.append(c)
            in_pot = tc == "P"
            code = self._code(obj, in_pot)
            if code is None:
                return False
            if in_pot:
                if code < O_SOUP:
                    return False
                tick = obj._cooking_tick
                ct = self._soup[code][1]
                if tick >= 0 and (ct is None or tick > ct):
                    return False
                buf[8 + self.pot_slot[c]] = tick + 1
            buf[16 + c] = code  # plane 1 + (c >> 4), byte c & 15 of a one-env array
        return True

    # -- bytes -> state -----------------------------------------------------------------------
    def _obj(self, code, pos, tick=None):
        if code < O_SOUP:
            o = ObjectState.__new__(ObjectState)
            o.name, o._position = _SIMPLE_NAME[code], pos
            return o
        ings, ct = self._soup[code]
        if tick is None:
            tick = ct  # soups outside pots are cooked
        o = SoupState.__new__(SoupState)
        o.name, o._position = "soup", pos
        o._ingredients, o._cooking_tick, o._cook_time = list(ings), tick, (None if tick < 0 else ct)
        return o

    def unpack_lazy(self, buf):
        """The state of `buf` (16 * n_planes bytes) as a _LazyState: built when somebody looks."""
        st = _LazyState.__new__(_LazyState)
        d = st.__dict__
        d["_packed"] = bytes(buf)
        d["_codec"] = self
        d["_bonus_orders"], d["_all_orders"] = self.bonus_orders, self.all_orders
        d["timestep"] = buf[6] | (buf[7] << 8)
        return st

    def unpack(self, buf):
        st = OvercookedState.__new__(OvercookedState)
        st._bonus_orders, st._all_orders = self.bonus_orders, self.all_orders
        st.timestep = buf[6] | (buf[7] << 8)
        return self.fill(st, buf)

    def fill(self, st, buf):
        """players / objects of `st` from the packed bytes (timestep and orders are the caller's)."""
        xy = self.xy
        players = []
        for p in range(self.num_players):
            pos = xy[buf[3 * p]]
            code = buf[3 * p + 2]
            pl = PlayerState.__new__(PlayerState)
            pl.position, pl.orientation = pos, self.or_of[buf[3 * p + 1]]
            pl.held_object = self._obj(code, pos) if code else None
            players.append(pl)
        objects = {}
        cells = buf[16:16 + self.W * self.H]
        if any(cells):
            for c, code in enumerate(cells):
                if code:
                    tick = buf[8 + self.pot_slot[c]] - 1 if c in self.pot_slot else None
                    objects[xy[c]] = self._obj(code, xy[c], tick)
        st.players, st.objects = tuple(players), objects
        return st
