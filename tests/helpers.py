"""Shared test utilities: random valid packed states, layout dict helpers."""
import numpy as np

from overcooked_ai_amd import layouts as L

CANONICAL_5 = ["cramped_room", "asymmetric_advantages", "coordination_ring", "forced_coordination", "counter_circuit"]


def random_packed_states(spec, n, rng, timestep_max=399, counter_fill=None):
    """Random VALID states of `spec` in the packed wire format [n_planes, n, 16] (include/oc_amd.h):
    players on distinct floor cells, random held objects (cooked soups only), random objects on counters (each with
    probability 0 / 0.1 / 0.35 drawn per env, or `counter_fill(e)` when given), pots empty / idle / cooking / ready with
    ticks within the recipe's cook time."""
    W, H = spec.width, spec.height
    n_planes = 1 + (W * H + 15) // 16
    out = np.zeros((n_planes, n, 16), np.uint8)
    floor = [y * W + x for (x, y) in spec.cells_of(" ")]
    counters = [y * W + x for (x, y) in spec.cells_of("X")]
    pots = [y * W + x for (x, y) in spec.cells_of("P")]
    has_tomato = bool(spec.cells_of("T"))

    def soup_code():
        k = int(rng.integers(1, 4))
        bits = int(rng.integers(0, 1 << k)) if (has_tomato or rng.random() < 0.2) else 0
        return 0x80 | (k << 3) | bits, k, bits

    def loose_obj():
        r = rng.random()
        if r < 0.3:
            return 1
        if r < 0.45:
            return 2
        if r < 0.75:
            return 3
        return soup_code()[0]

    for e in range(n):
        cells = rng.choice(len(floor), size=spec.num_players, replace=False)
        for p in range(spec.num_players):
            out[0, e, 3 * p] = floor[int(cells[p])]
            out[0, e, 3 * p + 1] = rng.integers(0, 4)
            out[0, e, 3 * p + 2] = loose_obj() if rng.random() < 0.6 else 0
        if spec.num_players == 1:
            out[0, e, 3] = 0xFF
        t = int(rng.integers(0, timestep_max + 1))
        out[0, e, 6], out[0, e, 7] = t & 0xFF, t >> 8
        p_counter = rng.choice([0.0, 0.1, 0.35]) if counter_fill is None else counter_fill(e)
        for c in counters:
            if rng.random() < p_counter:
                out[1 + (c >> 4), e, c & 15] = loose_obj()
        for k, c in enumerate(pots):
            if rng.random() < 0.25:
                continue
            code, cnt, bits = soup_code()
            n_t = bin(bits).count("1")
            ct = int(spec.recipe_time((cnt - n_t, n_t)))
            r = rng.random()
            tick = -1 if r < 0.4 else ct if r < 0.6 else max(0, ct - 1) if r < 0.75 else int(rng.integers(0, ct + 1))
            out[1 + (c >> 4), e, c & 15] = code
            out[0, e, 8 + k] = tick + 1
    return out


def spec_from_fixture(layout_dict):
    return L.LayoutSpec(layout_dict)


ROLLOUT_KERNELS = ("default", "lane_pair", "predicate_interact")


def select_kernel(env, kernel):
    """Pick the rollout kernel family a VecOvercookedEnv launches: "default" / None = k_rollout4 (what ships), the others
    are the cross-check families (the lane-pair kernel, the predicate-network kernel)."""
    assert kernel in (None, "step") + ROLLOUT_KERNELS, kernel
    for name in ROLLOUT_KERNELS[1:]:
        setattr(env, name, name == kernel)
