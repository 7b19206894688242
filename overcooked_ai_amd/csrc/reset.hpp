// reset.hpp — start states: k_reset, k_reset_random
// Part of liboc_amd.so: included by oc_amd.hip inside its anonymous namespace, in this order:
//   common, reset, step_predicate, step_table, step_one, step_lut4, rollout_pair, encode, rollout_encode, featurize, potential, shaping.
#pragma once

// ------------------------------------------------------------------------------------------
// k_reset
// ------------------------------------------------------------------------------------------
template <int NOBJ>
__global__ __launch_bounds__(BLOCK) void k_reset(const OcLayout* __restrict__ g_layouts,
                                                 const uint16_t* __restrict__ layout_id, uint4* st,
                                                 const uint8_t* __restrict__ mask, float4* __restrict__ ep_returns,
                                                 int64_t n) {
    const int64_t e = (int64_t)blockIdx.x * BLOCK + threadIdx.x;
    if (e >= n) return;
    if (mask && !mask[e]) return;
    const uint32_t lid = layout_id ? layout_id[e] : 0u;
    const uint8_t* base = reinterpret_cast<const uint8_t*>(g_layouts) + (size_t)lid * 256u;
    const uint32_t pos0 = base[L_START_POS], pos1 = base[L_START_POS + 1];
    const uint32_t or0 = base[L_START_OR], or1 = pos1 == 0xFFu ? 0u : base[L_START_OR + 1];
    st[e] = make_uint4(pos0 | (or0 << 8) | (pos1 << 24), or1, 0u, 0u);
    const uint4 z = make_uint4(0, 0, 0, 0);
#pragma unroll
    for (int p = 0; p < NOBJ; ++p) st[(int64_t)(1 + p) * n + e] = z;
    if (ep_returns) ep_returns[e] = make_float4(0.f, 0.f, 0.f, 0.f);
}

// ------------------------------------------------------------------------------------------
// k_reset_random: get_random_start_state_fn (mdp.py:1307-1369) drawn in-kernel.  Same distribution as the reference
// — joint player positions uniform over ordered tuples of distinct free cells (when random_start_pos), every pot
// non-empty with probability thresh (1..3 onions, then 0..3-n tomatoes, cooking from tick 0 with probability thresh,
// else idle), every player holding something with probability thresh (dish 0.2 / onion 0.6 / finished soup 0.2 with
// the same ingredient draw) — but from this library's own counter-based stream instead of numpy's global
// generator: block b of env g at reset epoch ep is philox4x32_10({ep, g_lo, g_hi, b}, {seed_lo, seed_hi ^ "RST!"});
// block 0 word 0 picks the joint position, block 1 + i = player i {p, kind, n, m}, block 3 + k = pot k {p, n, m, q}.
// Integer maps: "u < thresh" is word < floor(thresh * 2^32); randint(lo, hi) is lo + mulhi(word, hi - lo).
// oracle_reset_random restates it; overcooked_ai_amd.mdp.get_random_start_state_fn keeps the numpy-exact host path.
// ------------------------------------------------------------------------------------------
constexpr uint32_t RESET_KEY_TWEAK = 0x52535421u;  // "RST!"

// What a randomized start consists of; players face NORTH, nothing lies on the counters, timestep 0.
struct StartDraw {
    uint32_t pos0, pos1;      // cells (pos1 = 0xFF on one-player layouts)
    uint32_t held0, held1;    // object codes
    uint32_t ticks0, ticks1;  // header bytes 8..15: cooking_tick + 1 per pot slot
    uint32_t pots0, pots1;    // soup code per pot slot, one byte each (0 = empty)
    // (plain words, no arrays: a dynamically indexed member array becomes a stack object — scratch in the step kernels, LDS
    //  plus a read of the dispatch packet in host memory, ~10 us per launch, in k_reset_random; tools/launch_floor.hip)
    __device__ __forceinline__ uint32_t held(uint32_t i) const { return i ? held1 : held0; }
    __device__ __forceinline__ uint32_t tick(uint32_t k) const { return ((k < 4u ? ticks0 : ticks1) >> (8u * (k & 3u))) & 0xFFu; }
    __device__ __forceinline__ uint32_t pot_obj(uint32_t k) const { return ((k < 4u ? pots0 : pots1) >> (8u * (k & 3u))) & 0xFFu; }
};

// (StartArgs — the launch-time form of OcStartSpec — and EvArgs are defined in shared.hpp: they cross translation units)

// The layout of an env's NEXT episode (OvercookedEnv.reset(regen_mdp=True) with a generator that returns a different mdp
// every time, env.py:288-302; layout_generator.py:110-160): block 15 of the reset stream documented above, word 0.
constexpr uint32_t REGEN_BLOCK = 15u;
__device__ __forceinline__ uint32_t draw_layout(const StartArgs& sa, uint64_t g, uint32_t epoch) {
    uint32_t r[4];
    philox4x32_10(epoch, (uint32_t)g, (uint32_t)(g >> 32), REGEN_BLOCK, sa.seed_lo, sa.seed_hi ^ 0x52535421u, r);
    return sa.regen_first + __umulhi(r[0], sa.regen_count);
}

constexpr int N_EVENT_TYPES = 25;

// add one step's events to the env's counters; at the end of an episode publish them and start from zero
__device__ __forceinline__ void count_events(const EvArgs& ea, int64_t e, uint64_t ev, bool episode_ends, bool restarts) {
    if (!ea.counts) return;
    uint32_t* c = ea.counts + e * N_EVENT_TYPES;
    while (ev) {
        const int b = __ffsll((long long)ev) - 1;
        c[b >> 1] += 1u << (16 * (b & 1));  // this lane owns the env: a plain read-modify-write
        ev &= ev - 1ull;
    }
    if (episode_ends) {
        for (int k = 0; k < N_EVENT_TYPES; ++k) {
            if (ea.counts_done) ea.counts_done[e * N_EVENT_TYPES + k] = c[k];
            if (restarts || ea.clear_on_done) c[k] = 0;
        }
    }
}

__device__ __forceinline__ StartDraw draw_start(const Lay L, uint64_t g, uint32_t epoch, uint32_t seed_lo, uint32_t seed_hi,
                                                int random_start_pos, uint64_t thresh) {
    StartDraw d;
    const uint32_t g_lo = (uint32_t)g, g_hi = (uint32_t)(g >> 32);
    const uint32_t k1 = seed_hi ^ RESET_KEY_TWEAK;
    const uint32_t cells = L.u8(L_NCELLS), np = L.n_players();
    uint32_t r[4];
    d.pos0 = L.u8(L_START_POS); d.pos1 = L.u8(L_START_POS + 1);
    if (random_start_pos) {
        uint32_t n_floor = 0;
        for (uint32_t c = 0; c < cells; ++c) n_floor += (L.terrain(c) & 7u) == OC_T_FLOOR ? 1u : 0u;
        philox4x32_10(epoch, g_lo, g_hi, 0u, seed_lo, k1, r);
        // index into itertools.product(valid, repeat=n_players) without overlaps (mdp.py:1736-1747), row-major
        const uint32_t n_joint = np == 2u ? n_floor * (n_floor - 1u) : n_floor;
        const uint32_t idx = __umulhi(r[0], n_joint);
        uint32_t a = idx, b = 0xFFFFFFFFu;
        if (np == 2u) {
            a = idx / (n_floor - 1u);
            b = idx - a * (n_floor - 1u);
            b += b >= a ? 1u : 0u;
        }
        uint32_t seen = 0;
        for (uint32_t c = 0; c < cells; ++c) {
            if ((L.terrain(c) & 7u) != OC_T_FLOOR) continue;
            if (seen == a) d.pos0 = c;
            if (seen == b) d.pos1 = c;
            ++seen;
        }
    }
    d.held0 = d.held1 = 0u;
    d.ticks0 = d.ticks1 = 0u;
    d.pots0 = d.pots1 = 0u;
    const uint32_t n_pots = L.n_pots();
    auto soup_code = [](uint32_t n_on, uint32_t n_to) {  // onions first, then tomatoes (SoupState.get_soup, mdp.py:664-693)
        return OC_O_SOUP | ((n_on + n_to) << 3) | (((1u << n_to) - 1u) << n_on);
    };
    if (thresh != 0ull) {
        for (uint32_t i = 0; i < np; ++i) {
            philox4x32_10(epoch, g_lo, g_hi, 1u + i, seed_lo, k1, r);
            if ((uint64_t)r[0] < thresh) {
                const uint32_t n_on = 1u + __umulhi(r[2], 3u), n_to = __umulhi(r[3], 4u - n_on);
                const uint32_t h = r[1] < 858993459u ? (uint32_t)OC_O_DISH : r[1] < 3435973836u ? (uint32_t)OC_O_ONION : soup_code(n_on, n_to);
                if (i) d.held1 = h; else d.held0 = h;
            }
        }
        for (uint32_t k = 0; k < n_pots; ++k) {
            philox4x32_10(epoch, g_lo, g_hi, 3u + k, seed_lo, k1, r);
            if ((uint64_t)r[0] < thresh) {
                const uint32_t n_on = 1u + __umulhi(r[1], 3u), n_to = __umulhi(r[2], 4u - n_on);
                const uint32_t sh = 8u * (k & 3u), code = (soup_code(n_on, n_to) & 0xFFu) << sh;
                const uint32_t tick = ((uint64_t)r[3] < thresh) ? 1u << sh : 0u;  // cooking_tick 0 -> stored 1
                if (k < 4u) { d.pots0 |= code; d.ticks0 |= tick; } else { d.pots1 |= code; d.ticks1 |= tick; }
            }
        }
    }
    if (np < 2u) d.pos1 = 0xFFu;
    return d;
}

__global__ __launch_bounds__(BLOCK) void k_reset_random(const OcLayout* __restrict__ g_layouts,
                                                        const uint16_t* __restrict__ layout_id, uint4* st,
                                                        const uint8_t* __restrict__ mask,
                                                        float4* __restrict__ ep_returns, int64_t n, int n_obj,
                                                        uint32_t seed_lo, uint32_t seed_hi, int64_t env_offset,
                                                        uint32_t epoch, int random_start_pos, uint64_t thresh) {
    const int64_t e = (int64_t)blockIdx.x * BLOCK + threadIdx.x;
    if (e >= n) return;
    if (mask && !mask[e]) return;
    const uint32_t lid = layout_id ? layout_id[e] : 0u;
    const Lay L{reinterpret_cast<const uint8_t*>(g_layouts) + (size_t)lid * 256u};
    const StartDraw d = draw_start(L, (uint64_t)(env_offset + e), epoch, seed_lo, seed_hi, random_start_pos, thresh);
    const uint32_t np = L.n_players(), n_pots = L.n_pots();
    st[e] = make_uint4(d.pos0 | (d.held0 << 16) | (d.pos1 << 24), np == 2u ? (d.held1 << 8) : 0u, d.ticks0, d.ticks1);
    for (int p = 0; p < n_obj; ++p) {
        uint32_t w0 = 0u, w1 = 0u, w2 = 0u, w3 = 0u;
        for (uint32_t k = 0; k < n_pots; ++k) {
            const uint32_t c = L.pot_cell((int)k), q = (c >> 2) & 3u;
            const uint32_t v = (int)(c >> 4) == p ? d.pot_obj(k) << (8u * (c & 3u)) : 0u;
            w0 |= q == 0u ? v : 0u; w1 |= q == 1u ? v : 0u; w2 |= q == 2u ? v : 0u; w3 |= q == 3u ? v : 0u;
        }
        st[(int64_t)(1 + p) * n + e] = make_uint4(w0, w1, w2, w3);
    }
    if (ep_returns) ep_returns[e] = make_float4(0.f, 0.f, 0.f, 0.f);
}

// k_regen_layouts: new layout ids for the selected envs (explicit resets with regen_mdp semantics; the caller then resets
// those envs' states with oc_reset / oc_reset_random under the same mask)
__global__ __launch_bounds__(BLOCK) void k_regen_layouts(uint16_t* layout_id, const uint8_t* __restrict__ mask, uint8_t mask_bits,
                                                         int64_t n, StartArgs sa) {
    const int64_t e = (int64_t)blockIdx.x * BLOCK + threadIdx.x;
    if (e >= n) return;
    if (mask && !(mask[e] & mask_bits)) return;
    layout_id[e] = (uint16_t)draw_layout(sa, (uint64_t)(sa.env_offset + e), sa.epoch);
}
