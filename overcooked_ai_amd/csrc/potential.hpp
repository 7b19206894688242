// potential.hpp — potential_function: k_potential, k_potential2
// Part of liboc_amd.so: included by oc_amd.hip inside its anonymous namespace, in this order:
//   common, reset, step_predicate, step_table, step_one, step_lut4, rollout_pair, encode, rollout_encode, featurize, potential, shaping.
#pragma once

// ------------------------------------------------------------------------------------------
// k_potential: potential_function (mdp.py:2920-3238), phi(s) of potential-based reward shaping.  One lane per env.
// Everything that depends only on the layout and gamma — the steady-state value, the best completion of every
// ingredient multiset (_get_optimal_possible_recipe, mdp.py:1976-2016), gamma ** k — comes from the per-layout
// record built on the host (overcooked_ai_amd/potential.py); motion costs come from the planner tables of
// k_featurize (+1 for the interact action, planners.py:418-419).  The kernel multiplies and adds those float64
// values in the reference's order with contraction off (__dmul_rn / __dadd_rn), so phi is bit-identical to the
// reference's Python float.  The one order the reference leaves to its runtime — `list(set().union(...))` of the
// partially full pots (mdp.py:1882-1890), which breaks ties of the greedy pot/ingredient matching — is CPython's
// set iteration order, restated in py_set_order below (tuple hash + open addressing, CPython 3.8-3.12).
// ------------------------------------------------------------------------------------------
constexpr int PHI_BYTES = 472 + 8 * 512;
constexpr uint32_t COST_INF = 0xFFFFu;

struct Phi {
    const uint8_t* b;
    __device__ __forceinline__ double f64(int off) const { return *reinterpret_cast<const double*>(b + off); }
    __device__ __forceinline__ int i32(int off) const { return *reinterpret_cast<const int*>(b + off); }
    __device__ __forceinline__ double steady() const { return f64(0); }
    __device__ __forceinline__ double onion_value() const { return f64(8); }
    __device__ __forceinline__ double tomato_value() const { return f64(16); }
    __device__ __forceinline__ uint32_t max_delivery() const { return (uint32_t)i32(24); }
    __device__ __forceinline__ uint32_t max_pickup() const { return (uint32_t)i32(28); }
    __device__ __forceinline__ uint32_t pot_onion() const { return (uint32_t)i32(32); }
    __device__ __forceinline__ uint32_t pot_tomato() const { return (uint32_t)i32(36); }
    __device__ __forceinline__ double sort_value(uint32_t k) const { return f64(40 + 8 * (int)k); }
    __device__ __forceinline__ double opt_value_max1(uint32_t k) const { return f64(168 + 8 * (int)k); }
    __device__ __forceinline__ double value_max1(uint32_t k) const { return f64(296 + 8 * (int)k); }
    __device__ __forceinline__ uint32_t opt_key(uint32_t k) const { return b[424 + k]; }
    __device__ __forceinline__ uint32_t opt_time(uint32_t k) const { return b[440 + k]; }
    __device__ __forceinline__ uint32_t n_serve() const { return b[456]; }  // 255: not listed, scan the terrain
    __device__ __forceinline__ uint32_t serve_cell(uint32_t i) const { return b[457 + i]; }
    __device__ __forceinline__ double pw(uint32_t k) const { return f64(472 + 8 * (int)k); }  // gamma ** k
};

// hash((x, y)) of CPython's tuplehash for two small non-negative ints
__device__ __forceinline__ uint64_t py_tuple2_hash(uint64_t x, uint64_t y) {
    const uint64_t P1 = 11400714785074694791ull, P2 = 14029467366897019727ull, P5 = 2870177450012600261ull;
    uint64_t acc = P5;
    acc += x * P2; acc = (acc << 31) | (acc >> 33); acc *= P1;
    acc += y * P2; acc = (acc << 31) | (acc >> 33); acc *= P1;
    acc += 2ull ^ (P5 ^ 3527539ull);
    return acc == ~0ull ? 1546275796ull : acc;
}

// Iteration order of `set().union(...)` after inserting the n (<= 8) distinct cells in the given order.
__device__ void py_set_order(const uint32_t* cells, int n, uint32_t W, uint32_t* out) {
    uint64_t th[32];
    int tv[32], tv2[32];
    uint64_t th2[32];
    for (int i = 0; i < 32; ++i) { tv[i] = -1; tv2[i] = -1; }
    uint32_t mask = 7u, fill = 0;
    bool grown = false;
    const uint32_t inv_w = 65536u / W + 1u;
    for (int k = 0; k < n; ++k) {
        const uint32_t cy = (cells[k] * inv_w) >> 16, cx = cells[k] - cy * W;
        const uint64_t hash = py_tuple2_hash(cx, cy);
        uint64_t perturb = hash;
        uint32_t i = (uint32_t)hash & mask;
        int* V = grown ? tv2 : tv;
        uint64_t* Hh = grown ? th2 : th;
        for (;;) {  // set_add_entry: keys are distinct, only the free-slot search remains
            uint32_t probes = (i + 9u <= mask) ? 9u : 0u, j = i;
            bool found = false;
            for (uint32_t q = 0; q <= probes; ++q, ++j)
                if (V[j] < 0) { found = true; break; }
            if (found) { V[j] = (int)cells[k]; Hh[j] = hash; break; }
            perturb >>= 5;
            i = (uint32_t)((uint64_t)i * 5u + 1u + perturb) & mask;
        }
        ++fill;
        if (!grown && fill * 5u >= mask * 3u) {  // set_table_resize(so, used * 4): 5 entries -> 32 slots
            for (uint32_t s = 0; s <= mask; ++s) {
                if (tv[s] < 0) continue;
                const uint64_t h = th[s];
                uint64_t pb = h;
                uint32_t ii = (uint32_t)h & 31u;
                for (;;) {  // set_insert_clean
                    uint32_t j = ii;
                    bool found = tv2[j] < 0;
                    if (!found && ii + 9u <= 31u)
                        for (uint32_t q = 0; q < 9u; ++q) { ++j; if (tv2[j] < 0) { found = true; break; } }
                    if (found) { tv2[j] = tv[s]; th2[j] = h; break; }
                    pb >>= 5;
                    ii = (uint32_t)((uint64_t)ii * 5u + 1u + pb) & 31u;
                }
            }
            grown = true;
            mask = 31u;
        }
    }
    const int* V = grown ? tv2 : tv;
    int k = 0;
    for (uint32_t s = 0; s <= mask; ++s)
        if (V[s] >= 0) out[k++] = (uint32_t)V[s];
}

__global__ __launch_bounds__(BLOCK) void k_potential(const OcLayout* __restrict__ g_layouts,
                                                     const uint16_t* __restrict__ layout_id,
                                                     const uint8_t* __restrict__ plan_blob,
                                                     const uint32_t* __restrict__ plan_off,
                                                     const uint8_t* __restrict__ phi_tables,
                                                     const uint4* __restrict__ st, double* __restrict__ out, int64_t n,
                                                     int W, int H) {
#pragma clang fp contract(off)  // __dmul_rn / __dadd_rn are plain * and + in ROCm's headers: keep them unfused
    const int64_t e = (int64_t)blockIdx.x * BLOCK + threadIdx.x;
    if (e >= n) return;
    const uint32_t lid = layout_id ? layout_id[e] : 0u;
    const Lay L{reinterpret_cast<const uint8_t*>(g_layouts + lid)};
    const Phi T{phi_tables + (size_t)lid * PHI_BYTES};
    const uint8_t* plan = plan_blob + plan_off[lid];
    const uint32_t cells = (uint32_t)(W * H);
    const uint4 hw = st[e];
    const uint32_t hdr[4] = {hw.x, hw.y, hw.z, hw.w};
    auto hbyte = [&](uint32_t i) { return (hdr[i >> 2] >> (8u * (i & 3u))) & 0xFFu; };
    auto obj_at = [&](uint32_t c) {
        return (uint32_t) reinterpret_cast<const uint8_t*>(st + (int64_t)(1 + (c >> 4)) * n + e)[c & 15u];
    };
    const uint32_t np = hbyte(3) == 0xFFu ? 1u : 2u;
    uint32_t held[2] = {hbyte(2), np > 1u ? hbyte(5) : 0xFFu};
    const uint8_t* cost_row[2];
    const uint32_t row_stride = (cells + 15u) & ~15u;
    cost_row[0] = plan + 128 + ((uint32_t)plan[hbyte(0)] * 4u + hbyte(1)) * row_stride;
    cost_row[1] = np > 1u ? plan + 128 + ((uint32_t)plan[hbyte(3)] * 4u + hbyte(4)) * row_stride : cost_row[0];
    auto cost = [&](uint32_t p, uint32_t c) {  // min_cost_to_feature(player, [c]); COST_INF = np.inf
        const uint32_t v = cost_row[p][c];
        return v == 255u ? COST_INF : v + 1u;
    };
    const uint32_t max_del = T.max_delivery(), max_pick = T.max_pickup();

    // get_pot_states (mdp.py:1809-1838), pots in get_pot_locations order = slot order
    const uint32_t n_pots = L.n_pots();
    enum { EMPTY = 0, COOKING = 4, READY = 5 };  // 1..3 = idle with that many ingredients
    uint32_t pcell[OC_MAX_POTS], pkey[OC_MAX_POTS], pcls[OC_MAX_POTS], prem[OC_MAX_POTS];
    for (uint32_t k = 0; k < n_pots; ++k) {
        const uint32_t c = L.pot_cell((int)k), o = obj_at(c), tk = hbyte(8u + k);
        const uint32_t key = o ? recipe_idx(o) : 0u, ct = L.cook_time(key), cnt = (o >> 3) & 3u;
        pcell[k] = c; pkey[k] = key;
        pcls[k] = o == 0u ? (uint32_t)EMPTY : tk == 0u ? cnt : (tk - 1u >= ct ? (uint32_t)READY : (uint32_t)COOKING);
        prem[k] = pcls[k] == COOKING ? ct - (tk - 1u) : 0u;  // cook_time - _cooking_tick
    }

    double phi = T.steady();  // mdp.py:2985-3001

    // non-idle soups: cooking then ready, each in pot order, with their default value (mdp.py:3026-3046)
    uint32_t ni[OC_MAX_POTS], n_ni = 0;
    double ni_val[OC_MAX_POTS];
    for (uint32_t cls = COOKING; cls <= READY; ++cls)
        for (uint32_t k = 0; k < n_pots; ++k)
            if (pcls[k] == cls) {
                ni_val[n_ni] = __dmul_rn(T.pw(max_del + max(max_pick, prem[k])), T.value_max1(pkey[k]));
                ni[n_ni++] = k;
            }

    bool has_onion[2] = {held[0] == OC_O_ONION, held[1] == OC_O_ONION};
    bool has_tomato[2] = {held[0] == OC_O_TOMATO, held[1] == OC_O_TOMATO};

    // step 4: players holding a soup walk to the closest serving cell (mdp.py:3078-3090)
    for (uint32_t p = 0; p < np; ++p) {
        if (held[p] == 0xFFu || !(held[p] & OC_O_SOUP)) continue;
        uint32_t d = COST_INF;
        const uint32_t n_serve = T.n_serve();
        if (n_serve != 255u) {
            for (uint32_t i = 0; i < n_serve; ++i) d = min(d, cost(p, T.serve_cell(i)));
        } else {
            for (uint32_t c = 0; c < cells; ++c)
                if ((L.terrain(c) & 7u) == OC_T_SERVE) d = min(d, cost(p, c));
        }
        phi = __dadd_rn(phi, __dmul_rn(T.pw(min(d, max_del)), T.value_max1(recipe_idx(held[p]))));
    }

    // step 3: players holding a dish pursue the non-idle soup that is worth most to them (mdp.py:3092-3133)
    for (uint32_t p = 0; p < np; ++p) {
        if (held[p] != OC_O_DISH) continue;
        int best = -1;
        double best_value = 0.0;
        for (uint32_t i = 0; i < n_ni; ++i) {
            const uint32_t k = ni[i], d = cost(p, pcell[k]);
            const double soup_value = __dmul_rn(T.pw(max_del), T.value_max1(pkey[k]));
            const double value = __dmul_rn(T.pw(max(prem[k], min(d, max_pick))), soup_value);
            if (d != COST_INF && value > best_value) { best = (int)i; best_value = value; }
        }
        if (best >= 0 && best_value > ni_val[best]) ni_val[best] = best_value;
    }
    for (uint32_t i = 0; i < n_ni; ++i) phi = __dadd_rn(phi, ni_val[i]);

    // idle soups: full-but-not-cooking (pot order), then partially full (CPython set order), stable-sorted by the
    // value of their best completion, highest first (mdp.py:3003-3024)
    uint32_t idle[OC_MAX_POTS], n_idle = 0;
    for (uint32_t k = 0; k < n_pots; ++k)
        if (pcls[k] == 3u) idle[n_idle++] = k;
    {
        uint32_t part_cells[OC_MAX_POTS], ordered[OC_MAX_POTS], n_part = 0;
        for (uint32_t items = 1; items < 3u; ++items)
            for (uint32_t k = 0; k < n_pots; ++k)
                if (pcls[k] == items) part_cells[n_part++] = pcell[k];
        if (n_part > 1u) py_set_order(part_cells, (int)n_part, (uint32_t)W, ordered);
        else if (n_part == 1u) ordered[0] = part_cells[0];
        for (uint32_t j = 0; j < n_part; ++j)
            for (uint32_t k = 0; k < n_pots; ++k)
                if (pcell[k] == ordered[j]) idle[n_idle++] = k;
    }
    for (uint32_t a = 1; a < n_idle; ++a) {  // insertion sort, strict compare = Python's stable sorted(reverse=True)
        const uint32_t k = idle[a];
        const double key = T.sort_value(pkey[k]);
        int b = (int)a - 1;
        while (b >= 0 && T.sort_value(pkey[idle[b]]) < key) { idle[b + 1] = idle[b]; --b; }
        idle[b + 1] = k;
    }

    // step 2 (mdp.py:3135-3211)
    for (uint32_t a = 0; a < n_idle; ++a) {
        const uint32_t k = idle[a], key = pkey[k], ok = T.opt_key(key);
        const uint32_t missing_onions = (ok & 3u) - (key & 3u), missing_tomatoes = (ok >> 2) - (key >> 2);
        double disc = T.pw(max(max_pick, T.opt_time(key)) + max_del);
        for (uint32_t j = 0; j < missing_onions + missing_tomatoes; ++j) {
            bool* pertinent = j < missing_onions ? has_onion : has_tomato;
            uint32_t dist = COST_INF;
            int closest = -1;
            for (uint32_t p = 0; p < np; ++p) {
                if (!pertinent[p]) continue;
                const uint32_t cur = cost(p, pcell[k]);
                if (cur < dist) { dist = cur; closest = (int)p; }
            }
            disc = __dmul_rn(disc, T.pw(min(dist, j < missing_onions ? T.pot_onion() : T.pot_tomato())));
            if (closest >= 0) pertinent[closest] = false;
        }
        if (missing_onions + missing_tomatoes) disc = __dmul_rn(disc, T.pw(1));
        else {
            uint32_t cook_dist = COST_INF;
            for (uint32_t p = 0; p < np; ++p)
                if (held[p] == 0u) cook_dist = min(cook_dist, cost(p, pcell[k]));
            disc = __dmul_rn(disc, T.pw(min(cook_dist, max_pick)));
        }
        phi = __dadd_rn(phi, __dmul_rn(disc, T.opt_value_max1(key)));
    }

    // step 1: ingredients left over go to the closest empty pot, tomatoes first (mdp.py:3213-3245)
    for (uint32_t pass = 0; pass < 2u; ++pass) {
        const bool* holding = pass == 0u ? has_tomato : has_onion;
        for (uint32_t p = 0; p < np; ++p) {
            if (!holding[p]) continue;
            uint32_t dist = COST_INF;
            for (uint32_t k = 0; k < n_pots; ++k)
                if (pcls[k] == EMPTY) dist = min(dist, cost(p, pcell[k]));
            if (dist == COST_INF) continue;  // is_useful == 0: adds 0.0
            const double disc = T.pw(min(pass == 0u ? T.pot_tomato() : T.pot_onion(), dist) + max_pick + max_del);
            phi = __dadd_rn(phi, __dmul_rn(disc, pass == 0u ? T.tomato_value() : T.onion_value()));
        }
    }
    out[e] = phi;
}

// k_potential2: the same function for layout tables with at most two pots (every layout the reference ships).
// The pot lists of the reference (non-idle soups: cooking then ready; idle soups: full, then partially full in set
// order, stable-sorted by the value of their best completion) have at most two entries, so they are kept as
// (first, second) pairs in registers and the matching loops are straight-line code; the set order of two partially
// full pots is precomputed per layout on the host (two bits in the record: which pot comes out first for either
// insertion order).  All motion costs are fetched up front so that their latencies overlap.  potential2_core works
// from registers so that the fused training step (shaping.hpp) can call it on the state it has just computed.
// phi of one env whose dynamic state is already in registers: the players (cell, orientation, held; n_players = 1 or
// 2), the object byte and tick byte (cooking_tick + 1) of the first two pot slots.
__device__ __forceinline__ double potential2_core(const Lay L, const Phi T, const uint8_t* __restrict__ plan,
                                                  uint32_t cells, uint32_t np, uint32_t pos0, uint32_t or0, uint32_t held0,
                                                  uint32_t pos1, uint32_t or1, uint32_t held1_in, uint32_t oA, uint32_t oB,
                                                  uint32_t tkA, uint32_t tkB) {
#pragma clang fp contract(off)
    const uint32_t row_stride = (cells + 15u) & ~15u;
    const uint32_t held1 = np > 1u ? held1_in : 0xFFu;
    const uint8_t* row0 = plan + 128 + ((uint32_t)plan[pos0] * 4u + or0) * row_stride;
    const uint8_t* row1 = np > 1u ? plan + 128 + ((uint32_t)plan[pos1] * 4u + or1) * row_stride : row0;
    const uint32_t n_pots = L.n_pots();
    const uint32_t cellA = L.pot_cell(0), cellB = n_pots > 1u ? L.pot_cell(1) : cellA;
    // everything that comes from memory, issued together
    const uint32_t rA0 = row0[cellA], rA1 = row1[cellA], rB0 = row0[cellB], rB1 = row1[cellB];
    uint32_t serve0 = 255u, serve1 = 255u;  // min over the serving cells (255 = unreachable stays the maximum)
    const uint32_t n_serve = T.n_serve();
    if (n_serve != 255u) {  // the host listed them: no terrain scan
        for (uint32_t i = 0; i < n_serve; ++i) {
            const uint32_t c = T.serve_cell(i);
            serve0 = min(serve0, (uint32_t)row0[c]); serve1 = min(serve1, (uint32_t)row1[c]);
        }
    } else {
        for (uint32_t c = 0; c < cells; ++c)
            if ((L.terrain(c) & 7u) == OC_T_SERVE) { serve0 = min(serve0, (uint32_t)row0[c]); serve1 = min(serve1, (uint32_t)row1[c]); }
    }
    auto fin = [](uint32_t v) { return v == 255u ? COST_INF : v + 1u; };  // + the interact; COST_INF = np.inf
    // cost[player][pot]
    const uint32_t cA[2] = {fin(rA0), fin(rA1)}, cB[2] = {fin(rB0), fin(rB1)};
    const uint32_t dserve[2] = {fin(serve0), fin(serve1)};
    const uint32_t held[2] = {held0, held1};
    const uint32_t max_del = T.max_delivery(), max_pick = T.max_pickup();

    enum { EMPTY = 0, COOKING = 4, READY = 5, ABSENT = 7 };
    auto classify = [&](uint32_t o, uint32_t tk, uint32_t& key, uint32_t& rem) {
        key = o ? recipe_idx(o) : 0u;
        const uint32_t ct = L.cook_time(key), cnt = (o >> 3) & 3u;
        const uint32_t cls = o == 0u ? (uint32_t)EMPTY : tk == 0u ? cnt : (tk - 1u >= ct ? (uint32_t)READY : (uint32_t)COOKING);
        rem = cls == COOKING ? ct - (tk - 1u) : 0u;
        return cls;
    };
    uint32_t key[2], rem[2], cls[2];
    cls[0] = classify(oA, tkA, key[0], rem[0]);
    cls[1] = classify(oB, tkB, key[1], rem[1]);
    if (n_pots < 2u) cls[1] = ABSENT;
    auto pcost = [&](uint32_t p, uint32_t k) { return k == 0u ? cA[p] : cB[p]; };

    double phi = T.steady();

    // non-idle soups: cooking before ready, pot order inside a class (mdp.py:3026-3046)
    const bool ni0 = cls[0] == COOKING || cls[0] == READY, ni1 = cls[1] == COOKING || cls[1] == READY;
    const bool swap_ni = ni0 && ni1 && cls[0] == READY && cls[1] == COOKING;
    const uint32_t n_ni = (ni0 ? 1u : 0u) + (ni1 ? 1u : 0u);
    uint32_t nk[2];  // pots in list order
    nk[0] = ni0 ? (swap_ni ? 1u : 0u) : 1u;
    nk[1] = swap_ni ? 0u : 1u;
    double nv[2];
#pragma unroll
    for (int i = 0; i < 2; ++i) nv[i] = __dmul_rn(T.pw(max_del + max(max_pick, rem[nk[i]])), T.value_max1(key[nk[i]]));

    bool has_onion[2] = {held[0] == OC_O_ONION, held[1] == OC_O_ONION};
    bool has_tomato[2] = {held[0] == OC_O_TOMATO, held[1] == OC_O_TOMATO};

    // step 4 (mdp.py:3078-3090)
#pragma unroll
    for (uint32_t p = 0; p < 2u; ++p) {
        if (p >= np || held[p] == 0xFFu || !(held[p] & OC_O_SOUP)) continue;
        phi = __dadd_rn(phi, __dmul_rn(T.pw(min(dserve[p], max_del)), T.value_max1(recipe_idx(held[p]))));
    }
    // step 3 (mdp.py:3092-3133)
#pragma unroll
    for (uint32_t p = 0; p < 2u; ++p) {
        if (p >= np || held[p] != OC_O_DISH) continue;
        int best = -1;
        double best_value = 0.0;
#pragma unroll
        for (uint32_t i = 0; i < 2u; ++i) {
            if (i >= n_ni) continue;
            const uint32_t k = nk[i], d = pcost(p, k);
            const double soup_value = __dmul_rn(T.pw(max_del), T.value_max1(key[k]));
            const double value = __dmul_rn(T.pw(max(rem[k], min(d, max_pick))), soup_value);
            if (d != COST_INF && value > best_value) { best = (int)i; best_value = value; }
        }
        if (best == 0 && best_value > nv[0]) nv[0] = best_value;
        if (best == 1 && best_value > nv[1]) nv[1] = best_value;
    }
    if (n_ni > 0u) phi = __dadd_rn(phi, nv[0]);
    if (n_ni > 1u) phi = __dadd_rn(phi, nv[1]);

    // idle soups in processing order (mdp.py:3003-3024)
    const bool id0 = cls[0] >= 1u && cls[0] <= 3u, id1 = cls[1] >= 1u && cls[1] <= 3u;
    const uint32_t n_idle = (id0 ? 1u : 0u) + (id1 ? 1u : 0u);
    bool b_first = false;  // with both idle: does pot B precede pot A before the sort?
    if (id0 && id1) {
        if (cls[0] == 3u || cls[1] == 3u) b_first = cls[0] != 3u;  // full pots first, pot order among them
        else {
            const uint32_t bits = T.opt_key(0);  // host-computed CPython set order of the two pot positions
            b_first = (cls[0] == 2u && cls[1] == 1u) ? (bits & 2u) != 0u : (bits & 1u) != 0u;
        }
        const double sa = T.sort_value(key[0]), sb = T.sort_value(key[1]);
        if (b_first ? sa > sb : sb > sa) b_first = !b_first;  // stable descending sort of two
    }
    uint32_t ik[2];
    ik[0] = id0 ? (b_first ? 1u : 0u) : 1u;
    ik[1] = b_first ? 0u : 1u;
    // step 2 (mdp.py:3135-3211)
#pragma unroll
    for (uint32_t a = 0; a < 2u; ++a) {
        if (a >= n_idle) continue;
        const uint32_t k = ik[a], kk = key[k], ok = T.opt_key(kk);
        const uint32_t missing_onions = (ok & 3u) - (kk & 3u), missing_tomatoes = (ok >> 2) - (kk >> 2);
        double disc = T.pw(max(max_pick, T.opt_time(kk)) + max_del);
        for (uint32_t j = 0; j < missing_onions + missing_tomatoes; ++j) {
            const bool onion = j < missing_onions;
            const bool av0 = onion ? has_onion[0] : has_tomato[0], av1 = (np > 1u) && (onion ? has_onion[1] : has_tomato[1]);
            uint32_t dist = COST_INF;
            int closest = -1;
            if (av0 && pcost(0, k) < dist) { dist = pcost(0, k); closest = 0; }
            if (av1 && pcost(1, k) < dist) { dist = pcost(1, k); closest = 1; }
            disc = __dmul_rn(disc, T.pw(min(dist, onion ? T.pot_onion() : T.pot_tomato())));
            if (closest == 0) { if (onion) has_onion[0] = false; else has_tomato[0] = false; }
            if (closest == 1) { if (onion) has_onion[1] = false; else has_tomato[1] = false; }
        }
        if (missing_onions + missing_tomatoes) disc = __dmul_rn(disc, T.pw(1));
        else {
            uint32_t cook_dist = COST_INF;
            if (held[0] == 0u) cook_dist = min(cook_dist, pcost(0, k));
            if (np > 1u && held[1] == 0u) cook_dist = min(cook_dist, pcost(1, k));
            disc = __dmul_rn(disc, T.pw(min(cook_dist, max_pick)));
        }
        phi = __dadd_rn(phi, __dmul_rn(disc, T.opt_value_max1(kk)));
    }
    // step 1 (mdp.py:3213-3245)
#pragma unroll
    for (uint32_t pass = 0; pass < 2u; ++pass) {
#pragma unroll
        for (uint32_t p = 0; p < 2u; ++p) {
            if (p >= np || !(pass == 0u ? has_tomato[p] : has_onion[p])) continue;
            uint32_t dist = COST_INF;
            if (cls[0] == EMPTY) dist = min(dist, cA[p]);
            if (cls[1] == EMPTY) dist = min(dist, cB[p]);
            if (dist == COST_INF) continue;
            const double disc = T.pw(min(pass == 0u ? T.pot_tomato() : T.pot_onion(), dist) + max_pick + max_del);
            phi = __dadd_rn(phi, __dmul_rn(disc, pass == 0u ? T.tomato_value() : T.onion_value()));
        }
    }
    return phi;
}

__global__ __launch_bounds__(BLOCK) void k_potential2(const OcLayout* __restrict__ g_layouts,
                                                      const uint16_t* __restrict__ layout_id,
                                                      const uint8_t* __restrict__ plan_blob,
                                                      const uint32_t* __restrict__ plan_off,
                                                      const uint8_t* __restrict__ phi_tables,
                                                      const uint4* __restrict__ st, double* __restrict__ out, int64_t n,
                                                      int W, int H) {
    const int64_t e = (int64_t)blockIdx.x * BLOCK + threadIdx.x;
    if (e >= n) return;
    const uint32_t lid = layout_id ? layout_id[e] : 0u;
    const Lay L{reinterpret_cast<const uint8_t*>(g_layouts + lid)};
    const Phi T{phi_tables + (size_t)lid * PHI_BYTES};
    const uint4 hw = st[e];
    const uint32_t np = (hw.x >> 24) == 0xFFu ? 1u : 2u;
    const uint32_t cellA = L.pot_cell(0), cellB = L.n_pots() > 1u ? L.pot_cell(1) : cellA;
    const uint32_t oA = reinterpret_cast<const uint8_t*>(st + (int64_t)(1 + (cellA >> 4)) * n + e)[cellA & 15u];
    const uint32_t oB = reinterpret_cast<const uint8_t*>(st + (int64_t)(1 + (cellB >> 4)) * n + e)[cellB & 15u];
    out[e] = potential2_core(L, T, plan_blob + plan_off[lid], (uint32_t)(W * H), np, hw.x & 0xFFu, (hw.x >> 8) & 0xFFu,
                             (hw.x >> 16) & 0xFFu, np > 1u ? hw.x >> 24 : 0u, hw.y & 0xFFu, (hw.y >> 8) & 0xFFu, oA, oB,
                             hw.z & 0xFFu, (hw.z >> 8) & 0xFFu);
}
