// step_predicate.hpp — get_state_transition with the predicate-network interact (emits event_infos): k_step, k_rollout
// Part of liboc_amd.so: included by oc_amd.hip inside its anonymous namespace, in this order:
//   common, reset, step_predicate, step_table, step_one, step_lut4, rollout_pair, encode, rollout_encode, featurize, potential, shaping.
#pragma once

// EVENT_TYPES bit helpers (mdp.py:1027-1058): bit 2*k + player
enum {
    EV_TOMATO_PICKUP = 0, EV_USEFUL_TOMATO_PICKUP, EV_TOMATO_DROP, EV_USEFUL_TOMATO_DROP, EV_POTTING_TOMATO,
    EV_ONION_PICKUP, EV_USEFUL_ONION_PICKUP, EV_ONION_DROP, EV_USEFUL_ONION_DROP, EV_POTTING_ONION,
    EV_DISH_PICKUP, EV_USEFUL_DISH_PICKUP, EV_DISH_DROP, EV_USEFUL_DISH_DROP,
    EV_SOUP_PICKUP, EV_SOUP_DELIVERY, EV_SOUP_DROP,
    EV_OPTIMAL_ONION_POTTING, EV_OPTIMAL_TOMATO_POTTING, EV_VIABLE_ONION_POTTING, EV_VIABLE_TOMATO_POTTING,
    EV_CATASTROPHIC_ONION_POTTING, EV_CATASTROPHIC_TOMATO_POTTING, EV_USELESS_ONION_POTTING, EV_USELESS_TOMATO_POTTING
};
__device__ __forceinline__ uint64_t evbit(bool cond, int k, int p) { return cond ? (1ull << (2 * k + p)) : 0ull; }

// ------------------------------------------------------------------------------------------
// event_infos of ONE player's interact (log_object_pickup / drop / potting and their usefulness predicates,
// mdp.py:2121-2308; EVENT_TYPES, mdp.py:1027-1058) from the outcome of the interact, whichever formulation produced it:
//   type     terrain type of the faced cell,  h / o  hand and faced object BEFORE the interact (o: the soup for a pot)
//   swapX    counter pick-up or drop,  take  something taken from a dispenser,  place / plate / serve  as named
//   other_h  the other player's LIVE hand,  dish_useful  is_dish_pickup_useful for this player,  n_full  pots that are
//   cooking, ready or hold max_num_ingredients items (pot_states before any interact),  n_pots  pots of the layout
// ------------------------------------------------------------------------------------------
template <int P>
__device__ __forceinline__ uint64_t interact_events(const LayC& C, uint32_t type, uint32_t h, uint32_t o, bool swapX,
                                                    bool take, bool place, bool plate, bool serve, uint32_t other_h,
                                                    bool dish_useful, uint32_t n_full, bool two) {
    const bool hz = h == 0u;
    const uint32_t n = (o >> 3) & 3u;
    const bool isD = type == OC_T_DISH_DISP;
    const bool all_full = C.n_pots == n_full;
    const bool other_dish = other_h == OC_O_DISH, other_onion = other_h == OC_O_ONION;
    const bool ing_pick_useful = two & !(all_full & !other_dish);
    const bool ing_drop_useful = two & all_full & !other_dish;
    const bool dish_drop_useful = two & (n_full == 0u) & !other_onion;
    const bool pickX = swapX & hz, dropX = swapX & !hz;
    const bool takeO = take & (type == OC_T_ONION_DISP), takeD = take & isD;
    uint64_t e = 0;
    const bool pk_on = (pickX & (o == OC_O_ONION)) | takeO, pk_to = pickX & (o == OC_O_TOMATO);
    const bool pk_di = (pickX & (o == OC_O_DISH)) | takeD;
    e |= evbit(pk_on, EV_ONION_PICKUP, P) | evbit(pk_on & ing_pick_useful, EV_USEFUL_ONION_PICKUP, P);
    e |= evbit(pk_to, EV_TOMATO_PICKUP, P) | evbit(pk_to & ing_pick_useful, EV_USEFUL_TOMATO_PICKUP, P);
    e |= evbit(pk_di, EV_DISH_PICKUP, P) | evbit(pk_di & dish_useful, EV_USEFUL_DISH_PICKUP, P);
    e |= evbit((pickX & ((o & OC_O_SOUP) != 0u)) | plate, EV_SOUP_PICKUP, P);
    e |= evbit(dropX & (h == OC_O_ONION), EV_ONION_DROP, P) | evbit(dropX & (h == OC_O_ONION) & ing_drop_useful, EV_USEFUL_ONION_DROP, P);
    e |= evbit(dropX & (h == OC_O_TOMATO), EV_TOMATO_DROP, P) | evbit(dropX & (h == OC_O_TOMATO) & ing_drop_useful, EV_USEFUL_TOMATO_DROP, P);
    e |= evbit(dropX & (h == OC_O_DISH), EV_DISH_DROP, P) | evbit(dropX & (h == OC_O_DISH) & dish_drop_useful, EV_USEFUL_DISH_DROP, P);
    e |= evbit(dropX & ((h & OC_O_SOUP) != 0u), EV_SOUP_DROP, P);
    e |= evbit(serve, EV_SOUP_DELIVERY, P);
    // potting: class nibble of (old soup, ingredient): 1 optimal, 2 viable, 4 catastrophic, 8 useless
    const uint32_t nt = __popc(o & 7u), no = n - nt, pi = no + 3u * nt;  // old soup has <= 2 ingredients here
    const uint32_t pw = pi < 4u ? C.pclass[0] : C.pclass[1];
    const uint32_t tom = h == OC_O_TOMATO ? 1u : 0u;
    const uint32_t nib = (pw >> (8u * (pi & 3u) + 4u * tom)) & 0xFu;
    e |= evbit(place, EV_POTTING_ONION, P) >> (10u * tom);  // EV_POTTING_TOMATO = EV_POTTING_ONION - 5
    e |= evbit(place & ((nib & 1u) != 0u), EV_OPTIMAL_ONION_POTTING, P) << (2u * tom);
    e |= evbit(place & ((nib & 2u) != 0u), EV_VIABLE_ONION_POTTING, P) << (2u * tom);
    e |= evbit(place & ((nib & 4u) != 0u), EV_CATASTROPHIC_ONION_POTTING, P) << (2u * tom);
    e |= evbit(place & ((nib & 8u) != 0u), EV_USELESS_ONION_POTTING, P) << (2u * tom);
    return e;
}

// ------------------------------------------------------------------------------------------
// INTERACT of player P (resolve_interacts, mdp.py:1432-1579) as a pure function of its inputs, written
// without data-dependent branches: every outcome is a predicate, the new hand / cell / tick are selects.
//   h, other_h   this player's hand and the other player's LIVE hand
//   dcount       live number of loose dishes on counters
//   cell16       LDS word of the faced cell (object | terrain << 8)
//   ps, tk       pot registers the player sees
// ------------------------------------------------------------------------------------------
struct IOut {
    uint32_t new_h;    // hand after the interact
    uint32_t cell_obj; // object byte the faced cell holds afterwards (unchanged unless swapX)
    uint32_t slot, new_o, new_tk;  // pot slot touched and its registers afterwards (valid when pot_upd)
    bool swapX, pot_upd;
    bool take_dish;    // a dish was taken from a dispenser (its shaped reward may be added by the caller)
    int32_t ddelta;    // change of the loose-dish count
    float sparse, shaped;
    uint64_t ev;
};

template <int MAXP, bool EVENTS, int P, bool DEFER_DISH = false>
__device__ __forceinline__ IOut interact(const LayC& C, const Lay L, bool act, uint32_t h, uint32_t other_h,
                                         int32_t dcount, uint32_t cell16, const uint32_t (&ps)[MAXP],
                                         const uint32_t (&tkr)[MAXP], uint32_t useful_pots, uint32_t n_full, bool two) {
    IOut r;
    const uint32_t tc = cell16 >> 8;
    const uint32_t type = act ? (tc & 7u) : 7u;  // 7 matches no terrain: a lane that does not interact falls through
    const uint32_t slot = tc >> 3;
    const bool isP = type == OC_T_POT;
    uint32_t tkv = 0, pso = 0;
#pragma unroll
    for (int k = 0; k < MAXP; ++k) {
        const bool sel = slot == (uint32_t)k;
        tkv = sel ? tkr[k] : tkv;
        pso = sel ? ps[k] : pso;
    }
    const uint32_t o_cell = cell16 & 0xFFu;
    const uint32_t o = isP ? pso : o_cell;
    const uint32_t tk = isP ? tkv : 0u;
    const bool hz = h == 0u, oz = o == 0u;
    const uint32_t n = (o >> 3) & 3u;
    // counter: drop (mdp.py:1459-1471) or pick up (1473-1485) = swap hand and cell when exactly one is empty
    const bool swapX = (type == OC_T_COUNTER) & (hz != oz);
    // dispensers (mdp.py:1487-1513)
    const bool isD = type == OC_T_DISH_DISP;
    const bool take = hz & ((type == OC_T_ONION_DISP) | (type == OC_T_TOMATO_DISP) | isD);
    const uint32_t disp_obj = type == OC_T_ONION_DISP ? (uint32_t)OC_O_ONION
                              : type == OC_T_TOMATO_DISP ? (uint32_t)OC_O_TOMATO : (uint32_t)OC_O_DISH;
    // is_dish_pickup_useful (mdp.py:2180-2204): live hands and counters, stale pot_states
    const bool dish_useful = two & (((other_h == OC_O_DISH) ? 1u : 0u) < useful_pots) & (dcount == 0);
    // pot (mdp.py:1515-1568)
    const bool idle = tk == 0u;
    const bool start = isP & hz & (C.old_dyn == 0u) & (!oz) & idle & (n > 0u);       // begin_cooking -> tick 0
    const uint32_t ct = cook_of(C, o);
    const bool ready = (!idle) & ((tk - 1u) >= ct);
    const bool plate = isP & (h == OC_O_DISH) & (!oz) & ready;                       // soup pickup
    const bool is_ing = (h == OC_O_ONION) | (h == OC_O_TOMATO);
    const bool place = isP & is_ing & idle & (n < 3u);                               // not is_full (mdp.py:547-551)
    const uint32_t soup_new = OC_O_SOUP | ((n + 1u) << 3) | (o & 7u) | ((h == OC_O_TOMATO ? 1u : 0u) << n);
    // serving (mdp.py:1570-1577); deliver_soup / get_recipe_value (1631-1642, 1595-1602)
    const bool serve = (type == OC_T_SERVE) & ((h & OC_O_SOUP) != 0u);
    const float value = L.value(recipe_idx(h) & 15u);  // unconditional LUT read keeps the step straight-line

    r.new_h = swapX ? o : take ? disp_obj : plate ? o : (place | serve) ? 0u : h;
    r.cell_obj = swapX ? h : o_cell;
    r.slot = slot;
    r.new_o = plate ? 0u : place ? soup_new : o;
    r.new_tk = start ? 1u : plate ? 0u : tk;
    r.pot_upd = start | plate | place;
    r.swapX = swapX;
    r.ddelta = swapX ? ((h == OC_O_DISH ? 1 : 0) - (o == OC_O_DISH ? 1 : 0)) : 0;
    r.take_dish = take & isD;
    r.shaped = (place ? C.rew_place : 0.f) + (plate ? C.rew_soup : 0.f) +
               ((!DEFER_DISH & r.take_dish & dish_useful) ? C.rew_dish : 0.f);
    r.sparse = serve ? value : 0.f;
    r.ev = 0;
    if (EVENTS) {
        r.ev = interact_events<P>(C, type, h, o, swapX, take, place, plate, serve, other_h, dish_useful, n_full, two);
    }
    return r;
}

// ------------------------------------------------------------------------------------------
// One joint transition: get_state_transition (mdp.py:1375-1430).
//
// The reference applies player 0's interact before player 1's (mdp.py:1446).  Here both are computed from the
// pre-step pots and cells, which gives the scheduler two independent dependency chains to interleave (one
// wavefront per SIMD has nobody else to hide latency behind); the only ways player 0 can change what player 1
// sees are the same counter cell or the same pot, and those lanes (well under 1 % of env-steps) replay player
// 1's interact on the live state.  Hands and the loose-dish count flow from player 0 to player 1 as plain data.
// ------------------------------------------------------------------------------------------
template <int MAXP, bool EVENTS>
__device__ __forceinline__ void env_step(const LayC& C, const Lay L, uint32_t* cellw, EnvW<MAXP>& s, uint32_t delta4,
                                         uint32_t a0, uint32_t a1, float4& r, uint64_t& ev) {
    const bool two = s.pos1 != 0xFFu;
    // cells this step looks at: the two faced cells (pre-move pose, mdp.py:1452-1454) and the two move targets
    const uint32_t f0 = step_cell(s.pos0, s.or0, delta4);
    const uint32_t f1 = two ? step_cell(s.pos1, s.or1, delta4) : f0;
    const uint32_t m0 = a0 < 4u ? step_cell(s.pos0, a0, delta4) : s.pos0;
    const uint32_t m1 = (two & (a1 < 4u)) ? step_cell(s.pos1, a1, delta4) : (two ? s.pos1 : s.pos0);
    const uint32_t c_f0 = rd_cell16(cellw, f0), c_f1 = rd_cell16(cellw, f1);
    const uint32_t c_m0 = rd_cell16(cellw, m0), c_m1 = rd_cell16(cellw, m1);

    // pot_states, once before any interact (mdp.py:1439): pots that are ready / cooking / hold 1..2 idle items
    uint32_t useful_pots = 0, n_full = 0;
#pragma unroll
    for (int k = 0; k < MAXP; ++k) {
        const uint32_t o = s.ps[k], n = (o >> 3) & 3u;
        const bool nz = o != 0u, hot = s.tk[k] != 0u;
        useful_pots += (nz & (hot | ((n - 1u) < 2u))) ? 1u : 0u;
        if (EVENTS) n_full += (nz & (hot | (n == 3u))) ? 1u : 0u;
    }

    const bool act0 = a0 == OC_A_INTERACT, act1 = two & (a1 == OC_A_INTERACT);
    const IOut r0 = interact<MAXP, EVENTS, 0>(C, L, act0, s.held0, s.held1, s.dcount, c_f0, s.ps, s.tk, useful_pots,
                                              n_full, two);
    IOut r1 = interact<MAXP, EVENTS, 1>(C, L, act1, s.held1, r0.new_h, s.dcount + r0.ddelta, c_f1, s.ps, s.tk,
                                        useful_pots, n_full, two);
    // apply player 0
    s.held0 = r0.new_h;
    s.dcount += r0.ddelta;
#pragma unroll
    for (int k = 0; k < MAXP; ++k) {
        const bool upd = r0.pot_upd & (r0.slot == (uint32_t)k);
        s.ps[k] = upd ? r0.new_o : s.ps[k];
        s.tk[k] = upd ? r0.new_tk : s.tk[k];
    }
    // what player 1 faces after player 0's turn
    const bool same_cell = f1 == f0;
    const uint32_t c_f1_live = (same_cell & r0.swapX) ? ((c_f1 & 0xFF00u) | r0.cell_obj) : c_f1;
    const bool conflict = act1 & ((same_cell & r0.swapX) | (r0.pot_upd & (((c_f1 >> 8) & 7u) == OC_T_POT) &
                                                            ((c_f1 >> 11) == r0.slot)));
    if (__builtin_expect(conflict, 0)) {
        r1 = interact<MAXP, EVENTS, 1>(C, L, act1, s.held1, r0.new_h, s.dcount, c_f1_live, s.ps, s.tk, useful_pots, n_full,
                                       two);
    }
    // apply player 1
    s.held1 = r1.new_h;
    s.dcount += r1.ddelta;
#pragma unroll
    for (int k = 0; k < MAXP; ++k) {
        const bool upd = r1.pot_upd & (r1.slot == (uint32_t)k);
        s.ps[k] = upd ? r1.new_o : s.ps[k];
        s.tk[k] = upd ? r1.new_tk : s.tk[k];
    }
    // counter cells: unconditional byte stores (unchanged cells rewrite their own value); player 1 after player 0
    wr_cell_obj(cellw, f0, r0.cell_obj);
    wr_cell_obj(cellw, f1, r1.swapX ? r1.cell_obj : (c_f1_live & 0xFFu));
    r = make_float4(r0.sparse, r1.sparse, r0.shaped, r1.shaped);
    if (EVENTS) ev |= r0.ev | r1.ev;

    // resolve_movement (mdp.py:1644-1727): orientation follows the action even when blocked;
    // same target cell or swapped cells -> nobody moves (is_transition_collision, 1673-1683)
    const bool mv0 = a0 < 4u, mv1 = two & (a1 < 4u);
    const uint32_t np0 = (mv0 & (((c_m0 >> 8) & 7u) == OC_T_FLOOR)) ? m0 : s.pos0;
    const uint32_t np1 = (mv1 & (((c_m1 >> 8) & 7u) == OC_T_FLOOR)) ? m1 : s.pos1;
    s.or0 = mv0 ? a0 : s.or0;
    s.or1 = mv1 ? a1 : s.or1;
    const bool collide = two & ((np0 == np1) | ((np0 == s.pos1) & (np1 == s.pos0)));
    s.pos0 = collide ? s.pos0 : np0;
    s.pos1 = collide ? s.pos1 : np1;

    // step_environment_effects (mdp.py:1691-1703)
    s.t += 1u;
#pragma unroll
    for (int k = 0; k < MAXP; ++k) {
        const uint32_t o = s.ps[k], n = (o >> 3) & 3u;
        uint32_t tk = s.tk[k];
        const bool nz = o != 0u;
        tk = ((C.old_dyn != 0u) & nz & (tk == 0u) & (n == 3u)) ? 1u : tk;       // auto begin_cooking (old dynamics)
        const bool cooking = nz & (tk != 0u) & ((tk - 1u) < cook_of(C, o));      // is_cooking -> cook()
        s.tk[k] = tk + (cooking ? 1u : 0u);
    }
}

// exact count of bytes equal to OC_O_DISH in a dword
__device__ __forceinline__ uint32_t count_dish_bytes(uint32_t w) {
    const uint32_t x = w ^ 0x03030303u;  // dish bytes -> 0
    const uint32_t t = ((x & 0x7F7F7F7Fu) + 0x7F7F7F7Fu) | x | 0x7F7F7F7Fu;
    return __popc(~t);
}

// load one env from its HBM planes into registers + the LDS cell words
template <int MAXP>
__device__ __forceinline__ void load_env(const Lay L, const uint4* __restrict__ st, int64_t n, int64_t e, int n_obj,
                                         uint32_t n_pots, EnvW<MAXP>& s, uint32_t* cellw) {
    const uint4 h = st[e];
    s.pos0 = h.x & 0xFF; s.or0 = (h.x >> 8) & 0xFF; s.held0 = (h.x >> 16) & 0xFF; s.pos1 = h.x >> 24;
    s.or1 = h.y & 0xFF; s.held1 = (h.y >> 8) & 0xFF; s.t = h.y >> 16;
    int32_t dishes = 0;
    for (int p = 0; p < n_obj; ++p) {
        const uint4 v = st[(int64_t)(1 + p) * n + e];
        const uint32_t ow[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const uint32_t T = L.u32(L_TERRAIN + 16 * p + 4 * q);
            dishes += (int32_t)count_dish_bytes(ow[q]);
            // interleave object and terrain bytes: cells 4q..4q+3 of this plane -> two dwords of (obj | terrain << 8)
            cellw[(8 * p + 2 * q) * BLOCK] = __builtin_amdgcn_perm(T, ow[q], 0x05010400u);
            cellw[(8 * p + 2 * q + 1) * BLOCK] = __builtin_amdgcn_perm(T, ow[q], 0x07030602u);
        }
    }
    s.dcount = dishes;
#pragma unroll
    for (int k = 0; k < MAXP; ++k) {
        s.ps[k] = 0; s.tk[k] = 0;
        if ((uint32_t)k < n_pots) {
            s.ps[k] = rd_cell16(cellw, L.pot_cell(k)) & 0xFFu;
            s.tk[k] = ((k < 4 ? h.z : h.w) >> (8 * (k & 3))) & 0xFFu;
        }
    }
}

template <int MAXP>
__device__ __forceinline__ void store_env(const Lay L, uint4* __restrict__ st, int64_t n, int64_t e, int n_obj,
                                          uint32_t n_pots, const EnvW<MAXP>& s, uint32_t* cellw) {
    uint4 h;
    h.x = s.pos0 | (s.or0 << 8) | (s.held0 << 16) | (s.pos1 << 24);
    h.y = s.or1 | (s.held1 << 8) | (min(s.t, 0xFFFFu) << 16);  // the wire format's u16 timestep saturates
    h.z = 0; h.w = 0;
#pragma unroll
    for (int k = 0; k < MAXP; ++k) {
        if ((uint32_t)k < n_pots) {
            wr_cell_obj(cellw, L.pot_cell(k), s.ps[k]);
            if (k < 4) h.z |= s.tk[k] << (8 * (k & 3));
            else h.w |= s.tk[k] << (8 * (k & 3));
        }
    }
    st[e] = h;
    for (int p = 0; p < n_obj; ++p) {
        uint32_t ow[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const uint32_t lo = cellw[(8 * p + 2 * q) * BLOCK], hi = cellw[(8 * p + 2 * q + 1) * BLOCK];
            ow[q] = __builtin_amdgcn_perm(hi, lo, 0x06040200u);  // object bytes of 4 cells
        }
        st[(int64_t)(1 + p) * n + e] = make_uint4(ow[0], ow[1], ow[2], ow[3]);
    }
}

// start state of a layout (mdp.py:1297-1305, 939-950)
template <int MAXP>
__device__ __forceinline__ void env_reset(const Lay L, int n_obj, EnvW<MAXP>& s, uint32_t* cellw) {
    s.pos0 = L.u8(L_START_POS); s.pos1 = L.u8(L_START_POS + 1);
    s.or0 = L.u8(L_START_OR); s.or1 = s.pos1 == 0xFFu ? 0u : L.u8(L_START_OR + 1);
    s.held0 = s.held1 = 0; s.t = 0; s.dcount = 0;
#pragma unroll
    for (int k = 0; k < MAXP; ++k) { s.ps[k] = 0; s.tk[k] = 0; }
    for (int d = 0; d < n_obj * 8; ++d) cellw[d * BLOCK] &= 0xFF00FF00u;  // clear objects, keep terrain
}

template <int MAXP>
__device__ __forceinline__ uint32_t finish_step(const Lay L, int n_obj, uint32_t* cellw, EnvW<MAXP>& s, int horizon,
                                                uint32_t options, const float4& r, float4& ep) {
    ep.x += r.x; ep.y += r.y; ep.z += r.z; ep.w += r.w;
    uint32_t fl = 0;
    if ((int)s.t >= horizon) {
        fl |= OC_F_DONE;
        if (options & OC_OPT_AUTO_RESET) {
            env_reset<MAXP>(L, n_obj, s, cellw);
            ep = make_float4(0.f, 0.f, 0.f, 0.f);
            fl |= OC_F_RESET;
        }
    }
    return fl;
}

// ------------------------------------------------------------------------------------------
// k_step: one transition per launch, actions supplied by the caller.
// ------------------------------------------------------------------------------------------
template <bool UNIFORM, int MAXP, bool LAY_LDS, bool EVENTS>
__global__ __launch_bounds__(BLOCK) void k_step(const OcLayout* __restrict__ g_layouts, int n_layouts,
                                                const uint16_t* __restrict__ layout_id, const uint4* st_in,
                                                uint4* st_out, const uint8_t* __restrict__ actions,
                                                float4* __restrict__ rewards, uint8_t* __restrict__ flags,
                                                float4* __restrict__ ep_returns, uint64_t* __restrict__ events,
                                                int64_t n, int W, int n_obj, int horizon, uint32_t options) {
    extern __shared__ __attribute__((aligned(16))) uint32_t s_cells[];  // [n_obj * 8][BLOCK]
    __shared__ uint4 s_lay[LAY_LDS ? (UNIFORM ? 16 : LDS_LAYOUT_MAX * 16) : 1];  // one 256-byte record when the batch has one layout
    const int64_t e = (int64_t)blockIdx.x * BLOCK + threadIdx.x;
    const bool active = e < n;
    const Lay L = stage_layouts<LAY_LDS>(g_layouts, n_layouts, layout_id, e, active, s_lay);
    if (!active) return;
    uint32_t* cellw = s_cells + threadIdx.x;
    const LayC C = load_consts<UNIFORM>(L);
    const uint32_t delta4 = make_delta4(W);
    EnvW<MAXP> s;
    load_env<MAXP>(L, st_in, n, e, n_obj, C.n_pots, s, cellw);
    const uint32_t a01 = reinterpret_cast<const uint16_t*>(actions)[e];
    const uint32_t a0 = a01 & 0xFFu, a1 = a01 >> 8;
    float4 r = make_float4(0.f, 0.f, 0.f, 0.f);
    uint64_t ev = 0;
    uint32_t fl;
    float4 ep = ep_returns ? ep_returns[e] : make_float4(0.f, 0.f, 0.f, 0.f);
    if (a0 > 5u || a1 > 5u) {
        fl = OC_F_BAD_ACTION;  // get_state_transition raises ValueError (mdp.py:1394-1398): leave the env untouched
    } else {
        env_step<MAXP, EVENTS>(C, L, cellw, s, delta4, a0, a1, r, ev);
        fl = finish_step<MAXP>(L, n_obj, cellw, s, horizon, options, r, ep);
    }
    store_env<MAXP>(L, st_out, n, e, n_obj, C.n_pots, s, cellw);
    rewards[e] = r;
    flags[e] = (uint8_t)fl;
    if (ep_returns) ep_returns[e] = ep;
    if (EVENTS) events[e] = ev;
}

// ------------------------------------------------------------------------------------------
// k_rollout: n_steps fused transitions per launch under the uniform random policy; the env lives in
// registers + LDS between steps and only the per-step outputs (17 B per env-step) go to HBM.
// Action stream: one Philox4x32-10 block feeds 8 consecutive steps — word s/2 of block t/8 is
// expanded into base-6 digits by multiply-high (digit = mulhi(x, 6), x <- x * 6), two digits (player
// 0, player 1) per step.  oracle_random_actions restates the same mapping.
// ------------------------------------------------------------------------------------------
__device__ __forceinline__ uint32_t bitsel(uint32_t mask, uint32_t a, uint32_t b) { return (a & mask) | (b & ~mask); }

__device__ __forceinline__ void draw_actions(const uint32_t (&rnd)[4], uint32_t s8, uint32_t& a0, uint32_t& a1) {
    // s8 is wave-uniform; masks + v_bfi keep the step free of (uniform) branches, the multiplier stays scalar
    uint32_t w = rnd[0];
    w = bitsel(0u - (uint32_t)(s8 >= 2u), rnd[1], w);
    w = bitsel(0u - (uint32_t)(s8 >= 4u), rnd[2], w);
    w = bitsel(0u - (uint32_t)(s8 >= 6u), rnd[3], w);
    const uint32_t x = w * ((s8 & 1u) ? 36u : 1u);
    a0 = __umulhi(x, 6u);
    a1 = __umulhi(x * 6u, 6u);
}

template <bool UNIFORM, int MAXP, bool LAY_LDS>
__global__ __launch_bounds__(BLOCK) void k_rollout(const OcLayout* __restrict__ g_layouts, int n_layouts,
                                                   const uint16_t* __restrict__ layout_id, uint4* st,
                                                   float4* __restrict__ rewards, uint8_t* __restrict__ flags,
                                                   float4* __restrict__ ep_returns, int64_t n, int W, int n_obj,
                                                   int horizon, uint32_t options, uint32_t seed_lo, uint32_t seed_hi,
                                                   int64_t env_offset, int64_t t0, int n_steps) {
    extern __shared__ __attribute__((aligned(16))) uint32_t s_cells[];
    __shared__ uint4 s_lay[LAY_LDS ? (UNIFORM ? 16 : LDS_LAYOUT_MAX * 16) : 1];  // one 256-byte record when the batch has one layout
    const int64_t e = (int64_t)blockIdx.x * BLOCK + threadIdx.x;
    const bool active = e < n;
    const Lay L = stage_layouts<LAY_LDS>(g_layouts, n_layouts, layout_id, e, active, s_lay);
    if (!active) return;
    uint32_t* cellw = s_cells + threadIdx.x;
    const LayC C = load_consts<UNIFORM>(L);
    const uint32_t delta4 = make_delta4(W);
    EnvW<MAXP> s;
    load_env<MAXP>(L, st, n, e, n_obj, C.n_pots, s, cellw);
    float4 ep = ep_returns ? ep_returns[e] : make_float4(0.f, 0.f, 0.f, 0.f);
    const uint64_t g = (uint64_t)(env_offset + e);
    const uint32_t g_lo = (uint32_t)g, g_hi = (uint32_t)(g >> 32);
    uint32_t rnd[4] = {0, 0, 0, 0};
    for (int k = 0; k < n_steps; ++k) {
        const uint64_t t = (uint64_t)(t0 + k);
        const uint32_t s8 = (uint32_t)t & 7u;
        if (k == 0 || s8 == 0u) {
            const uint64_t blk = t >> 3;
            philox4x32_10((uint32_t)blk, g_lo, g_hi, (uint32_t)(blk >> 32), seed_lo, seed_hi, rnd);
        }
        uint32_t a0, a1;
        draw_actions(rnd, s8, a0, a1);
        float4 r;
        uint64_t ev = 0;
        env_step<MAXP, false>(C, L, cellw, s, delta4, a0, a1, r, ev);
        const uint32_t fl = finish_step<MAXP>(L, n_obj, cellw, s, horizon, options, r, ep);
        if (rewards) rewards[(int64_t)k * n + e] = r;
        if (flags) flags[(int64_t)k * n + e] = (uint8_t)fl;
    }
    store_env<MAXP>(L, st, n, e, n_obj, C.n_pots, s, cellw);
    if (ep_returns) ep_returns[e] = ep;
}
