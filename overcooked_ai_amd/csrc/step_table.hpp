// step_table.hpp — get_state_transition with the table-driven interact (8-byte LUT): k_step3, and the device functions
// k_step1 / k_train_step / k_rollout_encode build on (round 4: its rollout kernel k_rollout3 is gone — k_rollout4 runs every
// launch length)
// Part of liboc_amd.so: included by oc_amd.hip inside its anonymous namespace, in this order:
//   common, reset, step_predicate, step_table, step_one, step_lut4, rollout_pair, encode, rollout_encode, featurize, potential, shaping.
#pragma once

// ==========================================================================================
// v3: table-driven interact.
//
// SQ counters show the step kernels are bound by instruction issue (one wavefront per SIMD issues an integer VALU /
// SALU instruction every ~4-6 cycles), and that INTERACT is more than half of the stream.  v3 replaces the
// predicate network by one 8-byte look-up per player:
//     key   = (terrain type of the faced cell or 7 when the player does not interact,
//              class of the hand        {none, onion, tomato, dish, soup},
//              class of what is faced   counter: {empty, dish, other};
//                                       pot: {empty, idle 1, idle 2, idle 3 items, cooking, ready})
//     entry = three byte selectors into the pool {hand, faced object, soup+ingredient, tick, 1, 2, 3, 0}
//             (ONE v_perm_b32 yields the new hand, the new faced object and the new tick), event flags,
//             the pot's new class and the change of the loose-dish count.
// The table is layout independent (two variants: new / old dynamics) and generated at compile time.  Pot classes
// are kept in registers and advanced by the env effects, so "ready" costs no cook-time look-up in the interact.
// Cells live in LDS as u16[cell][lane] (address = cell << 9 | lane << 1: one v_lshl_add per access).
// ==========================================================================================
enum { PC_EMPTY = 0, PC_IDLE1 = 1, PC_IDLE2 = 2, PC_IDLE3 = 3, PC_COOKING = 4, PC_READY = 5 };
enum { LF_SWAP = 1, LF_POT_UPD = 2, LF_SERVE = 4, LF_TAKE_DISH = 8, LF_PLACE = 16, LF_PLATE = 32, LF_START = 64 };
constexpr int LUT_ENTRIES = 8 * 5 * 6;  // type x hand class x faced class

struct LutEntry { uint32_t lo, hi; };  // lo: sel_h | sel_o << 8 | sel_tk << 16 | 0x0C << 24 ; hi: flags | new_pc << 8 | (dd + 1) << 16

constexpr LutEntry lut_entry(int old_dyn, int type, int hc, int oc) {
    // pool selectors: 0 hand, 1 faced object, 2 soup + ingredient, 3 tick, 4 const 1 (onion / tick 0), 5 const 2
    // (tomato), 6 const 3 (dish), 7 const 0
    int sel_h = 0, sel_o = 1, sel_tk = 3, flags = 0, new_pc = oc, dd = 0;
    if (type == OC_T_COUNTER) {
        if (hc == 0 && (oc == 1 || oc == 2)) { sel_h = 1; sel_o = 0; flags = LF_SWAP; dd = (oc == 1) ? -1 : 0; }  // pick up
        else if (hc != 0 && oc == 0) { sel_h = 1; sel_o = 0; flags = LF_SWAP; dd = (hc == 3) ? 1 : 0; }            // drop
    } else if (type == OC_T_ONION_DISP) {
        if (hc == 0) sel_h = 4;
    } else if (type == OC_T_TOMATO_DISP) {
        if (hc == 0) sel_h = 5;
    } else if (type == OC_T_DISH_DISP) {
        if (hc == 0) { sel_h = 6; flags = LF_TAKE_DISH; }
    } else if (type == OC_T_POT) {
        if (hc == 0 && oc >= PC_IDLE1 && oc <= PC_IDLE3 && !old_dyn) {           // begin_cooking (mdp.py:1515-1522)
            sel_tk = 4; flags = LF_POT_UPD | LF_START; new_pc = PC_COOKING;
        } else if (hc == 3 && oc == PC_READY) {                                  // soup pickup (mdp.py:1525-1539)
            sel_h = 1; sel_o = 7; sel_tk = 7; flags = LF_POT_UPD | LF_PLATE; new_pc = PC_EMPTY;
        } else if ((hc == 1 || hc == 2) && oc <= PC_IDLE2) {                     // add ingredient (mdp.py:1541-1568)
            sel_h = 7; sel_o = 2; flags = LF_POT_UPD | LF_PLACE; new_pc = oc + 1;
        }
    } else if (type == OC_T_SERVE) {
        if (hc == 4) { sel_h = 7; flags = LF_SERVE; }                            // deliver (mdp.py:1570-1577)
    }
    return LutEntry{(uint32_t)(sel_h | (sel_o << 8) | (sel_tk << 16) | (0x0C << 24)),
                    (uint32_t)(flags | (new_pc << 8) | ((dd + 1) << 16))};
}

struct LutTable { LutEntry e[2 * LUT_ENTRIES]; };
constexpr LutTable make_lut() {
    LutTable t{};
    for (int od = 0; od < 2; ++od)
        for (int type = 0; type < 8; ++type)
            for (int hc = 0; hc < 5; ++hc)
                for (int oc = 0; oc < 6; ++oc) t.e[od * LUT_ENTRIES + (type * 5 + hc) * 6 + oc] = lut_entry(od, type, hc, oc);
    return t;
}
__device__ const LutTable g_lut = make_lut();

template <int MAXP>
struct Env3 {
    uint32_t pos0, or0, held0, pos1, or1, held1, t;
    uint32_t tk[MAXP], ps[MAXP], pc[MAXP];  // per pot slot: tick + 1, soup code, class
    int32_t dcount;
};

__device__ __forceinline__ uint32_t rd_cell3(const uint16_t* cells, uint32_t c) { return cells[c * BLOCK]; }
__device__ __forceinline__ void wr_obj3(uint16_t* cells, uint32_t c, uint32_t v) {
    reinterpret_cast<uint8_t*>(cells + c * BLOCK)[0] = (uint8_t)v;
}

__device__ __forceinline__ uint32_t pot_class(const LayC& C, uint32_t o, uint32_t tk) {
    const uint32_t n = (o >> 3) & 3u;
    const uint32_t hot = (tk - 1u) >= cook_of(C, o) ? (uint32_t)PC_READY : (uint32_t)PC_COOKING;
    return o == 0u ? (uint32_t)PC_EMPTY : (tk == 0u ? n : hot);  // idle with n = 1..3 items (n = 0: an empty soup object)
}

struct IOut3 {
    uint32_t new_h, new_o, new_tk, new_pc, slot, flags, cell_obj;
    int32_t ddelta;
    float sparse;
};

// one player's INTERACT through the table; `s_lut` = this lane's table variant in LDS, `c16` the faced cell word
template <int MAXP, bool EAGER_VALUE = true>
__device__ __forceinline__ IOut3 interact3(const Lay L, const uint8_t* s_lut, bool act, uint32_t h, uint32_t c16,
                                           const uint32_t (&ps)[MAXP], const uint32_t (&tkr)[MAXP],
                                           const uint32_t (&pcr)[MAXP]) {
    IOut3 r;
    const uint32_t tc = c16 >> 8;
    const uint32_t type = act ? (tc & 7u) : 7u;  // 7 = no interact: every entry of that row is a no-op
    const uint32_t slot = tc >> 3;
    uint32_t pso = 0, tkv = 0, pcv = 0;
    if (MAXP == 1) {  // a single pot: slot is 0 for every cell, the values are only used when the cell is that pot
        pso = ps[0]; tkv = tkr[0]; pcv = pcr[0];
    } else {
#pragma unroll
        for (int k = 0; k < MAXP; ++k) {
            const bool sel = slot == (uint32_t)k;
            pso = sel ? ps[k] : pso;
            tkv = sel ? tkr[k] : tkv;
            pcv = sel ? pcr[k] : pcv;
        }
    }
    const bool isP = type == OC_T_POT;
    const uint32_t o_cell = c16 & 0xFFu;
    const uint32_t o = isP ? pso : o_cell;
    const uint32_t oc = isP ? pcv : (o_cell == 0u ? 0u : o_cell == OC_O_DISH ? 1u : 2u);
    const uint32_t hc = min(h, 4u);
    const uint32_t key = (type * 5u + hc) * 6u + oc;
    const uint2 ent = *reinterpret_cast<const uint2*>(s_lut + key * 8u);
    // pool {hand, faced object, soup + ingredient, tick | 1, 2, 3, 0}: one v_perm_b32 picks all three results
    const uint32_t n = (o >> 3) & 3u;
    const uint32_t soup_new = OC_O_SOUP | ((n + 1u) << 3) | (o & 7u) | ((h == OC_O_TOMATO ? 1u : 0u) << n);
    const uint32_t pool = h | (o << 8) | (soup_new << 16) | (tkv << 24);
    const uint32_t res = __builtin_amdgcn_perm(0x00030201u, pool, ent.x);
    r.new_h = res & 0xFFu;
    r.new_o = (res >> 8) & 0xFFu;
    r.new_tk = (res >> 16) & 0xFFu;
    r.flags = ent.y & 0xFFu;
    r.new_pc = (ent.y >> 8) & 0xFFu;
    r.ddelta = (int32_t)((ent.y >> 16) & 3u) - 1;
    r.slot = slot;
    r.cell_obj = (r.flags & LF_SWAP) ? r.new_o : o_cell;
    if (EAGER_VALUE) {
        const float value = L.value(recipe_idx(h) & 15u);  // unconditional read: keeps both players' look-ups in one block
        r.sparse = (r.flags & LF_SERVE) ? value : 0.f;      // deliver_soup (mdp.py:1631-1642)
    } else {
        r.sparse = 0.f;  // the caller looks the value up, and only when some lane of the wavefront delivers
    }
    return r;
}

template <int MAXP>
__device__ __forceinline__ void apply_pot3(Env3<MAXP>& s, const IOut3& r) {
#pragma unroll
    for (int k = 0; k < MAXP; ++k) {
        const bool upd = ((r.flags & LF_POT_UPD) != 0u) & (MAXP == 1 || r.slot == (uint32_t)k);
        s.ps[k] = upd ? r.new_o : s.ps[k];
        s.tk[k] = upd ? r.new_tk : s.tk[k];
        s.pc[k] = upd ? r.new_pc : s.pc[k];
    }
}

// get_state_transition (mdp.py:1375-1430), table-driven, in three pieces so that the rollout loop can issue the
// LDS reads of step k+1 before the tail (env effects, bookkeeping, output stores) of step k:
//   probe3      the four cells a step looks at: the faced cells (pre-move pose, mdp.py:1452-1454) and the move targets
//   step3_main  resolve_interacts + resolve_movement.  Same sequencing argument as env_step: both interacts are
//               evaluated on the pre-step pots/cells, player 1 replays only when player 0 touched its cell or pot
//   step3_env   step_environment_effects
struct Probe3 {
    uint32_t f0, f1, m0, m1;
    uint32_t c_f0, c_f1, c_m0, c_m1;
};

// FAST levels: 0 generic; 1 = every layout of the table has two players ("is there a second player" folds away; the
// rollout loop is unrolled over the Philox block); 2 = 1 + one layout for the whole batch with at most 64 cells: the
// floor test of resolve_movement becomes a bit test against a wave-uniform 64-bit mask instead of two LDS reads;
// 3 = 2 with the whole single-player move in one LDS look-up: MOVE[cell * 8 + action] = the cell the player ends on
// (the target if it is free, else the cell itself; STAY / INTERACT: the cell itself), built once per workgroup.
template <int MAXP, int FAST>
__device__ __forceinline__ Probe3 probe3(const uint16_t* cells, const Env3<MAXP>& s, uint32_t delta4, uint32_t a0,
                                         uint32_t a1, const uint8_t* s_move) {
    Probe3 q;
    const bool two = FAST >= 1 || s.pos1 != 0xFFu;
    q.f0 = step_cell(s.pos0, s.or0, delta4);
    q.f1 = two ? step_cell(s.pos1, s.or1, delta4) : q.f0;
    if (FAST == 3) {
        q.m0 = s_move[s.pos0 * 8u + a0];
        q.m1 = s_move[s.pos1 * 8u + a1];
    } else {
        q.m0 = a0 < 4u ? step_cell(s.pos0, a0, delta4) : s.pos0;
        q.m1 = (two & (a1 < 4u)) ? step_cell(s.pos1, a1, delta4) : (two ? s.pos1 : s.pos0);
    }
    q.c_f0 = rd_cell3(cells, q.f0); q.c_f1 = rd_cell3(cells, q.f1);
    if (FAST < 2) { q.c_m0 = rd_cell3(cells, q.m0); q.c_m1 = rd_cell3(cells, q.m1); }
    else { q.c_m0 = 0; q.c_m1 = 0; }
    return q;
}

template <int MAXP, int FAST, bool EVENTS = false>
__device__ __forceinline__ void step3_main(const LayC& C, const Lay L, const uint8_t* s_lut, uint16_t* cells,
                                           Env3<MAXP>& s, uint32_t a0, uint32_t a1, const Probe3& q, float4& r,
                                           uint64_t floor_mask, uint64_t* ev = nullptr) {
    const bool two = FAST >= 1 || s.pos1 != 0xFFu;
    const bool mv0 = a0 < 4u, mv1 = two & (a1 < 4u);
    const uint32_t f0 = q.f0, f1 = q.f1, c_f0 = q.c_f0, c_f1 = q.c_f1;

    // pot_states before any interact (mdp.py:1439): ready / cooking / 1..2 idle items  <=>  class not in {empty, idle 3}
    uint32_t useful_pots = 0, n_full = 0;
#pragma unroll
    for (int k = 0; k < MAXP; ++k) {
        useful_pots += ((s.pc[k] != PC_EMPTY) & (s.pc[k] != PC_IDLE3)) ? 1u : 0u;
        if (EVENTS) n_full += (((uint32_t)k < C.n_pots) & (s.pc[k] >= PC_IDLE3)) ? 1u : 0u;
    }
    uint32_t ps_before[MAXP];
    if (EVENTS) {
#pragma unroll
        for (int k = 0; k < MAXP; ++k) ps_before[k] = s.ps[k];
    }

    const bool act0 = a0 == OC_A_INTERACT, act1 = two & (a1 == OC_A_INTERACT);
    const uint32_t h0_before = s.held0, h1_before = s.held1;
    const IOut3 r0 = interact3<MAXP, false>(L, s_lut, act0, s.held0, c_f0, s.ps, s.tk, s.pc);
    IOut3 r1 = interact3<MAXP, false>(L, s_lut, act1, s.held1, c_f1, s.ps, s.tk, s.pc);
    // shaped rewards; is_dish_pickup_useful (mdp.py:2180-2204) sees the live hands / counters and the stale pots
    const bool du0 = two & (((s.held1 == OC_O_DISH) ? 1u : 0u) < useful_pots) & (s.dcount == 0);
    const float sh0 = ((r0.flags & LF_PLACE) ? C.rew_place : 0.f) + ((r0.flags & LF_PLATE) ? C.rew_soup : 0.f) +
                      ((((r0.flags & LF_TAKE_DISH) != 0u) & du0) ? C.rew_dish : 0.f);
    s.held0 = r0.new_h;
    s.dcount += r0.ddelta;
    apply_pot3<MAXP>(s, r0);
    const bool same_cell = f1 == f0;
    const bool swap0 = (r0.flags & LF_SWAP) != 0u;
    const uint32_t c_f1_live = (same_cell & swap0) ? ((c_f1 & 0xFF00u) | r0.cell_obj) : c_f1;
    const bool conflict = act1 & ((same_cell & swap0) | (((r0.flags & LF_POT_UPD) != 0u) &
                                                          (((c_f1 >> 8) & 7u) == OC_T_POT) & ((c_f1 >> 11) == r0.slot)));
    if (__builtin_expect(conflict, 0)) r1 = interact3<MAXP, false>(L, s_lut, act1, s.held1, c_f1_live, s.ps, s.tk, s.pc);
    const bool du1 = two & (((s.held0 == OC_O_DISH) ? 1u : 0u) < useful_pots) & (s.dcount == 0);
    const float sh1 = ((r1.flags & LF_PLACE) ? C.rew_place : 0.f) + ((r1.flags & LF_PLATE) ? C.rew_soup : 0.f) +
                      ((((r1.flags & LF_TAKE_DISH) != 0u) & du1) ? C.rew_dish : 0.f);
    if (EVENTS) {  // event_infos (mdp.py:2121-2308) from the two outcomes
        auto faced = [&](uint32_t c16, const uint32_t (&ps)[MAXP]) {  // the object an interact sees: the soup for a pot
            uint32_t o = c16 & 0xFFu;
            if (((c16 >> 8) & 7u) == OC_T_POT) {
#pragma unroll
                for (int k = 0; k < MAXP; ++k) o = (c16 >> 11) == (uint32_t)k ? ps[k] : o;
            }
            return o;
        };
        uint32_t ps_mid[MAXP];  // pots after player 0 (what a replayed player 1 saw)
#pragma unroll
        for (int k = 0; k < MAXP; ++k) ps_mid[k] = s.ps[k];
        const uint32_t t0_ = act0 ? (c_f0 >> 8) & 7u : 7u, t1_ = act1 ? (c_f1 >> 8) & 7u : 7u;
        const uint32_t o0 = faced(c_f0, ps_before), o1 = conflict ? faced(c_f1_live, ps_mid) : faced(c_f1, ps_before);
        const bool disp0 = (t0_ == OC_T_ONION_DISP) | (t0_ == OC_T_TOMATO_DISP) | (t0_ == OC_T_DISH_DISP);
        const bool disp1 = (t1_ == OC_T_ONION_DISP) | (t1_ == OC_T_TOMATO_DISP) | (t1_ == OC_T_DISH_DISP);
        const int32_t dc_before = s.dcount - r0.ddelta;
        const bool du0e = two & (((h1_before == OC_O_DISH) ? 1u : 0u) < useful_pots) & (dc_before == 0);
        *ev = interact_events<0>(C, t0_, h0_before, o0, (r0.flags & LF_SWAP) != 0u, disp0 & (h0_before == 0u) & (r0.new_h != 0u),
                                 (r0.flags & LF_PLACE) != 0u, (r0.flags & LF_PLATE) != 0u, (r0.flags & LF_SERVE) != 0u,
                                 h1_before, du0e, n_full, two) |
              interact_events<1>(C, t1_, h1_before, o1, (r1.flags & LF_SWAP) != 0u, disp1 & (h1_before == 0u) & (r1.new_h != 0u),
                                 (r1.flags & LF_PLACE) != 0u, (r1.flags & LF_PLATE) != 0u, (r1.flags & LF_SERVE) != 0u,
                                 s.held0, du1, n_full, two);
    }
    s.held1 = r1.new_h;
    s.dcount += r1.ddelta;
    apply_pot3<MAXP>(s, r1);
    wr_obj3(cells, f0, r0.cell_obj);
    wr_obj3(cells, f1, (r1.flags & LF_SWAP) ? r1.cell_obj : (c_f1_live & 0xFFu));
    // deliver_soup (mdp.py:1631-1642): deliveries are rare, so the recipe-value look-ups sit behind a wave-uniform branch
    float sp0 = 0.f, sp1 = 0.f;
    const bool serve0 = (r0.flags & LF_SERVE) != 0u, serve1 = (r1.flags & LF_SERVE) != 0u;
    if (__builtin_expect(__builtin_amdgcn_ballot_w64(serve0 | serve1) != 0ull, 0)) {
        sp0 = serve0 ? L.value(recipe_idx(h0_before) & 15u) : 0.f;
        sp1 = serve1 ? L.value(recipe_idx(h1_before) & 15u) : 0.f;
    }
    r = make_float4(sp0, sp1, sh0, sh1);

    // resolve_movement (mdp.py:1644-1727)
    uint32_t np0, np1;
    if (FAST == 3) {  // the move table already holds "target if free, else stay"
        np0 = q.m0;
        np1 = q.m1;
    } else {
        const bool fl0 = FAST == 2 ? ((floor_mask >> q.m0) & 1ull) != 0ull : ((q.c_m0 >> 8) & 7u) == OC_T_FLOOR;
        const bool fl1 = FAST == 2 ? ((floor_mask >> q.m1) & 1ull) != 0ull : ((q.c_m1 >> 8) & 7u) == OC_T_FLOOR;
        np0 = (mv0 & fl0) ? q.m0 : s.pos0;
        np1 = (mv1 & fl1) ? q.m1 : s.pos1;
    }
    s.or0 = mv0 ? a0 : s.or0;
    s.or1 = mv1 ? a1 : s.or1;
    const bool collide = two & ((np0 == np1) | ((np0 == s.pos1) & (np1 == s.pos0)));
    s.pos0 = collide ? s.pos0 : np0;
    s.pos1 = collide ? s.pos1 : np1;
}

// step_environment_effects (mdp.py:1691-1703): advance cooking pots, promote them to ready
template <int MAXP>
__device__ __forceinline__ void step3_env(const LayC& C, Env3<MAXP>& s) {
    s.t += 1u;
#pragma unroll
    for (int k = 0; k < MAXP; ++k) {
        uint32_t pc = s.pc[k], tk = s.tk[k];
        const bool autostart = (C.old_dyn != 0u) & (pc == PC_IDLE3);  // old dynamics: 3 idle items start by themselves
        pc = autostart ? (uint32_t)PC_COOKING : pc;
        tk = autostart ? 1u : tk;
        const bool cooking = pc == PC_COOKING;
        tk += cooking ? 1u : 0u;
        pc = (cooking & ((tk - 1u) >= cook_of(C, s.ps[k]))) ? (uint32_t)PC_READY : pc;
        s.pc[k] = pc;
        s.tk[k] = tk;
    }
}

template <int MAXP, int FAST = 0, bool EVENTS = false>
__device__ __forceinline__ void env_step3(const LayC& C, const Lay L, const uint8_t* s_lut, uint16_t* cells,
                                          Env3<MAXP>& s, uint32_t delta4, uint32_t a0, uint32_t a1, float4& r,
                                          uint64_t floor_mask = 0, const uint8_t* s_move = nullptr, uint64_t* ev = nullptr) {
    const Probe3 q = probe3<MAXP, FAST>(cells, s, delta4, a0, a1, s_move);
    step3_main<MAXP, FAST, EVENTS>(C, L, s_lut, cells, s, a0, a1, q, r, floor_mask, ev);
    step3_env<MAXP>(C, s);
}

// bit c set <=> cell c is floor; wave-uniform (layouts of at most 64 cells, one layout per batch)
__device__ __forceinline__ uint64_t make_floor_mask(const Lay L, int n_cells) {
    uint32_t lo = 0, hi = 0;
    for (int c = 0; c < n_cells && c < 32; ++c) lo |= ((L.terrain(c) & 7u) == OC_T_FLOOR ? 1u : 0u) << c;
    for (int c = 32; c < n_cells && c < 64; ++c) hi |= ((L.terrain(c) & 7u) == OC_T_FLOOR ? 1u : 0u) << (c - 32);
    lo = (uint32_t)__builtin_amdgcn_readfirstlane((int)lo);
    hi = (uint32_t)__builtin_amdgcn_readfirstlane((int)hi);
    return ((uint64_t)hi << 32) | lo;
}

template <int MAXP>
__device__ __forceinline__ void load_env3(const LayC& C, const Lay L, const uint4* __restrict__ st, int64_t n, int64_t e,
                                          int n_obj, Env3<MAXP>& s, uint16_t* cells) {
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
            const uint32_t lo = __builtin_amdgcn_perm(T, ow[q], 0x05010400u);  // cells 4q, 4q+1: obj | terrain << 8
            const uint32_t hi = __builtin_amdgcn_perm(T, ow[q], 0x07030602u);  // cells 4q+2, 4q+3
            const int c = 16 * p + 4 * q;
            cells[(c + 0) * BLOCK] = (uint16_t)lo;
            cells[(c + 1) * BLOCK] = (uint16_t)(lo >> 16);
            cells[(c + 2) * BLOCK] = (uint16_t)hi;
            cells[(c + 3) * BLOCK] = (uint16_t)(hi >> 16);
        }
    }
    s.dcount = dishes;
#pragma unroll
    for (int k = 0; k < MAXP; ++k) {
        s.ps[k] = 0; s.tk[k] = 0; s.pc[k] = PC_EMPTY;
        if ((uint32_t)k < C.n_pots) {
            s.ps[k] = rd_cell3(cells, L.pot_cell(k)) & 0xFFu;
            s.tk[k] = ((k < 4 ? h.z : h.w) >> (8 * (k & 3))) & 0xFFu;
            s.pc[k] = pot_class(C, s.ps[k], s.tk[k]);
        }
    }
}

template <int MAXP>
__device__ __forceinline__ void store_env3(const LayC& C, const Lay L, uint4* __restrict__ st, int64_t n, int64_t e,
                                           int n_obj, const Env3<MAXP>& s, uint16_t* cells) {
    uint4 h;
    h.x = s.pos0 | (s.or0 << 8) | (s.held0 << 16) | (s.pos1 << 24);
    h.y = s.or1 | (s.held1 << 8) | (min(s.t, 0xFFFFu) << 16);  // the wire format's u16 timestep saturates
    h.z = 0; h.w = 0;
#pragma unroll
    for (int k = 0; k < MAXP; ++k) {
        if ((uint32_t)k < C.n_pots) {
            wr_obj3(cells, L.pot_cell(k), s.ps[k]);
            if (k < 4) h.z |= s.tk[k] << (8 * (k & 3));
            else h.w |= s.tk[k] << (8 * (k & 3));
        }
    }
    st[e] = h;
    for (int p = 0; p < n_obj; ++p) {
        uint32_t ow[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int c = 16 * p + 4 * q;
            const uint32_t b0 = cells[(c + 0) * BLOCK] & 0xFFu, b1 = cells[(c + 1) * BLOCK] & 0xFFu;
            const uint32_t b2 = cells[(c + 2) * BLOCK] & 0xFFu, b3 = cells[(c + 3) * BLOCK] & 0xFFu;
            ow[q] = b0 | (b1 << 8) | (b2 << 16) | (b3 << 24);
        }
        st[(int64_t)(1 + p) * n + e] = make_uint4(ow[0], ow[1], ow[2], ow[3]);
    }
}

template <int MAXP>
__device__ __forceinline__ void env_reset3(const Lay L, int n_obj, Env3<MAXP>& s, uint16_t* cells) {
    s.pos0 = L.u8(L_START_POS); s.pos1 = L.u8(L_START_POS + 1);
    s.or0 = L.u8(L_START_OR); s.or1 = s.pos1 == 0xFFu ? 0u : L.u8(L_START_OR + 1);
    s.held0 = s.held1 = 0; s.t = 0; s.dcount = 0;
#pragma unroll
    for (int k = 0; k < MAXP; ++k) { s.ps[k] = 0; s.tk[k] = 0; s.pc[k] = PC_EMPTY; }
    for (int c = 0; c < n_obj * 16; ++c) reinterpret_cast<uint8_t*>(cells + c * BLOCK)[0] = 0;  // clear objects, keep terrain
}

// a randomized start (get_random_start_state_fn, mdp.py:1307-1369) drawn by draw_start, in Env3 form
template <int MAXP>
__device__ __forceinline__ void env_reset3_draw(const LayC& C, const Lay L, int n_obj, Env3<MAXP>& s, uint16_t* cells,
                                                const StartDraw& d) {
    env_reset3<MAXP>(L, n_obj, s, cells);
    s.pos0 = d.pos0; s.pos1 = d.pos1;
    s.held0 = d.held0; s.held1 = d.held1;
#pragma unroll
    for (int k = 0; k < MAXP; ++k) {
        if ((uint32_t)k < C.n_pots) {
            s.ps[k] = d.pot_obj((uint32_t)k);
            s.tk[k] = d.tick((uint32_t)k);
            s.pc[k] = pot_class(C, s.ps[k], s.tk[k]);
        }
    }
}

// sa / g / epoch: the start_state_fn of the batch (disabled: the standard start state), this env's global index and the
// epoch of a restart at this step
// before_restart(): called right before the env is put back to a start state — the caller's hook for moving the env to
// another layout first (regen_layout below): C and L are read again after it.
template <int MAXP, typename F>
__device__ __forceinline__ uint32_t finish_step3(const LayC& C, const Lay& L, int n_obj, uint16_t* cells, Env3<MAXP>& s,
                                                 int horizon, uint32_t options, const float4& r, float4& ep,
                                                 const StartArgs& sa, uint64_t g, uint32_t epoch, F&& before_restart) {
    ep.x += r.x; ep.y += r.y; ep.z += r.z; ep.w += r.w;
    uint32_t fl = 0;
    if (__builtin_expect((int)s.t >= horizon, 0)) {  // once per episode: keep the restart out of the straight-line path
        fl |= OC_F_DONE;
        if (options & OC_OPT_AUTO_RESET) {
            before_restart();
            if (sa.enabled)
                env_reset3_draw<MAXP>(C, L, n_obj, s, cells, draw_start(L, g, epoch, sa.seed_lo, sa.seed_hi, sa.random_start_pos, sa.thresh));
            else
                env_reset3<MAXP>(L, n_obj, s, cells);
            ep = make_float4(0.f, 0.f, 0.f, 0.f);
            fl |= OC_F_RESET;
        }
    }
    return fl;
}

template <int MAXP>
__device__ __forceinline__ uint32_t finish_step3(const LayC& C, const Lay& L, int n_obj, uint16_t* cells, Env3<MAXP>& s,
                                                 int horizon, uint32_t options, const float4& r, float4& ep,
                                                 const StartArgs& sa, uint64_t g, uint32_t epoch) {
    return finish_step3<MAXP>(C, L, n_obj, cells, s, horizon, options, r, ep, sa, g, epoch, []() {});
}

// A restart with StartArgs.regen_count > 0 (OvercookedEnv.reset(regen_mdp=True) over a generator of layouts, env.py:288-302):
// draw the env's next layout, record it, and switch this lane's layout pointer and constants to it.  False: nothing changed.
template <bool UNIFORM, bool LAY_LDS>
__device__ __forceinline__ bool regen_layout(const StartArgs& sa, uint64_t g, uint32_t epoch, int64_t e, const uint4* s_lay,
                                             const OcLayout* g_layouts, Lay& L, LayC& C, uint32_t* lid) {
    if (UNIFORM || !sa.regen_count) return false;
    *lid = draw_layout(sa, g, epoch);
    sa.layout_ids[e] = (uint16_t)*lid;
    L = LAY_LDS ? Lay{reinterpret_cast<const uint8_t*>(s_lay) + *lid * 256u}
                : Lay{reinterpret_cast<const uint8_t*>(g_layouts) + (size_t)*lid * 256u};
    C = load_consts<UNIFORM>(L);
    return true;
}

// stage the interact table (both variants, 3 840 bytes) in LDS; returns this lane's variant
__device__ __forceinline__ const uint8_t* stage_lut(uint2* s_lut, uint32_t old_dyn) {
    const uint2* src = reinterpret_cast<const uint2*>(&g_lut);
    for (int i = threadIdx.x; i < 2 * LUT_ENTRIES; i += BLOCK) s_lut[i] = src[i];
    return reinterpret_cast<const uint8_t*>(s_lut) + (old_dyn ? LUT_ENTRIES * 8 : 0);
}

// k_step3: transitions with caller-supplied actions (one per launch for oc_step, K for oc_step_many), table-driven
// interact (no event logging;
// oc_step with d_events != NULL uses k_step, whose predicate-network interact produces the event bits)
template <bool UNIFORM, int MAXP, bool LAY_LDS, bool FAST = false, bool EVENTS = false>
__global__ __launch_bounds__(BLOCK) void k_step3(const OcLayout* __restrict__ g_layouts, int n_layouts,
                                                 const uint16_t* layout_id, const uint4* st_in,
                                                 uint4* st_out, const uint8_t* __restrict__ actions,
                                                 float4* __restrict__ rewards, uint8_t* __restrict__ flags,
                                                 float4* __restrict__ ep_returns, int64_t n, int W, int n_obj,
                                                 int horizon, uint32_t options, int n_steps, StartArgs sa, EvArgs ea) {
    extern __shared__ __attribute__((aligned(16))) uint16_t s_cells3[];  // [n_obj * 16][BLOCK]
    __shared__ uint4 s_lay[LAY_LDS ? (UNIFORM ? 16 : LDS_LAYOUT_MAX * 16) : 1];  // one 256-byte record when the batch has one layout
    __shared__ uint2 s_lut[2 * LUT_ENTRIES];
    const uint32_t blk = xcd_block();  // (common.hpp)
    const int64_t e = (int64_t)blk * BLOCK + threadIdx.x;
    const bool active = e < n;
    for (int i = threadIdx.x; i < 2 * LUT_ENTRIES; i += BLOCK) s_lut[i] = reinterpret_cast<const uint2*>(&g_lut)[i];
    Lay L = stage_layouts<LAY_LDS>(g_layouts, n_layouts, layout_id, e, active, s_lay);  // contains the barrier
    if (!active) return;
    uint16_t* cells = s_cells3 + threadIdx.x;
    LayC C = load_consts<UNIFORM>(L);  // (L, C, lut and the floor mask follow the env to another layout at a restart with regen_count)
    const uint8_t* lut = reinterpret_cast<const uint8_t*>(s_lut) + (C.old_dyn ? LUT_ENTRIES * 8 : 0);
    const uint32_t delta4 = make_delta4(W);
    Env3<MAXP> s;
    load_env3<MAXP>(C, L, st_in, n, e, n_obj, s, cells);
    const uint64_t g = (uint64_t)(sa.env_offset + e);
    uint64_t floor_mask = FAST ? make_floor_mask(L, (int)L.u8(L_NCELLS)) : 0ull;
    float4 ep = ep_returns ? ep_returns[e] : make_float4(0.f, 0.f, 0.f, 0.f);
    // n_steps transitions with the caller's actions [n_steps][n][2] (oc_step: one; oc_step_many: K in one launch, the
    // env staying on chip in between).  The next step's actions are fetched while the current step runs.
    const uint16_t* act_k = reinterpret_cast<const uint16_t*>(actions) + (int64_t)blk * BLOCK;  // wave-uniform rows
    float4* rew_k = rewards + (int64_t)blk * BLOCK;
    uint8_t* flg_k = flags + (int64_t)blk * BLOCK;
    // The actions of eight steps are fetched together into a 128-bit queue: s_waitcnt vmcnt counts loads AND stores in
    // issue order, so a per-step look-ahead load makes every step wait for the previous step's output stores as well
    // (~1 us per step; measured on oc_step_many).  One such wait per eight steps instead.
    uint32_t q0 = 0, q1 = 0, q2 = 0, q3 = 0;
    for (int k = 0; k < n_steps; ++k) {
        if ((k & 7) == 0) {
            uint32_t v[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) v[j] = (k + j < n_steps) ? (act_k + (int64_t)j * n)[threadIdx.x] : 0u;
            act_k += 8 * n;
            q0 = v[0] | (v[1] << 16); q1 = v[2] | (v[3] << 16); q2 = v[4] | (v[5] << 16); q3 = v[6] | (v[7] << 16);
        }
        const uint32_t a01 = q0 & 0xFFFFu;
        q0 = __builtin_amdgcn_alignbit(q1, q0, 16); q1 = __builtin_amdgcn_alignbit(q2, q1, 16);
        q2 = __builtin_amdgcn_alignbit(q3, q2, 16); q3 >>= 16;
        const uint32_t a0 = a01 & 0xFFu, a1 = a01 >> 8;
        float4 r = make_float4(0.f, 0.f, 0.f, 0.f);
        uint32_t fl;
        if (__builtin_expect(a0 > 5u || a1 > 5u, 0)) {
            fl = OC_F_BAD_ACTION;  // get_state_transition raises ValueError (mdp.py:1394-1398): leave the env untouched
        } else {
            uint64_t ev = 0;
            env_step3<MAXP, FAST ? 2 : 0, EVENTS>(C, L, lut, cells, s, delta4, a0, a1, r, floor_mask, nullptr, &ev);
            fl = finish_step3<MAXP>(C, L, n_obj, cells, s, horizon, options, r, ep, sa, g, sa.epoch + (uint32_t)k, [&]() {
                uint32_t lid;
                if (regen_layout<UNIFORM, LAY_LDS>(sa, g, sa.epoch + (uint32_t)k, e, s_lay, g_layouts, L, C, &lid)) {
                    lut = reinterpret_cast<const uint8_t*>(s_lut) + (C.old_dyn ? LUT_ENTRIES * 8 : 0);
                    if (FAST) floor_mask = make_floor_mask(L, (int)L.u8(L_NCELLS));
                    // the cell words carry the terrain in their upper byte: the new layout's (the restart clears the objects)
                    for (int c = 0; c < n_obj * 16; ++c) cells[c * BLOCK] = (uint16_t)(L.terrain((uint32_t)c) << 8);
                }
            });
            if (EVENTS) {
                if (ea.events) ea.events[(int64_t)k * n + e] = ev;
                count_events(ea, e, ev, (fl & OC_F_DONE) != 0u, (fl & OC_F_RESET) != 0u);
            }
        }
        if (EVENTS && ea.events && (fl & OC_F_BAD_ACTION)) ea.events[(int64_t)k * n + e] = 0;
        rew_k[threadIdx.x] = r;
        flg_k[threadIdx.x] = (uint8_t)fl;
        rew_k += n;
        flg_k += n;
    }
    store_env3<MAXP>(C, L, st_out, n, e, n_obj, s, cells);
    if (ep_returns) ep_returns[e] = ep;
}
