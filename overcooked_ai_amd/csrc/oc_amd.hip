// oc_amd.hip — MI355X (gfx950 / CDNA4) kernels and C-ABI for the batched Overcooked hot path.
//
// What runs here (reference = HumanCompatibleAI/overcooked_ai, "mdp.py" = src/overcooked_ai_py/mdp/overcooked_mdp.py):
//   env_step()            get_state_transition            mdp.py:1375  (interacts 1432 -> movement 1644 -> env effects 1691)
//   k_step / k_rollout    OvercookedEnv.step bookkeeping  overcooked_env.py:244-274, is_done 321-325
//   k_encode              lossless_state_encoding         mdp.py:2385-2561
//   k_reset               get_standard_start_state        mdp.py:1297-1305
//
// Execution model: one lane per env, 64-lane wavefronts, 256-lane workgroups.  This is integer /
// indexing work (no MFMA).  Per-env state arrives as coalesced 16-byte planes (1 KiB per wavefront
// per plane), the object bytes of the grid are staged in LDS in [dword][lane] order — bank =
// lane % 32 whatever cell a lane touches, so divergent per-lane cell indices never conflict — and
// the compiled layout table (terrain tile, pot cells, recipe LUTs) is staged in LDS once per
// workgroup.  See DESIGN.md for the data layout and the roofline of each kernel.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "../../include/oc_amd.h"

namespace {

constexpr int BLOCK = 256;
constexpr int LDS_LAYOUT_MAX = 32;  // layout tables up to this many entries are staged in LDS (8 KiB)

// byte offsets inside OcLayout (include/oc_amd.h)
constexpr int L_NCELLS = 2, L_NPOTS = 3, L_NPLAYERS = 4, L_OLDDYN = 5, L_START_POS = 8, L_START_OR = 10, L_POT_CELL = 16,
              L_PCLASS = 24, L_REW = 32, L_COOK = 48, L_VALUE = 64, L_TERRAIN = 128;
static_assert(sizeof(OcLayout) == 256, "OcLayout must be 256 bytes");

thread_local char g_err[256] = "";

// ------------------------------------------------------------------------------------------
// Layout accessors.  `base` points at one 256-byte OcLayout, either in LDS or in global memory;
// after inlining the compiler resolves the address space from the pointer's origin.
// ------------------------------------------------------------------------------------------
struct Lay {
    const uint8_t* base;
    __device__ __forceinline__ uint32_t u8(int off) const { return base[off]; }
    __device__ __forceinline__ uint32_t u32(int off) const { return *reinterpret_cast<const uint32_t*>(base + off); }
    __device__ __forceinline__ uint32_t n_pots() const { return u8(L_NPOTS); }
    __device__ __forceinline__ uint32_t n_players() const { return u8(L_NPLAYERS); }
    __device__ __forceinline__ bool old_dynamics() const { return u8(L_OLDDYN) != 0; }
    __device__ __forceinline__ uint32_t pot_cell(int k) const { return u8(L_POT_CELL + k); }
    __device__ __forceinline__ uint32_t terrain(uint32_t c) const { return u8(L_TERRAIN + c); }
    __device__ __forceinline__ uint32_t cook_time(uint32_t idx) const { return u8(L_COOK + idx); }
    __device__ __forceinline__ float f32(int off) const { return *reinterpret_cast<const float*>(base + off); }
    __device__ __forceinline__ float value(uint32_t idx) const { return f32(L_VALUE + 4 * idx); }
    __device__ __forceinline__ float rew_placement() const { return f32(L_REW); }
    __device__ __forceinline__ float rew_dish() const { return f32(L_REW + 4); }
    __device__ __forceinline__ float rew_soup() const { return f32(L_REW + 8); }
};

// ------------------------------------------------------------------------------------------
// Working representation of one env inside the step / rollout kernels.
//
//  * registers: both players, the timestep, and — per pot slot — the soup code and cooking tick
//    (pots are the only cells whose content is needed every step: stale pot_states, env effects);
//    the number of loose dishes on counters (is_dish_pickup_useful needs "no dish on any counter");
//  * LDS: one 16-bit word per grid cell = object code (low byte, wire format) | terrain byte (high
//    byte: type | pot slot << 3), stored as dwords[cell / 2][lane].  One ds_read_u16 answers "what
//    terrain is there and what lies on it"; bank = lane % 32 for every cell, so the divergent
//    per-lane cell indices of a wavefront never conflict.  Per-env (divergent) terrain costs nothing
//    extra in the step loop.  The object bytes of pot cells are stale while the kernel runs (the
//    registers are authoritative) and are written back before the planes are stored.
// ------------------------------------------------------------------------------------------
template <int MAXP>
struct EnvW {
    uint32_t pos0, or0, held0, pos1, or1, held1, t;
    uint32_t tk[MAXP];  // cooking_tick + 1 per pot slot (0 = idle)
    uint32_t ps[MAXP];  // soup code per pot slot (0 = empty pot)
    int32_t dcount;     // loose dishes lying on counters
};

// Per-layout constants the step loop needs every iteration.  With a single layout for the whole batch
// they are made wave-uniform (SGPRs) via readfirstlane.
struct LayC {
    uint32_t old_dyn, n_pots;
    float rew_place, rew_dish, rew_soup;
    uint32_t cook[4];    // cook_time[n_onion + 4*n_tomato] as 4 dwords: dword n_tomato, byte n_onion
    uint32_t pclass[2];  // potting class nibbles (events only)
};

template <bool UNIFORM>
__device__ __forceinline__ uint32_t uni(uint32_t v) {
    return UNIFORM ? (uint32_t)__builtin_amdgcn_readfirstlane((int)v) : v;
}
template <bool UNIFORM>
__device__ __forceinline__ float unif(float v) {
    return UNIFORM ? __uint_as_float((uint32_t)__builtin_amdgcn_readfirstlane((int)__float_as_uint(v))) : v;
}

template <bool UNIFORM>
__device__ __forceinline__ LayC load_consts(const Lay L) {
    LayC C;
    C.old_dyn = uni<UNIFORM>(L.u8(L_OLDDYN));
    C.n_pots = uni<UNIFORM>(L.u8(L_NPOTS));
    C.rew_place = unif<UNIFORM>(L.rew_placement());
    C.rew_dish = unif<UNIFORM>(L.rew_dish());
    C.rew_soup = unif<UNIFORM>(L.rew_soup());
#pragma unroll
    for (int i = 0; i < 4; ++i) C.cook[i] = uni<UNIFORM>(L.u32(L_COOK + 4 * i));
    C.pclass[0] = uni<UNIFORM>(L.u32(L_PCLASS));
    C.pclass[1] = uni<UNIFORM>(L.u32(L_PCLASS + 4));
    return C;
}

template <int STRIDE = BLOCK>
__device__ __forceinline__ uint32_t rd_cell16(const uint32_t* cellw, uint32_t c) {
    return reinterpret_cast<const uint16_t*>(cellw + (c >> 1) * STRIDE)[c & 1];
}
template <int STRIDE = BLOCK>
__device__ __forceinline__ void wr_cell_obj(uint32_t* cellw, uint32_t c, uint32_t v) {
    reinterpret_cast<uint8_t*>(cellw + (c >> 1) * STRIDE)[(c & 1) * 2] = (uint8_t)v;
}

// recipe index n_onion + 4*n_tomato of a soup code
__device__ __forceinline__ uint32_t recipe_idx(uint32_t soup) {
    const uint32_t n = (soup >> 3) & 3u, nt = __popc(soup & 7u);
    return (n - nt) + 4u * nt;
}

// Recipe.time of a soup code through the 16-byte LUT held in 4 registers
__device__ __forceinline__ uint32_t cook_of(const LayC& C, uint32_t soup) {
    const uint32_t n = (soup >> 3) & 3u, nt = __popc(soup & 7u), no = n - nt;
    // byte (n_onion + 4*(n_tomato & 1)) of the dword pair {cook[2j+1], cook[2j]}: v_perm_b32 with selector
    // 0x0C (constant 0) in the upper lanes picks it in one instruction per pair
    const uint32_t sel = 0x0C0C0C00u | no | ((nt & 1u) << 2);
    const uint32_t lo = __builtin_amdgcn_perm(C.cook[1], C.cook[0], sel);
    const uint32_t hi = __builtin_amdgcn_perm(C.cook[3], C.cook[2], sel);
    return nt >= 2u ? hi : lo;
}

// cell-index delta of direction d (0..3 = N,S,E,W) from a packed table of 4 signed bytes
__device__ __forceinline__ uint32_t step_cell(uint32_t c, uint32_t d, uint32_t delta4) {
    return c + (uint32_t)__builtin_amdgcn_sbfe((int)delta4, 8u * d, 8u);
}

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
        // log_object_pickup / drop / potting and their usefulness predicates (mdp.py:2121-2308)
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
        r.ev = e;
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
    h.y = s.or1 | (s.held1 << 8) | (s.t << 16);
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

// Philox4x32-10 (Salmon et al., SC'11).  Same constants/rounds as oracle_philox4x32_10.
__device__ __forceinline__ void philox4x32_10(uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3, uint32_t k0,
                                              uint32_t k1, uint32_t (&out)[4]) {
#pragma unroll
    for (int r = 0; r < 10; ++r) {
        // one 32 x 32 -> 64 multiply (v_mad_u64_u32) per product instead of a mul_hi / mul_lo pair
        const uint64_t p0 = (uint64_t)0xD2511F53u * c0, p1 = (uint64_t)0xCD9E8D57u * c2;
        const uint32_t hi0 = (uint32_t)(p0 >> 32), lo0 = (uint32_t)p0;
        const uint32_t hi1 = (uint32_t)(p1 >> 32), lo1 = (uint32_t)p1;
        const uint32_t n0 = hi1 ^ c1 ^ k0, n2 = hi0 ^ c3 ^ k1;
        c0 = n0; c1 = lo1; c2 = n2; c3 = lo0;
        k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
    }
    out[0] = c0; out[1] = c1; out[2] = c2; out[3] = c3;
}

// ------------------------------------------------------------------------------------------
// Workgroup prologue shared by the kernels: stage the layout table in LDS and return this lane's
// layout pointer.
// ------------------------------------------------------------------------------------------
template <bool LAY_LDS>
__device__ __forceinline__ Lay stage_layouts(const OcLayout* __restrict__ g_layouts, int n_layouts,
                                             const uint16_t* __restrict__ layout_id, int64_t e, bool active,
                                             uint4* s_lay) {
    uint32_t lid = 0;
    if (layout_id != nullptr && active) lid = layout_id[e];
    if (LAY_LDS) {
        const uint4* src = reinterpret_cast<const uint4*>(g_layouts);
        for (int i = threadIdx.x; i < n_layouts * 16; i += BLOCK) s_lay[i] = src[i];
        __syncthreads();
        return Lay{reinterpret_cast<const uint8_t*>(s_lay) + lid * 256u};
    } else {
        return Lay{reinterpret_cast<const uint8_t*>(g_layouts) + (size_t)lid * 256u};
    }
}

__device__ __forceinline__ uint32_t make_delta4(int W) {
    // signed byte deltas of N, S, E, W for row-major cells (actions.py:12-16)
    return ((uint32_t)(-W) & 0xFFu) | (((uint32_t)W & 0xFFu) << 8) | (1u << 16) | (0xFFu << 24);
}

// post-transition bookkeeping shared by k_step and k_rollout (env.py:266-267, 321-325, 387-392)
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
    __shared__ uint4 s_lay[LAY_LDS ? LDS_LAYOUT_MAX * 16 : 1];
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
    __shared__ uint4 s_lay[LAY_LDS ? LDS_LAYOUT_MAX * 16 : 1];
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
template <int MAXP>
__device__ __forceinline__ IOut3 interact3(const Lay L, const uint8_t* s_lut, bool act, uint32_t h, uint32_t c16,
                                           const uint32_t (&ps)[MAXP], const uint32_t (&tkr)[MAXP],
                                           const uint32_t (&pcr)[MAXP]) {
    IOut3 r;
    const uint32_t tc = c16 >> 8;
    const uint32_t type = act ? (tc & 7u) : 7u;  // 7 = no interact: every entry of that row is a no-op
    const uint32_t slot = tc >> 3;
    uint32_t pso = 0, tkv = 0, pcv = 0;
#pragma unroll
    for (int k = 0; k < MAXP; ++k) {
        const bool sel = slot == (uint32_t)k;
        pso = sel ? ps[k] : pso;
        tkv = sel ? tkr[k] : tkv;
        pcv = sel ? pcr[k] : pcv;
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
    const float value = L.value(recipe_idx(h) & 15u);  // unconditional read: keeps both players' look-ups in one block
    r.sparse = (r.flags & LF_SERVE) ? value : 0.f;      // deliver_soup (mdp.py:1631-1642)
    return r;
}

template <int MAXP>
__device__ __forceinline__ void apply_pot3(Env3<MAXP>& s, const IOut3& r) {
#pragma unroll
    for (int k = 0; k < MAXP; ++k) {
        const bool upd = ((r.flags & LF_POT_UPD) != 0u) & (r.slot == (uint32_t)k);
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

// FAST = one layout for the whole batch, two players, at most 64 cells: "is there a second player" folds away and the
// floor test of resolve_movement becomes a bit test against a wave-uniform 64-bit mask instead of two LDS reads.
template <int MAXP, bool FAST>
__device__ __forceinline__ Probe3 probe3(const uint16_t* cells, const Env3<MAXP>& s, uint32_t delta4, uint32_t a0,
                                         uint32_t a1) {
    Probe3 q;
    const bool two = FAST || s.pos1 != 0xFFu;
    q.f0 = step_cell(s.pos0, s.or0, delta4);
    q.f1 = two ? step_cell(s.pos1, s.or1, delta4) : q.f0;
    q.m0 = a0 < 4u ? step_cell(s.pos0, a0, delta4) : s.pos0;
    q.m1 = (two & (a1 < 4u)) ? step_cell(s.pos1, a1, delta4) : (two ? s.pos1 : s.pos0);
    q.c_f0 = rd_cell3(cells, q.f0); q.c_f1 = rd_cell3(cells, q.f1);
    if (!FAST) { q.c_m0 = rd_cell3(cells, q.m0); q.c_m1 = rd_cell3(cells, q.m1); }
    else { q.c_m0 = 0; q.c_m1 = 0; }
    return q;
}

template <int MAXP, bool FAST>
__device__ __forceinline__ void step3_main(const LayC& C, const Lay L, const uint8_t* s_lut, uint16_t* cells,
                                           Env3<MAXP>& s, uint32_t a0, uint32_t a1, const Probe3& q, float4& r,
                                           uint64_t floor_mask) {
    const bool two = FAST || s.pos1 != 0xFFu;
    const bool mv0 = a0 < 4u, mv1 = two & (a1 < 4u);
    const uint32_t f0 = q.f0, f1 = q.f1, c_f0 = q.c_f0, c_f1 = q.c_f1;

    // pot_states before any interact (mdp.py:1439): ready / cooking / 1..2 idle items  <=>  class not in {empty, idle 3}
    uint32_t useful_pots = 0;
#pragma unroll
    for (int k = 0; k < MAXP; ++k) useful_pots += ((s.pc[k] != PC_EMPTY) & (s.pc[k] != PC_IDLE3)) ? 1u : 0u;

    const bool act0 = a0 == OC_A_INTERACT, act1 = two & (a1 == OC_A_INTERACT);
    const IOut3 r0 = interact3<MAXP>(L, s_lut, act0, s.held0, c_f0, s.ps, s.tk, s.pc);
    IOut3 r1 = interact3<MAXP>(L, s_lut, act1, s.held1, c_f1, s.ps, s.tk, s.pc);
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
    if (__builtin_expect(conflict, 0)) r1 = interact3<MAXP>(L, s_lut, act1, s.held1, c_f1_live, s.ps, s.tk, s.pc);
    const bool du1 = two & (((s.held0 == OC_O_DISH) ? 1u : 0u) < useful_pots) & (s.dcount == 0);
    const float sh1 = ((r1.flags & LF_PLACE) ? C.rew_place : 0.f) + ((r1.flags & LF_PLATE) ? C.rew_soup : 0.f) +
                      ((((r1.flags & LF_TAKE_DISH) != 0u) & du1) ? C.rew_dish : 0.f);
    s.held1 = r1.new_h;
    s.dcount += r1.ddelta;
    apply_pot3<MAXP>(s, r1);
    wr_obj3(cells, f0, r0.cell_obj);
    wr_obj3(cells, f1, (r1.flags & LF_SWAP) ? r1.cell_obj : (c_f1_live & 0xFFu));
    r = make_float4(r0.sparse, r1.sparse, sh0, sh1);

    // resolve_movement (mdp.py:1644-1727)
    const bool fl0 = FAST ? ((floor_mask >> q.m0) & 1ull) != 0ull : ((q.c_m0 >> 8) & 7u) == OC_T_FLOOR;
    const bool fl1 = FAST ? ((floor_mask >> q.m1) & 1ull) != 0ull : ((q.c_m1 >> 8) & 7u) == OC_T_FLOOR;
    const uint32_t np0 = (mv0 & fl0) ? q.m0 : s.pos0;
    const uint32_t np1 = (mv1 & fl1) ? q.m1 : s.pos1;
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

template <int MAXP, bool FAST = false>
__device__ __forceinline__ void env_step3(const LayC& C, const Lay L, const uint8_t* s_lut, uint16_t* cells,
                                          Env3<MAXP>& s, uint32_t delta4, uint32_t a0, uint32_t a1, float4& r,
                                          uint64_t floor_mask = 0) {
    const Probe3 q = probe3<MAXP, FAST>(cells, s, delta4, a0, a1);
    step3_main<MAXP, FAST>(C, L, s_lut, cells, s, a0, a1, q, r, floor_mask);
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
    h.y = s.or1 | (s.held1 << 8) | (s.t << 16);
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

template <int MAXP>
__device__ __forceinline__ uint32_t finish_step3(const Lay L, int n_obj, uint16_t* cells, Env3<MAXP>& s, int horizon,
                                                 uint32_t options, const float4& r, float4& ep) {
    ep.x += r.x; ep.y += r.y; ep.z += r.z; ep.w += r.w;
    uint32_t fl = 0;
    if ((int)s.t >= horizon) {
        fl |= OC_F_DONE;
        if (options & OC_OPT_AUTO_RESET) {
            env_reset3<MAXP>(L, n_obj, s, cells);
            ep = make_float4(0.f, 0.f, 0.f, 0.f);
            fl |= OC_F_RESET;
        }
    }
    return fl;
}

// stage the interact table (both variants, 3 840 bytes) in LDS; returns this lane's variant
__device__ __forceinline__ const uint8_t* stage_lut(uint2* s_lut, uint32_t old_dyn) {
    const uint2* src = reinterpret_cast<const uint2*>(&g_lut);
    for (int i = threadIdx.x; i < 2 * LUT_ENTRIES; i += BLOCK) s_lut[i] = src[i];
    return reinterpret_cast<const uint8_t*>(s_lut) + (old_dyn ? LUT_ENTRIES * 8 : 0);
}

template <bool UNIFORM, int MAXP, bool LAY_LDS, bool FAST = false>
__global__ __launch_bounds__(BLOCK) void k_rollout3(const OcLayout* __restrict__ g_layouts, int n_layouts,
                                                    const uint16_t* __restrict__ layout_id, uint4* st,
                                                    float4* __restrict__ rewards, uint8_t* __restrict__ flags,
                                                    float4* __restrict__ ep_returns, int64_t n, int W, int n_obj,
                                                    int horizon, uint32_t options, uint32_t seed_lo, uint32_t seed_hi,
                                                    int64_t env_offset, int64_t t0, int n_steps) {
    extern __shared__ __attribute__((aligned(16))) uint16_t s_cells3[];  // [n_obj * 16][BLOCK]
    __shared__ uint4 s_lay[LAY_LDS ? LDS_LAYOUT_MAX * 16 : 1];
    __shared__ uint2 s_lut[2 * LUT_ENTRIES];
    const int64_t e = (int64_t)blockIdx.x * BLOCK + threadIdx.x;
    const bool active = e < n;
    for (int i = threadIdx.x; i < 2 * LUT_ENTRIES; i += BLOCK) s_lut[i] = reinterpret_cast<const uint2*>(&g_lut)[i];
    const Lay L = stage_layouts<LAY_LDS>(g_layouts, n_layouts, layout_id, e, active, s_lay);  // contains the barrier
    if (!active) return;
    uint16_t* cells = s_cells3 + threadIdx.x;
    const LayC C = load_consts<UNIFORM>(L);
    const uint8_t* lut = reinterpret_cast<const uint8_t*>(s_lut) + (C.old_dyn ? LUT_ENTRIES * 8 : 0);
    const uint32_t delta4 = make_delta4(W);
    Env3<MAXP> s;
    load_env3<MAXP>(C, L, st, n, e, n_obj, s, cells);
    const uint64_t floor_mask = FAST ? make_floor_mask(L, (int)L.u8(L_NCELLS)) : 0ull;
    float4 ep = ep_returns ? ep_returns[e] : make_float4(0.f, 0.f, 0.f, 0.f);
    const uint64_t g = (uint64_t)(env_offset + e);
    const uint32_t g_lo = (uint32_t)g, g_hi = (uint32_t)(g >> 32);
    uint32_t rnd[4] = {0, 0, 0, 0};
    // outputs of step k live at [k][e]: a wave-uniform base per step (SALU) + this lane's 32-bit offset
    float4* const rew_blk = rewards ? rewards + (int64_t)blockIdx.x * BLOCK : nullptr;
    uint8_t* const flg_blk = flags ? flags + (int64_t)blockIdx.x * BLOCK : nullptr;
    // (issuing step k+1's cell reads before step k's tail was tried and measured: no gain — the loop is bound by
    //  instruction issue, not by LDS latency)
    if (FAST) {
        // One Philox block = 8 steps: the loop is unrolled over the block so that the word / digit position of every
        // step is a compile-time constant (x runs through w, 6w, 36w, 216w: no word select, no x36 multiply) and the
        // refresh test and the back edge are paid once per 8 steps.  A launch may start and end inside a block.
        int k = 0;
        uint32_t s8 = (uint32_t)t0 & 7u;
        uint64_t blk = (uint64_t)t0 >> 3;
        philox4x32_10((uint32_t)blk, g_lo, g_hi, (uint32_t)(blk >> 32), seed_lo, seed_hi, rnd);
        uint32_t x = (s8 & 1u) ? rnd[s8 >> 1] * 36u : 0u;
#define OC_STEP(S8)                                                                                      \
    {                                                                                                    \
        if (((S8) & 1u) == 0u) x = rnd[(S8) >> 1];                                                       \
        const uint32_t a0 = __umulhi(x, 6u);                                                             \
        x *= 6u;                                                                                         \
        const uint32_t a1 = __umulhi(x, 6u);                                                             \
        x *= 6u;                                                                                         \
        float4 r;                                                                                        \
        env_step3<MAXP, FAST>(C, L, lut, cells, s, delta4, a0, a1, r, floor_mask);                       \
        const uint32_t fl = finish_step3<MAXP>(L, n_obj, cells, s, horizon, options, r, ep);             \
        if (rew_blk) (rew_blk + (int64_t)k * n)[threadIdx.x] = r;                                        \
        if (flg_blk) (flg_blk + (int64_t)k * n)[threadIdx.x] = (uint8_t)fl;                              \
        if (++k == n_steps) break;                                                                       \
    }
        for (;;) {
            switch (s8) {
                case 0: OC_STEP(0u)  // fall through: the rest of the block
                case 1: OC_STEP(1u)
                case 2: OC_STEP(2u)
                case 3: OC_STEP(3u)
                case 4: OC_STEP(4u)
                case 5: OC_STEP(5u)
                case 6: OC_STEP(6u)
                default: OC_STEP(7u)
            }
            if (k == n_steps) break;
            s8 = 0u;
            ++blk;
            philox4x32_10((uint32_t)blk, g_lo, g_hi, (uint32_t)(blk >> 32), seed_lo, seed_hi, rnd);
        }
#undef OC_STEP
    } else {
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
            env_step3<MAXP, FAST>(C, L, lut, cells, s, delta4, a0, a1, r, floor_mask);
            const uint32_t fl = finish_step3<MAXP>(L, n_obj, cells, s, horizon, options, r, ep);
            if (rew_blk) (rew_blk + (int64_t)k * n)[threadIdx.x] = r;
            if (flg_blk) (flg_blk + (int64_t)k * n)[threadIdx.x] = (uint8_t)fl;
        }
    }
    store_env3<MAXP>(C, L, st, n, e, n_obj, s, cells);
    if (ep_returns) ep_returns[e] = ep;
}

// k_step3: one transition per launch with caller-supplied actions, table-driven interact (no event logging;
// oc_step with d_events != NULL uses k_step, whose predicate-network interact produces the event bits)
template <bool UNIFORM, int MAXP, bool LAY_LDS, bool FAST = false>
__global__ __launch_bounds__(BLOCK) void k_step3(const OcLayout* __restrict__ g_layouts, int n_layouts,
                                                 const uint16_t* __restrict__ layout_id, const uint4* st_in,
                                                 uint4* st_out, const uint8_t* __restrict__ actions,
                                                 float4* __restrict__ rewards, uint8_t* __restrict__ flags,
                                                 float4* __restrict__ ep_returns, int64_t n, int W, int n_obj,
                                                 int horizon, uint32_t options) {
    extern __shared__ __attribute__((aligned(16))) uint16_t s_cells3[];  // [n_obj * 16][BLOCK]
    __shared__ uint4 s_lay[LAY_LDS ? LDS_LAYOUT_MAX * 16 : 1];
    __shared__ uint2 s_lut[2 * LUT_ENTRIES];
    const int64_t e = (int64_t)blockIdx.x * BLOCK + threadIdx.x;
    const bool active = e < n;
    for (int i = threadIdx.x; i < 2 * LUT_ENTRIES; i += BLOCK) s_lut[i] = reinterpret_cast<const uint2*>(&g_lut)[i];
    const Lay L = stage_layouts<LAY_LDS>(g_layouts, n_layouts, layout_id, e, active, s_lay);  // contains the barrier
    if (!active) return;
    uint16_t* cells = s_cells3 + threadIdx.x;
    const LayC C = load_consts<UNIFORM>(L);
    const uint8_t* lut = reinterpret_cast<const uint8_t*>(s_lut) + (C.old_dyn ? LUT_ENTRIES * 8 : 0);
    const uint32_t delta4 = make_delta4(W);
    Env3<MAXP> s;
    load_env3<MAXP>(C, L, st_in, n, e, n_obj, s, cells);
    const uint32_t a01 = reinterpret_cast<const uint16_t*>(actions)[e];
    const uint32_t a0 = a01 & 0xFFu, a1 = a01 >> 8;
    float4 r = make_float4(0.f, 0.f, 0.f, 0.f);
    uint32_t fl;
    float4 ep = ep_returns ? ep_returns[e] : make_float4(0.f, 0.f, 0.f, 0.f);
    if (a0 > 5u || a1 > 5u) {
        fl = OC_F_BAD_ACTION;  // get_state_transition raises ValueError (mdp.py:1394-1398): leave the env untouched
    } else {
        env_step3<MAXP, FAST>(C, L, lut, cells, s, delta4, a0, a1, r,
                              FAST ? make_floor_mask(L, (int)L.u8(L_NCELLS)) : 0ull);
        fl = finish_step3<MAXP>(L, n_obj, cells, s, horizon, options, r, ep);
    }
    store_env3<MAXP>(C, L, st_out, n, e, n_obj, s, cells);
    rewards[e] = r;
    flags[e] = (uint8_t)fl;
    if (ep_returns) ep_returns[e] = ep;
}

// ------------------------------------------------------------------------------------------
// k_rollout_pair: the fused random-policy rollout with TWO lanes per env (lane parity = player index = pot
// slot owned).  One wavefront per SIMD issues at most one instruction every four cycles, so with 65 536 envs
// (1 024 lane-per-env wavefronts on 1 024 SIMDs) the lane-per-env kernel is bound by the length of its own
// instruction stream.  Splitting each env over a lane pair halves that stream and doubles the wavefronts per
// SIMD.  The players exchange what the other needs with DPP quad-permutes (v_mov_b32_dpp, no LDS):
//   - before the interacts: hand, position, faced cell, pot registers;
//   - after them: the packed result of the interact (new hand, counter byte, pot update, dish-count delta).
// Interact order (player 0 before player 1, mdp.py:1446) is kept exactly: both lanes evaluate `interact` on the
// pre-step pots and cells; player 1's inputs that player 0 can change (hand, dish count) arrive as data, and
// the pairs where player 0 changed the very cell or pot player 1 uses replay player 1's interact on the live
// state.  Requires 2-player layouts with at most 2 pots (every layout shipped by the reference).
// ------------------------------------------------------------------------------------------
constexpr int PAIR_ENVS = BLOCK / 2;

__device__ __forceinline__ uint32_t xchg(uint32_t v) {  // value held by the other lane of the pair
    return (uint32_t)__builtin_amdgcn_mov_dpp((int)v, 0xB1 /* quad_perm [1,0,3,2] */, 0xF, 0xF, true);
}

struct PairW {
    uint32_t pos, ori, held, t;
    uint32_t ps, tk, pc;  // the pot slot this lane owns (slot index = lane parity): soup code, tick + 1, class
    int32_t dcount;       // loose dishes on counters (kept identical in both lanes)
};

// interact result packed for the partner: new hand | counter byte | new pot soup | new tick, and
// flags | slot << 8 | new class << 12 | (dish delta + 1) << 16
__device__ __forceinline__ uint32_t pack_res1(const IOut3& r) {
    return r.new_h | (r.cell_obj << 8) | (r.new_o << 16) | (r.new_tk << 24);
}
__device__ __forceinline__ uint32_t pack_res2(const IOut3& r) {
    return r.flags | (r.slot << 8) | (r.new_pc << 12) | ((uint32_t)(r.ddelta + 1) << 16);
}

__device__ __forceinline__ void pair_step(const LayC& C, const Lay L, const uint8_t* s_lut, uint32_t* cellw, uint32_t p,
                                          PairW& s, uint32_t delta4, uint32_t a, float& sparse, float& shaped) {
    const bool lane1 = p != 0u;
    const bool mv = a < 4u;
    const uint32_t f = step_cell(s.pos, s.ori, delta4);
    const uint32_t m = mv ? step_cell(s.pos, a, delta4) : s.pos;
    const uint32_t c_f = rd_cell16<PAIR_ENVS>(cellw, f), c_m = rd_cell16<PAIR_ENVS>(cellw, m);
    // the partner's pre-step view
    const uint32_t held_o = xchg(s.held), pos_o = xchg(s.pos), f_o = xchg(f);
    const uint32_t pot_own = s.ps | (s.tk << 8) | (s.pc << 16);
    const uint32_t pot_oth = xchg(pot_own);
    // pot_states before any interact (mdp.py:1439): class not in {empty, idle with 3 items}
    const uint32_t u_own = ((s.pc != PC_EMPTY) & (s.pc != PC_IDLE3)) ? 1u : 0u;
    const uint32_t useful_pots = u_own + xchg(u_own);
    const uint32_t pot0 = lane1 ? pot_oth : pot_own, pot1 = lane1 ? pot_own : pot_oth;  // by slot
    uint32_t ps_arr[2] = {pot0 & 0xFFu, pot1 & 0xFFu};
    uint32_t tk_arr[2] = {(pot0 >> 8) & 0xFFu, (pot1 >> 8) & 0xFFu};
    uint32_t pc_arr[2] = {pot0 >> 16, pot1 >> 16};
    const bool act = a == OC_A_INTERACT;
    IOut3 r = interact3<2>(L, s_lut, act, s.held, c_f, ps_arr, tk_arr, pc_arr);
    // hand the result to the partner
    uint32_t o1 = xchg(pack_res1(r)), o2 = xchg(pack_res2(r));
    const bool same_cell = f == f_o;
    {
        // player 1 replays when player 0 changed the counter cell or the pot it uses (player 0's lane follows it
        // into the branch only to receive the final result)
        const bool o_swapX = (o2 & LF_SWAP) != 0u, o_pot_upd = (o2 & LF_POT_UPD) != 0u;
        const uint32_t o_slot = (o2 >> 8) & 7u;
        const bool conflict = lane1 & act & ((same_cell & o_swapX) |
                                            (o_pot_upd & (((c_f >> 8) & 7u) == OC_T_POT) & ((c_f >> 11) == o_slot)));
        const bool cpair = conflict | (xchg(conflict ? 1u : 0u) != 0u);
        if (__builtin_expect(cpair, 0)) {
            const uint32_t o_new_o = (o1 >> 16) & 0xFFu, o_new_tk = o1 >> 24, o_new_pc = (o2 >> 12) & 7u;
#pragma unroll
            for (int k = 0; k < 2; ++k) {
                const bool upd = o_pot_upd & (o_slot == (uint32_t)k);
                ps_arr[k] = upd ? o_new_o : ps_arr[k];
                tk_arr[k] = upd ? o_new_tk : tk_arr[k];
                pc_arr[k] = upd ? o_new_pc : pc_arr[k];
            }
            const uint32_t c_f_live = (same_cell & o_swapX) ? ((c_f & 0xFF00u) | ((o1 >> 8) & 0xFFu)) : c_f;
            const IOut3 r2 = interact3<2>(L, s_lut, act, s.held, c_f_live, ps_arr, tk_arr, pc_arr);
            if (lane1) r = r2;
            const uint32_t n1 = xchg(pack_res1(r)), n2 = xchg(pack_res2(r));
            if (!lane1) { o1 = n1; o2 = n2; }
        }
    }
    const uint32_t o_new_h = o1 & 0xFFu, o_cell_obj = (o1 >> 8) & 0xFFu, o_new_o = (o1 >> 16) & 0xFFu, o_new_tk = o1 >> 24;
    const bool o_swapX = (o2 & LF_SWAP) != 0u, o_pot_upd = (o2 & LF_POT_UPD) != 0u;
    const uint32_t o_slot = (o2 >> 8) & 7u, o_new_pc = (o2 >> 12) & 7u;
    const int32_t o_dd = (int32_t)((o2 >> 16) & 3u) - 1;
    // is_dish_pickup_useful (mdp.py:2180-2204) on the live hands/counters: player 1 sees player 0's new hand
    const uint32_t other_live = lane1 ? o_new_h : held_o;
    const int32_t dcount_live = s.dcount + (lane1 ? o_dd : 0);
    const bool dish_useful = (((other_live == OC_O_DISH) ? 1u : 0u) < useful_pots) & (dcount_live == 0);
    sparse = r.sparse;
    shaped = ((r.flags & LF_PLACE) ? C.rew_place : 0.f) + ((r.flags & LF_PLATE) ? C.rew_soup : 0.f) +
             ((((r.flags & LF_TAKE_DISH) != 0u) & dish_useful) ? C.rew_dish : 0.f);
    // apply: hand, dish count, the pot slot this lane owns (player 1's update wins when both hit it: it was replayed)
    s.held = r.new_h;
    s.dcount += r.ddelta + o_dd;
    {
        const bool mine = ((r.flags & LF_POT_UPD) != 0u) & (r.slot == p), theirs = o_pot_upd & (o_slot == p);
        const bool hit1 = lane1 ? mine : theirs, hit0 = lane1 ? theirs : mine;
        const uint32_t mine_pk = r.new_o | (r.new_tk << 8) | (r.new_pc << 16);
        const uint32_t theirs_pk = o_new_o | (o_new_tk << 8) | (o_new_pc << 16);
        const uint32_t pk_1 = lane1 ? mine_pk : theirs_pk, pk_0 = lane1 ? theirs_pk : mine_pk;
        const uint32_t fin = hit1 ? pk_1 : hit0 ? pk_0 : pot_own;
        s.ps = fin & 0xFFu; s.tk = (fin >> 8) & 0xFFu; s.pc = fin >> 16;
    }
    {
        // counter byte of the faced cell; when both face one cell both lanes store the same final value
        const bool my_swap = (r.flags & LF_SWAP) != 0u;
        const bool sw1 = lane1 ? my_swap : o_swapX, sw0 = lane1 ? o_swapX : my_swap;
        const uint32_t ob1 = lane1 ? r.cell_obj : o_cell_obj, ob0 = lane1 ? o_cell_obj : r.cell_obj;
        const uint32_t final_same = sw1 ? ob1 : sw0 ? ob0 : (c_f & 0xFFu);
        wr_cell_obj<PAIR_ENVS>(cellw, f, same_cell ? final_same : r.cell_obj);
    }
    // resolve_movement (mdp.py:1644-1727)
    const uint32_t np = (mv & (((c_m >> 8) & 7u) == OC_T_FLOOR)) ? m : s.pos;
    const uint32_t np_o = xchg(np);
    const bool collide = (np == np_o) | ((np == pos_o) & (np_o == s.pos));
    s.ori = mv ? a : s.ori;
    s.pos = collide ? s.pos : np;
    // step_environment_effects (mdp.py:1691-1703) for the pot this lane owns
    s.t += 1u;
    {
        uint32_t pc = s.pc, tk = s.tk;
        const bool autostart = (C.old_dyn != 0u) & (pc == PC_IDLE3);
        pc = autostart ? (uint32_t)PC_COOKING : pc;
        tk = autostart ? 1u : tk;
        const bool cooking = pc == PC_COOKING;
        tk += cooking ? 1u : 0u;
        pc = (cooking & ((tk - 1u) >= cook_of(C, s.ps))) ? (uint32_t)PC_READY : pc;
        s.pc = pc; s.tk = tk;
    }
}

template <bool UNIFORM, bool LAY_LDS>
__global__ __launch_bounds__(BLOCK) void k_rollout_pair(const OcLayout* __restrict__ g_layouts, int n_layouts,
                                                        const uint16_t* __restrict__ layout_id, uint4* st,
                                                        float4* __restrict__ rewards, uint8_t* __restrict__ flags,
                                                        float4* __restrict__ ep_returns, int64_t n, int W, int n_obj,
                                                        int horizon, uint32_t options, uint32_t seed_lo,
                                                        uint32_t seed_hi, int64_t env_offset, int64_t t0, int n_steps) {
    extern __shared__ __attribute__((aligned(16))) uint32_t s_cells[];  // [n_obj * 8][PAIR_ENVS]
    __shared__ uint4 s_lay[LAY_LDS ? LDS_LAYOUT_MAX * 16 : 1];
    __shared__ uint2 s_lut[2 * LUT_ENTRIES];
    for (int i = threadIdx.x; i < 2 * LUT_ENTRIES; i += BLOCK) s_lut[i] = reinterpret_cast<const uint2*>(&g_lut)[i];
    const uint32_t p = threadIdx.x & 1u, el = threadIdx.x >> 1;
    const bool lane1 = p != 0u;
    const int64_t e = (int64_t)blockIdx.x * PAIR_ENVS + el;
    const bool active = e < n;
    const Lay L = stage_layouts<LAY_LDS>(g_layouts, n_layouts, layout_id, e, active, s_lay);
    uint32_t* cellw = s_cells + el;
    const uint32_t delta4 = make_delta4(W);
    PairW s = {};
    LayC C = {};
    float ep_sp = 0.f, ep_sh = 0.f;
    if (active) {
        C = load_consts<UNIFORM>(L);
        const uint4 h = st[e];
        s.pos = lane1 ? (h.x >> 24) : (h.x & 0xFFu);
        s.ori = lane1 ? (h.y & 0xFFu) : ((h.x >> 8) & 0xFFu);
        s.held = lane1 ? ((h.y >> 8) & 0xFFu) : ((h.x >> 16) & 0xFFu);
        s.t = h.y >> 16;
        s.tk = (p < C.n_pots) ? ((h.z >> (8u * p)) & 0xFFu) : 0u;
        // each lane stages half of every object plane (dwords 2p, 2p+1) into the LDS cell words
        int32_t dishes = 0;
        for (int pl = 0; pl < n_obj; ++pl) {
            const uint4 v = st[(int64_t)(1 + pl) * n + e];
            const uint32_t ow[2] = {lane1 ? v.z : v.x, lane1 ? v.w : v.y};
#pragma unroll
            for (int q = 0; q < 2; ++q) {
                const uint32_t qq = 2u * p + (uint32_t)q;
                const uint32_t T = L.u32(L_TERRAIN + 16 * pl + 4 * (int)qq);
                dishes += (int32_t)count_dish_bytes(ow[q]);
                cellw[(8 * pl + 2 * (int)qq) * PAIR_ENVS] = __builtin_amdgcn_perm(T, ow[q], 0x05010400u);
                cellw[(8 * pl + 2 * (int)qq + 1) * PAIR_ENVS] = __builtin_amdgcn_perm(T, ow[q], 0x07030602u);
            }
        }
        s.dcount = dishes + (int32_t)xchg((uint32_t)dishes);
        if (ep_returns) {
            const float4 ep = ep_returns[e];
            ep_sp = lane1 ? ep.y : ep.x;
            ep_sh = lane1 ? ep.w : ep.z;
        }
    }
    __syncthreads();  // the partner lane staged the other half of the cells
    if (!active) return;
    s.ps = (p < C.n_pots) ? (rd_cell16<PAIR_ENVS>(cellw, L.pot_cell((int)p)) & 0xFFu) : 0u;
    s.pc = pot_class(C, s.ps, s.tk);
    const uint8_t* lut = reinterpret_cast<const uint8_t*>(s_lut) + (C.old_dyn ? LUT_ENTRIES * 8 : 0);
    const uint64_t g = (uint64_t)(env_offset + e);
    const uint32_t g_lo = (uint32_t)g, g_hi = (uint32_t)(g >> 32);
    const uint32_t mul_p = lane1 ? 6u : 1u;  // player 1 reads the next base-6 digit
    uint32_t rnd[4] = {0, 0, 0, 0};
    for (int k = 0; k < n_steps; ++k) {
        const uint64_t t = (uint64_t)(t0 + k);
        const uint32_t s8 = (uint32_t)t & 7u;
        if (k == 0 || s8 == 0u) {
            const uint64_t blk = t >> 3;
            philox4x32_10((uint32_t)blk, g_lo, g_hi, (uint32_t)(blk >> 32), seed_lo, seed_hi, rnd);
        }
        uint32_t w = rnd[0];
        w = bitsel(0u - (uint32_t)(s8 >= 2u), rnd[1], w);
        w = bitsel(0u - (uint32_t)(s8 >= 4u), rnd[2], w);
        w = bitsel(0u - (uint32_t)(s8 >= 6u), rnd[3], w);
        const uint32_t x = w * ((s8 & 1u) ? 36u : 1u) * mul_p;
        const uint32_t a = __umulhi(x, 6u);
        float sp, sh;
        pair_step(C, L, lut, cellw, p, s, delta4, a, sp, sh);
        ep_sp += sp; ep_sh += sh;
        uint32_t fl = 0;
        if ((int)s.t >= horizon) {  // is_done (env.py:321-325); both lanes agree
            fl |= OC_F_DONE;
            if (options & OC_OPT_AUTO_RESET) {
                s.pos = L.u8(L_START_POS + (int)p);
                s.ori = L.u8(L_START_OR + (int)p);
                s.held = 0; s.t = 0; s.ps = 0; s.tk = 0; s.pc = PC_EMPTY; s.dcount = 0;
                ep_sp = 0.f; ep_sh = 0.f;
                for (int d = (int)p; d < n_obj * 8; d += 2) cellw[d * PAIR_ENVS] &= 0xFF00FF00u;
                fl |= OC_F_RESET;
            }
        }
        if (rewards) {
            float* base = reinterpret_cast<float*>(rewards + ((int64_t)k * n + e));
            base[p] = sp;       // sparse_reward_by_agent[p]
            base[2 + p] = sh;   // shaped_reward_by_agent[p]
        }
        if (flags) flags[(int64_t)k * n + e] = (uint8_t)fl;  // both lanes store the same byte
    }
    // write back: pot soups into their cells, then header (lane 0) and alternating object planes
    if (p < C.n_pots) wr_cell_obj<PAIR_ENVS>(cellw, L.pot_cell((int)p), s.ps);
    const uint32_t pos_o = xchg(s.pos), ori_o = xchg(s.ori), held_o = xchg(s.held), tk_o = xchg(s.tk);
    const float ep_sp_o = __uint_as_float(xchg(__float_as_uint(ep_sp)));
    const float ep_sh_o = __uint_as_float(xchg(__float_as_uint(ep_sh)));
    if (!lane1) {
        uint4 h;
        h.x = s.pos | (s.ori << 8) | (s.held << 16) | (pos_o << 24);
        h.y = ori_o | (held_o << 8) | (s.t << 16);
        h.z = s.tk | (tk_o << 8);
        h.w = 0;
        st[e] = h;
        if (ep_returns) ep_returns[e] = make_float4(ep_sp, ep_sp_o, ep_sh, ep_sh_o);
    }
    __syncthreads();
    for (int pl = (int)p; pl < n_obj; pl += 2) {
        uint32_t ow[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const uint32_t lo = cellw[(8 * pl + 2 * q) * PAIR_ENVS], hi = cellw[(8 * pl + 2 * q + 1) * PAIR_ENVS];
            ow[q] = __builtin_amdgcn_perm(hi, lo, 0x06040200u);
        }
        st[(int64_t)(1 + pl) * n + e] = make_uint4(ow[0], ow[1], ow[2], ow[3]);
    }
}

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
    const uint64_t g = (uint64_t)(env_offset + e);
    const uint32_t g_lo = (uint32_t)g, g_hi = (uint32_t)(g >> 32);
    const uint32_t k1 = seed_hi ^ RESET_KEY_TWEAK;
    const uint32_t cells = L.u8(L_NCELLS), np = L.n_players();
    uint32_t r[4];
    uint32_t pos0 = L.u8(L_START_POS), pos1 = L.u8(L_START_POS + 1);
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
            if (seen == a) pos0 = c;
            if (seen == b) pos1 = c;
            ++seen;
        }
    }
    uint32_t held[2] = {0u, 0u};
    uint32_t ticks[2] = {0u, 0u};  // header bytes 8..15
    uint8_t pot_obj[OC_MAX_POTS];
    const uint32_t n_pots = L.n_pots();
    for (uint32_t k = 0; k < (uint32_t)OC_MAX_POTS; ++k) pot_obj[k] = 0;
    auto soup_code = [](uint32_t n_on, uint32_t n_to) {  // onions first, then tomatoes (SoupState.get_soup, mdp.py:664-693)
        return OC_O_SOUP | ((n_on + n_to) << 3) | (((1u << n_to) - 1u) << n_on);
    };
    if (thresh != 0ull) {
        for (uint32_t i = 0; i < np; ++i) {
            philox4x32_10(epoch, g_lo, g_hi, 1u + i, seed_lo, k1, r);
            if ((uint64_t)r[0] < thresh) {
                const uint32_t n_on = 1u + __umulhi(r[2], 3u), n_to = __umulhi(r[3], 4u - n_on);
                held[i] = r[1] < 858993459u ? (uint32_t)OC_O_DISH : r[1] < 3435973836u ? (uint32_t)OC_O_ONION : soup_code(n_on, n_to);
            }
        }
        for (uint32_t k = 0; k < n_pots; ++k) {
            philox4x32_10(epoch, g_lo, g_hi, 3u + k, seed_lo, k1, r);
            if ((uint64_t)r[0] < thresh) {
                const uint32_t n_on = 1u + __umulhi(r[1], 3u), n_to = __umulhi(r[2], 4u - n_on);
                pot_obj[k] = (uint8_t)soup_code(n_on, n_to);
                if ((uint64_t)r[3] < thresh) ticks[k >> 2] |= 1u << (8u * (k & 3u));  // cooking_tick 0 -> stored 1
            }
        }
    }
    if (np < 2u) pos1 = 0xFFu;
    st[e] = make_uint4(pos0 | (held[0] << 16) | (pos1 << 24), np == 2u ? (held[1] << 8) : 0u, ticks[0], ticks[1]);
    for (int p = 0; p < n_obj; ++p) {
        uint32_t w[4] = {0u, 0u, 0u, 0u};
        for (uint32_t k = 0; k < n_pots; ++k) {
            const uint32_t c = L.pot_cell((int)k);
            if ((int)(c >> 4) == p) w[(c >> 2) & 3u] |= (uint32_t)pot_obj[k] << (8u * (c & 3u));
        }
        st[(int64_t)(1 + p) * n + e] = make_uint4(w[0], w[1], w[2], w[3]);
    }
    if (ep_returns) ep_returns[e] = make_float4(0.f, 0.f, 0.f, 0.f);
}

// ------------------------------------------------------------------------------------------
// k_encode: lossless_state_encoding (mdp.py:2385-2561) for both players of every env.
//
// Output per env: [2][W][H][26] values, i.e. 2*W*H "items" of 26 consecutive values each, item i of
// view v describing cell (x, y) = (i / H, i % H).  The encoding is >95 % zeros, so a workgroup that owns E
// consecutive envs (E * row ~ 40 KiB of LDS) (1) zero-fills an LDS image of its slice of the output with
// 16-byte stores, (2) scatters the few non-zero values — one task per (env, grid cell) for the terrain /
// object / urgency layers and one per (env, player) for the location / orientation / held-object layers —
// and (3) streams the image to HBM as contiguous 16-byte stores.  The output is the only real traffic:
// 2*W*H*26*sizeof(T) bytes per env against <= 144 bytes of state.
// ------------------------------------------------------------------------------------------
template <typename T>
__device__ __forceinline__ void enc_object_layers(T* item, uint32_t o, bool in_pot, uint32_t tk, uint32_t ct) {
    if (o & OC_O_SOUP) {
        const uint32_t n = (o >> 3) & 3u, nt = __popc(o & 7u), no = n - nt;
        if (in_pot && tk == 0u) { item[16] = (T)no; item[17] = (T)nt; }           // idle: *_in_pot (mdp.py:2490-2497)
        else {
            item[18] = (T)no; item[19] = (T)nt;                                     // mdp.py:2499-2525
            if (in_pot) {
                item[20] = (T)(ct - (tk - 1u));                                     // cook_time - _cooking_tick
                item[21] = (T)((tk - 1u) >= ct ? 1u : 0u);
            } else item[21] = (T)1;
        }
    } else if (o == OC_O_DISH) item[22] = (T)1;
    else if (o == OC_O_ONION) item[23] = (T)1;
    else if (o == OC_O_TOMATO) item[24] = (T)1;
}

template <typename T, bool LAY_LDS>
__global__ __launch_bounds__(BLOCK) void k_encode(const OcLayout* __restrict__ g_layouts, int n_layouts,
                                                  const uint16_t* __restrict__ layout_id,
                                                  const uint4* __restrict__ st, T* __restrict__ obs, int64_t n,
                                                  int W, int H, int n_planes, int envs_per_block, int horizon) {
    extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
    __shared__ uint4 s_lay[LAY_LDS ? LDS_LAYOUT_MAX * 16 : 1];
    const int cells = W * H;
    const int64_t e0 = (int64_t)blockIdx.x * envs_per_block;
    const int ne = (int)min((int64_t)envs_per_block, n - e0);
    // LDS carve: [envs_per_block][n_planes] uint4 state, then the output image
    uint4* s_state = reinterpret_cast<uint4*>(smem);
    const int state_bytes = envs_per_block * n_planes * 16;
    T* s_out = reinterpret_cast<T*>(smem + state_bytes);
    const int items_per_env = 2 * cells;
    const size_t env_bytes = (size_t)items_per_env * OC_NUM_LAYERS * sizeof(T);
    const size_t total = env_bytes * ne;  // multiple of 4; multiple of 16 unless this is a ragged tail block

    if (LAY_LDS) {
        const uint4* src = reinterpret_cast<const uint4*>(g_layouts);
        for (int i = threadIdx.x; i < n_layouts * 16; i += BLOCK) s_lay[i] = src[i];
    }
    // issue the state loads first, zero-fill the image while they are in flight, then park them in LDS
    const int n_ld = ne * n_planes;
    uint4 ld0 = make_uint4(0, 0, 0, 0), ld1 = ld0;
    const int i0 = threadIdx.x, i1 = threadIdx.x + BLOCK;
    if (i0 < n_ld) ld0 = st[(int64_t)(i0 / ne) * n + e0 + (i0 % ne)];  // consecutive lanes: consecutive envs of a plane
    if (i1 < n_ld) ld1 = st[(int64_t)(i1 / ne) * n + e0 + (i1 % ne)];
    {
        uint4* img = reinterpret_cast<uint4*>(s_out);
        const size_t n16 = (total + 15) / 16;
        const uint4 z = make_uint4(0, 0, 0, 0);
        for (size_t i = threadIdx.x; i < n16; i += BLOCK) img[i] = z;
    }
    if (i0 < n_ld) s_state[(i0 % ne) * n_planes + (i0 / ne)] = ld0;
    if (i1 < n_ld) s_state[(i1 % ne) * n_planes + (i1 / ne)] = ld1;
    for (int i = threadIdx.x + 2 * BLOCK; i < n_ld; i += BLOCK)
        s_state[(i % ne) * n_planes + (i / ne)] = st[(int64_t)(i / ne) * n + e0 + (i % ne)];
    __syncthreads();

    const uint32_t inv_w = 65536u / (uint32_t)W + 1u;  // y = c / W for c < 128 (exact: fractional parts are >= 1/W)
    const int tasks_per_env = cells + 2;
    for (int q = threadIdx.x; q < ne * tasks_per_env; q += BLOCK) {
        const int le = q / tasks_per_env;
        const int j = q - le * tasks_per_env;
        const uint8_t* se = reinterpret_cast<const uint8_t*>(s_state + le * n_planes);
        uint32_t lid = 0;
        if (layout_id != nullptr) lid = layout_id[e0 + le];
        const Lay L = LAY_LDS ? Lay{reinterpret_cast<const uint8_t*>(s_lay) + lid * 256u}
                              : Lay{reinterpret_cast<const uint8_t*>(g_layouts) + (size_t)lid * 256u};
        T* env_img = s_out + (size_t)le * items_per_env * OC_NUM_LAYERS;
        if (j < cells) {
            // terrain (mdp.py:2449-2465), urgency (2446-2447) and the object lying on this cell (2482-2534)
            const uint32_t c = (uint32_t)j;
            const uint32_t y = (c * inv_w) >> 16, x = c - y * (uint32_t)W;
            const uint32_t i = x * (uint32_t)H + y;
            const uint32_t tc = L.terrain(c), type = tc & 7u;
            const uint32_t o = se[16 + c];
            const uint32_t t = se[6] | ((uint32_t)se[7] << 8);
            const bool urgent = (horizon - (int)t) < 40;
            if (type != OC_T_FLOOR || urgent || o) {
                // layer of each terrain code: P(4)->10, X(1)->11, O(2)->12, T(3)->13, D(5)->14, S(6)->15
                const uint32_t layer = (0x0F0E0A0D0C0B00ull >> (8u * type)) & 0xFFu;
                uint32_t tk = 0, ct = 0;
                const bool in_pot = type == OC_T_POT;
                if (in_pot && o) { tk = se[8 + (tc >> 3)]; ct = L.cook_time(recipe_idx(o)); }
#pragma unroll
                for (int v = 0; v < 2; ++v) {
                    T* item = env_img + ((size_t)v * cells + i) * OC_NUM_LAYERS;
                    if (type != OC_T_FLOOR) item[layer] = (T)1;
                    if (urgent) item[25] = (T)1;
                    if (o) enc_object_layers<T>(item, o, in_pot, tk, ct);
                }
            }
        } else {
            // player layers (mdp.py:2468-2479, ordering 2423-2434) and the held object (all_objects_list, 876-879)
            const int pl = j - cells;
            const uint32_t pos = se[3 * pl], ori = se[3 * pl + 1], held = se[3 * pl + 2];
            if (pos != 0xFFu) {
                const uint32_t y = (pos * inv_w) >> 16, x = pos - y * (uint32_t)W;
                const uint32_t i = x * (uint32_t)H + y;
#pragma unroll
                for (int v = 0; v < 2; ++v) {
                    T* item = env_img + ((size_t)v * cells + i) * OC_NUM_LAYERS;
                    const int k = (pl == v) ? 0 : 1;  // the view's own player comes first
                    item[k] = (T)1;
                    item[2 + 4 * k + ori] = (T)1;
                    if (held) enc_object_layers<T>(item, held, false, 0u, 0u);
                }
            }
        }
    }
    __syncthreads();

    // stream the image out: contiguous, 16 B per lane per store
    uint8_t* gdst = reinterpret_cast<uint8_t*>(obs) + env_bytes * (size_t)e0;
    const uint8_t* ssrc = reinterpret_cast<const uint8_t*>(s_out);
    const size_t n16 = total / 16;
    for (size_t i = threadIdx.x; i < n16; i += BLOCK)
        reinterpret_cast<uint4*>(gdst)[i] = reinterpret_cast<const uint4*>(ssrc)[i];
    const size_t rem4 = (total - n16 * 16) / 4;
    if (threadIdx.x < rem4)
        reinterpret_cast<uint32_t*>(gdst + n16 * 16)[threadIdx.x] =
            reinterpret_cast<const uint32_t*>(ssrc + n16 * 16)[threadIdx.x];
}

// ------------------------------------------------------------------------------------------
// k_encode_uniform: k_encode specialised for a single layout shared by the whole batch (BASELINE configs[1-2]).
// The generic kernel above is instruction-issue bound, not HBM bound (SQ counters: ~630 instructions per
// wavefront per 4.7 KB of output, most of them the branchy scatter of the *static* terrain layers).  With one
// layout those layers are the same for every env, so persistent workgroups build them ONCE into an LDS template;
// per group of envs they copy template -> image (16-byte LDS moves), scatter only the dynamic values (one task
// per player and one per non-empty object dword; urgency only for envs in their last 40 steps) and stream the
// image out.  The template covers UNIT consecutive envs so that its size is a multiple of 16 bytes.
// ------------------------------------------------------------------------------------------
template <typename T>
__global__ __launch_bounds__(BLOCK) void k_encode_uniform(const OcLayout* __restrict__ g_layouts,
                                                          const uint4* __restrict__ st, T* __restrict__ obs,
                                                          int64_t n, int W, int H, int n_planes, int unit,
                                                          int units_per_group, int horizon) {
    extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
    __shared__ uint4 s_lay[16];
    const int cells = W * H;
    const int items_per_env = 2 * cells;
    const size_t env_bytes = (size_t)items_per_env * OC_NUM_LAYERS * sizeof(T);
    const size_t unit_bytes = env_bytes * unit;                 // multiple of 16 by construction
    const int unit_chunks = (int)(unit_bytes / 16);
    const int epg = unit * units_per_group;                     // envs per group
    // LDS carve: template (one unit), image (one group), state planes of the group
    uint4* s_tmpl = reinterpret_cast<uint4*>(smem);
    uint4* s_img = s_tmpl + unit_chunks;
    uint4* s_state = s_img + (size_t)unit_chunks * units_per_group;
    T* tmpl = reinterpret_cast<T*>(s_tmpl);
    T* img = reinterpret_cast<T*>(s_img);
    const uint32_t inv_w = 65536u / (uint32_t)W + 1u;

    if (threadIdx.x < 16) s_lay[threadIdx.x] = reinterpret_cast<const uint4*>(g_layouts)[threadIdx.x];
    for (int i = threadIdx.x; i < unit_chunks; i += BLOCK) s_tmpl[i] = make_uint4(0, 0, 0, 0);
    __syncthreads();
    const Lay L{reinterpret_cast<const uint8_t*>(s_lay)};
    // static terrain layers (mdp.py:2449-2465) of `unit` envs, both views
    for (int q = threadIdx.x; q < unit * cells; q += BLOCK) {
        const int u = q / cells;
        const uint32_t c = (uint32_t)(q - u * cells);
        const uint32_t type = L.terrain(c) & 7u;
        if (type != OC_T_FLOOR) {
            const uint32_t y = (c * inv_w) >> 16, x = c - y * (uint32_t)W, i = x * (uint32_t)H + y;
            const uint32_t layer = (0x0F0E0A0D0C0B00ull >> (8u * type)) & 0xFFu;  // P->10 X->11 O->12 T->13 D->14 S->15
            T* base = tmpl + (size_t)u * items_per_env * OC_NUM_LAYERS;
            base[((size_t)i) * OC_NUM_LAYERS + layer] = (T)1;
            base[((size_t)cells + i) * OC_NUM_LAYERS + layer] = (T)1;
        }
    }
    __syncthreads();

    const int64_t n_groups = (n + epg - 1) / epg;
    const int obj_dwords = (n_planes - 1) * 4;
    const int tasks_per_env = obj_dwords + 2;
    for (int64_t g = blockIdx.x; g < n_groups; g += gridDim.x) {
        const int64_t e0 = g * epg;
        const int ne = (int)min((int64_t)epg, n - e0);
        // state planes of the group (issued first, parked after the template copy)
        const int n_ld = ne * n_planes;
        uint4 ld0 = make_uint4(0, 0, 0, 0);
        if ((int)threadIdx.x < n_ld) ld0 = st[(int64_t)(threadIdx.x / ne) * n + e0 + (threadIdx.x % ne)];
        for (int i = threadIdx.x; i < unit_chunks; i += BLOCK) {
            const uint4 v = s_tmpl[i];
            for (int u = 0; u < units_per_group; ++u) s_img[(size_t)u * unit_chunks + i] = v;
        }
        if ((int)threadIdx.x < n_ld) s_state[(threadIdx.x % ne) * n_planes + (threadIdx.x / ne)] = ld0;
        for (int i = threadIdx.x + BLOCK; i < n_ld; i += BLOCK)
            s_state[(i % ne) * n_planes + (i / ne)] = st[(int64_t)(i / ne) * n + e0 + (i % ne)];
        __syncthreads();

        // dynamic values: players, objects
        for (int q = threadIdx.x; q < ne * tasks_per_env; q += BLOCK) {
            const int le = q / tasks_per_env;
            const int j = q - le * tasks_per_env;
            const uint8_t* se = reinterpret_cast<const uint8_t*>(s_state + le * n_planes);
            T* env_img = img + (size_t)le * items_per_env * OC_NUM_LAYERS;
            if (j < obj_dwords) {
                const uint32_t w = reinterpret_cast<const uint32_t*>(se + 16)[j];
                if (w != 0u) {
#pragma unroll
                    for (int b = 0; b < 4; ++b) {
                        const uint32_t o = (w >> (8 * b)) & 0xFFu;
                        if (o) {
                            const uint32_t c = 4u * (uint32_t)j + (uint32_t)b;
                            const uint32_t y = (c * inv_w) >> 16, x = c - y * (uint32_t)W, i = x * (uint32_t)H + y;
                            const uint32_t tc = L.terrain(c);
                            const bool in_pot = (tc & 7u) == OC_T_POT;
                            uint32_t tk = 0, ct = 0;
                            if (in_pot) { tk = se[8 + (tc >> 3)]; ct = L.cook_time(recipe_idx(o)); }
                            enc_object_layers<T>(env_img + (size_t)i * OC_NUM_LAYERS, o, in_pot, tk, ct);
                            enc_object_layers<T>(env_img + ((size_t)cells + i) * OC_NUM_LAYERS, o, in_pot, tk, ct);
                        }
                    }
                }
            } else {
                const int pl = j - obj_dwords;
                const uint32_t pos = se[3 * pl], ori = se[3 * pl + 1], held = se[3 * pl + 2];
                if (pos != 0xFFu) {
                    const uint32_t y = (pos * inv_w) >> 16, x = pos - y * (uint32_t)W, i = x * (uint32_t)H + y;
#pragma unroll
                    for (int v = 0; v < 2; ++v) {
                        T* item = env_img + ((size_t)v * cells + i) * OC_NUM_LAYERS;
                        const int k = (pl == v) ? 0 : 1;  // the view's own player comes first (mdp.py:2422-2434)
                        item[k] = (T)1;
                        item[2 + 4 * k + ori] = (T)1;
                        if (held) enc_object_layers<T>(item, held, false, 0u, 0u);
                    }
                }
            }
        }
        // urgency layer (mdp.py:2446-2447) for envs in their last 40 steps
        for (int q = threadIdx.x; q < ne * cells; q += BLOCK) {
            const int le = q / cells;
            const int c = q - le * cells;
            const uint8_t* se = reinterpret_cast<const uint8_t*>(s_state + le * n_planes);
            const uint32_t t = se[6] | ((uint32_t)se[7] << 8);
            if ((horizon - (int)t) < 40) {
                T* env_img = img + (size_t)le * items_per_env * OC_NUM_LAYERS;
                env_img[(size_t)c * OC_NUM_LAYERS + 25] = (T)1;
                env_img[((size_t)cells + c) * OC_NUM_LAYERS + 25] = (T)1;
            }
        }
        __syncthreads();

        // stream the image out: contiguous 16-byte stores (ragged tails in dwords)
        const size_t total = env_bytes * ne;
        uint8_t* gdst = reinterpret_cast<uint8_t*>(obs) + env_bytes * (size_t)e0;
        const size_t n16 = total / 16;
        for (size_t i = threadIdx.x; i < n16; i += BLOCK) reinterpret_cast<uint4*>(gdst)[i] = s_img[i];
        const size_t rem4 = (total - n16 * 16) / 4;
        if (threadIdx.x < rem4)
            reinterpret_cast<uint32_t*>(gdst + n16 * 16)[threadIdx.x] =
                reinterpret_cast<const uint32_t*>(reinterpret_cast<const uint8_t*>(s_img) + n16 * 16)[threadIdx.x];
        __syncthreads();
    }
}

// ------------------------------------------------------------------------------------------
// k_featurize: featurize_state (mdp.py:2579-2898), the hand-crafted feature vector used by the behaviour-cloning
// agents.  Two lanes per env (lane parity = player).  Each lane walks the grid once; for every feature cell it
// reads COST[state][cell] (fewest actions from the player's (cell, orientation) to a goal of that feature,
// precomputed on the host from the reference's MotionPlanner semantics, overcooked_ai_amd/planner.py) and keeps the
// arg-min per category.  min_cost_to_feature (planners.py:391-423) breaks ties by list order — dispensers before
// counter objects, row-major inside each group — which is the lexicographic minimum of (cost, group, cell).
// The 2 x (2*(num_pots*10+26)+4) floats of an env are assembled in an LDS image and streamed out coalesced.
// ------------------------------------------------------------------------------------------
constexpr int FEAT_ENVS = BLOCK / 2;

__device__ __forceinline__ uint32_t feat_key(uint32_t cost, uint32_t group, uint32_t cell) {
    return (cost << 9) | (group << 8) | cell;  // cost < 255, cell < 128
}

template <bool LAY_LDS>
__global__ __launch_bounds__(BLOCK) void k_featurize(const OcLayout* __restrict__ g_layouts, int n_layouts,
                                                     const uint16_t* __restrict__ layout_id,
                                                     const uint8_t* __restrict__ plan_blob,
                                                     const uint32_t* __restrict__ plan_off,
                                                     const uint4* __restrict__ st, float* __restrict__ out, int64_t n,
                                                     int W, int H, int n_planes, int num_pots) {
    extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
    __shared__ uint4 s_lay[LAY_LDS ? LDS_LAYOUT_MAX * 16 : 1];
    const int per = num_pots * 10 + 26, total = 2 * per + 4;  // floats per player block / per (env, player) row
    // every feature is a small integer (deltas, counts, flags, cook time left < 255): the LDS image holds int16 and
    // the copy-out converts.  Row stride total + 2 shorts = an odd number of dwords: lanes writing column k of
    // consecutive rows hit 32 different banks.
    const int rs = total + 2;
    uint4* s_state = reinterpret_cast<uint4*>(smem);           // [FEAT_ENVS][n_planes]
    int16_t* s_img = reinterpret_cast<int16_t*>(smem + (size_t)FEAT_ENVS * n_planes * 16);  // [FEAT_ENVS][2][rs]
    const uint32_t p = threadIdx.x & 1u, el = threadIdx.x >> 1;
    const int64_t e0 = (int64_t)blockIdx.x * FEAT_ENVS;
    const int ne = (int)min((int64_t)FEAT_ENVS, n - e0);
    const int64_t e = e0 + el;
    const bool active = (int)el < ne;
    for (int i = threadIdx.x; i < ne * n_planes; i += BLOCK)
        s_state[(i % ne) * n_planes + (i / ne)] = st[(int64_t)(i / ne) * n + e0 + (i % ne)];
    const Lay L = stage_layouts<LAY_LDS>(g_layouts, n_layouts, layout_id, e, active, s_lay);  // contains the barrier
    if (active) {
        const uint8_t* se = reinterpret_cast<const uint8_t*>(s_state + el * n_planes);
        const uint32_t lid = layout_id ? layout_id[e] : 0u;
        const uint8_t* plan = plan_blob + plan_off[lid];
        const uint32_t pos = se[3 * p], ori = se[3 * p + 1], held = se[3 * p + 2];
        const uint32_t opos = se[3 * (1 - p)];
        const uint32_t cells = (uint32_t)(W * H);
        const uint32_t inv_w = 65536u / (uint32_t)W + 1u;
        const uint32_t py = (pos * inv_w) >> 16, px = pos - py * (uint32_t)W;
        const uint4* cost_row = reinterpret_cast<const uint4*>(plan + 128 + ((uint32_t)plan[pos] * 4u + ori) * (uint32_t)(n_planes - 1) * 16u);
        // arg-min keys: 0 onion, 1 tomato, 2 dish, 3 counter soup, 4 serving, 5 empty counter; two best pots
        uint32_t best[6] = {~0u, ~0u, ~0u, ~0u, ~0u, ~0u};
        uint32_t pot1 = ~0u, pot2 = ~0u, pot3 = ~0u, pot4 = ~0u;
        // 16 cells per iteration: their costs (one 16-byte global load, the next one already in flight), terrain
        // and objects (LDS) arrive as words; the terrain branches are wave-uniform when the batch has one layout
        uint4 cw4 = cost_row[0];
        for (int pl = 0; pl < n_planes - 1; ++pl) {
            const uint4 cur = cw4;
            if (pl + 2 < n_planes) cw4 = cost_row[pl + 1];
            const uint4 tw4 = *reinterpret_cast<const uint4*>(L.base + L_TERRAIN + 16 * pl);
            const uint4 ow4 = *reinterpret_cast<const uint4*>(se + 16 + 16 * pl);
            const uint32_t cw[4] = {cur.x, cur.y, cur.z, cur.w}, tw[4] = {tw4.x, tw4.y, tw4.z, tw4.w},
                           ow[4] = {ow4.x, ow4.y, ow4.z, ow4.w};
#pragma unroll
            for (int j = 0; j < 16; ++j) {
                const uint32_t c = (uint32_t)(16 * pl + j);
                if (c >= cells) break;
                const uint32_t type = (tw[j >> 2] >> (8 * (j & 3))) & 7u;
                if (type == OC_T_FLOOR) continue;
                const uint32_t cost = (cw[j >> 2] >> (8 * (j & 3))) & 0xFFu;
                if (cost == 255u) continue;
                const uint32_t o = (ow[j >> 2] >> (8 * (j & 3))) & 0xFFu;
                if (type == OC_T_ONION_DISP) best[0] = min(best[0], feat_key(cost, 0, c));
                else if (type == OC_T_TOMATO_DISP) best[1] = min(best[1], feat_key(cost, 0, c));
                else if (type == OC_T_DISH_DISP) best[2] = min(best[2], feat_key(cost, 0, c));
                else if (type == OC_T_SERVE) best[4] = min(best[4], feat_key(cost, 0, c));
                else if (type == OC_T_POT) {
                    const uint32_t k = feat_key(cost, 0, c);  // keep the four smallest keys in order
                    if (k < pot1) { pot4 = pot3; pot3 = pot2; pot2 = pot1; pot1 = k; }
                    else if (k < pot2) { pot4 = pot3; pot3 = pot2; pot2 = k; }
                    else if (k < pot3) { pot4 = pot3; pot3 = k; }
                    else if (k < pot4) pot4 = k;
                } else {  // counter
                    if (o == 0u) best[5] = min(best[5], feat_key(cost, 0, c));
                    else if (o == OC_O_ONION) best[0] = min(best[0], feat_key(cost, 1, c));
                    else if (o == OC_O_TOMATO) best[1] = min(best[1], feat_key(cost, 1, c));
                    else if (o == OC_O_DISH) best[2] = min(best[2], feat_key(cost, 1, c));
                    else best[3] = min(best[3], feat_key(cost, 0, c));
                }
            }
        }
        int16_t* own = s_img + ((size_t)el * 2 + p) * rs;              // this player's row: own block first
        int16_t* oth = s_img + ((size_t)el * 2 + (1 - p)) * rs + per;  // the other row carries it second
        int k = 0;
        auto put = [&](int v) { own[k] = (int16_t)v; oth[k] = (int16_t)v; ++k; };
        for (uint32_t d = 0; d < 4; ++d) put(ori == d ? 1 : 0);
        // IDX_TO_OBJ = [onion, soup, dish, tomato] (mdp.py:2733)
        put(held == OC_O_ONION ? 1 : 0); put((held & OC_O_SOUP) ? 1 : 0);
        put(held == OC_O_DISH ? 1 : 0); put(held == OC_O_TOMATO ? 1 : 0);
        const bool held_is[6] = {held == OC_O_ONION, held == OC_O_TOMATO, held == OC_O_DISH, (held & OC_O_SOUP) != 0u,
                                 false, false};
#pragma unroll
        for (int q = 0; q < 6; ++q) {
            int dx = 0, dy = 0;
            uint32_t soup = 0;
            if (held_is[q]) { soup = held; }  // a held object of this kind: deltas (0, 0) (mdp.py:2629-2632)
            else if (best[q] != ~0u) {
                const uint32_t c = best[q] & 0x7Fu;
                const uint32_t cy = (c * inv_w) >> 16, cx = c - cy * (uint32_t)W;
                dx = (int)cx - (int)px; dy = (int)cy - (int)py;
                soup = se[16 + c];
            }
            put(dx); put(dy);
            if (q == 3) {  // ingredient counts of the closest (or held) soup
                const uint32_t nn = (soup & OC_O_SOUP) ? ((soup >> 3) & 3u) : 0u;
                const uint32_t nt = (soup & OC_O_SOUP) ? __popc(soup & 7u) : 0u;
                put((int)(nn - nt)); put((int)nt);
            }
        }
        const uint32_t potk[4] = {pot1, pot2, pot3, pot4};
        for (int j = 0; j < num_pots; ++j) {
            const uint32_t key = j < 4 ? potk[j] : ~0u;
            if (key == ~0u) { for (int z = 0; z < 10; ++z) put(0); continue; }
            const uint32_t c = key & 0x7Fu;
            const uint32_t cy = (c * inv_w) >> 16, cx = c - cy * (uint32_t)W;
            const uint32_t o = se[16 + c], tk = se[8 + (L.terrain(c) >> 3)];
            const uint32_t nn = (o >> 3) & 3u, nt = __popc(o & 7u);
            const uint32_t ct = L.cook_time((nn - nt) + 4u * nt);
            const bool empty = o == 0u, idle = tk == 0u;
            const bool ready = !empty && !idle && (tk - 1u) >= ct, cooking = !empty && !idle && !ready;
            const bool full = cooking || ready || (!empty && nn == 3u);
            const uint32_t remaining = (empty || idle || ready) ? 0u : ct - (tk - 1u);
            put(1); put(empty ? 1 : 0); put(full ? 1 : 0); put(cooking ? 1 : 0); put(ready ? 1 : 0);
            put(empty ? 0 : (int)(nn - nt)); put(empty ? 0 : (int)nt); put((int)remaining);
            put((int)cx - (int)px); put((int)cy - (int)py);
        }
        for (uint32_t d = 0; d < 4; ++d) {  // walls (mdp.py:2831-2838)
            const uint32_t c = pos + (uint32_t)(d == 0 ? -W : d == 1 ? W : d == 2 ? 1 : -1);
            put((L.terrain(c) & 7u) == OC_T_FLOOR ? 0 : 1);
        }
        const uint32_t oy = (opos * inv_w) >> 16, ox = opos - oy * (uint32_t)W;
        int16_t* row = s_img + ((size_t)el * 2 + p) * rs;
        row[2 * per + 0] = (int16_t)((int)ox - (int)px);  // other player's position relative to this one
        row[2 * per + 1] = (int16_t)((int)oy - (int)py);
        row[2 * per + 2] = (int16_t)px;
        row[2 * per + 3] = (int16_t)py;
    }
    __syncthreads();
    // rows are contiguous in the output: stream them out as 16-byte stores (total is a multiple of 4)
    const uint32_t q_per_row = (uint32_t)total / 4u, n_q = (uint32_t)ne * 2u * q_per_row;
    const uint32_t magic = 0xFFFFFFFFu / q_per_row + 1u;  // i / q_per_row == mulhi(i, magic) for i < 2^16
    float4* gdst = reinterpret_cast<float4*>(out + (size_t)e0 * 2 * total);
    for (uint32_t i = threadIdx.x; i < n_q; i += BLOCK) {
        const uint32_t row = __umulhi(i, magic), col = i - row * q_per_row;
        const uint32_t* src = reinterpret_cast<const uint32_t*>(s_img + (size_t)row * rs + 4u * col);
        const uint32_t w0 = src[0], w1 = src[1];
        gdst[i] = make_float4((float)(int16_t)(w0 & 0xFFFFu), (float)((int32_t)w0 >> 16),
                              (float)(int16_t)(w1 & 0xFFFFu), (float)((int32_t)w1 >> 16));
    }
}

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
constexpr int PHI_BYTES = 456 + 8 * 512;
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
    __device__ __forceinline__ double pw(uint32_t k) const { return f64(456 + 8 * (int)k); }  // gamma ** k
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
        for (uint32_t c = 0; c < cells; ++c)
            if ((L.terrain(c) & 7u) == OC_T_SERVE) d = min(d, cost(p, c));
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
// insertion order).  All motion costs are fetched up front so that their latencies overlap.
__global__ __launch_bounds__(BLOCK) void k_potential2(const OcLayout* __restrict__ g_layouts,
                                                      const uint16_t* __restrict__ layout_id,
                                                      const uint8_t* __restrict__ plan_blob,
                                                      const uint32_t* __restrict__ plan_off,
                                                      const uint8_t* __restrict__ phi_tables,
                                                      const uint4* __restrict__ st, double* __restrict__ out, int64_t n,
                                                      int W, int H) {
#pragma clang fp contract(off)
    const int64_t e = (int64_t)blockIdx.x * BLOCK + threadIdx.x;
    if (e >= n) return;
    const uint32_t lid = layout_id ? layout_id[e] : 0u;
    const Lay L{reinterpret_cast<const uint8_t*>(g_layouts + lid)};
    const Phi T{phi_tables + (size_t)lid * PHI_BYTES};
    const uint8_t* plan = plan_blob + plan_off[lid];
    const uint32_t cells = (uint32_t)(W * H), row_stride = (cells + 15u) & ~15u;
    const uint4 hw = st[e];
    const uint32_t np = (hw.x >> 24) == 0xFFu ? 1u : 2u;
    const uint32_t held0 = (hw.x >> 16) & 0xFFu, held1 = np > 1u ? (hw.y >> 8) & 0xFFu : 0xFFu;
    const uint8_t* row0 = plan + 128 + ((uint32_t)plan[hw.x & 0xFFu] * 4u + ((hw.x >> 8) & 0xFFu)) * row_stride;
    const uint8_t* row1 = np > 1u ? plan + 128 + ((uint32_t)plan[hw.x >> 24] * 4u + (hw.y & 0xFFu)) * row_stride : row0;
    const uint32_t n_pots = L.n_pots();
    const uint32_t cellA = L.pot_cell(0), cellB = n_pots > 1u ? L.pot_cell(1) : cellA;
    // everything that comes from memory, issued together
    const uint32_t rA0 = row0[cellA], rA1 = row1[cellA], rB0 = row0[cellB], rB1 = row1[cellB];
    const uint32_t oA = reinterpret_cast<const uint8_t*>(st + (int64_t)(1 + (cellA >> 4)) * n + e)[cellA & 15u];
    const uint32_t oB = reinterpret_cast<const uint8_t*>(st + (int64_t)(1 + (cellB >> 4)) * n + e)[cellB & 15u];
    uint32_t serve0 = 255u, serve1 = 255u;  // min over the serving cells (255 = unreachable stays the maximum)
    for (uint32_t c = 0; c < cells; ++c)
        if ((L.terrain(c) & 7u) == OC_T_SERVE) { serve0 = min(serve0, (uint32_t)row0[c]); serve1 = min(serve1, (uint32_t)row1[c]); }
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
    cls[0] = classify(oA, hw.z & 0xFFu, key[0], rem[0]);
    cls[1] = classify(oB, (hw.z >> 8) & 0xFFu, key[1], rem[1]);
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
    out[e] = phi;
}

// ------------------------------------------------------------------------------------------
// k_shape_rewards: the per-agent training reward of the RLlib environment (OvercookedMultiAgent.step,
// human_aware_rl/rllib/rllib.py:293-342): sparse_reward + reward_shaping_factor * dense_reward[i], with
// sparse_reward = sum of both agents' sparse rewards (env.py:273) and dense = phi(s') - phi(s) for both agents
// (use_phi) or shaped_r_by_agent.  float64 like the reference's Python floats.  It also carries phi forward
// (phi(s) of the next step = phi(s'), or the start state's potential where the episode ended) and emits the done
// byte mask that oc_reset takes.
// ------------------------------------------------------------------------------------------
__global__ __launch_bounds__(BLOCK) void k_shape_rewards(const float4* __restrict__ rewards,
                                                         const uint8_t* __restrict__ flags,
                                                         const uint16_t* __restrict__ layout_id,
                                                         const double* __restrict__ phi_next, double* __restrict__ phi_cur,
                                                         const double* __restrict__ phi_start, double factor,
                                                         double* __restrict__ out, uint8_t* __restrict__ done, int64_t n) {
#pragma clang fp contract(off)
    const int64_t e = (int64_t)blockIdx.x * BLOCK + threadIdx.x;
    if (e >= n) return;
    const float4 r = rewards[e];
    const double sparse = (double)r.x + (double)r.y;
    const bool is_done = (flags[e] & OC_F_DONE) != 0;
    double d0 = (double)r.z, d1 = (double)r.w;
    if (phi_next) {
        const double pn = phi_next[e], pc = phi_cur[e];
        d0 = d1 = pn - pc;
        phi_cur[e] = is_done ? phi_start[layout_id ? layout_id[e] : 0] : pn;
    }
    reinterpret_cast<double2*>(out)[e] = make_double2(sparse + factor * d0, sparse + factor * d1);
    if (done) done[e] = is_done ? 1 : 0;
}

// ------------------------------------------------------------------------------------------
// host side
// ------------------------------------------------------------------------------------------
int fail(int code, const char* msg) {
    snprintf(g_err, sizeof(g_err), "%s", msg);
    return code;
}

int check_launch(const char* what) {
    hipError_t err = hipGetLastError();
    if (err != hipSuccess) {
        snprintf(g_err, sizeof(g_err), "%s: %s", what, hipGetErrorString(err));
        return OC_ELAUNCH;
    }
    return OC_OK;
}

int check_batch(const OcBatch* b, int* n_obj) {
    if (!b) return fail(OC_EINVAL, "batch is NULL");
    if (!b->d_layouts) return fail(OC_EINVAL, "batch.d_layouts is NULL");
    if (b->n_envs < 0) return fail(OC_EINVAL, "batch.n_envs < 0");
    if (b->n_layouts < 1 || b->n_layouts > 65536) return fail(OC_EINVAL, "batch.n_layouts out of range (1..65536)");
    if (b->n_layouts > 1 && !b->d_layout_id) return fail(OC_EINVAL, "d_layout_id required when n_layouts > 1");
    if (b->width < 3 || b->height < 3 || b->width * b->height > OC_MAX_CELLS)
        return fail(OC_EINVAL, "grid shape out of range (3x3 .. 128 cells)");
    *n_obj = (b->width * b->height + 15) / 16;
    return OC_OK;
}

inline int enc_lds_budget() {
    static int v = []() { const char* e = getenv("OC_ENC_LDS"); return e ? atoi(e) : 40 * 1024; }();
    return v;
}

inline unsigned grid_for(int64_t n) { return (unsigned)((n + BLOCK - 1) / BLOCK); }

// SIMDs of the current device (4 per CU); cached per thread
inline int64_t simd_count() {
    thread_local int cached_dev = -1;
    thread_local int64_t cached = 1024;
    int dev = 0;
    if (hipGetDevice(&dev) == hipSuccess && dev != cached_dev) {
        int cus = 0;
        if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess && cus > 0) cached = 4 * (int64_t)cus;
        cached_dev = dev;
    }
    return cached;
}

#define DISPATCH_NOBJ(NOBJ_VALUE, ...)                                  \
    switch (NOBJ_VALUE) {                                               \
        case 1: { constexpr int NOBJ = 1; __VA_ARGS__; } break;         \
        case 2: { constexpr int NOBJ = 2; __VA_ARGS__; } break;         \
        case 3: { constexpr int NOBJ = 3; __VA_ARGS__; } break;         \
        case 4: { constexpr int NOBJ = 4; __VA_ARGS__; } break;         \
        case 5: { constexpr int NOBJ = 5; __VA_ARGS__; } break;         \
        case 6: { constexpr int NOBJ = 6; __VA_ARGS__; } break;         \
        case 7: { constexpr int NOBJ = 7; __VA_ARGS__; } break;         \
        default: { constexpr int NOBJ = 8; __VA_ARGS__; } break;        \
    }

// kernel variant selection: UNIFORM (one layout for the whole batch -> layout constants in SGPRs),
// MAXP (pot slots kept in registers: 1 for single-pot batches of one layout such as cramped_room, 2 covers every
// canonical layout, 8 is the format's maximum),
// LAY_LDS (layout table staged in LDS vs read from HBM/L2 for tables of more than 32 layouts)
template <bool EVENTS>
void launch_step(const OcBatch* b, int n_obj, const void* d_state_in, void* d_state_out, const uint8_t* d_actions,
                 float* d_rewards, uint8_t* d_flags, float* d_ep_returns, uint64_t* d_events, int horizon,
                 uint32_t options, hipStream_t s) {
    const bool uniform = b->n_layouts == 1;
    const bool lds = b->n_layouts <= LDS_LAYOUT_MAX;
    const bool small = b->max_pots >= 1 && b->max_pots <= 2;
    const bool fast = (b->batch_flags & OC_BATCH_TWO_PLAYERS) != 0 && b->width * b->height <= 64;
    const size_t smem = (size_t)n_obj * 8 * BLOCK * sizeof(uint32_t);
    const dim3 grid(grid_for(b->n_envs)), block(BLOCK);
    if (!EVENTS && !(options & OC_OPT_PREDICATE_INTERACT)) {
#define GO3(U, MP, LL, ...)                                                                                          \
    do {                                                                                                             \
        if (smem > 40 * 1024)                                                                                        \
            (void)hipFuncSetAttribute((const void*)k_step3<U, MP, LL, ##__VA_ARGS__>,                                \
                                      hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);                        \
        hipLaunchKernelGGL((k_step3<U, MP, LL, ##__VA_ARGS__>), grid, block, smem, s, b->d_layouts, b->n_layouts, b->d_layout_id,   \
                           (const uint4*)d_state_in, (uint4*)d_state_out, d_actions, (float4*)d_rewards, d_flags,    \
                           (float4*)d_ep_returns, b->n_envs, b->width, n_obj, horizon, options);                     \
    } while (0)
        if (uniform && fast && b->max_pots == 1) GO3(true, 1, true, true);
        else if (uniform && fast && small) GO3(true, 2, true, true);
        else if (uniform) { if (b->max_pots == 1) GO3(true, 1, true); else if (small) GO3(true, 2, true); else GO3(true, 8, true); }
        else if (lds) { if (small) GO3(false, 2, true); else GO3(false, 8, true); }
        else { if (small) GO3(false, 2, false); else GO3(false, 8, false); }
#undef GO3
        return;
    }
#define GO(U, MP, LL)                                                                                                    do {                                                                                                                     if (smem > 48 * 1024)                                                                                                    (void)hipFuncSetAttribute((const void*)k_step<U, MP, LL, EVENTS>,                                                                              hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);                                hipLaunchKernelGGL((k_step<U, MP, LL, EVENTS>), grid, block, smem, s, b->d_layouts, b->n_layouts,                                       b->d_layout_id, (const uint4*)d_state_in, (uint4*)d_state_out, d_actions,                                            (float4*)d_rewards, d_flags, (float4*)d_ep_returns, d_events, b->n_envs, b->width, n_obj,                            horizon, options);                                                                            } while (0)
    if (uniform) { if (small) GO(true, 2, true); else GO(true, 8, true); }
    else if (lds) { if (small) GO(false, 2, true); else GO(false, 8, true); }
    else { if (small) GO(false, 2, false); else GO(false, 8, false); }
#undef GO
}

}  // namespace

extern "C" {

int oc_abi_version(void) { return OC_ABI_VERSION; }
size_t oc_layout_size(void) { return sizeof(OcLayout); }
const char* oc_last_error(void) { return g_err; }
int oc_state_planes(int width, int height) { return 1 + (width * height + 15) / 16; }

int oc_step(const OcBatch* b, const void* d_state_in, void* d_state_out, const uint8_t* d_actions, float* d_rewards,
            uint8_t* d_flags, float* d_ep_returns, uint64_t* d_events, int horizon, uint32_t options, void* stream) {
    int n_obj = 0;
    if (int rc = check_batch(b, &n_obj)) return rc;
    if (!d_state_in || !d_state_out || !d_actions || !d_rewards || !d_flags)
        return fail(OC_EINVAL, "oc_step: NULL state/actions/rewards/flags pointer");
    if (horizon < 1 || horizon > 65535) return fail(OC_EINVAL, "oc_step: horizon must be in 1..65535");
    if (b->n_envs == 0) return OC_OK;
    hipStream_t s = (hipStream_t)stream;
    if (d_events)
        launch_step<true>(b, n_obj, d_state_in, d_state_out, d_actions, d_rewards, d_flags, d_ep_returns, d_events,
                          horizon, options, s);
    else
        launch_step<false>(b, n_obj, d_state_in, d_state_out, d_actions, d_rewards, d_flags, d_ep_returns, nullptr,
                           horizon, options, s);
    return check_launch("oc_step");
}

int oc_step_many(const OcBatch* b, void* d_state, const uint8_t* d_actions, float* d_rewards, uint8_t* d_flags,
                 float* d_ep_returns, int n_steps, int horizon, uint32_t options, void* stream) {
    if (n_steps < 0) return fail(OC_EINVAL, "oc_step_many: n_steps < 0");
    if (!b) return fail(OC_EINVAL, "batch is NULL");
    for (int k = 0; k < n_steps; ++k) {
        const int64_t off = (int64_t)k * b->n_envs;
        if (int rc = oc_step(b, d_state, d_state, d_actions + 2 * off, d_rewards + 4 * off, d_flags + off, d_ep_returns,
                             nullptr, horizon, options, stream))
            return rc;
    }
    return OC_OK;
}

int oc_rollout_random(const OcBatch* b, void* d_state, float* d_rewards, uint8_t* d_flags, float* d_ep_returns,
                      int horizon, uint32_t options, uint64_t seed, int64_t env_offset, int64_t t0, int n_steps,
                      void* stream) {
    int n_obj = 0;
    if (int rc = check_batch(b, &n_obj)) return rc;
    if (!d_state) return fail(OC_EINVAL, "oc_rollout_random: NULL state pointer");
    if (horizon < 1 || horizon > 65535) return fail(OC_EINVAL, "oc_rollout_random: horizon must be in 1..65535");
    if (n_steps < 0) return fail(OC_EINVAL, "oc_rollout_random: n_steps < 0");
    if (b->n_envs == 0 || n_steps == 0) return OC_OK;
    hipStream_t s = (hipStream_t)stream;
    const bool uniform = b->n_layouts == 1;
    const bool lds = b->n_layouts <= LDS_LAYOUT_MAX;
    const bool small = b->max_pots >= 1 && b->max_pots <= 2;
    // Lane pairs pay ~1.5x the total VALU work of one lane per env but halve the per-wavefront instruction stream:
    // they win while one lane per env leaves half of the SIMDs without a wavefront (measured on MI355X, us per
    // batched step, pair vs lane: 0.81 vs 1.02 at 32 768 envs, 1.11 vs 1.03 at 40 960, 1.13 vs 1.01 at 65 536).
    const bool pair_ok = small && (b->batch_flags & OC_BATCH_TWO_PLAYERS) != 0;
    const bool want_pair = (options & OC_OPT_LANE_PAIR) ||
                           (!(options & (OC_OPT_LANE_PER_ENV | OC_OPT_PREDICATE_INTERACT)) && b->n_envs <= 32 * simd_count());
    if (pair_ok && want_pair) {
        // two lanes per env (k_rollout_pair)
        const size_t smem2 = (size_t)n_obj * 8 * PAIR_ENVS * sizeof(uint32_t);
        const dim3 grid2((unsigned)((b->n_envs + PAIR_ENVS - 1) / PAIR_ENVS)), block2(BLOCK);
        if (uniform)
            hipLaunchKernelGGL((k_rollout_pair<true, true>), grid2, block2, smem2, s, b->d_layouts, b->n_layouts,
                               b->d_layout_id, (uint4*)d_state, (float4*)d_rewards, d_flags, (float4*)d_ep_returns,
                               b->n_envs, b->width, n_obj, horizon, options, (uint32_t)seed, (uint32_t)(seed >> 32),
                               env_offset, t0, n_steps);
        else if (lds)
            hipLaunchKernelGGL((k_rollout_pair<false, true>), grid2, block2, smem2, s, b->d_layouts, b->n_layouts,
                               b->d_layout_id, (uint4*)d_state, (float4*)d_rewards, d_flags, (float4*)d_ep_returns,
                               b->n_envs, b->width, n_obj, horizon, options, (uint32_t)seed, (uint32_t)(seed >> 32),
                               env_offset, t0, n_steps);
        else
            hipLaunchKernelGGL((k_rollout_pair<false, false>), grid2, block2, smem2, s, b->d_layouts, b->n_layouts,
                               b->d_layout_id, (uint4*)d_state, (float4*)d_rewards, d_flags, (float4*)d_ep_returns,
                               b->n_envs, b->width, n_obj, horizon, options, (uint32_t)seed, (uint32_t)(seed >> 32),
                               env_offset, t0, n_steps);
        return check_launch("oc_rollout_random");
    }
    if ((options & OC_OPT_PREDICATE_INTERACT) == 0) {
        const bool fast = (b->batch_flags & OC_BATCH_TWO_PLAYERS) != 0 && b->width * b->height <= 64;
        const size_t smem3 = (size_t)n_obj * 16 * BLOCK * sizeof(uint16_t);
        const dim3 grid3(grid_for(b->n_envs)), block3(BLOCK);
#define GO3(U, MP, LL, ...)                                                                                          \
    do {                                                                                                             \
        if (smem3 > 40 * 1024)                                                                                       \
            (void)hipFuncSetAttribute((const void*)k_rollout3<U, MP, LL, ##__VA_ARGS__>,                             \
                                      hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem3);                       \
        hipLaunchKernelGGL((k_rollout3<U, MP, LL, ##__VA_ARGS__>), grid3, block3, smem3, s, b->d_layouts, b->n_layouts,             \
                           b->d_layout_id, (uint4*)d_state, (float4*)d_rewards, d_flags, (float4*)d_ep_returns,     \
                           b->n_envs, b->width, n_obj, horizon, options, (uint32_t)seed, (uint32_t)(seed >> 32),    \
                           env_offset, t0, n_steps);                                                                 \
    } while (0)
        if (uniform && fast && b->max_pots == 1) GO3(true, 1, true, true);
        else if (uniform && fast && small) GO3(true, 2, true, true);
        else if (uniform) { if (b->max_pots == 1) GO3(true, 1, true); else if (small) GO3(true, 2, true); else GO3(true, 8, true); }
        else if (lds) { if (small) GO3(false, 2, true); else GO3(false, 8, true); }
        else { if (small) GO3(false, 2, false); else GO3(false, 8, false); }
#undef GO3
        return check_launch("oc_rollout_random");
    }
    const size_t smem = (size_t)n_obj * 8 * BLOCK * sizeof(uint32_t);
    const dim3 grid(grid_for(b->n_envs)), block(BLOCK);
#define GO(U, MP, LL)                                                                                                    do {                                                                                                                     if (smem > 48 * 1024)                                                                                                    (void)hipFuncSetAttribute((const void*)k_rollout<U, MP, LL>, hipFuncAttributeMaxDynamicSharedMemorySize,                                       (int)smem);                                                                            hipLaunchKernelGGL((k_rollout<U, MP, LL>), grid, block, smem, s, b->d_layouts, b->n_layouts, b->d_layout_id,                            (uint4*)d_state, (float4*)d_rewards, d_flags, (float4*)d_ep_returns, b->n_envs, b->width,                            n_obj, horizon, options, (uint32_t)seed, (uint32_t)(seed >> 32), env_offset, t0, n_steps);     } while (0)
    if (uniform) { if (small) GO(true, 2, true); else GO(true, 8, true); }
    else if (lds) { if (small) GO(false, 2, true); else GO(false, 8, true); }
    else { if (small) GO(false, 2, false); else GO(false, 8, false); }
#undef GO
    return check_launch("oc_rollout_random");
}

int oc_featurize(const OcBatch* b, const uint8_t* d_plan_blob, const uint32_t* d_plan_off, const void* d_state,
                 float* d_features, int num_pots, void* stream) {
    int n_obj = 0;
    if (int rc = check_batch(b, &n_obj)) return rc;
    if (!d_plan_blob || !d_plan_off || !d_state || !d_features) return fail(OC_EINVAL, "oc_featurize: NULL pointer");
    if (num_pots < 0 || num_pots > 4) return fail(OC_EINVAL, "oc_featurize: num_pots must be in 0..4");
    if (!(b->batch_flags & OC_BATCH_TWO_PLAYERS)) return fail(OC_EINVAL, "oc_featurize: needs 2-player layouts");
    if (((uintptr_t)d_features & 15u) != 0) return fail(OC_EINVAL, "oc_featurize: d_features must be 16-byte aligned");
    if (b->n_envs == 0) return OC_OK;
    hipStream_t s = (hipStream_t)stream;
    const int n_planes = 1 + n_obj;
    const int total = 2 * (num_pots * 10 + 26) + 4;
    const size_t smem = (size_t)FEAT_ENVS * n_planes * 16 + (size_t)FEAT_ENVS * 2 * (total + 2) * sizeof(int16_t);
    const dim3 grid((unsigned)((b->n_envs + FEAT_ENVS - 1) / FEAT_ENVS)), block(BLOCK);
    if (b->n_layouts <= LDS_LAYOUT_MAX) {
        (void)hipFuncSetAttribute((const void*)k_featurize<true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
        hipLaunchKernelGGL((k_featurize<true>), grid, block, smem, s, b->d_layouts, b->n_layouts, b->d_layout_id, d_plan_blob,
                           d_plan_off, (const uint4*)d_state, d_features, b->n_envs, b->width, b->height, n_planes, num_pots);
    } else {
        (void)hipFuncSetAttribute((const void*)k_featurize<false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
        hipLaunchKernelGGL((k_featurize<false>), grid, block, smem, s, b->d_layouts, b->n_layouts, b->d_layout_id, d_plan_blob,
                           d_plan_off, (const uint4*)d_state, d_features, b->n_envs, b->width, b->height, n_planes, num_pots);
    }
    return check_launch("oc_featurize");
}

int oc_phi_table_size(void) { return PHI_BYTES; }

int oc_shape_rewards(const OcBatch* b, const float* d_rewards, const uint8_t* d_flags, const double* d_phi_next,
                     double* d_phi_cur, const double* d_phi_start, double reward_shaping_factor, double* d_out,
                     uint8_t* d_done, void* stream) {
    if (!b) return fail(OC_EINVAL, "batch is NULL");
    if (!d_rewards || !d_flags || !d_out) return fail(OC_EINVAL, "oc_shape_rewards: NULL rewards/flags/out pointer");
    if (d_phi_next && (!d_phi_cur || !d_phi_start)) return fail(OC_EINVAL, "oc_shape_rewards: phi_next needs phi_cur and phi_start");
    if (((uintptr_t)d_out & 15u) != 0) return fail(OC_EINVAL, "oc_shape_rewards: d_out must be 16-byte aligned");
    if (b->n_envs == 0) return OC_OK;
    hipLaunchKernelGGL(k_shape_rewards, dim3(grid_for(b->n_envs)), dim3(BLOCK), 0, (hipStream_t)stream,
                       (const float4*)d_rewards, d_flags, b->n_layouts > 1 ? b->d_layout_id : nullptr, d_phi_next, d_phi_cur,
                       d_phi_start, reward_shaping_factor, d_out, d_done, b->n_envs);
    return check_launch("oc_shape_rewards");
}

int oc_potential(const OcBatch* b, const uint8_t* d_plan_blob, const uint32_t* d_plan_off, const uint8_t* d_phi_tables,
                 const void* d_state, double* d_phi, void* stream) {
    int n_obj = 0;
    if (int rc = check_batch(b, &n_obj)) return rc;
    if (!d_plan_blob || !d_plan_off || !d_phi_tables || !d_state || !d_phi) return fail(OC_EINVAL, "oc_potential: NULL pointer");
    if (((uintptr_t)d_phi_tables & 7u) != 0 || ((uintptr_t)d_phi & 7u) != 0)
        return fail(OC_EINVAL, "oc_potential: d_phi_tables / d_phi must be 8-byte aligned");
    if (b->n_envs == 0) return OC_OK;
    if (b->max_pots >= 1 && b->max_pots <= 2)
        hipLaunchKernelGGL(k_potential2, dim3(grid_for(b->n_envs)), dim3(BLOCK), 0, (hipStream_t)stream, b->d_layouts,
                           b->d_layout_id, d_plan_blob, d_plan_off, d_phi_tables, (const uint4*)d_state, d_phi, b->n_envs,
                           b->width, b->height);
    else
        hipLaunchKernelGGL(k_potential, dim3(grid_for(b->n_envs)), dim3(BLOCK), 0, (hipStream_t)stream, b->d_layouts,
                           b->d_layout_id, d_plan_blob, d_plan_off, d_phi_tables, (const uint4*)d_state, d_phi, b->n_envs,
                           b->width, b->height);
    return check_launch("oc_potential");
}

int oc_reset(const OcBatch* b, void* d_state, const uint8_t* d_mask, float* d_ep_returns, void* stream) {
    int n_obj = 0;
    if (int rc = check_batch(b, &n_obj)) return rc;
    if (!d_state) return fail(OC_EINVAL, "oc_reset: NULL state pointer");
    if (b->n_envs == 0) return OC_OK;
    hipStream_t s = (hipStream_t)stream;
    DISPATCH_NOBJ(n_obj, {
        hipLaunchKernelGGL((k_reset<NOBJ>), dim3(grid_for(b->n_envs)), dim3(BLOCK), 0, s, b->d_layouts, b->d_layout_id,
                           (uint4*)d_state, d_mask, (float4*)d_ep_returns, b->n_envs);
    });
    return check_launch("oc_reset");
}

int oc_reset(const OcBatch* b, void* d_state, const uint8_t* d_mask, float* d_ep_returns, void* stream);

int oc_multi_agent_step(const OcBatch* b, void* d_state, const uint8_t* d_actions, float* d_rewards, uint8_t* d_flags,
                        float* d_ep_returns, float* d_ep_returns_out, const uint8_t* d_plan_blob,
                        const uint32_t* d_plan_off, const uint8_t* d_phi_tables, double* d_phi_next, double* d_phi_cur,
                        const double* d_phi_start, double reward_shaping_factor, double* d_shaped, uint8_t* d_done,
                        void* d_obs, int obs_dtype, int horizon, void* stream) {
    if (!d_done) return fail(OC_EINVAL, "oc_multi_agent_step: d_done is required (it is the reset mask)");
    if (int rc = oc_step(b, d_state, d_state, d_actions, d_rewards, d_flags, d_ep_returns, nullptr, horizon, 0u, stream)) return rc;
    if (d_phi_tables) {
        if (int rc = oc_potential(b, d_plan_blob, d_plan_off, d_phi_tables, d_state, d_phi_next, stream)) return rc;
    }
    if (int rc = oc_shape_rewards(b, d_rewards, d_flags, d_phi_tables ? d_phi_next : nullptr, d_phi_cur, d_phi_start,
                                  reward_shaping_factor, d_shaped, d_done, stream))
        return rc;
    if (d_ep_returns && d_ep_returns_out && b->n_envs > 0) {
        if (hipMemcpyAsync(d_ep_returns_out, d_ep_returns, (size_t)b->n_envs * 4 * sizeof(float), hipMemcpyDeviceToDevice,
                           (hipStream_t)stream) != hipSuccess)
            return fail(OC_ELAUNCH, "oc_multi_agent_step: copy of the episode returns failed");
    }
    if (int rc = oc_reset(b, d_state, d_done, d_ep_returns, stream)) return rc;
    if (d_obs) return oc_encode_lossless(b, d_state, d_obs, obs_dtype, horizon, stream);
    return OC_OK;
}

int oc_reset_random(const OcBatch* b, void* d_state, const uint8_t* d_mask, float* d_ep_returns, uint64_t seed,
                    int64_t env_offset, uint32_t epoch, int random_start_pos, double rnd_obj_prob_thresh, void* stream) {
    int n_obj = 0;
    if (int rc = check_batch(b, &n_obj)) return rc;
    if (!d_state) return fail(OC_EINVAL, "oc_reset_random: NULL state pointer");
    if (!(rnd_obj_prob_thresh >= 0.0 && rnd_obj_prob_thresh <= 1.0))
        return fail(OC_EINVAL, "oc_reset_random: rnd_obj_prob_thresh must be in [0, 1]");
    if (b->n_envs == 0) return OC_OK;
    const uint64_t thresh = (uint64_t)(rnd_obj_prob_thresh * 4294967296.0);  // floor(thresh * 2^32); 1.0 -> 2^32: always
    hipLaunchKernelGGL(k_reset_random, dim3(grid_for(b->n_envs)), dim3(BLOCK), 0, (hipStream_t)stream, b->d_layouts,
                       b->d_layout_id, (uint4*)d_state, d_mask, (float4*)d_ep_returns, b->n_envs, n_obj, (uint32_t)seed,
                       (uint32_t)(seed >> 32), env_offset, epoch, random_start_pos, thresh);
    return check_launch("oc_reset_random");
}

int oc_encode_lossless(const OcBatch* b, const void* d_state, void* d_obs, int obs_dtype, int horizon, void* stream) {
    int n_obj = 0;
    if (int rc = check_batch(b, &n_obj)) return rc;
    if (!d_state || !d_obs) return fail(OC_EINVAL, "oc_encode_lossless: NULL pointer");
    if (obs_dtype != OC_OBS_U8 && obs_dtype != OC_OBS_F32) return fail(OC_EINVAL, "oc_encode_lossless: bad obs_dtype");
    if (((uintptr_t)d_obs & 15u) != 0) return fail(OC_EINVAL, "oc_encode_lossless: d_obs must be 16-byte aligned");
    if (b->n_envs == 0) return OC_OK;
    hipStream_t s = (hipStream_t)stream;
    const int n_planes = 1 + n_obj;
    const int cells = b->width * b->height;
    const size_t elem = obs_dtype == OC_OBS_U8 ? 1 : 4;
    const size_t env_bytes = (size_t)2 * cells * OC_NUM_LAYERS * elem;
    // envs per workgroup: fill ~40 KiB of LDS; a multiple of 4 keeps every block's byte range 16-byte aligned
    int epb = (int)((size_t)enc_lds_budget() / (env_bytes + (size_t)n_planes * 16));
    if (epb >= 4) epb &= ~3;
    if (epb < 1) epb = 1;
    if (epb > 32) epb = 32;
    if (obs_dtype == OC_OBS_U8 && (epb & 3) != 0 && (env_bytes & 15u) != 0) {
        epb = 4;  // u8 rows of odd cell counts are only 4-byte multiples: keep blocks 16-byte aligned
    }
    const size_t smem = (size_t)epb * n_planes * 16 + (((size_t)epb * env_bytes + 15) & ~(size_t)15);
    if (smem > 160 * 1024) return fail(OC_EINVAL, "oc_encode_lossless: grid too large for LDS staging");
    // single layout + u8: persistent template kernel (measured 30.2 vs 32.4 us on 65 536 asymmetric_advantages envs;
    // f32 is HBM-write bound either way and the generic kernel is marginally faster there: 112 vs 116 us)
    if (b->n_layouts == 1 && obs_dtype == OC_OBS_U8 && !getenv("OC_ENC_GENERIC")) {
        int unit = 1;
        while (((env_bytes * unit) & 15u) != 0) unit *= 2;              // 1, 2 or 4 envs per template
        const size_t unit_bytes = env_bytes * unit;
        int upg = (int)((size_t)enc_lds_budget() / unit_bytes);          // units per group
        if (upg < 1) upg = 1;
        if (upg * unit > 32) upg = 32 / unit > 0 ? 32 / unit : 1;
        const size_t smem_u = unit_bytes + unit_bytes * upg + (size_t)unit * upg * n_planes * 16;
        if (smem_u <= 150 * 1024) {
            const int64_t n_groups = (b->n_envs + (int64_t)unit * upg - 1) / ((int64_t)unit * upg);
            int per_cu = (int)((150 * 1024) / (smem_u + 512));
            if (per_cu > 8) per_cu = 8;
            if (per_cu < 1) per_cu = 1;
            int64_t grid_u = (simd_count() / 4) * per_cu;
            if (grid_u > n_groups) grid_u = n_groups;
            if (obs_dtype == OC_OBS_U8) {
                if (smem_u > 48 * 1024)
                    (void)hipFuncSetAttribute((const void*)k_encode_uniform<uint8_t>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem_u);
                hipLaunchKernelGGL((k_encode_uniform<uint8_t>), dim3((unsigned)grid_u), dim3(BLOCK), smem_u, s, b->d_layouts,
                                   (const uint4*)d_state, (uint8_t*)d_obs, b->n_envs, b->width, b->height, n_planes, unit, upg, horizon);
            } else {
                if (smem_u > 48 * 1024)
                    (void)hipFuncSetAttribute((const void*)k_encode_uniform<float>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem_u);
                hipLaunchKernelGGL((k_encode_uniform<float>), dim3((unsigned)grid_u), dim3(BLOCK), smem_u, s, b->d_layouts,
                                   (const uint4*)d_state, (float*)d_obs, b->n_envs, b->width, b->height, n_planes, unit, upg, horizon);
            }
            return check_launch("oc_encode_lossless");
        }
    }
    const unsigned grid = (unsigned)((b->n_envs + epb - 1) / epb);
    const bool lds = b->n_layouts <= LDS_LAYOUT_MAX;
#define LAUNCH_ENC(T, LDSFLAG)                                                                                        \
    do {                                                                                                              \
        if (smem > 48 * 1024)                                                                                         \
            (void)hipFuncSetAttribute((const void*)k_encode<T, LDSFLAG>, hipFuncAttributeMaxDynamicSharedMemorySize, \
                                      (int)smem);                                                                     \
        hipLaunchKernelGGL((k_encode<T, LDSFLAG>), dim3(grid), dim3(BLOCK), smem, s, b->d_layouts, b->n_layouts,      \
                           b->d_layout_id, (const uint4*)d_state, (T*)d_obs, b->n_envs, b->width, b->height,          \
                           n_planes, epb, horizon);                                                                   \
    } while (0)
    if (obs_dtype == OC_OBS_U8) {
        if (lds) LAUNCH_ENC(uint8_t, true); else LAUNCH_ENC(uint8_t, false);
    } else {
        if (lds) LAUNCH_ENC(float, true); else LAUNCH_ENC(float, false);
    }
#undef LAUNCH_ENC
    return check_launch("oc_encode_lossless");
}

}  // extern "C"
