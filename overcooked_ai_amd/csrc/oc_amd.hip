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
#include <string.h>

#include "../../include/oc_amd.h"

namespace {

constexpr int BLOCK = 256;
constexpr int LDS_LAYOUT_MAX = 32;  // layout tables up to this many entries are staged in LDS (8 KiB)

// byte offsets inside OcLayout (include/oc_amd.h)
constexpr int L_NPOTS = 3, L_NPLAYERS = 4, L_OLDDYN = 5, L_START_POS = 8, L_START_OR = 10, L_POT_CELL = 16,
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

__device__ __forceinline__ uint32_t rd_cell16(const uint32_t* cellw, uint32_t c) {
    return reinterpret_cast<const uint16_t*>(cellw + (c >> 1) * BLOCK)[c & 1];
}
__device__ __forceinline__ void wr_cell_obj(uint32_t* cellw, uint32_t c, uint32_t v) {
    reinterpret_cast<uint8_t*>(cellw + (c >> 1) * BLOCK)[(c & 1) * 2] = (uint8_t)v;
}

// recipe index n_onion + 4*n_tomato of a soup code
__device__ __forceinline__ uint32_t recipe_idx(uint32_t soup) {
    const uint32_t n = (soup >> 3) & 3u, nt = __popc(soup & 7u);
    return (n - nt) + 4u * nt;
}

// Recipe.time of a soup code through the 16-byte LUT held in 4 registers
__device__ __forceinline__ uint32_t cook_of(const LayC& C, uint32_t soup) {
    const uint32_t n = (soup >> 3) & 3u, nt = __popc(soup & 7u), no = n - nt;
    const uint32_t w = nt == 0u ? C.cook[0] : nt == 1u ? C.cook[1] : nt == 2u ? C.cook[2] : C.cook[3];
    return (w >> (8u * no)) & 0xFFu;
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
// INTERACT of player P (resolve_interacts, mdp.py:1432-1579), written without data-dependent
// branches: every outcome is a predicate, the new hand / cell / tick are selects.  `cell16` is the
// LDS word of the faced cell, `fwd_*` forwards player 0's counter write when both face one cell.
// ------------------------------------------------------------------------------------------
template <int MAXP, bool EVENTS, int P>
__device__ __forceinline__ void interact(const LayC& C, const Lay L, EnvW<MAXP>& s, bool act, uint32_t f,
                                         uint32_t cell16, uint32_t useful_pots, uint32_t n_full, bool two,
                                         bool& wr, uint32_t& wr_val, float& sparse, float& shaped, uint64_t& ev) {
    uint32_t h = P ? s.held1 : s.held0;
    const uint32_t other_h = P ? s.held0 : s.held1;
    const uint32_t tc = cell16 >> 8;
    const uint32_t type = act ? (tc & 7u) : 7u;  // 7 matches no terrain: a lane that does not interact falls through
    const uint32_t slot = tc >> 3;
    const bool isP = type == OC_T_POT;
    uint32_t tkv = 0, pso = 0;
#pragma unroll
    for (int k = 0; k < MAXP; ++k) {
        const bool sel = slot == (uint32_t)k;
        tkv = sel ? s.tk[k] : tkv;
        pso = sel ? s.ps[k] : pso;
    }
    const uint32_t o = isP ? pso : (cell16 & 0xFFu);
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
    const bool dish_useful = two & (((other_h == OC_O_DISH) ? 1u : 0u) < useful_pots) & (s.dcount == 0);
    // pot (mdp.py:1515-1568)
    const bool idle = tk == 0u;
    const bool start = isP & hz & (C.old_dyn == 0u) & (!oz) & idle & (n > 0u);       // begin_cooking -> tick 0
    const uint32_t ct = cook_of(C, o);
    const bool ready = (!idle) & ((tk - 1u) >= ct);
    const bool plate = isP & (h == OC_O_DISH) & (!oz) & ready;                       // soup pickup
    const bool is_ing = (h == OC_O_ONION) | (h == OC_O_TOMATO);
    const bool place = isP & is_ing & idle & (n < 3u);                               // not is_full (mdp.py:547-551)
    const uint32_t soup_new = OC_O_SOUP | ((n + 1u) << 3) | (o & 7u) | ((h == OC_O_TOMATO ? 1u : 0u) << n);
    // serving (mdp.py:1570-1577)
    const bool serve = (type == OC_T_SERVE) & ((h & OC_O_SOUP) != 0u);

    const uint32_t new_h = swapX ? o : take ? disp_obj : plate ? o : (place | serve) ? 0u : h;
    const uint32_t new_o = swapX ? h : plate ? 0u : place ? soup_new : o;
    const uint32_t new_tk = start ? 1u : plate ? 0u : tk;
    const bool pot_upd = start | plate | place;
#pragma unroll
    for (int k = 0; k < MAXP; ++k) {
        const bool upd = pot_upd & (slot == (uint32_t)k);
        s.ps[k] = upd ? new_o : s.ps[k];
        s.tk[k] = upd ? new_tk : s.tk[k];
    }
    s.dcount += swapX ? ((h == OC_O_DISH ? 1 : 0) - (o == OC_O_DISH ? 1 : 0)) : 0;
    if (P) s.held1 = new_h; else s.held0 = new_h;
    wr = swapX;
    wr_val = new_o;
    shaped = (place ? C.rew_place : 0.f) + (plate ? C.rew_soup : 0.f) + ((take & isD & dish_useful) ? C.rew_dish : 0.f);
    sparse = 0.f;
    if (serve) sparse = L.value(recipe_idx(h));  // deliver_soup / get_recipe_value (mdp.py:1631-1642, 1595-1602); rare

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
        e |= evbit(place, tom ? EV_POTTING_TOMATO : EV_POTTING_ONION, P);
        e |= evbit(place & ((nib & 1u) != 0u), EV_OPTIMAL_ONION_POTTING + (int)0, P) << (2 * tom);
        e |= evbit(place & ((nib & 2u) != 0u), EV_VIABLE_ONION_POTTING, P) << (2 * tom);
        e |= evbit(place & ((nib & 4u) != 0u), EV_CATASTROPHIC_ONION_POTTING, P) << (2 * tom);
        e |= evbit(place & ((nib & 8u) != 0u), EV_USELESS_ONION_POTTING, P) << (2 * tom);
        ev |= e;
    }
}

// ------------------------------------------------------------------------------------------
// One joint transition: get_state_transition (mdp.py:1375-1430).
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

    // resolve_interacts: player 0 fully, then player 1 (mdp.py:1446)
    bool wr0, wr1;
    uint32_t v0, v1;
    float sp0, sh0, sp1, sh1;
    interact<MAXP, EVENTS, 0>(C, L, s, a0 == OC_A_INTERACT, f0, c_f0, useful_pots, n_full, two, wr0, v0, sp0, sh0, ev);
    // player 1 sees player 0's counter write when both face the same cell
    const uint32_t c_f1_live = (wr0 & (f1 == f0)) ? ((c_f1 & 0xFF00u) | v0) : c_f1;
    interact<MAXP, EVENTS, 1>(C, L, s, two & (a1 == OC_A_INTERACT), f1, c_f1_live, useful_pots, n_full, two, wr1, v1, sp1,
                              sh1, ev);
    if (wr0) wr_cell_obj(cellw, f0, v0);
    if (wr1) wr_cell_obj(cellw, f1, v1);
    r = make_float4(sp0, sp1, sh0, sh1);

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
        const uint32_t hi0 = __umulhi(0xD2511F53u, c0), lo0 = 0xD2511F53u * c0;
        const uint32_t hi1 = __umulhi(0xCD9E8D57u, c2), lo1 = 0xCD9E8D57u * c2;
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
__device__ __forceinline__ void draw_actions(const uint32_t (&rnd)[4], uint32_t s8, uint32_t& a0, uint32_t& a1) {
    const uint32_t w = s8 < 2u ? rnd[0] : s8 < 4u ? rnd[1] : s8 < 6u ? rnd[2] : rnd[3];  // s8 is wave-uniform
    const uint32_t x = (s8 & 1u) ? w * 36u : w;
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
// k_encode: lossless_state_encoding (mdp.py:2385-2561) for both players of every env.
//
// Output per env: [2][W][H][26] values, i.e. 2*W*H "items" of 26 consecutive values each, item j of
// view v describing cell (x, y) = (j / H, j % H).  A workgroup owns ENVS consecutive envs: it stages
// their packed state in LDS, every lane computes whole items into an LDS image of the output, and
// the image is streamed to HBM as contiguous 16-byte stores (the output is the dominant traffic:
// 2*W*H*26*sizeof(T) bytes per env against <= 144 bytes of state).
// ------------------------------------------------------------------------------------------
struct EncEnv {
    uint32_t pos[2], ori[2], held[2];
    uint32_t urgent;
};

__device__ __forceinline__ void encode_item(const Lay L, const uint8_t* st /* LDS: this env's planes, 16 B each */,
                                            int W, int H, uint32_t view, uint32_t j, uint32_t urgent,
                                            uint32_t (&val)[OC_NUM_LAYERS]) {
    const uint32_t x = j / (uint32_t)H, y = j - x * (uint32_t)H;
    const uint32_t c = y * (uint32_t)W + x;
    const uint32_t pos_a = st[view ? 3 : 0], or_a = st[view ? 4 : 1];   // primary agent (mdp.py:2422-2434)
    const uint32_t pos_b = st[view ? 0 : 3], or_b = st[view ? 1 : 4];   // other agent
#pragma unroll
    for (int l = 0; l < OC_NUM_LAYERS; ++l) val[l] = 0;
    const bool here_a = pos_a == c, here_b = pos_b == c;
    val[0] = here_a; val[1] = here_b;
#pragma unroll
    for (int d = 0; d < 4; ++d) {
        val[2 + d] = here_a && or_a == (uint32_t)d;
        val[6 + d] = here_b && or_b == (uint32_t)d;
    }
    const uint32_t tc = L.terrain(c);
    const uint32_t type = tc & 7u;
    val[10] = type == OC_T_POT; val[11] = type == OC_T_COUNTER; val[12] = type == OC_T_ONION_DISP;
    val[13] = type == OC_T_TOMATO_DISP; val[14] = type == OC_T_DISH_DISP; val[15] = type == OC_T_SERVE;
    // object lying on this cell, or held by a player standing on it (all_objects_list, mdp.py:876-879)
    uint32_t o = st[16 + c];
    if (st[0] == c) o = st[2];
    if (st[3] == c) o = st[5];
    if (o & OC_O_SOUP) {
        const uint32_t n = (o >> 3) & 3u, nt = __popc(o & 7u), no = n - nt;
        if (type == OC_T_POT) {
            const uint32_t tk = st[8 + (tc >> 3)];
            if (tk == 0u) { val[16] = no; val[17] = nt; }                 // idle: *_in_pot (mdp.py:2490-2497)
            else {
                const uint32_t ct = L.cook_time(no + 4u * nt);
                val[18] = no; val[19] = nt;
                val[20] = ct - (tk - 1u);                                   // cook_time - _cooking_tick (mdp.py:2505-2509)
                val[21] = (tk - 1u) >= ct;
            }
        } else { val[18] = no; val[19] = nt; val[21] = 1; }                 // mdp.py:2515-2525
    } else {
        val[22] = o == OC_O_DISH; val[23] = o == OC_O_ONION; val[24] = o == OC_O_TOMATO;
    }
    val[25] = urgent;                                                       // mdp.py:2446-2447
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

    if (LAY_LDS) {
        const uint4* src = reinterpret_cast<const uint4*>(g_layouts);
        for (int i = threadIdx.x; i < n_layouts * 16; i += BLOCK) s_lay[i] = src[i];
    }
    for (int i = threadIdx.x; i < ne * n_planes; i += BLOCK) {
        const int le = i % ne, p = i / ne;  // consecutive lanes read consecutive envs of one plane
        s_state[le * n_planes + p] = st[(int64_t)p * n + e0 + le];
    }
    __syncthreads();

    const int items_per_env = 2 * cells;
    const int pairs = ne * cells;  // two consecutive items per lane -> 52 B (u8) / 208 B (f32) contiguous
    for (int q = threadIdx.x; q < pairs; q += BLOCK) {
        const int le = q / cells;
        const int pj = q - le * cells;
        const uint8_t* se = reinterpret_cast<const uint8_t*>(s_state + le * n_planes);
        uint32_t lid = 0;
        if (layout_id != nullptr) lid = layout_id[e0 + le];
        const Lay L = LAY_LDS ? Lay{reinterpret_cast<const uint8_t*>(s_lay) + lid * 256u}
                              : Lay{reinterpret_cast<const uint8_t*>(g_layouts) + (size_t)lid * 256u};
        const uint32_t t = se[6] | ((uint32_t)se[7] << 8);
        const uint32_t urgent = (horizon - (int)t) < 40 ? 1u : 0u;
        uint32_t va[OC_NUM_LAYERS], vb[OC_NUM_LAYERS];
        const uint32_t i0 = 2u * pj, i1 = i0 + 1u;
        encode_item(L, se, W, H, i0 >= (uint32_t)cells, i0 >= (uint32_t)cells ? i0 - cells : i0, urgent, va);
        encode_item(L, se, W, H, i1 >= (uint32_t)cells, i1 >= (uint32_t)cells ? i1 - cells : i1, urgent, vb);
        T* dst = s_out + ((size_t)le * items_per_env + i0) * OC_NUM_LAYERS;
        if (sizeof(T) == 1) {
            uint32_t* d32 = reinterpret_cast<uint32_t*>(dst);  // 52-byte pair -> 13 aligned dwords
            uint32_t b[52];
#pragma unroll
            for (int l = 0; l < 26; ++l) { b[l] = va[l]; b[26 + l] = vb[l]; }
#pragma unroll
            for (int w = 0; w < 13; ++w)
                d32[w] = b[4 * w] | (b[4 * w + 1] << 8) | (b[4 * w + 2] << 16) | (b[4 * w + 3] << 24);
        } else {
            float2* d64 = reinterpret_cast<float2*>(dst);      // 208-byte pair, 16-byte aligned
#pragma unroll
            for (int w = 0; w < 13; ++w) d64[w] = make_float2((float)va[2 * w], (float)va[2 * w + 1]);
#pragma unroll
            for (int w = 0; w < 13; ++w) d64[13 + w] = make_float2((float)vb[2 * w], (float)vb[2 * w + 1]);
        }
    }
    __syncthreads();

    // stream the image out: contiguous, 16 B per lane per store
    const size_t env_bytes = (size_t)items_per_env * OC_NUM_LAYERS * sizeof(T);
    const size_t total = env_bytes * ne;  // multiple of 4; multiple of 16 unless this is a ragged tail block
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

inline unsigned grid_for(int64_t n) { return (unsigned)((n + BLOCK - 1) / BLOCK); }

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
// MAXP (pot slots kept in registers: 2 covers every canonical layout, 8 is the format's maximum),
// LAY_LDS (layout table staged in LDS vs read from HBM/L2 for tables of more than 32 layouts)
template <bool EVENTS>
void launch_step(const OcBatch* b, int n_obj, const void* d_state_in, void* d_state_out, const uint8_t* d_actions,
                 float* d_rewards, uint8_t* d_flags, float* d_ep_returns, uint64_t* d_events, int horizon,
                 uint32_t options, hipStream_t s) {
    const bool uniform = b->n_layouts == 1;
    const bool lds = b->n_layouts <= LDS_LAYOUT_MAX;
    const bool small = b->max_pots >= 1 && b->max_pots <= 2;
    const size_t smem = (size_t)n_obj * 8 * BLOCK * sizeof(uint32_t);
    const dim3 grid(grid_for(b->n_envs)), block(BLOCK);
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
    const size_t smem = (size_t)n_obj * 8 * BLOCK * sizeof(uint32_t);
    const dim3 grid(grid_for(b->n_envs)), block(BLOCK);
#define GO(U, MP, LL)                                                                                                    do {                                                                                                                     if (smem > 48 * 1024)                                                                                                    (void)hipFuncSetAttribute((const void*)k_rollout<U, MP, LL>, hipFuncAttributeMaxDynamicSharedMemorySize,                                       (int)smem);                                                                            hipLaunchKernelGGL((k_rollout<U, MP, LL>), grid, block, smem, s, b->d_layouts, b->n_layouts, b->d_layout_id,                            (uint4*)d_state, (float4*)d_rewards, d_flags, (float4*)d_ep_returns, b->n_envs, b->width,                            n_obj, horizon, options, (uint32_t)seed, (uint32_t)(seed >> 32), env_offset, t0, n_steps);     } while (0)
    if (uniform) { if (small) GO(true, 2, true); else GO(true, 8, true); }
    else if (lds) { if (small) GO(false, 2, true); else GO(false, 8, true); }
    else { if (small) GO(false, 2, false); else GO(false, 8, false); }
#undef GO
    return check_launch("oc_rollout_random");
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
    int epb = (int)((40 * 1024) / (env_bytes + (size_t)n_planes * 16));
    if (epb >= 4) epb &= ~3;
    if (epb < 1) epb = 1;
    if (epb > 32) epb = 32;
    if (obs_dtype == OC_OBS_U8 && (epb & 3) != 0 && (env_bytes & 15u) != 0) {
        epb = 4;  // u8 rows of odd cell counts are only 4-byte multiples: keep blocks 16-byte aligned
    }
    const size_t smem = (size_t)epb * n_planes * 16 + (size_t)epb * env_bytes;
    if (smem > 160 * 1024) return fail(OC_EINVAL, "oc_encode_lossless: grid too large for LDS staging");
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
