// step_lut4.hpp — get_state_transition, fourth formulation: k_rollout4 (the default rollout path)
// Part of liboc_amd.so: included by oc_amd.hip inside its anonymous namespace after step_table.hpp.
#pragma once

// ==========================================================================================
// v4: everything a step looks at is one look-up away.
//
// At 65 536 envs every SIMD holds ONE wavefront, so a batched step costs (instructions per step) x (issue time):
// the round-1 kernel (k_rollout3, retired in round 4) spent ~235 instructions per env-step (199 VALU).  v4 restructures the data so that the common step is
// ~90 instructions:
//
//  * cell word (LDS, u16 [cell][lane]) = object code (low byte, wire format) | KEY BYTE (high byte) with
//        key byte = 30 * terrain type + class of what is there
//                   counter: {empty, dish, other}; pot: {empty, idle 1, idle 2, idle 3, cooking, ready}
//    The key byte is kept current by whoever changes the cell (the interact writes a whole new word; the env
//    effects rewrite a pot's key byte when it turns ready).  Pots are ordinary cells: no "is it a pot" selects.
//  * interact = LUT[key byte + 6 * hand class (+ 240 when the player does not interact)], 16 bytes:
//        .x  byte selectors          res = v_perm(.y, pool, .x)  ->  [flags][new hand][new object][new key byte]
//                                    (the new cell word is the upper half: stored with ds_write_b16_d16_hi, no shift)
//        .y  constants               [new key byte][dispensed object][flags][change of the loose-dish count, signed]
//        .z  "add ingredient" term   pool = [hand][object][object][key byte] + .z  (byte 2 becomes the soup + ingredient:
//                                    the entry knows how many items the pot holds and what the hand is)
//        .w  shaped reward of the outcome: the float itself when the batch has one layout (patched into the LDS copy
//                                    of the table at kernel start), else its class {0 none, 1 potting, 2 soup pickup}
//    9 VALU per player: hand class, key, address, pool (perm + add), result perm, new hand, new cell word.
//  * pots: their class lives only in the cell word.  A countdown register per pot (steps until ready; "a long time" when
//    not cooking) is decremented every step and the lanes where it reaches zero write "ready" into the pot's key
//    byte; cook times are looked up only when cooking starts.  2 VALU per pot and step.
//  * rewards: the shaped reward of potting / soup pickup comes out of the LUT entry; everything else — cooking starts
//    (countdown), deliveries (recipe value), dish pick-ups that may be useful (is_dish_pickup_useful), the horizon —
//    happens in ~0.3 % of the lane-steps and sits behind one divergent branch.
//  * layouts with at most 7 free cells (cramped_room): the JOINT move table.  Joint pose J = both players' (cell,
//    orientation); row J holds, per joint action, the next joint pose with collisions and blocked moves resolved
//    (resolve_movement, mdp.py:1644-1727, is ONE LDS read) and the LDS offsets of the two faced cells.  The table is
//    built by the workgroup at kernel start (~1 % of a 400-step launch).  Other layouts move arithmetically.
// Semantics are those of step_table.hpp / the oracle: same conflict replay for player 1, stale pot states for the
// usefulness predicates, same restart at the horizon.
// ==========================================================================================
constexpr uint32_t KB_COUNTER = OC_T_COUNTER * 30u, KB_POT = OC_T_POT * 30u;
enum { F4_TAKE_DISH = 1, F4_PLACE = 2, F4_PLATE = 4, F4_START = 8, F4_SERVE = 16, F4_CHG = 128 };
constexpr uint32_t F4_POTBITS = F4_PLACE | F4_PLATE | F4_START;
constexpr int LUT4_KEYS = 240, LUT4_BYTES = 2 * LUT4_KEYS * 16;  // keys 240..479: the "does not interact" copy (all no-ops)
// Countdown register of a pot.  Cooking: steps until it is ready.  Never cooking during this launch: REM_IDLE minus the
// steps run (stays a large positive number: a launch runs at most 2^30 steps).  Cooked or started during the launch: at
// most the longest cook time, or negative once it is ready — rem_live() tells these two cases apart when the state is stored.
constexpr uint32_t REM_IDLE = 0x7FFFFF00u;
__device__ __forceinline__ bool rem_live(uint32_t rem) { return (int32_t)rem < 0x100; }

struct Lut4Entry { uint32_t sel, cst, add, rew; };
enum { RW4_NONE = 0, RW4_PLACE = 1, RW4_PLATE = 2 };

constexpr Lut4Entry lut4_entry(int old_dyn, int type, int hc, int oc) {
    // selectors: 0 hand, 1 object, 2 object + .z, 3 key byte | 4 new key byte, 5 dispensed object, 6 flags | 0x0C zero
    uint32_t sel_h = 0, sel_o = 1, sel_kb = 3, nkb = 0, cobj = 0, flags = 0, add = 0, rew = RW4_NONE;
    int dd = 0;
    if (type == OC_T_COUNTER) {
        if (hc == 0 && (oc == 1 || oc == 2)) {          // pick up from a counter (mdp.py:1473-1485)
            sel_h = 1; sel_o = 0x0C; sel_kb = 4; nkb = KB_COUNTER; flags = F4_CHG; dd = oc == 1 ? -1 : 0;
        } else if (hc != 0 && oc == 0) {                // drop on a counter (mdp.py:1459-1471)
            sel_h = 0x0C; sel_o = 0; sel_kb = 4; nkb = KB_COUNTER + (hc == 3 ? 1 : 2); flags = F4_CHG; dd = hc == 3 ? 1 : 0;
        }
    } else if (type == OC_T_ONION_DISP) {
        if (hc == 0) { sel_h = 5; cobj = OC_O_ONION; }
    } else if (type == OC_T_TOMATO_DISP) {
        if (hc == 0) { sel_h = 5; cobj = OC_O_TOMATO; }
    } else if (type == OC_T_DISH_DISP) {
        if (hc == 0) { sel_h = 5; cobj = OC_O_DISH; flags = F4_TAKE_DISH; }
    } else if (type == OC_T_POT) {
        if (hc == 0 && oc >= PC_IDLE1 && oc <= PC_IDLE3 && !old_dyn) {   // begin_cooking (mdp.py:1515-1522)
            sel_kb = 4; nkb = KB_POT + PC_COOKING; flags = F4_CHG | F4_START;
        } else if (hc == 3 && oc == PC_READY) {                          // soup pickup (mdp.py:1525-1539)
            sel_h = 1; sel_o = 0x0C; sel_kb = 4; nkb = KB_POT + PC_EMPTY; flags = F4_CHG | F4_PLATE; rew = RW4_PLATE;
        } else if ((hc == 1 || hc == 2) && oc <= PC_IDLE2) {             // add ingredient (mdp.py:1541-1568)
            // wire code of the soup afterwards = old code + 8 (one more item) + tomato bit at position n (+ 0x80 for
            // the first item: the pot's object byte is 0 then)
            sel_h = 0x0C; sel_o = 2; sel_kb = 4; nkb = KB_POT + oc + 1; flags = F4_CHG | F4_PLACE; rew = RW4_PLACE;
            add = 8u + ((hc == 2 ? 1u : 0u) << oc) + (oc == 0 ? 0x80u : 0u);
        }
    } else if (type == OC_T_SERVE) {
        if (hc == 4) { sel_h = 0x0C; flags = F4_SERVE; }                 // deliver (mdp.py:1570-1577)
    }
    return Lut4Entry{6u | (sel_h << 8) | (sel_o << 16) | (sel_kb << 24),
                     nkb | (cobj << 8) | (flags << 16) | (((uint32_t)dd & 0xFFu) << 24), add << 16, rew};
}

struct Lut4Table { Lut4Entry e[2][2 * LUT4_KEYS]; };  // [old_dynamics][key]
constexpr Lut4Table make_lut4() {
    Lut4Table t{};
    for (int od = 0; od < 2; ++od)
        for (int type = 0; type < 8; ++type)
            for (int hc = 0; hc < 5; ++hc)
                for (int oc = 0; oc < 6; ++oc) {
                    t.e[od][type * 30 + hc * 6 + oc] = lut4_entry(od, type, hc, oc);
                    t.e[od][LUT4_KEYS + type * 30 + hc * 6 + oc] = lut4_entry(od, 7, hc, oc);
                }
    return t;
}
__device__ const Lut4Table g_lut4 = make_lut4();

constexpr uint32_t DC0 = 0xFFFFFFFFu;
template <int MAXP>
struct Env4 {
    uint32_t h0, h1;                 // hands: wire object code in BYTE 1 (the other bytes are whatever the last interact left)
    uint32_t J;                      // JOINT: byte offset of this joint pose's row in the move table
    uint32_t pos0, or0, pos1, or1;   // arithmetic movement: cell index / orientation
    uint32_t tleft, over;            // timestep = horizon - 1 - tleft + over (over > 0: running past the horizon)
    uint32_t dcount;                 // loose dishes on counters MINUS ONE (from DC0): "none" is the sign bit
    uint32_t rem[MAXP];              // steps until the pot is ready (REM_IDLE when it is not cooking)
    uint32_t tk[MAXP];               // wire tick byte the pot arrived with (its tick when the countdown never ran)
    uint32_t poff[MAXP];             // byte offset of the pot's cell word in this lane's LDS column
    uint32_t exotic;                 // bit k: pot k arrived holding an ingredient-less soup object (kept as "empty")
    uint32_t pending;                // bit k: old dynamics, pot k arrived idle with 3 items: it starts in the first step's env effects
};

// LDS accesses by absolute byte address.  k_rollout4 has no static __shared__, so its dynamic region starts at LDS
// address 0 and table offsets ARE addresses: nothing is added per access, constants fold into the DS offset field.
#define OC_LDS __attribute__((address_space(3)))
typedef uint32_t oc_u32x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ uint32_t lds_rd16(uint32_t a) {
    const uint32_t v = *(const OC_LDS uint16_t*)(uintptr_t)a;
    __builtin_assume(v <= 0xFFFFu);  // ds_read_u16 zero-extends: spare the re-masking of values carried across steps
    return v;
}
__device__ __forceinline__ uint32_t lds_rd32(uint32_t a) { return *(const OC_LDS uint32_t*)(uintptr_t)a; }
__device__ __forceinline__ uint4 lds_rd128(uint32_t a) {
    const oc_u32x4 v = *(const OC_LDS oc_u32x4*)(uintptr_t)a;
    return make_uint4(v.x, v.y, v.z, v.w);
}
typedef uint32_t oc_u32x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ uint2 lds_rd64(uint32_t a) {
    const oc_u32x2 v = *(const OC_LDS oc_u32x2*)(uintptr_t)a;
    return make_uint2(v.x, v.y);
}
// 12-byte records at a 12-byte stride (ds_read_b96 / ds_write_b96: gfx950 takes them at 4-byte alignment)
typedef uint32_t oc_u32x3 __attribute__((ext_vector_type(3)));
struct oc_rec3 { uint32_t x, y, z; };
__device__ __forceinline__ oc_rec3 lds_rd96(uint32_t a) {
    typedef oc_u32x3 __attribute__((aligned(4))) u32x3_a4;
    const oc_u32x3 v = *(const OC_LDS u32x3_a4*)(uintptr_t)a;
    return oc_rec3{v.x, v.y, v.z};
}
__device__ __forceinline__ void lds_wr96(uint32_t a, uint32_t x, uint32_t y, uint32_t z) {
    typedef oc_u32x3 __attribute__((aligned(4))) u32x3_a4;
    const oc_u32x3 v = {x, y, z};
    *(OC_LDS u32x3_a4*)(uintptr_t)a = v;
}
__device__ __forceinline__ void lds_wr64(uint32_t a, uint32_t x, uint32_t y) {
    const oc_u32x2 v = {x, y};
    *(OC_LDS oc_u32x2*)(uintptr_t)a = v;
}
// Progress counters between the two wavefronts of a MODE 3 pair.  poll: a fresh wave-uniform read (never a value the
// compiler has kept in a register); post: lane 0 writes — the LDS executes a wavefront's instructions in order, so whatever
// the wavefront wrote (or read) before is done when the other side sees the new count.
__device__ __forceinline__ uint32_t lds_poll32(uint32_t a) {
    uint32_t v;
    asm volatile("ds_read_b32 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(v) : "v"(a) : "memory");
    return (uint32_t)__builtin_amdgcn_readfirstlane((int)v);
}
__device__ __forceinline__ void lds_post32(uint32_t a, uint32_t v) {
    if ((threadIdx.x & 63u) == 0u) asm volatile("ds_write_b32 %0, %1" : : "v"(a), "v"(v) : "memory");
}
__device__ __forceinline__ void lds_wr16(uint32_t a, uint32_t v) { *(OC_LDS uint16_t*)(uintptr_t)a = (uint16_t)v; }
__device__ __forceinline__ void lds_wr32(uint32_t a, uint32_t v) { *(OC_LDS uint32_t*)(uintptr_t)a = v; }
__device__ __forceinline__ void lds_wr8(uint32_t a, uint32_t v) { *(OC_LDS uint8_t*)(uintptr_t)a = (uint8_t)v; }

// Cell words, [cell][lane] in LDS.  CW = 2: u16 = object code | key byte << 8.  CW = 4 (the instance that runs one wavefront
// per SIMD on one small layout): u32 with the same two bytes in its UPPER half — the word an interact produces
// ([flags][new hand][new object][new key byte]) is stored as it is, values carried across steps are full dwords (16-bit
// loads carried through the loop are re-masked every step) and the lanes of a wavefront fall into distinct banks whatever
// cells they touch (two lanes share a bank with u16 words).
template <int CW> __device__ __forceinline__ uint32_t cw_rd(uint32_t a) { return CW == 2 ? lds_rd16(a) : lds_rd32(a); }
template <int CW> __device__ __forceinline__ uint32_t cw_kb(uint32_t c) { return CW == 2 ? c >> 8 : c >> 24; }
template <int CW> __device__ __forceinline__ uint32_t cw_obj(uint32_t c) { return CW == 2 ? (c & 0xFFu) : ((c >> 16) & 0xFFu); }
template <int CW> __device__ __forceinline__ uint32_t cw_make(uint32_t obj, uint32_t kb) { return (obj | (kb << 8)) << (CW == 2 ? 0 : 16); }
template <int CW> __device__ __forceinline__ void cw_wr(uint32_t a, uint32_t w) { if (CW == 2) lds_wr16(a, w); else lds_wr32(a, w); }
template <int CW> __device__ __forceinline__ void cw_wr_kb(uint32_t a, uint32_t kb) { lds_wr8(a + (CW == 2 ? 1u : 3u), kb); }
// the cell word an interact leaves behind, from its result [flags][new hand][new object][new key byte] (CW = 2: the upper
// half, stored with ds_write_b16_d16_hi)
template <int CW> __device__ __forceinline__ uint32_t cw_of_result(uint32_t r) { return CW == 2 ? r >> 16 : r; }

__device__ __forceinline__ uint32_t key_byte_of(uint32_t terrain_type, uint32_t o) {
    const uint32_t cls = terrain_type == OC_T_COUNTER ? (o == 0u ? 0u : o == OC_O_DISH ? 1u : 2u) : 0u;
    return terrain_type * 30u + cls;
}

// one player's INTERACT: `ent` = the LUT entry of (faced cell word, hand, does it interact); h carries the hand in byte 1
template <int CW>
__device__ __forceinline__ uint32_t interact4(const uint4 ent, uint32_t h, uint32_t cw) {
    const uint32_t pool = __builtin_amdgcn_perm(cw, h, CW == 2 ? 0x05040401u : 0x07060601u) + ent.z;  // [hand][object][object + add][key byte]
    return __builtin_amdgcn_perm(ent.y, pool, ent.x);                           // [flags][new hand][new object][new key byte]
}
template <int CW>
__device__ __forceinline__ uint32_t lut4_addr(uint32_t off, uint32_t h, uint32_t cw) {
    return (min((h >> 8) & 0xFFu, 4u) * 6u + cw_kb<CW>(cw)) * 16u + off;
}

// Two spare cell words per lane behind the grid: row n_obj * 16 takes the "ready" stores of pots that are not ripe, row
// n_obj * 16 + 1 is what the unused pot slots of a lane point at (an empty pot for ever: no pot slot needs a validity test)
template <int CW> __device__ __forceinline__ uint32_t nopot_off(int n_obj) { return ((uint32_t)n_obj * 16u + 1u) * (BLOCK * CW); }
template <int MAXP, int CW>
__device__ __forceinline__ void load_env4(const LayC& C, const Lay L, const uint4* __restrict__ st, int64_t n, int64_t e,
                                          int n_obj, int horizon, Env4<MAXP>& s, uint32_t col) {
    const uint4 h = st[e];
    s.pos0 = h.x & 0xFF; s.or0 = (h.x >> 8) & 0xFF; s.h0 = (h.x >> 8) & 0xFF00u; s.pos1 = h.x >> 24;
    s.or1 = h.y & 0xFF; s.h1 = h.y & 0xFF00u;
    const uint32_t t = h.y >> 16;
    s.tleft = t < (uint32_t)horizon ? (uint32_t)horizon - 1u - t : 0u;
    s.over = t < (uint32_t)horizon ? 0u : t - ((uint32_t)horizon - 1u);
    uint32_t dishes = 0;
    for (int p = 0; p < n_obj; ++p) {
        const uint4 v = st[(int64_t)(1 + p) * n + e];
        const uint32_t ow[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const uint32_t T = L.u32(L_TERRAIN + 16 * p + 4 * q);
#pragma unroll
            for (int b = 0; b < 4; ++b) {
                const uint32_t o = (ow[q] >> (8 * b)) & 0xFFu, type = (T >> (8 * b)) & 7u;
                dishes += (type == OC_T_COUNTER && o == OC_O_DISH) ? 1u : 0u;
                cw_wr<CW>(col + (uint32_t)(16 * p + 4 * q + b) * (BLOCK * CW), cw_make<CW>(o, key_byte_of(type, o)));
            }
        }
    }
    s.dcount = dishes + DC0;
    s.exotic = 0; s.pending = 0;
    cw_wr<CW>(col + nopot_off<CW>(n_obj), cw_make<CW>(0u, KB_POT + PC_EMPTY));  // what unused pot slots read: an empty pot, never written
#pragma unroll
    for (int k = 0; k < MAXP; ++k) {
        s.rem[k] = REM_IDLE; s.tk[k] = 0; s.poff[k] = nopot_off<CW>(n_obj);
        if ((uint32_t)k < C.n_pots) {
            s.poff[k] = L.pot_cell(k) * (BLOCK * CW);
            uint32_t o = cw_obj<CW>(cw_rd<CW>(col + s.poff[k]));
            const uint32_t tkb = ((k < 4 ? h.z : h.w) >> (8 * (k & 3))) & 0xFFu;
            const uint32_t pc = pot_class(C, o, tkb);
            if (o == OC_O_SOUP) { s.exotic |= 1u << k; o = 0; }  // a soup object without ingredients behaves as an empty pot
            if (C.old_dyn && pc == PC_IDLE3) s.pending |= 1u << k;
            s.tk[k] = tkb;
            s.rem[k] = pc == PC_COOKING ? cook_of(C, o) - (tkb - 1u) : REM_IDLE;
            cw_wr<CW>(col + s.poff[k], cw_make<CW>(o, KB_POT + pc));
        }
    }
}

template <int MAXP, int CW>
__device__ __forceinline__ void store_env4(const LayC& C, const Lay L, uint4* __restrict__ st, int64_t n, int64_t e,
                                           int n_obj, int horizon, const Env4<MAXP>& s, uint32_t col) {
    uint4 h;
    const uint32_t t = min((uint32_t)horizon - 1u - s.tleft + s.over, 0xFFFFu);  // the wire format's u16: saturates
    h.x = s.pos0 | (s.or0 << 8) | ((s.h0 & 0xFF00u) << 8) | (s.pos1 << 24);
    h.y = s.or1 | (s.h1 & 0xFF00u) | (t << 16);
    h.z = 0; h.w = 0;
    uint32_t pot_fix_cell[MAXP], pot_fix_obj[MAXP];
#pragma unroll
    for (int k = 0; k < MAXP; ++k) {
        pot_fix_cell[k] = 0xFFFFFFFFu; pot_fix_obj[k] = 0;
        if ((uint32_t)k < C.n_pots) {
            const uint32_t cw = cw_rd<CW>(col + s.poff[k]), o = cw_obj<CW>(cw), pc = cw_kb<CW>(cw) - KB_POT;
            uint32_t tkb = s.tk[k];
            const bool live = rem_live(s.rem[k]);
            if (live) {
                const uint32_t cook = cook_of(C, o);
                tkb = (pc == PC_COOKING ? cook - s.rem[k] : cook) + 1u;
            }
            if (pc < PC_COOKING) tkb = 0;  // empty or idle
            if (k < 4) h.z |= tkb << (8 * (k & 3));
            else h.w |= tkb << (8 * (k & 3));
            if (o == 0u && ((s.exotic >> k) & 1u) && !live) { pot_fix_cell[k] = L.pot_cell(k); pot_fix_obj[k] = OC_O_SOUP; }
        }
    }
    st[e] = h;
    for (int p = 0; p < n_obj; ++p) {
        uint32_t ow[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            ow[q] = 0;
#pragma unroll
            for (int b = 0; b < 4; ++b) {
                const uint32_t c = (uint32_t)(16 * p + 4 * q + b);
                uint32_t o = cw_obj<CW>(cw_rd<CW>(col + c * (BLOCK * CW)));
#pragma unroll
                for (int k = 0; k < MAXP; ++k) o = c == pot_fix_cell[k] ? pot_fix_obj[k] : o;
                ow[q] |= o << (8 * b);
            }
        }
        st[(int64_t)(1 + p) * n + e] = make_uint4(ow[0], ow[1], ow[2], ow[3]);
    }
}

// restart at the horizon (OvercookedEnv.reset with the standard start state, env.py:288-319)
template <int MAXP, int CW>
__device__ __forceinline__ void env_reset4(const LayC& C, const Lay L, int n_obj, int horizon, Env4<MAXP>& s, uint32_t col) {
    s.pos0 = L.u8(L_START_POS); s.pos1 = L.u8(L_START_POS + 1);
    s.or0 = L.u8(L_START_OR); s.or1 = s.pos1 == 0xFFu ? 0u : L.u8(L_START_OR + 1);
    s.h0 = s.h1 = 0; s.dcount = DC0; s.exotic = 0; s.pending = 0;
    s.tleft = (uint32_t)horizon - 1u; s.over = 0;
    for (int c = 0; c < n_obj * 16; ++c) cw_wr<CW>(col + (uint32_t)c * (BLOCK * CW), cw_make<CW>(0u, (L.terrain((uint32_t)c) & 7u) * 30u));
#pragma unroll
    for (int k = 0; k < MAXP; ++k) {
        s.rem[k] = REM_IDLE; s.tk[k] = 0;
    }
}

// ------------------------------------------------------------------------------------------
// JOINT move table (built per workgroup).  Row J = (fi0 * 4 + or0) * NP + (fi1 * 4 + or1), fi = index of the player's
// cell among the layout's free cells (row-major), NP = 4 * #free cells; a row is 38 u16:
//    [ja] for ja = 6 * a0 + a1: byte offset (J' * 76) of the joint pose after resolve_movement
//    [36], [37]: LDS byte offsets (cell * 512) of the cells the two players face in pose J
// With 32-bit cell words (CW = 4) the row is 37 u32: [ja] = J' * 148, [36] = both faced-cell offsets (cell * 1024), u16 each
// (grids of at most 64 cells).
// ------------------------------------------------------------------------------------------
constexpr int JOINT_MAX_FLOOR = 7;
template <int CW> struct Mvj {
    static constexpr int ROW_BYTES = CW == 2 ? 76 : 148;  // bytes of a row
    static constexpr int FACES = CW == 2 ? 72 : 144;      // byte offset of the two faced-cell offsets (one u32) inside a row
};

// Built by every workgroup at launch (~2 us): the free cells are ranked by the first wavefront (ballot + prefix count),
// the single-player moves SP[pose][action] = (cell, pose) after _move_if_direction by 168 lanes, and a joint row then
// only combines two of them per entry (collision test + index arithmetic).  `sp` = 2 * 28 * 6 bytes of scratch LDS.
// `mvj` = the table, `mvj_addr` = its LDS address (rows hold LDS addresses of rows).
// FACE_STRIDE: bytes between two cells' words in a lane's column (the faced-cell offsets are multiples of it) — BLOCK * CW;
// MODE 4 keeps the compact 16-bit rows (CW = 2 here) next to 32-bit cell words: faces = cell * 1024 still fit 16 bits.
template <int CW, int FACE_STRIDE = BLOCK * CW>
__device__ __forceinline__ void build_joint_table(const Lay L, int W, uint8_t* mvj, uint32_t mvj_addr, uint8_t* s_fl, uint8_t* s_fi,
                                                  uint8_t* sp) {
    static_assert(CW == 4 || 63 * FACE_STRIDE < 65536, "16-bit faced-cell offsets");
    const int nc = (int)L.u8(L_NCELLS);
    if (threadIdx.x < 64) {  // free cells in row-major order: s_fi[cell] = rank (0xFF: not free / beyond the table), s_fl[rank] = cell
        int base = 0;
        for (int c0 = 0; c0 < nc; c0 += 64) {
            const int c = c0 + (int)threadIdx.x;
            const bool fl = c < nc && (L.terrain((uint32_t)c) & 7u) == OC_T_FLOOR;
            const uint64_t m = __ballot(fl);
            const int rank = base + __popcll(m & ((1ull << threadIdx.x) - 1ull));
            if (c < nc) s_fi[c] = (fl && rank < JOINT_MAX_FLOOR) ? (uint8_t)rank : (uint8_t)0xFF;
            if (fl && rank < JOINT_MAX_FLOOR) s_fl[rank] = (uint8_t)c;
            base += __popcll(m);
        }
        if (threadIdx.x == 0) s_fl[JOINT_MAX_FLOOR] = (uint8_t)min(base, JOINT_MAX_FLOOR);
    }
    __syncthreads();
    const int nf = s_fl[JOINT_MAX_FLOOR], NP = 4 * nf, NJ = NP * NP;
    auto ahead = [&](int c, int d) {  // cell in direction d, or c itself when that leaves the grid
        const int t = c + (d == 0 ? -W : d == 1 ? W : d == 2 ? 1 : -1);
        return (t >= 0 && t < nc) ? t : c;
    };
    if ((int)threadIdx.x < NP * 6) {  // one player alone: _move_if_direction (mdp.py:1718-1727)
        const int P = threadIdx.x / 6, a = threadIdx.x - P * 6;
        const int c = s_fl[P >> 2], o = P & 3;
        const int t = a < 4 ? ahead(c, a) : c;
        const int q = (a < 4 && s_fi[t] != 0xFF) ? t : c;
        sp[2 * threadIdx.x] = (uint8_t)q;
        sp[2 * threadIdx.x + 1] = (uint8_t)(s_fi[q] * 4 + (a < 4 ? a : o));
    }
    __syncthreads();
    for (int J = threadIdx.x; J < NJ; J += BLOCK) {
        const int P0 = J / NP, P1 = J - P0 * NP;
        const int c0 = s_fl[P0 >> 2], c1 = s_fl[P1 >> 2];
        uint16_t* row16 = reinterpret_cast<uint16_t*>(mvj + J * Mvj<CW>::ROW_BYTES);
        uint32_t* row32 = reinterpret_cast<uint32_t*>(mvj + J * Mvj<CW>::ROW_BYTES);
        *reinterpret_cast<uint32_t*>(mvj + J * Mvj<CW>::ROW_BYTES + Mvj<CW>::FACES) =
            (uint32_t)(ahead(c0, P0 & 3) * FACE_STRIDE) | ((uint32_t)(ahead(c1, P1 & 3) * FACE_STRIDE) << 16);
        int q1[6], p1[6];
#pragma unroll
        for (int a = 0; a < 6; ++a) { q1[a] = sp[2 * (P1 * 6 + a)]; p1[a] = sp[2 * (P1 * 6 + a) + 1]; }
#pragma unroll
        for (int a0 = 0; a0 < 6; ++a0) {
            const int q0 = sp[2 * (P0 * 6 + a0)], p0 = sp[2 * (P0 * 6 + a0) + 1];
#pragma unroll
            for (int a1 = 0; a1 < 6; ++a1) {
                // same target cell or swapped cells: nobody moves, orientations still turn (mdp.py:1673-1683, Q6)
                const bool collide = q0 == q1[a1] || (q0 == c1 && q1[a1] == c0);
                const int n0 = collide ? ((P0 & ~3) | (p0 & 3)) : p0, n1 = collide ? ((P1 & ~3) | (p1[a1] & 3)) : p1[a1];
                const uint32_t next_row = mvj_addr + (uint32_t)((n0 * NP + n1) * Mvj<CW>::ROW_BYTES);
                if (CW == 2) row16[a0 * 6 + a1] = (uint16_t)next_row; else row32[a0 * 6 + a1] = next_row;
            }
        }
    }
}

// One Philox block = 8 steps (include/oc_amd.h): word s >> 1, top base-36 digit (= 6 * a0 + a1) for even s, second for odd s
struct Phx4 { uint32_t w0, w1, w2, w3; };
__device__ __forceinline__ Phx4 philox_words(uint64_t blk, uint32_t g_lo, uint32_t g_hi, uint32_t seed_lo, uint32_t seed_hi) {
    uint32_t r[4];
    philox4x32_10((uint32_t)blk, g_lo, g_hi, (uint32_t)(blk >> 32), seed_lo, seed_hi, r);
    return Phx4{r[0], r[1], r[2], r[3]};
}
// The same block computed a few rounds at a time (the unrolled step loop spreads the next block's ten rounds over the
// shadows of its steps' look-ups instead of paying for them in one lump)
struct PhxInc {
    uint32_t c0, c1, c2, c3, k0, k1;
    __device__ __forceinline__ void start(uint64_t blk, uint32_t g_lo, uint32_t g_hi, uint32_t seed_lo, uint32_t seed_hi) {
        c0 = (uint32_t)blk; c1 = g_lo; c2 = g_hi; c3 = (uint32_t)(blk >> 32); k0 = seed_lo; k1 = seed_hi;
    }
    __device__ __forceinline__ void round() {  // one round of philox4x32_10 (common.hpp)
        const uint64_t p0 = (uint64_t)0xD2511F53u * c0, p1 = (uint64_t)0xCD9E8D57u * c2;
        const uint32_t hi0 = (uint32_t)(p0 >> 32), lo0 = (uint32_t)p0, hi1 = (uint32_t)(p1 >> 32), lo1 = (uint32_t)p1;
        const uint32_t n0 = hi1 ^ c1 ^ k0, n2 = hi0 ^ c3 ^ k1;
        c0 = n0; c1 = lo1; c2 = n2; c3 = lo0;
        k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
    }
    __device__ __forceinline__ Phx4 words() const { return Phx4{c0, c1, c2, c3}; }
};
// (the words are passed by value and picked with arithmetic masks: a chain of selects between fields of one struct gets
//  folded into a dynamically indexed load, which pins the struct in scratch memory — and a scratch load in the step loop
//  waits, through vmcnt, for the output stores of the previous steps)
__device__ __forceinline__ uint32_t joint_action_of(uint32_t w0, uint32_t w1, uint32_t w2, uint32_t w3, uint32_t s8) {
    const uint32_t pick = s8 >> 1;
    const uint32_t w = (w0 & (0u - (uint32_t)(pick == 0u))) | (w1 & (0u - (uint32_t)(pick == 1u))) |
                       (w2 & (0u - (uint32_t)(pick == 2u))) | (w3 & (0u - (uint32_t)(pick == 3u)));
    return __umulhi(w * ((s8 & 1u) ? 36u : 1u), 36u);
}

// LDS map of k_rollout4 — ONE dynamic region at compile-time offsets, starting at LDS address 0 (the kernel has no
// static __shared__; checked at kernel start), so that table offsets are LDS addresses:
//    MVJ               JOINT move table (MODE 1 only; capacity for NF free cells): first with CW = 2, after CT with CW = 4
//    ACT               [2][40] u16 (u32 with CW = 4): LUT address of "this player does / does not interact" per joint action (MODE 1)
//    LUT               interact table: one dynamics variant (one layout) or both
//    LAY               staged layout records
//    FL / FI           free-cell list / cell -> free-cell index (MODE 1)
//    CT                [32] cook time by the low five bits of the soup code (one layout)
//    CELLS             cell words u16 / u32 [n_obj * 16 + 2][BLOCK]
// With u16 table entries (CW = 2) the move table comes first, so that row addresses and the LUT addresses in ACT fit 16 bits;
// with u32 entries (CW = 4, an 85 KB move table) the small tables come first instead, so that THEIR addresses stay below
// 64 KiB and fold into the 16-bit offset field of the DS instructions.
template <bool UNIFORM, bool LAY_LDS, int MODE, int NF, bool ONE_LUT = UNIFORM, int CW = 2>
struct Lds4 {
    static constexpr int MVJ_CAP = MODE == 1 ? ((16 * NF * NF * Mvj<CW>::ROW_BYTES + 15) & ~15) : 0;
    static constexpr bool TABLE_FIRST = CW == 2;
    // MODE 1: [player][40] u16 / u32 LUT addresses
    static constexpr int ACT = TABLE_FIRST ? MVJ_CAP : 0, ACT_P1 = 40 * CW, ACT_BYTES = MODE == 1 ? 2 * ACT_P1 : 0;
    static constexpr int LUT = ACT + ACT_BYTES, LUT_BYTES = ONE_LUT ? LUT4_BYTES : 2 * LUT4_BYTES;
    static constexpr int LAY = LUT + LUT_BYTES, LAY_BYTES = LAY_LDS ? (UNIFORM ? 256 : LDS_LAYOUT_MAX * 256) : 16;
    static constexpr int FL = LAY + LAY_BYTES, FI = FL + 16, CT = FI + (MODE == 1 ? OC_MAX_CELLS : 0);
    static constexpr int MVJ = TABLE_FIRST ? 0 : CT + 32;  // LDS address of the move table
    static constexpr int CELLS = TABLE_FIRST ? CT + 32 : MVJ + MVJ_CAP;
    static_assert(CW == 4 || MVJ_CAP + ACT_BYTES + LUT4_KEYS * 16 < 65536, "row / LUT addresses are u16 in the tables");
    static_assert(CW == 2 || CT + 32 < 65536, "the small tables' addresses must fit the DS offset field");
};

// flags[k][e] for the 64 envs of this wavefront: SGPR row pointer + lane offset, no 64-bit address arithmetic
__device__ __forceinline__ void store_flag_byte(uint8_t* row, uint32_t lane_off, uint32_t v) {
    asm volatile("global_store_byte %0, %1, %2" : : "v"(lane_off), "v"(v), "s"(row) : "memory");
}

// a randomized start (get_random_start_state_fn, mdp.py:1307-1369) drawn by draw_start, in Env4 / key-byte form
template <int MAXP, int CW>
__device__ __forceinline__ void env_reset4_draw(const LayC& C, const Lay L, int n_obj, int horizon, Env4<MAXP>& s, uint32_t col,
                                                const StartDraw& d) {
    env_reset4<MAXP, CW>(C, L, n_obj, horizon, s, col);
    s.pos0 = d.pos0; s.pos1 = d.pos1;
    s.h0 = d.held0 << 8; s.h1 = d.held1 << 8;
#pragma unroll
    for (int k = 0; k < MAXP; ++k) {
        if ((uint32_t)k < C.n_pots) {
            const uint32_t o = d.pot_obj((uint32_t)k), tkb = d.tick((uint32_t)k);
            const uint32_t pc = pot_class(C, o, tkb);
            s.tk[k] = tkb;
            s.rem[k] = pc == PC_COOKING ? cook_of(C, o) - (tkb - 1u) : REM_IDLE;
            // old dynamics: a pot drawn idle with three items starts in the new episode's first step, as one that arrives so in the
            // loaded state (round 6: this line was missing — such pots never started after an in-kernel restart)
            if (C.old_dyn && pc == PC_IDLE3) s.pending |= 1u << k;
            cw_wr<CW>(col + s.poff[k], cw_make<CW>(o, KB_POT + pc));
        }
    }
}

// MODE 0: arithmetic movement, any table; MODE 1: JOINT move table (one two-player layout with <= NF free cells);
// MODE 2: per-env terrain, two players everywhere, at most 64 cells: resolve_movement (mdp.py:1644-1727) needs only the
//   static terrain, so the pose runs ONE STEP AHEAD of the interacts on a per-lane 64-bit floor mask (no cell reads for the
//   move targets), the faced cells of the next step are read as soon as this step's cell writes are issued (PIPE), and the
//   ~25 VALU of the movement fill the shadow of this step's LUT reads
// (The per-env-terrain step split between a mover and an interact wavefront per 64 envs — round 5's MODE 3 / MODE 4 of this
//  kernel — is k_rollout5, step_duo5.hpp, since round 6.)
// RU: every layout of the table has the same shaping rewards and dynamics flag (hint OC_BATCH_UNIFORM_SHAPING): one LUT
//   variant whose entries carry the reward floats, as with a single layout
// OLD: some layout of the table may use old dynamics (auto-start of full pots in the env effects)
// EV: event_infos are logged (per-step masks and / or per-episode counters, EvArgs)
// NOCONF: no non-floor cell of the layout touches two floor cells (hint OC_BATCH_NO_SHARED_FACES; cramped_room): the two players
//   can never face the same cell, so player 1 never has to redo its interact on a cell player 0 has just changed
// CW: bytes of a cell word (2, or 4 for the one-wavefront-per-SIMD instance of small single layouts: see cw_rd)
// FT8 (MODE 1 / 2 instances; option OC_OPT_FLAGS_TILED8): the flags array is tiled by 8 steps — flags[k / 8][e][k % 8] — so that a
//   wavefront stores the flag bytes of a whole unrolled block as ONE 512-byte piece of full lines instead of eight 64-byte
//   pieces in eight far-apart rows (which cost the rollout ~6 % and its run-to-run spread: NOTEBOOK, round 4).  The launch
//   must start on a block boundary and run whole blocks (t0 and n_steps multiples of 8): no rolled steps
// PIPE (MODE 1, 2): the next step's faced cells are read one step ahead.  That hides the read behind the tail of the step
//   when a SIMD holds one wavefront (65 536 envs); with two or more wavefronts per SIMD the extra LDS traffic costs
//   more than the latency it hides (131 072 cramped_room envs: 0.48 vs 0.65 us per batched step), so big batches turn it off
template <bool UNIFORM, int MAXP, bool LAY_LDS, int MODE, bool OUT, bool OLD, int NF = JOINT_MAX_FLOOR, bool EV = false,
          bool PIPE = true, bool RU = false, int CW = 2, bool NOCONF = false, bool FT8 = false>
#ifndef OC_R4_WAVES_MAX
#define OC_R4_WAVES_MAX 4
#endif
__global__ __launch_bounds__(BLOCK) __attribute__((amdgpu_waves_per_eu(1, OC_R4_WAVES_MAX))) void k_rollout4(const OcLayout* __restrict__ g_layouts, int n_layouts,
                                                    const uint16_t* layout_id, uint4* st,
                                                    float4* __restrict__ rewards, uint8_t* __restrict__ flags,
                                                    float4* __restrict__ ep_returns, int64_t n, int W, int n_obj,
                                                    int horizon, uint32_t options, uint32_t seed_lo, uint32_t seed_hi,
                                                    int64_t env_offset, int64_t t0, int n_steps, StartArgs sa, EvArgs ea) {
    extern __shared__ __attribute__((aligned(16))) uint8_t s_dyn4[];
    constexpr bool RUX = UNIFORM || RU;  // one LUT variant, patched with the reward floats
    using M = Lds4<UNIFORM, LAY_LDS, MODE, NF, RUX, CW>;
    static_assert(CW == 2 || CW == 4, "cell words are u16 or u32");
    if ((uint32_t)(uintptr_t)(OC_LDS uint8_t*)s_dyn4 != 0u) __builtin_trap();  // folds away: the region starts at address 0
    uint4* const s_lay = reinterpret_cast<uint4*>(s_dyn4 + M::LAY);
    uint4* const s_lut = reinterpret_cast<uint4*>(s_dyn4 + M::LUT);
    uint8_t* const s_fl = s_dyn4 + M::FL;
    uint8_t* const s_fi = s_dyn4 + M::FI;
    const uint32_t tid = threadIdx.x;
    const uint32_t blk = xcd_block();  // (common.hpp: each XCD owns a contiguous eighth of the envs)
    const int64_t e = (int64_t)blk * BLOCK + tid;
    const bool active = e < n;
    Lay L = stage_layouts<LAY_LDS>(g_layouts, n_layouts, layout_id, e, active, s_lay);  // contains a barrier
    {
        const uint4* src = reinterpret_cast<const uint4*>(&g_lut4);
        const int first = RUX ? (L.old_dynamics() ? 2 * LUT4_KEYS : 0) : 0, count = RUX ? 2 * LUT4_KEYS : 4 * LUT4_KEYS;
        for (int i = threadIdx.x; i < count; i += BLOCK) {
            uint4 ent = src[first + i];
            if (RUX)  // one layout (or one set of shaping rewards for the whole table): the entry carries the shaped reward itself
                ent.w = ent.w == RW4_PLACE ? __float_as_uint(L.rew_placement()) : ent.w == RW4_PLATE ? __float_as_uint(L.rew_soup()) : 0u;
            s_lut[i] = ent;
        }
    }
    if (UNIFORM && threadIdx.x < 32) {
        const LayC Cs = load_consts<true>(L);
        s_dyn4[M::CT + threadIdx.x] = (uint8_t)cook_of(Cs, OC_O_SOUP | threadIdx.x);
    }
    if (MODE == 1) {
        if (threadIdx.x < 36) {  // the LUT's LDS address is folded into the offsets
            const uint32_t a0 = (uint32_t)(M::LUT + (threadIdx.x / 6 == 5 ? 0 : LUT4_KEYS * 16));
            const uint32_t a1 = (uint32_t)(M::LUT + (threadIdx.x % 6 == 5 ? 0 : LUT4_KEYS * 16));
            if (CW == 2) {
                reinterpret_cast<uint16_t*>(s_dyn4 + M::ACT)[threadIdx.x] = (uint16_t)a0;
                reinterpret_cast<uint16_t*>(s_dyn4 + M::ACT + M::ACT_P1)[threadIdx.x] = (uint16_t)a1;
            } else {
                reinterpret_cast<uint32_t*>(s_dyn4 + M::ACT)[threadIdx.x] = a0;
                reinterpret_cast<uint32_t*>(s_dyn4 + M::ACT + M::ACT_P1)[threadIdx.x] = a1;
            }
        }
        build_joint_table<CW>(L, W, s_dyn4 + M::MVJ, (uint32_t)M::MVJ, s_fl, s_fi, s_dyn4 + M::CELLS);  // (scratch: the cell words come later)
    }
    __syncthreads();
    if (!active) return;
    const uint32_t col = (uint32_t)M::CELLS + tid * (uint32_t)CW;  // LDS address of this lane's column of cell words
    // (L, C, lut_var, two and MODE 2's floor mask change when a restart moves the env to another layout: StartArgs.regen_count)
    // EV with per-episode counters (EvArgs.counts): [N_EVENT_TYPES][BLOCK] u32 behind the cell words — read once, kept in LDS for the
    // launch, written back at its end (round 6: a read-modify-write of the counters in HBM inside the step loop made every step wait
    // for the output stores before it: 44.7 G env-steps/s on cramped_room against 340 G without the event log)
    const uint32_t cnt0 = (uint32_t)M::CELLS + ((uint32_t)n_obj * 16u + 2u) * (uint32_t)(BLOCK * CW) + tid * 4u;
    if (EV && ea.counts)
        for (int k = 0; k < N_EVENT_TYPES; ++k) lds_wr32(cnt0 + (uint32_t)k * (BLOCK * 4u), ea.counts[e * N_EVENT_TYPES + k]);
    LayC C = load_consts<UNIFORM>(L);
    uint32_t lut_var = (uint32_t)M::LUT + (RUX ? 0u : (C.old_dyn ? (uint32_t)LUT4_BYTES : 0u));  // this lane's LUT
    const uint32_t delta4 = make_delta4(W);
    Env4<MAXP> s;
    load_env4<MAXP, CW>(C, L, st, n, e, n_obj, horizon, s, col);
    bool two = MODE == 1 || MODE == 2 || s.pos1 != 0xFFu;
    uint64_t fm = 0;  // MODE 2: bit c = cell c is floor (static per layout)
    auto floor_mask_of = [&](const Lay Lx) __attribute__((always_inline)) {
        uint64_t m = 0;
        for (int i = 0; i < n_obj * 4; ++i) {
            const uint32_t T = Lx.u32(L_TERRAIN + 4 * i);
#pragma unroll
            for (int b = 0; b < 4; ++b)
                if (((T >> (8 * b)) & 7u) == OC_T_FLOOR && (uint32_t)(4 * i + b) < Lx.u8(L_NCELLS)) m |= 1ull << (4 * i + b);
        }
        return m;
    };
    if (MODE == 2) fm = floor_mask_of(L);
    auto joint_row = [&]() {  // LDS address of the row of the joint pose (pos0, or0, pos1, or1)
        const uint32_t NP = 4u * s_fl[JOINT_MAX_FLOOR];
        return (uint32_t)M::MVJ + ((s_fi[s.pos0] * 4u + s.or0) * NP + (s_fi[s.pos1] * 4u + s.or1)) * (uint32_t)Mvj<CW>::ROW_BYTES;
    };
    if (MODE == 1) s.J = joint_row();
    float4 ep = ep_returns ? ep_returns[e] : make_float4(0.f, 0.f, 0.f, 0.f);
    typedef float f32x2 __attribute__((ext_vector_type(2)));
    typedef float f32x4 __attribute__((ext_vector_type(4)));
    // The episode's shaped returns live in one register pair and gain the upper half of every step's reward quad (a player
    // earns at most one shaped reward per step, so quad.zw IS what the returns gain; a restart leaves -quad.zw behind so
    // that the sum comes out as zero)
    f32x2 epsh = {ep.z, ep.w};
    const uint64_t g = (uint64_t)(env_offset + e);
    const uint32_t g_lo = (uint32_t)g, g_hi = (uint32_t)(g >> 32);
    float4* rew_k = rewards ? rewards + (int64_t)blk * BLOCK : nullptr;  // wave-uniform row pointers
    const uint32_t wave_base = (uint32_t)__builtin_amdgcn_readfirstlane((int)(tid & ~63u));  // first lane of this wavefront
    static_assert(!FT8 || ((MODE == 1 || MODE == 2) && OUT && !EV), "the tiled flags array is served by joint-table and per-env-terrain instances");
    uint8_t* flg_k = flags ? flags + ((int64_t)blk * BLOCK + wave_base) * (FT8 ? 8 : 1) : nullptr;  // (FT8: the tile row of 8 steps)
    uint32_t flt_lo = 0, flt_hi = 0;  // FT8: the flag bytes of the block's steps 0..3 / 4..7
    const uint32_t lane = tid & 63u;
    const float4 zero4 = make_float4(0.f, 0.f, 0.f, 0.f);
    uint32_t step_k = 0;  // index of the step within this launch (wave-uniform): the epoch offset of a restart

    // ---- resolve_interacts + env effects + bookkeeping for one step.  fo*: LDS addresses of the faced cells, off*: LUT
    //      address of (this lane's table, does the player interact), c*: the faced cell words (already read).
    //      m0..m3: the caller's movement result for this step, overwritten inside the horizon branch when the env is put
    //      back to its start state — MODE 1: (row of the next pose, its faced cells, row of the pose after that; ja2n = the
    //      next step's joint action * 2), MODE 0: (pos0, pos1, or0, or1), MODE 2: (pos0, -, pos1, or0, or1 = m4; the faced cells' LDS addresses in nf0 / nf1)
    //      of the next step's pose.
    //      pw[k]: pot k's cell word read BEFORE this step's interacts (its class is the "pot_states" of mdp.py:1439),
    //      cookv[k]: the cook time of what that pot holds (FAST_START only), MAXP <= 2.
    // (!PIPE = two or more wavefronts per SIMD: LDS operations are what the step is short of — no pot word / cook time
    //  reads in the straight line, the rare branch looks them up)
    constexpr bool PW = MAXP <= 2 && PIPE;
    constexpr bool FAST_START = UNIFORM && MAXP == 1 && !OLD && PIPE;  // a cooking start is two instructions of the straight line
    auto cook_time = [&](uint32_t soup) __attribute__((always_inline)) {
        return UNIFORM ? (uint32_t)*(const OC_LDS uint8_t*)(uintptr_t)((uint32_t)M::CT + (soup & 31u)) : cook_of(C, soup);
    };
    // some recipe of the batch's layout cooks in zero steps (FAST_START handles that start in the rare branch)
    bool zero_cook = false;
    if (FAST_START) {
#pragma unroll
        for (int no = 0; no <= 3; ++no)
#pragma unroll
            for (int nt = 0; nt + no <= 3; ++nt)
                if (no + nt > 0) zero_cook |= ((C.cook[nt] >> (8 * no)) & 0xFFu) == 0u;
    }
    //      MODE 1 also prefetches the next step's faced cells (nc0, nc1) and pot words (npw) as soon as this step's cell
    //      writes are issued — m1 holds the next pose's faced-cell offsets on entry.
    const uint32_t dummy = col + (uint32_t)n_obj * 16u * (BLOCK * CW);  // a spare cell word per lane (one row past the grid)
    auto rd_pots = [&](uint32_t (&out)[MAXP]) __attribute__((always_inline)) {
#pragma unroll
        for (int k = 0; k < MAXP; ++k) out[k] = PW ? cw_rd<CW>(col + s.poff[k]) : 0u;
    };
    // Output rows of a whole unrolled block of 8 steps (OUT instances): per-lane byte offsets of the block's eight reward
    // quads / flag bytes from the block's first row, computed once — the step then stores through (row of the block's
    // first step) + offset[k] and the two row pointers advance once per block instead of every step (4 scalar
    // instructions per step less).  Needs 8 rows within 4 GiB; larger batches stay in the rolled loop.
    const bool blocked_rows = OUT && n < ((int64_t)1 << 24);
    uint32_t rew_off[8], flg_off[8];
#pragma unroll
    for (int k8 = 0; k8 < 8; ++k8) {
        rew_off[k8] = (tid + (uint32_t)k8 * (uint32_t)n) * 16u;
        flg_off[k8] = lane + (uint32_t)k8 * (uint32_t)n;
    }
    // The look-ups of a step: both players' LUT entries against the pre-step cells (resolve_interacts, mdp.py:1432-1579)
    // and — FAST_START — the cook time of what the pot holds, should somebody start it.  Issued first; whatever the
    // caller has that does not depend on them runs while they are in flight.
    struct Looked { uint4 e0, e1; uint32_t cookv; };
    auto look_up = [&](uint32_t off0, uint32_t off1, uint32_t c0, uint32_t c1, const uint32_t (&pw)[MAXP]) __attribute__((always_inline)) {
        if (CW == 2) {  // (values read with ds_read_u16 a step earlier: tell the compiler they are still 16 bits wide)
            __builtin_assume(c0 <= 0xFFFFu); __builtin_assume(c1 <= 0xFFFFu);
            __builtin_assume(off0 <= 0xFFFFu); __builtin_assume(off1 <= 0xFFFFu);
#pragma unroll
            for (int k = 0; k < MAXP; ++k) __builtin_assume(pw[k] <= 0xFFFFu);
        }
        Looked q;
        q.e0 = lds_rd128(lut4_addr<CW>(off0, s.h0, c0));
        q.e1 = lds_rd128(lut4_addr<CW>(off1, s.h1, c1));
        q.cookv = FAST_START ? cook_time(cw_obj<CW>(pw[0])) : 0u;
        return q;
    };
    // Outputs of a step whose stores (and episode-return additions) are put off to the next step of the same unrolled block,
    // where they run while that step's look-ups are in flight.
    struct Pend { uint64_t lo, hi; uint32_t fl; };  // the reward quad as two register pairs (sparse, shaped) and the flag byte
    auto flush = [&](const Pend& p, int k8) __attribute__((always_inline)) {
        // One block, straight from (row SGPR pair, lane offset) — no copy of the offset to keep the compiler from folding it
        // into a 64-bit address: the quad (its two pairs sit in one four-register tuple), the flag byte, and the episode's
        // shaped returns as one packed add.  A store of more than 8 bytes needs two wait states before its data registers
        // may be rewritten: the two instructions behind it.
        typedef uint64_t u64x2 __attribute__((ext_vector_type(2)));
        const u64x2 q = {p.lo, p.hi};
        // Two or more wavefronts per SIMD (the one-wavefront-per-env-group instances on big batches: BASELINE configs[4] at
        // 131 072 envs per GPU) store the quads with `sc1 nt` — streaming, written through: a step's row is written once and read by
        // nobody on this GPU before the launch ends.  Measured: 304 -> 315 G env-steps/s there (tools/store_rollout.hip: the store
        // path alone 5.36 -> 5.52 TB/s; the same bits on the flag tiles cost it all again).
        // -DOC_R4_PLAIN_QUADS: plain stores everywhere, for A/B.
#ifdef OC_R4_PLAIN_QUADS
#define OC_R4_QUAD_POLICY ""
#else
#define OC_R4_QUAD_POLICY " sc1 nt"
#endif
        if (FT8) {  // (the flag byte has gone into the block's tile)
            asm volatile("global_store_dwordx4 %1, %2, %4" OC_R4_QUAD_POLICY "\n\tv_pk_add_f32 %0, %0, %3\n\ts_nop 0"
                         : "+v"(epsh) : "v"(rew_off[k8 & 7]), "v"(q), "v"(p.hi), "s"(rew_k) : "memory");
        } else {
            asm volatile("global_store_dwordx4 %1, %2, %4" OC_R4_QUAD_POLICY "\n\tglobal_store_byte %5, %6, %7\n\t"
                         "v_pk_add_f32 %0, %0, %3"
                         : "+v"(epsh)
                         : "v"(rew_off[k8 & 7]), "v"(q), "v"(p.hi), "s"(rew_k), "v"(flg_off[k8 & 7]), "v"(p.fl), "s"(flg_k)
                         : "memory");
        }
    };
    // k8: index of the step inside an unrolled block of 8 (stores through the block's row + offset), or -1 (rolled step);
    // defer: where to leave the step's outputs instead of storing them (flush() follows), or nullptr
    auto core = [&](uint32_t fo0, uint32_t fo1, uint32_t off0, uint32_t off1, uint32_t c0, uint32_t c1, uint32_t ja2n,
                    const uint32_t (&pw)[MAXP], uint32_t& m0, uint32_t& m1, uint32_t& m2, uint32_t& m3, uint32_t& m4,
                    uint32_t& nf0, uint32_t& nf1, uint32_t& nc0, uint32_t& nc1, uint32_t (&npw)[MAXP], int k8,
                    const Looked& looked, Pend* defer = nullptr) __attribute__((always_inline)) {
        // ---- the straight line: everything a step does when nothing rare happens -------------------------------------
        const uint32_t cookv = looked.cookv;
        const uint4 e0 = looked.e0;
        uint4 e1 = looked.e1;
        const uint32_t r0 = interact4<CW>(e0, s.h0, c0);
        uint32_t r1 = interact4<CW>(e1, s.h1, c1);
        const uint32_t cw0 = cw_of_result<CW>(r0);  // player 0's faced cell afterwards
        cw_wr<CW>(fo0, cw0);
        cw_wr<CW>(fo1, cw_of_result<CW>(r1));
        const uint32_t h0_before = s.h0, h1_before = s.h1, dc_before = s.dcount;
        const uint32_t dc_mid = dc_before + (uint32_t)((int32_t)e0.y >> 24);  // loose dishes after player 0's interact
        uint32_t dcount = dc_mid + (uint32_t)((int32_t)e1.y >> 24);
        // step_environment_effects (mdp.py:1691-1703): the countdowns; a finished one turns the pot ready — every lane
        // stores, the others into their spare cell word.  begin_cooking (mdp.py:1515-1522) = load the countdown: tick 0
        // now, cooked once by this step's env effects (Q4).
        uint32_t rem_before[MAXP];
        bool ripe[MAXP];
#pragma unroll
        for (int k = 0; k < MAXP; ++k) {
            rem_before[k] = s.rem[k];
            if (FAST_START) s.rem[k] = ((r0 | r1) & F4_START) ? cookv : s.rem[k];
            s.rem[k] -= 1u;
            ripe[k] = s.rem[k] == 0u;
            if (PIPE) {
                if (MAXP <= 2 || (uint32_t)k < C.n_pots) cw_wr_kb<CW>(ripe[k] ? col + s.poff[k] : dummy, KB_POT + PC_READY);
            } else if (ripe[k]) {  // only the lanes concerned store
                cw_wr_kb<CW>(col + s.poff[k], KB_POT + PC_READY);
            }
        }
        if (MODE == 1) {  // LDS addresses of the next step's faced cells (its fo0 / fo1); MODE 2: the caller has put them in nf0 / nf1
            nf0 = col + (m1 & 0xFFFFu);
            nf1 = col + (m1 >> 16);
        }
        if ((MODE == 1 || MODE == 2) && PIPE) {  // the next step's cells: everything this step writes to the grid has been issued
            nc0 = cw_rd<CW>(nf0);
            nc1 = cw_rd<CW>(nf1);
            rd_pots(npw);
        }
        // shaped reward of potting / soup pickup straight from the entries (class -> this lane's layout when the table is mixed)
        auto shaped_of = [&](uint32_t w) __attribute__((always_inline)) {
            if (RUX) return __uint_as_float(w);
            // class -> this lane's layout, with masks (left as a ternary chain the compiler builds four exec-mask branches)
            return __uint_as_float((__float_as_uint(C.rew_place) & (0u - (uint32_t)(w == RW4_PLACE))) |
                                   (__float_as_uint(C.rew_soup) & (0u - (uint32_t)(w == RW4_PLATE))));
        };
        const float sh0 = shaped_of(e0.w);
        float sh1 = shaped_of(e1.w);
        const bool done = s.tleft == 0u;
        const bool conflict = NOCONF ? false : (fo0 == fo1) & ((r0 & F4_CHG) != 0u);
        // ---- ONE branch for everything rare; its test is integer arithmetic up to one compare (no SGPR hand-offs) ----
        // bit 0 (= F4_TAKE_DISH) of `take`: a dish taken from the dispenser may be "useful" — some pot was (pot_states
        // before the interacts: class idle 1, idle 2, cooking or ready) and no dish lay on a counter, before or after
        // player 0's own pick-up
        static_assert(F4_TAKE_DISH == 1, "the gate is built in bit 0");
        constexpr uint32_t USEFUL_CLASSES = (1u << PC_IDLE1) | (1u << PC_IDLE2) | (1u << PC_COOKING) | (1u << PC_READY);
        uint32_t take = (uint32_t)min((int32_t)dc_before, (int32_t)dc_mid) >> 31;  // (counts run from -1)
        if (PW) {
            uint32_t ub = 0;
#pragma unroll
            // (every slot reads a pot word — nopot_off — so the key byte is in the pot range; the & 31 only keeps a foreign
            //  key byte from being an undefined C++ shift: the hardware shift takes the low five bits anyway, no instruction)
            for (int k = 0; k < MAXP; ++k) ub |= USEFUL_CLASSES >> ((cw_kb<CW>(pw[k]) - KB_POT) & 31u);
            take &= ub;
        }
        uint32_t gate = (uint32_t)F4_SERVE;
        if (!FAST_START) gate |= F4_START;
        else gate |= zero_cook ? (uint32_t)F4_START : 0u;
        if (OLD) gate |= C.old_dyn ? (uint32_t)F4_PLACE : 0u;  // old dynamics: the third item starts the pot (Q11)
        uint32_t rare_bits;
        bool rare;
        {
            const uint32_t tleft_new = s.tleft - 1u;  // the horizon: the sign bit (tleft < 2^31)
            s.tleft = tleft_new;
            rare_bits = ((r0 | r1) & (take | gate)) | (tleft_new & 0x80000000u);
            if (OLD) rare_bits |= s.pending;
            rare = (rare_bits != 0u) | conflict;
        }
        uint32_t nh0 = r0, nh1 = r1;  // the hands after the step
        float4 rw = make_float4(0.f, 0.f, sh0, sh1);  // this step's reward quad and flag byte: stored ONCE, after the branch
        uint64_t q_lo = 0, q_hi = 0;  // the same quad as two register pairs (what the unrolled blocks store; the rare branch rewrites both whole)
        if (RUX) {  // the entries carry the floats in .w: (e0.w, e1.w) -> one register pair with v_pk_mov_b32, the zeros with v_mov_b64
            const uint64_t a64 = ((uint64_t)e0.w << 32) | e0.z, b64 = ((uint64_t)e1.w << 32) | e1.z;
            asm("v_pk_mov_b32 %0, %1, %2 op_sel:[1,1] op_sel_hi:[1,1]" : "=v"(q_hi) : "v"(a64), "v"(b64));
            asm("v_mov_b64 %0, 0" : "=v"(q_lo));
        } else {
            q_hi = ((uint64_t)__float_as_uint(sh1) << 32) | __float_as_uint(sh0);
        }
        uint32_t fl = 0;                              // (a second store to the same address would wait for the first)
        if (__builtin_expect(rare, 0)) {
            bool grid_changed = false;  // something below wrote to the grid after the prefetch
            if (conflict) {
                e1 = lds_rd128(lut4_addr<CW>(off1, h1_before, cw0));
                r1 = interact4<CW>(e1, h1_before, cw0);
                nh1 = r1;
                cw_wr<CW>(fo1, cw_of_result<CW>(r1));
                dcount = dc_mid + (uint32_t)((int32_t)e1.y >> 24);
                sh1 = shaped_of(e1.w);
                grid_changed = true;
            }
            // cooking starts.  Old dynamics: a pot that has just received its third item — or arrived full and idle —
            // starts by itself in the env effects, same arithmetic (Q11).
            uint32_t smask = F4_START;
            if (OLD) smask |= C.old_dyn ? (uint32_t)F4_PLACE : 0u;
            bool starts = ((r0 | r1) & smask) != 0u;
            if (OLD) starts |= s.pending != 0u;
            if (starts || conflict) {
#pragma unroll
                for (int k = 0; k < MAXP; ++k) {
                    if (MAXP > 1 && (uint32_t)k >= C.n_pots) break;
                    const uint32_t pa = col + s.poff[k];
                    uint32_t soup = 0;
                    bool go = false;
#pragma unroll
                    for (int p = 0; p < 2; ++p) {
                        const uint32_t rr = p ? r1 : r0, fo = p ? fo1 : fo0;
                        const bool begins = (rr & F4_START) != 0u;
                        const bool fills = OLD && C.old_dyn && (rr & F4_PLACE) && (rr >> 24) == KB_POT + PC_IDLE3;
                        if ((begins | fills) && (MAXP == 1 || fo == pa)) { go = true; soup = (rr >> 16) & 0xFFu; }
                    }
                    if (OLD && ((s.pending >> k) & 1u)) { go = true; soup = cw_obj<CW>(cw_rd<CW>(pa)); }
                    // (the straight line may have loaded or not loaded the countdown from player 1's stale interact: redo)
                    uint32_t cook = 0;
                    if (go) cook = cook_time(soup);
                    s.rem[k] = (go ? cook : rem_before[k]) - 1u;
                    ripe[k] = s.rem[k] == 0u;
                    if (go) {  // (cook == 0: ready at once, never ticks)
                        s.exotic &= ~(1u << k);
                        cw_wr_kb<CW>(pa, (cook == 0u || ripe[k]) ? KB_POT + PC_READY : KB_POT + PC_COOKING);
                        grid_changed = true;
                    } else if (ripe[k]) {
                        cw_wr_kb<CW>(pa, KB_POT + PC_READY);
                        grid_changed = true;
                    }
                }
                if (OLD) s.pending = 0;
            }
            // deliveries and dish pick-ups
            const uint32_t hb0 = (h0_before >> 8) & 0xFFu, hb1 = (h1_before >> 8) & 0xFFu, hn0 = (r0 >> 8) & 0xFFu;
            if ((r0 | r1) & (F4_SERVE | F4_TAKE_DISH)) {
                // pot_states before any interact (mdp.py:1439): ready / cooking / 1..2 idle items <=> class not in {empty, idle 3}
                uint32_t useful_pots = 0;
#pragma unroll
                for (int k = 0; k < MAXP; ++k) {
                    if (MAXP > 1 && (uint32_t)k >= C.n_pots) break;
                    uint32_t kb;
                    if (PW) {
                        kb = cw_kb<CW>(pw[k]);
                    } else {  // the cells hold the classes after this step's interacts; a pot a player has just changed
                              // had the class of that player's faced cell word before (player 0's first)
                        const uint32_t pa = col + s.poff[k];
                        kb = cw_kb<CW>(cw_rd<CW>(pa));
                        kb = ((r1 & F4_POTBITS) && fo1 == pa) ? cw_kb<CW>(conflict ? cw0 : c1) : kb;
                        kb = ((r0 & F4_POTBITS) && fo0 == pa) ? cw_kb<CW>(c0) : kb;
                    }
                    useful_pots += (kb != KB_POT + PC_EMPTY && kb != KB_POT + PC_IDLE3) ? 1u : 0u;
                }
                // is_dish_pickup_useful (mdp.py:2180-2204): live hands / counters, stale pots
                const bool du0 = two & (((hb1 == OC_O_DISH) ? 1u : 0u) < useful_pots) & (dc_before == DC0);
                const bool du1 = two & (((hn0 == OC_O_DISH) ? 1u : 0u) < useful_pots) & (dc_mid == DC0);
                float4 r;
                r.z = (((r0 & F4_TAKE_DISH) != 0u) & du0) ? C.rew_dish : 0.f;
                r.w = (((r1 & F4_TAKE_DISH) != 0u) & du1) ? C.rew_dish : 0.f;
                r.x = r.y = 0.f;
                if ((r0 | r1) & F4_SERVE) {  // deliver_soup (mdp.py:1631-1642): the recipe value look-ups only where somebody serves
                    r.x = (r0 & F4_SERVE) ? L.value(recipe_idx(hb0) & 15u) : 0.f;
                    r.y = (r1 & F4_SERVE) ? L.value(recipe_idx(hb1) & 15u) : 0.f;
                }
                ep.x += r.x; ep.y += r.y;
                rw = make_float4(r.x, r.y, r.z + sh0, r.w + sh1);
            } else {
                rw.w = sh1;
            }
            if (done) {  // OvercookedEnv.step bookkeeping at the horizon (env.py:266-267, 321-325)
                fl = OC_F_DONE;
                if (options & OC_OPT_AUTO_RESET) {
                    if (sa.enabled) {  // the batch's start_state_fn: drawn from (seed, global env, epoch of this step)
                        if (!UNIFORM && sa.regen_count) {  // ... on a layout drawn for the new episode (regen_mdp, env.py:288-302)
                            const uint32_t lid = draw_layout(sa, g, sa.epoch + step_k);
                            sa.layout_ids[e] = (uint16_t)lid;
                            L = LAY_LDS ? Lay{reinterpret_cast<const uint8_t*>(s_lay) + lid * 256u}
                                        : Lay{reinterpret_cast<const uint8_t*>(g_layouts) + (size_t)lid * 256u};
                            C = load_consts<UNIFORM>(L);
                            lut_var = (uint32_t)M::LUT + (RUX ? 0u : (C.old_dyn ? (uint32_t)LUT4_BYTES : 0u));
#pragma unroll
                            for (int k = 0; k < MAXP; ++k) s.poff[k] = (uint32_t)k < C.n_pots ? L.pot_cell(k) * (BLOCK * CW) : nopot_off<CW>(n_obj);
                            if (MODE == 0) two = L.n_players() == 2u;
                            if (MODE == 2) fm = floor_mask_of(L);
                        }
                        env_reset4_draw<MAXP, CW>(C, L, n_obj, horizon, s, col,
                                              draw_start(L, g, sa.epoch + step_k, sa.seed_lo, sa.seed_hi, sa.random_start_pos, sa.thresh));
                        nh0 = s.h0; nh1 = s.h1;
                    } else {
                        env_reset4<MAXP, CW>(C, L, n_obj, horizon, s, col);
                        nh0 = nh1 = 0;
                    }
                    dcount = DC0;
                    ep = zero4;        // the episode ends with this step: its returns restart from zero,
                    epsh.x = -rw.z; epsh.y = -rw.w;
                    fl |= OC_F_RESET;
                    grid_changed = true;
                    if (MODE == 1) {  // redo the look-ahead from the start pose
                        m0 = joint_row();
                        m1 = lds_rd32(m0 + (uint32_t)Mvj<CW>::FACES);
                        m2 = CW == 2 ? lds_rd16(m0 + ja2n) : lds_rd32(m0 + ja2n);
                    } else if (MODE == 2) {
                        m0 = s.pos0; m2 = s.pos1; m3 = s.or0; m4 = s.or1;
                        nf0 = col + step_cell(s.pos0, s.or0, delta4) * (BLOCK * CW);
                        nf1 = col + step_cell(s.pos1, s.or1, delta4) * (BLOCK * CW);
                    } else {
                        m0 = s.pos0; m1 = s.pos1; m2 = s.or0; m3 = s.or1;
                    }
                } else {
                    s.tleft = 0u;
                    s.over += 1u;
                }
            }
            if (MODE == 1 && grid_changed) {  // (the restart may have changed the next pose)
                nf0 = col + (m1 & 0xFFFFu);
                nf1 = col + (m1 >> 16);
            }
            if ((MODE == 1 || MODE == 2) && PIPE && grid_changed) {  // read the next step's cells again
                nc0 = cw_rd<CW>(nf0);
                nc1 = cw_rd<CW>(nf1);
                rd_pots(npw);
            }
            q_lo = ((uint64_t)__float_as_uint(rw.y) << 32) | __float_as_uint(rw.x);
            q_hi = ((uint64_t)__float_as_uint(rw.w) << 32) | __float_as_uint(rw.z);
        }
        if (EV) {  // event_infos of the step (mdp.py:2121-2308), from the outcomes above
            uint64_t ev = 0;
            const bool acted = (((r0 | r1) & 0xFFu) != 0u) | ((((r0 ^ h0_before) | (r1 ^ h1_before)) & 0xFF00u) != 0u);
            if (acted) {
                const uint32_t hb0 = (h0_before >> 8) & 0xFFu, hb1 = (h1_before >> 8) & 0xFFu, hn0 = (r0 >> 8) & 0xFFu;
                const uint32_t hn1 = (r1 >> 8) & 0xFFu, cc1 = conflict ? cw0 : c1;
                uint32_t useful_pots = 0, n_full = 0;
#pragma unroll
                for (int k = 0; k < MAXP; ++k) {
                    if (MAXP > 1 && (uint32_t)k >= C.n_pots) break;
                    uint32_t kb;
                    if (PW) {
                        kb = cw_kb<CW>(pw[k]);
                    } else {
                        const uint32_t pa = col + s.poff[k];
                        kb = cw_kb<CW>(cw_rd<CW>(pa));
                        kb = ((r1 & F4_POTBITS) && fo1 == pa) ? cw_kb<CW>(cc1) : kb;
                        kb = ((r0 & F4_POTBITS) && fo0 == pa) ? cw_kb<CW>(c0) : kb;
                    }
                    useful_pots += (kb != KB_POT + PC_EMPTY && kb != KB_POT + PC_IDLE3) ? 1u : 0u;
                    n_full += kb >= KB_POT + PC_IDLE3 ? 1u : 0u;
                }
                const bool du0 = two & (((hb1 == OC_O_DISH) ? 1u : 0u) < useful_pots) & (dc_before == DC0);
                const bool du1 = two & (((hn0 == OC_O_DISH) ? 1u : 0u) < useful_pots) & (dc_mid == DC0);
                const uint32_t t0_ = (cw_kb<CW>(c0) * 137u) >> 12, t1_ = (cw_kb<CW>(cc1) * 137u) >> 12;  // key byte / 30 = terrain type
                const bool disp0 = (t0_ == OC_T_ONION_DISP) | (t0_ == OC_T_TOMATO_DISP) | (t0_ == OC_T_DISH_DISP);
                const bool disp1 = (t1_ == OC_T_ONION_DISP) | (t1_ == OC_T_TOMATO_DISP) | (t1_ == OC_T_DISH_DISP);
                ev = interact_events<0>(C, t0_, hb0, cw_obj<CW>(c0), ((r0 & F4_CHG) != 0u) & (t0_ == OC_T_COUNTER),
                                        disp0 & (hb0 == 0u) & (hn0 != 0u), (r0 & F4_PLACE) != 0u, (r0 & F4_PLATE) != 0u,
                                        (r0 & F4_SERVE) != 0u, hb1, du0, n_full, two) |
                     interact_events<1>(C, t1_, hb1, cw_obj<CW>(cc1), ((r1 & F4_CHG) != 0u) & (t1_ == OC_T_COUNTER),
                                        disp1 & (hb1 == 0u) & (hn1 != 0u), (r1 & F4_PLACE) != 0u, (r1 & F4_PLATE) != 0u,
                                        (r1 & F4_SERVE) != 0u, hn0, du1, n_full, two);
            }
            if (ea.events) ea.events[(int64_t)step_k * n + e] = ev;
            if (ea.counts) {  // the episode's counters live in LDS for the launch ([event type][lane]; count_events' packing)
                while (ev) {
                    const int b = __ffsll((long long)ev) - 1;
                    const uint32_t a = cnt0 + (uint32_t)(b >> 1) * (BLOCK * 4u);
                    lds_wr32(a, lds_rd32(a) + (1u << (16 * (b & 1))));
                    ev &= ev - 1ull;
                }
                if (done) {  // the episode ends: publish its counts, start the next one from zero
                    for (int k = 0; k < N_EVENT_TYPES; ++k) {
                        const uint32_t a = cnt0 + (uint32_t)k * (BLOCK * 4u);
                        if (ea.counts_done) ea.counts_done[e * N_EVENT_TYPES + k] = lds_rd32(a);
                        if ((options & OC_OPT_AUTO_RESET) || ea.clear_on_done) lds_wr32(a, 0u);
                    }
                }
            }
        }
        if (FT8 && k8 >= 0) {  // this step's byte of the block's flag tile (k8 is a constant of the unrolled step)
            uint32_t& half = (k8 & 4) ? flt_hi : flt_lo;
            half = (k8 & 3) == 0 ? fl : (half | (fl << (8 * (k8 & 3))));
        }
        const bool deferred = OUT && k8 >= 0 && defer != nullptr;
        if (deferred) {  // stored by the next step of the block, in the shadow of its look-ups
            defer->fl = fl;
            defer->lo = q_lo; defer->hi = q_hi;
        } else if (OUT && k8 >= 0) {  // a step of an unrolled block: the block's first row + this step's precomputed offset
            Pend now = {q_lo, q_hi, fl};
            flush(now, k8);  // (with the episode returns' packed add)
        } else {
            if (OUT || rew_k) rew_k[tid] = rw;
            if (OUT || flg_k) store_flag_byte(flg_k, lane, fl);
            if (OUT || rew_k) rew_k += n;
            if (OUT || flg_k) flg_k += n;
        }
        s.h0 = nh0;
        s.h1 = nh1;
        s.dcount = dcount;
        if (!(OUT && k8 >= 0)) { epsh.x += rw.z; epsh.y += rw.w; }
        step_k += 1u;
    };
    // after the eight steps of an unrolled block (FT8: first the block's flag tile — 8 bytes per env, 512 contiguous bytes
    // per wavefront)
    auto advance_rows = [&]() __attribute__((always_inline)) {
        if (FT8) {
            const uint64_t tile = ((uint64_t)flt_hi << 32) | flt_lo;
            asm volatile("global_store_dwordx2 %0, %1, %2" : : "v"(lane * 8u), "v"(tile), "s"(flg_k) : "memory");
        }
        if (OUT) { rew_k += 8 * n; flg_k += 8 * n; }
    };


    Phx4 w = {0, 0, 0, 0};  // the Philox block of the step being looked at
#define OC_JA_AT(T, FIRST)                                                                                   \
    ([&]() __attribute__((always_inline)) {                                                                  \
        const uint64_t t_ = (uint64_t)(T);                                                                   \
        if ((FIRST) || ((uint32_t)t_ & 7u) == 0u) w = philox_words(t_ >> 3, g_lo, g_hi, seed_lo, seed_hi);   \
        return joint_action_of(w.w0, w.w1, w.w2, w.w3, (uint32_t)t_ & 7u);                                                        \
    }())

    if (MODE == 1) {
        // Software pipeline over the move table: while step k resolves its interacts, the reads for step k + 1 (faced
        // cells of the next pose, the pose after that, who interacts) are already in flight.  State carried:
        //   Jc = row of the pose at step k, fa = its faced-cell offsets, off0/off1 = LUT addresses for step k's actions,
        //   Jn = row of the pose at step k + 1.
        uint32_t Jc = s.J;
        constexpr uint32_t JS = (uint32_t)CW;  // scale of a joint action as a table index: the tables' entries are u16 / u32
        auto row_rd = [&](uint32_t a) __attribute__((always_inline)) { return CW == 2 ? lds_rd16(a) : lds_rd32(a); };
        // JS * (top base-36 digit of x) = mulhi(x, 36 * JS) with the low bits cleared
        auto jsd = [&](uint32_t x) __attribute__((always_inline)) { return __umulhi(x, 36u * JS) & ~(JS - 1u); };
        const uint32_t ja0 = OC_JA_AT(t0, true) * JS;
        uint32_t fa = lds_rd32(Jc + (uint32_t)Mvj<CW>::FACES), Jn = row_rd(Jc + ja0);
        uint32_t off0 = row_rd((uint32_t)M::ACT + ja0), off1 = row_rd((uint32_t)(M::ACT + M::ACT_P1) + ja0);
        uint32_t fo0 = col + (fa & 0xFFFFu), fo1 = col + (fa >> 16);  // LDS addresses of the faced cells of step k
        uint32_t c0 = cw_rd<CW>(fo0), c1 = cw_rd<CW>(fo1);
        uint32_t pw[MAXP];
        rd_pots(pw);
        Pend pend = {0ull, 0ull, 0u};
        PhxInc inc = {0, 0, 0, 0, 0, 0};
        // One step.  On entry: Jc / Jn = rows of the poses of this step and the next, fo* = this step's faced cells, off* = the
        // LUT addresses for this step's actions, c* / pw = the cell words (PIPE).  xn = Philox word holding the NEXT step's
        // joint action (second digit when mul36), rounds = how many Philox rounds of the block after this one to run here.
        // Three regions, kept apart by scheduling barriers: the step's look-ups are issued; then everything that does not
        // depend on them (the previous step's output stores, a slice of the next Philox block, the next action and the
        // table reads it feeds) runs while they are in flight; then the interacts themselves.
        auto pstep = [&](uint32_t xn, bool mul36, int k8, int rounds) __attribute__((always_inline)) {
            if (!PIPE) {
                c0 = cw_rd<CW>(fo0);
                c1 = cw_rd<CW>(fo1);
                rd_pots(pw);
            }
            // the next step's joint action and the table reads it feeds: the next pose's faced cells, the pose after it, who interacts
            uint32_t ja2n = 0, fa_n = 0, Jnn = 0, off0n = 0, off1n = 0;
            auto next_tables = [&]() __attribute__((always_inline)) {
                ja2n = jsd(mul36 ? xn * 36u : xn);
                fa_n = lds_rd32(Jn + (uint32_t)Mvj<CW>::FACES);
                Jnn = row_rd(Jn + ja2n);
                off0n = row_rd((uint32_t)M::ACT + ja2n);
                off1n = row_rd((uint32_t)(M::ACT + M::ACT_P1) + ja2n);
            };
            if (!PIPE) {  // two or more wavefronts per SIMD hide each other's latency: no hand-made shadow, the look-ahead reads first
                next_tables();
                __builtin_amdgcn_sched_barrier(0);  // (left to itself the scheduler queues them behind the LUT reads the step waits for)
            }
            const Looked looked = look_up(off0, off1, c0, c1, pw);
            if (PIPE) {
                __builtin_amdgcn_sched_barrier(0);
                if (OUT && k8 >= 1) flush(pend, k8 - 1);
            }
#pragma unroll
            for (int r = 0; r < rounds; ++r) inc.round();
            if (PIPE) {
                next_tables();
                __builtin_amdgcn_sched_barrier(0);
            }
            uint32_t Jcn = Jn, unused = 0, unused4 = 0, nf0 = 0, nf1 = 0, nc0 = 0, nc1 = 0, npw[MAXP];
            core(fo0, fo1, off0, off1, c0, c1, ja2n, pw, Jcn, fa_n, Jnn, unused, unused4, nf0, nf1, nc0, nc1, npw, k8, looked,
                 (PIPE && k8 >= 0 && k8 < 7) ? &pend : nullptr);
            Jc = Jcn; Jn = Jnn; off0 = off0n; off1 = off1n; fo0 = nf0; fo1 = nf1;
            if (PIPE) {
                c0 = nc0; c1 = nc1;
#pragma unroll
                for (int k = 0; k < MAXP; ++k) pw[k] = npw[k];
            }
        };
        // the Philox word of global step t (w = the block it lies in) and whether t's action is the word's second digit
        auto word_of = [&](int64_t t) __attribute__((always_inline)) {
            const uint64_t t_ = (uint64_t)t;
            if (((uint32_t)t_ & 7u) == 0u) w = philox_words(t_ >> 3, g_lo, g_hi, seed_lo, seed_hi);
            const uint32_t pick = ((uint32_t)t_ & 7u) >> 1;
            return (w.w0 & (0u - (uint32_t)(pick == 0u))) | (w.w1 & (0u - (uint32_t)(pick == 1u))) |
                   (w.w2 & (0u - (uint32_t)(pick == 2u))) | (w.w3 & (0u - (uint32_t)(pick == 3u)));
        };
        int k = 0;
        const int head_end = min(n_steps, (int)((8u - ((uint32_t)t0 & 7u)) & 7u));
        // rolled steps up to the next block boundary (the last one loads the block the unrolled loop starts in)
        for (; k < head_end; ++k) { const uint32_t xn = word_of(t0 + k + 1); pstep(xn, ((t0 + k + 1) & 1) != 0, -1, 0); }
        for (; blocked_rows && n_steps - k >= 8; k += 8) {  // whole Philox blocks, unrolled; w = the block of these 8 steps
            inc.start(((uint64_t)(t0 + k) >> 3) + 1u, g_lo, g_hi, seed_lo, seed_hi);  // the next block: ten rounds over steps 0..6
            pstep(w.w0, true, 0, 2);
            pstep(w.w1, false, 1, 2);
            pstep(w.w1, true, 2, 2);
            pstep(w.w2, false, 3, 1);
            pstep(w.w2, true, 4, 1);
            pstep(w.w3, false, 5, 1);
            pstep(w.w3, true, 6, 1);
            w = inc.words();
            pstep(w.w0, false, 7, 0);  // the look-ahead digit of step 7 is the next block's first
            advance_rows();
        }
        for (; k < n_steps; ++k) { const uint32_t xn = word_of(t0 + k + 1); pstep(xn, ((t0 + k + 1) & 1) != 0, -1, 0); }  // the tail
        // joint pose -> cells / orientations
        const uint32_t NP = 4u * s_fl[JOINT_MAX_FLOOR], Jidx = (Jc - (uint32_t)M::MVJ) / (uint32_t)Mvj<CW>::ROW_BYTES, P0 = Jidx / NP, P1 = Jidx - P0 * NP;
        s.pos0 = s_fl[P0 >> 2]; s.or0 = P0 & 3u; s.pos1 = s_fl[P1 >> 2]; s.or1 = P1 & 3u;
    } else if (MODE == 2) {
        // Per-env terrain, pose one step ahead.  Carried across steps: the pose of the step about to run (P0, O0, P1, O1), the
        // LDS offsets of its two faced cells (fa, within this lane's column) and — PIPE — those cells' words and the pot words,
        // read right after the previous step's cell writes.
        // signed byte deltas of N, S, E, W; actions 4 and 5 (bytes 4, 5 of the pair {0, delta4}) move by 0: one v_perm_b32
        // picks the byte, the add sign-extends it
        auto ahead = [&](uint32_t c, uint32_t d) __attribute__((always_inline)) {
            return c + (uint32_t)(int32_t)(int8_t)(uint8_t)__builtin_amdgcn_perm(0u, delta4, d);
        };
        uint32_t P0 = s.pos0, O0 = s.or0, P1 = s.pos1, O1 = s.or1;
        uint32_t fo0 = col + ahead(P0, O0) * (BLOCK * CW), fo1 = col + ahead(P1, O1) * (BLOCK * CW);  // faced cells of the step about to run
        uint32_t c0 = 0, c1 = 0, pw[MAXP];
        if (PIPE) {
            c0 = cw_rd<CW>(fo0);
            c1 = cw_rd<CW>(fo1);
        }
        rd_pots(pw);
        constexpr uint32_t NOI = (uint32_t)(LUT4_KEYS * 16);
        auto mstep = [&](uint32_t a0, uint32_t a1, int k8) __attribute__((always_inline)) {  // the two actions of THIS step
            if (!PIPE) {
                c0 = cw_rd<CW>(fo0);
                c1 = cw_rd<CW>(fo1);
                rd_pots(pw);
            }
            const uint32_t off0 = lut_var + (a0 == 5u ? 0u : NOI), off1 = lut_var + (a1 == 5u ? 0u : NOI);
            const Looked looked = look_up(off0, off1, c0, c1, pw);
            // resolve_movement (mdp.py:1644-1727) on the static terrain: the pose of the NEXT step
            const uint32_t t0_ = ahead(P0, a0), t1_ = ahead(P1, a1);
            // (the floor bit is tested as a 32-bit value — the empty asm keeps the compiler from widening the test back to 64
            //  bits, which costs a zero register per operand and two 64-bit compares)
            uint32_t fb0 = (uint32_t)(fm >> t0_), fb1 = (uint32_t)(fm >> t1_);
            asm("" : "+v"(fb0));
            asm("" : "+v"(fb1));
            const uint32_t np0 = (fb0 & 1u) ? t0_ : P0, np1 = (fb1 & 1u) ? t1_ : P1;
            const bool collide = (np0 == np1) | ((np0 == P1) & (np1 == P0));
            uint32_t q0 = collide ? P0 : np0, q1 = collide ? P1 : np1;
            uint32_t o0 = a0 < 4u ? a0 : O0, o1 = a1 < 4u ? a1 : O1;
            uint32_t unused1 = 0, nc0 = 0, nc1 = 0, npw[MAXP];
            uint32_t nf0 = col + ahead(q0, o0) * (BLOCK * CW), nf1 = col + ahead(q1, o1) * (BLOCK * CW);  // the next pose's faced cells
            core(fo0, fo1, off0, off1, c0, c1, 0u, pw, q0, unused1, q1, o0, o1, nf0, nf1, nc0, nc1, npw, k8, looked);
            P0 = q0; P1 = q1; O0 = o0; O1 = o1; fo0 = nf0; fo1 = nf1;
            if (PIPE) {
                c0 = nc0; c1 = nc1;
#pragma unroll
                for (int k = 0; k < MAXP; ++k) pw[k] = npw[k];
            }
        };
        int k = 0;
        const int head_end = min(n_steps, (int)((8u - ((uint32_t)t0 & 7u)) & 7u));
        for (int phase = 0; phase < 2; ++phase) {
            const int upto = phase == 0 ? head_end : n_steps;
            for (; k < upto; ++k) {  // rolled steps: up to the next block boundary, and the tail
                const uint32_t ja = OC_JA_AT(t0 + k, k == 0), a0 = (ja * 43u) >> 8;
                mstep(a0, ja - 6u * a0, -1);
            }
            if (phase == 0) {
                // whole Philox blocks, unrolled: a word x holds two steps; the actions are the base-6 digits of x (and of 36 x):
                // player 0 = mulhi(x, 6), player 1 = mulhi(6 x, 6) — the joint action mulhi(x, 36) = 6 a0 + a1 taken apart
                auto xstep = [&](uint32_t x, int k8) __attribute__((always_inline)) { mstep(__umulhi(x, 6u), __umulhi(x * 6u, 6u), k8); };
                for (; blocked_rows && n_steps - k >= 8; k += 8) {
                    w = philox_words((uint64_t)(t0 + k) >> 3, g_lo, g_hi, seed_lo, seed_hi);
                    xstep(w.w0, 0); xstep(w.w0 * 36u, 1);
                    xstep(w.w1, 2); xstep(w.w1 * 36u, 3);
                    xstep(w.w2, 4); xstep(w.w2 * 36u, 5);
                    xstep(w.w3, 6); xstep(w.w3 * 36u, 7);
                    advance_rows();
                }
            }
        }
        s.pos0 = P0; s.or0 = O0; s.pos1 = P1; s.or1 = O1;
    } else {
        auto astep = [&](uint32_t a0, uint32_t a1) __attribute__((always_inline)) {
            const uint32_t f0 = step_cell(s.pos0, s.or0, delta4), f1 = two ? step_cell(s.pos1, s.or1, delta4) : f0;
            const uint32_t m0 = a0 < 4u ? step_cell(s.pos0, a0, delta4) : s.pos0;
            const uint32_t m1 = (two & (a1 < 4u)) ? step_cell(s.pos1, a1, delta4) : (two ? s.pos1 : s.pos0);
            const uint32_t fo0 = col + f0 * (BLOCK * CW), fo1 = col + f1 * (BLOCK * CW);
            const uint32_t off0 = lut_var + (a0 == OC_A_INTERACT ? 0u : (uint32_t)(LUT4_KEYS * 16));
            const uint32_t off1 = lut_var + ((two & (a1 == OC_A_INTERACT)) ? 0u : (uint32_t)(LUT4_KEYS * 16));
            const uint32_t c0 = cw_rd<CW>(fo0), c1 = cw_rd<CW>(fo1);
            const uint32_t cm0 = cw_rd<CW>(col + m0 * (BLOCK * CW)), cm1 = cw_rd<CW>(col + m1 * (BLOCK * CW));
            uint32_t pw[MAXP];
            rd_pots(pw);
            // resolve_movement (mdp.py:1644-1727): decided on the pre-step terrain, applied after the interacts
            const bool mv0 = a0 < 4u, mv1 = two & (a1 < 4u);
            const uint32_t np0 = (mv0 & (cw_kb<CW>(cm0) < 30u)) ? m0 : s.pos0, np1 = (mv1 & (cw_kb<CW>(cm1) < 30u)) ? m1 : s.pos1;
            const bool collide = two & ((np0 == np1) | ((np0 == s.pos1) & (np1 == s.pos0)));
            const uint32_t q0 = collide ? s.pos0 : np0, q1 = collide ? s.pos1 : np1;
            const uint32_t o0 = mv0 ? a0 : s.or0, o1 = mv1 ? a1 : s.or1;
            uint32_t p0 = q0, p1 = q1, d0 = o0, d1 = o1;
            uint32_t nc0 = 0, nc1 = 0, nf0 = 0, nf1 = 0, unused4 = 0, npw[MAXP];
            core(fo0, fo1, off0, off1, c0, c1, 0u, pw, p0, p1, d0, d1, unused4, nf0, nf1, nc0, nc1, npw, -1, look_up(off0, off1, c0, c1, pw));
            s.pos0 = p0; s.pos1 = p1; s.or0 = d0; s.or1 = d1;
        };
        int k = 0;
        const int head_end = min(n_steps, (int)((8u - ((uint32_t)t0 & 7u)) & 7u));
        for (int phase = 0; phase < 2; ++phase) {
            const int upto = phase == 0 ? head_end : n_steps;
            for (; k < upto; ++k) {  // rolled steps: up to the next block boundary, and the tail
                const uint32_t ja = OC_JA_AT(t0 + k, k == 0);
                const uint32_t a0 = ja / 6u;
                astep(a0, ja - 6u * a0);
            }
            if (phase == 0) {
                for (; n_steps - k >= 8; k += 8) {  // whole Philox blocks, unrolled
                    w = philox_words((uint64_t)(t0 + k) >> 3, g_lo, g_hi, seed_lo, seed_hi);
#pragma unroll
                    for (int wd = 0; wd < 4; ++wd) {
                        uint32_t x = wd == 0 ? w.w0 : wd == 1 ? w.w1 : wd == 2 ? w.w2 : w.w3;
#pragma unroll
                        for (int half = 0; half < 2; ++half) {
                            const uint32_t a0 = __umulhi(x, 6u);
                            x *= 6u;
                            const uint32_t a1 = __umulhi(x, 6u);
                            x *= 6u;
                            astep(a0, a1);
                        }
                    }
                }
            }
        }
    }
#undef OC_JA_AT
    store_env4<MAXP, CW>(C, L, st, n, e, n_obj, horizon, s, col);
    if (EV && ea.counts)
        for (int k = 0; k < N_EVENT_TYPES; ++k) ea.counts[e * N_EVENT_TYPES + k] = lds_rd32(cnt0 + (uint32_t)k * (BLOCK * 4u));
    ep.z = epsh.x; ep.w = epsh.y;
    if (ep_returns) ep_returns[e] = ep;
}
