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
              L_REW = 32, L_COOK = 48, L_VALUE = 64, L_TERRAIN = 128;
static_assert(sizeof(OcLayout) == 256, "OcLayout must be 256 bytes");

thread_local char g_err[256] = "";

// ------------------------------------------------------------------------------------------
// Layout accessors.  `base` points at one 256-byte OcLayout, either in LDS or in global memory;
// after inlining the compiler resolves the address space from the pointer's origin.
// ------------------------------------------------------------------------------------------
struct Lay {
    const uint8_t* base;
    __device__ __forceinline__ uint32_t u8(int off) const { return base[off]; }
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

// Per-lane view of the object bytes of one env, living in LDS as words[dword][lane].
template <int NOBJ>
struct ObjLds {
    uint32_t* col;  // &s_obj[0][tid]; dword w of this lane is col[w * BLOCK]
    __device__ __forceinline__ uint32_t get(uint32_t c) const {
        return reinterpret_cast<const uint8_t*>(col + (c >> 2) * BLOCK)[c & 3];
    }
    __device__ __forceinline__ void set(uint32_t c, uint32_t v) {
        reinterpret_cast<uint8_t*>(col + (c >> 2) * BLOCK)[c & 3] = (uint8_t)v;
    }
    __device__ __forceinline__ uint32_t word(int w) const { return col[w * BLOCK]; }
    __device__ __forceinline__ void set_word(int w, uint32_t v) { col[w * BLOCK] = v; }
    // any byte == OC_O_DISH in the grid?  Only counters can hold a loose dish (pots hold soups,
    // floor cells nothing), so this is get_counter_objects_dict(state)["dish"] != [] (mdp.py:2195).
    __device__ __forceinline__ bool any_dish() const {
        uint32_t acc = 0;
#pragma unroll
        for (int w = 0; w < NOBJ * 4; ++w) {
            uint32_t x = word(w) ^ 0x03030303u;                      // dish bytes become 0
            acc |= (x - 0x01010101u) & ~x & 0x80808080u;             // classic zero-byte detector
        }
        return acc != 0;
    }
};

struct Env {
    uint32_t pos0, or0, held0, pos1, or1, held1, t;
    uint32_t tk_lo, tk_hi;  // pot slots 0-3 / 4-7: cooking_tick + 1 per byte
    __device__ __forceinline__ uint32_t tick1(uint32_t slot) const {
        uint32_t w = slot < 4 ? tk_lo : tk_hi;
        return (w >> ((slot & 3) * 8)) & 0xFFu;
    }
    __device__ __forceinline__ void set_tick1(uint32_t slot, uint32_t v) {
        uint32_t sh = (slot & 3) * 8;
        uint32_t m = ~(0xFFu << sh);
        if (slot < 4) tk_lo = (tk_lo & m) | (v << sh);
        else tk_hi = (tk_hi & m) | (v << sh);
    }
    __device__ __forceinline__ void unpack(const uint4& h) {
        pos0 = h.x & 0xFF; or0 = (h.x >> 8) & 0xFF; held0 = (h.x >> 16) & 0xFF; pos1 = h.x >> 24;
        or1 = h.y & 0xFF; held1 = (h.y >> 8) & 0xFF; t = h.y >> 16;
        tk_lo = h.z; tk_hi = h.w;
    }
    __device__ __forceinline__ uint4 pack() const {
        uint4 h;
        h.x = pos0 | (or0 << 8) | (held0 << 16) | (pos1 << 24);
        h.y = or1 | (held1 << 8) | (t << 16);
        h.z = tk_lo; h.w = tk_hi;
        return h;
    }
};

// recipe index n_onion + 4*n_tomato of a soup code
__device__ __forceinline__ uint32_t recipe_idx(uint32_t soup) {
    uint32_t n = (soup >> 3) & 3u;
    uint32_t nt = __popc(soup & 7u);
    return (n - nt) + 4u * nt;
}

// orientation / motion action 0..3 (N,S,E,W) -> cell index delta for row-major cells (actions.py:12-16)
__device__ __forceinline__ int dir_delta(uint32_t d, int W) {
    return d == 0 ? -W : d == 1 ? W : d == 2 ? 1 : -1;
}

// ------------------------------------------------------------------------------------------
// One joint transition of one env.  a0/a1 in 0..5.  Rewards are accumulated into sp*/sh*.
// ------------------------------------------------------------------------------------------
template <int NOBJ>
__device__ __forceinline__ void env_step(const Lay L, ObjLds<NOBJ> obj, Env& s, int W, uint32_t a0, uint32_t a1,
                                         float& sp0, float& sp1, float& sh0, float& sh1) {
    const bool two = s.pos1 != 0xFFu;
    const uint32_t n_pots = L.n_pots();
    const bool old_dyn = L.old_dynamics();

    // pot_states, computed once before any interact (mdp.py:1439): the number of pots that are
    // ready, cooking or hold 1..2 idle ingredients (mdp.py:2199-2203; 3 idle items do not count).
    uint32_t useful_pots = 0;
    for (uint32_t k = 0; k < n_pots; ++k) {
        uint32_t o = obj.get(L.pot_cell(k));
        if (o) {
            uint32_t n = (o >> 3) & 3u;
            useful_pots += (s.tick1(k) != 0u || (n >= 1u && n < 3u)) ? 1u : 0u;
        }
    }

    // ---- resolve_interacts: player 0 fully, then player 1 (mdp.py:1446) ----
#pragma unroll
    for (int p = 0; p < 2; ++p) {
        const uint32_t a = p ? a1 : a0;
        if (a != OC_A_INTERACT || (p == 1 && !two)) continue;
        const uint32_t pos = p ? s.pos1 : s.pos0;
        const uint32_t ori = p ? s.or1 : s.or0;
        uint32_t h = p ? s.held1 : s.held0;
        const uint32_t f = pos + dir_delta(ori, W);  // facing cell; borders are never floor (mdp.py:2082-2088)
        const uint32_t tc = L.terrain(f);
        const uint32_t type = tc & 7u, slot = tc >> 3;
        const uint32_t o = obj.get(f);
        float shaped = 0.f, sparse = 0.f;
        if (type == OC_T_COUNTER) {
            if (h && !o) { obj.set(f, h); h = 0; }             // drop, mdp.py:1459-1471
            else if (!h && o) { h = o; obj.set(f, 0); }        // pick up, mdp.py:1473-1485
        } else if (type == OC_T_ONION_DISP) {
            if (!h) h = OC_O_ONION;                            // mdp.py:1487-1494
        } else if (type == OC_T_TOMATO_DISP) {
            if (!h) h = OC_O_TOMATO;                           // mdp.py:1496-1498
        } else if (type == OC_T_DISH_DISP) {
            if (!h) {                                          // mdp.py:1500-1513
                // is_dish_pickup_useful (mdp.py:2180-2204) sees the live hands (before this pickup)
                // and counters, but the stale pot_states.
                const uint32_t other_h = p ? s.held0 : s.held1;
                const uint32_t dishes_held = (two && other_h == OC_O_DISH) ? 1u : 0u;
                if (two && dishes_held < useful_pots && !obj.any_dish()) shaped += L.rew_dish();
                h = OC_O_DISH;
            }
        } else if (type == OC_T_POT) {
            const uint32_t tk = s.tick1(slot);
            if (!h) {
                // soup_to_be_cooked_at_location (mdp.py:1899-1908): idle soup with >= 1 ingredient
                if (!old_dyn && o && tk == 0u && ((o >> 3) & 3u) > 0u) s.set_tick1(slot, 1u);  // begin_cooking: tick 0
            } else if (h == OC_O_DISH) {
                if (o && tk != 0u && (tk - 1u) >= L.cook_time(recipe_idx(o))) {  // soup ready, mdp.py:1525-1539
                    h = o;
                    obj.set(f, 0);
                    s.set_tick1(slot, 0u);
                    shaped += L.rew_soup();
                }
            } else if (h == OC_O_ONION || h == OC_O_TOMATO) {  // mdp.py:1541-1568
                const uint32_t soup = o ? o : (uint32_t)OC_O_SOUP;
                const uint32_t n = (soup >> 3) & 3u;
                if (tk == 0u && n < 3u) {                      // not is_full (mdp.py:547-551)
                    const uint32_t bit = (h == OC_O_TOMATO) ? (1u << n) : 0u;
                    obj.set(f, OC_O_SOUP | ((n + 1u) << 3) | (soup & 7u) | bit);
                    h = 0;
                    shaped += L.rew_placement();
                }
            }
        } else if (type == OC_T_SERVE) {
            if (h & OC_O_SOUP) {                               // deliver_soup, mdp.py:1570-1577, 1631-1642
                sparse += L.value(recipe_idx(h));
                h = 0;
            }
        }
        if (p) { s.held1 = h; sp1 += sparse; sh1 += shaped; }
        else { s.held0 = h; sp0 += sparse; sh0 += shaped; }
    }

    // ---- resolve_movement (mdp.py:1644-1727) ----
    uint32_t np0 = s.pos0, np1 = s.pos1;
    if (a0 < 4u) {
        s.or0 = a0;                                            // orientation follows the action even when blocked
        const uint32_t c = s.pos0 + dir_delta(a0, W);
        if ((L.terrain(c) & 7u) == OC_T_FLOOR) np0 = c;
    }
    if (two && a1 < 4u) {
        s.or1 = a1;
        const uint32_t c = s.pos1 + dir_delta(a1, W);
        if ((L.terrain(c) & 7u) == OC_T_FLOOR) np1 = c;
    }
    // is_transition_collision (mdp.py:1673-1683): same target cell, or swapped cells -> nobody moves
    const bool collide = two && (np0 == np1 || (np0 == s.pos1 && np1 == s.pos0));
    if (!collide) { s.pos0 = np0; s.pos1 = np1; }

    // ---- step_environment_effects (mdp.py:1691-1703) ----
    s.t += 1u;
    for (uint32_t k = 0; k < n_pots; ++k) {
        const uint32_t o = obj.get(L.pot_cell(k));
        if (!o) continue;
        uint32_t tk = s.tick1(k);
        if (old_dyn && tk == 0u && ((o >> 3) & 3u) == 3u) tk = 1u;            // auto begin_cooking (old dynamics)
        if (tk != 0u && (tk - 1u) < L.cook_time(recipe_idx(o))) tk += 1u;     // is_cooking -> cook()
        s.set_tick1(k, tk);
    }
}

// start state of a layout (mdp.py:1297-1305, 939-950)
template <int NOBJ>
__device__ __forceinline__ void env_reset(const Lay L, ObjLds<NOBJ> obj, Env& s) {
    s.pos0 = L.u8(L_START_POS); s.pos1 = L.u8(L_START_POS + 1);
    s.or0 = L.u8(L_START_OR); s.or1 = L.u8(L_START_OR + 1);
    s.held0 = s.held1 = 0; s.t = 0; s.tk_lo = s.tk_hi = 0;
    if (s.pos1 == 0xFFu) s.or1 = 0;
#pragma unroll
    for (int w = 0; w < NOBJ * 4; ++w) obj.set_word(w, 0);
}

// Philox4x32-10 (Salmon et al., SC'11).  Same constants/rounds as oracle_philox4x32_10.
__device__ __forceinline__ void philox4x32_10(uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3, uint32_t k0,
                                              uint32_t k1, uint32_t& r0, uint32_t& r1) {
#pragma unroll
    for (int r = 0; r < 10; ++r) {
        const uint32_t hi0 = __umulhi(0xD2511F53u, c0), lo0 = 0xD2511F53u * c0;
        const uint32_t hi1 = __umulhi(0xCD9E8D57u, c2), lo1 = 0xCD9E8D57u * c2;
        const uint32_t n0 = hi1 ^ c1 ^ k0, n2 = hi0 ^ c3 ^ k1;
        c0 = n0; c1 = lo1; c2 = n2; c3 = lo0;
        k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
    }
    r0 = c0; r1 = c1;
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

template <int NOBJ>
__device__ __forceinline__ void load_env(const uint4* __restrict__ st, int64_t n, int64_t e, Env& s, ObjLds<NOBJ> obj) {
    uint4 h = st[e];
    uint4 v[NOBJ];
#pragma unroll
    for (int p = 0; p < NOBJ; ++p) v[p] = st[(int64_t)(1 + p) * n + e];
    s.unpack(h);
#pragma unroll
    for (int p = 0; p < NOBJ; ++p) {
        obj.set_word(4 * p + 0, v[p].x); obj.set_word(4 * p + 1, v[p].y);
        obj.set_word(4 * p + 2, v[p].z); obj.set_word(4 * p + 3, v[p].w);
    }
}

template <int NOBJ>
__device__ __forceinline__ void store_env(uint4* __restrict__ st, int64_t n, int64_t e, const Env& s, ObjLds<NOBJ> obj) {
    st[e] = s.pack();
#pragma unroll
    for (int p = 0; p < NOBJ; ++p) {
        uint4 v;
        v.x = obj.word(4 * p + 0); v.y = obj.word(4 * p + 1); v.z = obj.word(4 * p + 2); v.w = obj.word(4 * p + 3);
        st[(int64_t)(1 + p) * n + e] = v;
    }
}

// post-transition bookkeeping shared by k_step and k_rollout (env.py:266-267, 321-325, 387-392)
template <int NOBJ>
__device__ __forceinline__ uint32_t finish_step(const Lay L, ObjLds<NOBJ> obj, Env& s, int horizon, uint32_t options,
                                                float4 r, float4& ep) {
    ep.x += r.x; ep.y += r.y; ep.z += r.z; ep.w += r.w;
    uint32_t fl = 0;
    if ((int)s.t >= horizon) {
        fl |= OC_F_DONE;
        if (options & OC_OPT_AUTO_RESET) {
            env_reset<NOBJ>(L, obj, s);
            ep = make_float4(0.f, 0.f, 0.f, 0.f);
            fl |= OC_F_RESET;
        }
    }
    return fl;
}

// ------------------------------------------------------------------------------------------
// k_step: one transition per launch, actions supplied by the caller.
// ------------------------------------------------------------------------------------------
template <int NOBJ, bool LAY_LDS>
__global__ __launch_bounds__(BLOCK) void k_step(const OcLayout* __restrict__ g_layouts, int n_layouts,
                                                const uint16_t* __restrict__ layout_id, const uint4* st_in,
                                                uint4* st_out, const uint8_t* __restrict__ actions,
                                                float4* __restrict__ rewards, uint8_t* __restrict__ flags,
                                                float4* __restrict__ ep_returns, int64_t n, int W, int horizon,
                                                uint32_t options) {
    __shared__ uint32_t s_obj[NOBJ * 4][BLOCK];
    __shared__ uint4 s_lay[LAY_LDS ? LDS_LAYOUT_MAX * 16 : 1];
    const int64_t e = (int64_t)blockIdx.x * BLOCK + threadIdx.x;
    const bool active = e < n;
    const Lay L = stage_layouts<LAY_LDS>(g_layouts, n_layouts, layout_id, e, active, s_lay);
    if (!active) return;
    ObjLds<NOBJ> obj{&s_obj[0][threadIdx.x]};
    Env s;
    load_env<NOBJ>(st_in, n, e, s, obj);
    const uint32_t a01 = reinterpret_cast<const uint16_t*>(actions)[e];
    const uint32_t a0 = a01 & 0xFFu, a1 = a01 >> 8;
    float4 r = make_float4(0.f, 0.f, 0.f, 0.f);
    uint32_t fl;
    float4 ep = ep_returns ? ep_returns[e] : make_float4(0.f, 0.f, 0.f, 0.f);
    if (a0 > 5u || a1 > 5u) {
        fl = OC_F_BAD_ACTION;  // get_state_transition raises ValueError (mdp.py:1394-1398): leave the env untouched
    } else {
        env_step<NOBJ>(L, obj, s, W, a0, a1, r.x, r.y, r.z, r.w);
        fl = finish_step<NOBJ>(L, obj, s, horizon, options, r, ep);
    }
    store_env<NOBJ>(st_out, n, e, s, obj);
    rewards[e] = r;
    flags[e] = (uint8_t)fl;
    if (ep_returns) ep_returns[e] = ep;
}

// ------------------------------------------------------------------------------------------
// k_rollout: n_steps fused transitions, actions from Philox; state stays in registers/LDS.
// ------------------------------------------------------------------------------------------
template <int NOBJ, bool LAY_LDS>
__global__ __launch_bounds__(BLOCK) void k_rollout(const OcLayout* __restrict__ g_layouts, int n_layouts,
                                                   const uint16_t* __restrict__ layout_id, uint4* st,
                                                   float4* __restrict__ rewards, uint8_t* __restrict__ flags,
                                                   float4* __restrict__ ep_returns, int64_t n, int W, int horizon,
                                                   uint32_t options, uint32_t seed_lo, uint32_t seed_hi,
                                                   int64_t env_offset, int64_t t0, int n_steps) {
    __shared__ uint32_t s_obj[NOBJ * 4][BLOCK];
    __shared__ uint4 s_lay[LAY_LDS ? LDS_LAYOUT_MAX * 16 : 1];
    const int64_t e = (int64_t)blockIdx.x * BLOCK + threadIdx.x;
    const bool active = e < n;
    const Lay L = stage_layouts<LAY_LDS>(g_layouts, n_layouts, layout_id, e, active, s_lay);
    if (!active) return;
    ObjLds<NOBJ> obj{&s_obj[0][threadIdx.x]};
    Env s;
    load_env<NOBJ>(st, n, e, s, obj);
    float4 ep = ep_returns ? ep_returns[e] : make_float4(0.f, 0.f, 0.f, 0.f);
    const uint64_t g = (uint64_t)(env_offset + e);
    const uint32_t g_lo = (uint32_t)g, g_hi = (uint32_t)(g >> 32);
    for (int k = 0; k < n_steps; ++k) {
        const uint64_t t = (uint64_t)(t0 + k);
        uint32_t r0, r1;
        philox4x32_10((uint32_t)t, g_lo, g_hi, (uint32_t)(t >> 32), seed_lo, seed_hi, r0, r1);
        const uint32_t a0 = r0 % 6u, a1 = r1 % 6u;
        float4 r = make_float4(0.f, 0.f, 0.f, 0.f);
        env_step<NOBJ>(L, obj, s, W, a0, a1, r.x, r.y, r.z, r.w);
        const uint32_t fl = finish_step<NOBJ>(L, obj, s, horizon, options, r, ep);
        if (rewards) rewards[(int64_t)k * n + e] = r;
        if (flags) flags[(int64_t)k * n + e] = (uint8_t)fl;
    }
    store_env<NOBJ>(st, n, e, s, obj);
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
    Env s;
    s.pos0 = base[L_START_POS]; s.pos1 = base[L_START_POS + 1];
    s.or0 = base[L_START_OR]; s.or1 = s.pos1 == 0xFFu ? 0u : base[L_START_OR + 1];
    s.held0 = s.held1 = 0; s.t = 0; s.tk_lo = s.tk_hi = 0;
    st[e] = s.pack();
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

}  // namespace

extern "C" {

int oc_abi_version(void) { return OC_ABI_VERSION; }
size_t oc_layout_size(void) { return sizeof(OcLayout); }
const char* oc_last_error(void) { return g_err; }
int oc_state_planes(int width, int height) { return 1 + (width * height + 15) / 16; }

int oc_step(const OcBatch* b, const void* d_state_in, void* d_state_out, const uint8_t* d_actions, float* d_rewards,
            uint8_t* d_flags, float* d_ep_returns, int horizon, uint32_t options, void* stream) {
    int n_obj = 0;
    if (int rc = check_batch(b, &n_obj)) return rc;
    if (!d_state_in || !d_state_out || !d_actions || !d_rewards || !d_flags)
        return fail(OC_EINVAL, "oc_step: NULL state/actions/rewards/flags pointer");
    if (horizon < 1 || horizon > 65535) return fail(OC_EINVAL, "oc_step: horizon must be in 1..65535");
    if (b->n_envs == 0) return OC_OK;
    hipStream_t s = (hipStream_t)stream;
    const bool lds = b->n_layouts <= LDS_LAYOUT_MAX;
    DISPATCH_NOBJ(n_obj, {
        if (lds)
            hipLaunchKernelGGL((k_step<NOBJ, true>), dim3(grid_for(b->n_envs)), dim3(BLOCK), 0, s, b->d_layouts,
                               b->n_layouts, b->d_layout_id, (const uint4*)d_state_in, (uint4*)d_state_out, d_actions,
                               (float4*)d_rewards, d_flags, (float4*)d_ep_returns, b->n_envs, b->width, horizon,
                               options);
        else
            hipLaunchKernelGGL((k_step<NOBJ, false>), dim3(grid_for(b->n_envs)), dim3(BLOCK), 0, s, b->d_layouts,
                               b->n_layouts, b->d_layout_id, (const uint4*)d_state_in, (uint4*)d_state_out, d_actions,
                               (float4*)d_rewards, d_flags, (float4*)d_ep_returns, b->n_envs, b->width, horizon,
                               options);
    });
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
    const bool lds = b->n_layouts <= LDS_LAYOUT_MAX;
    DISPATCH_NOBJ(n_obj, {
        if (lds)
            hipLaunchKernelGGL((k_rollout<NOBJ, true>), dim3(grid_for(b->n_envs)), dim3(BLOCK), 0, s, b->d_layouts,
                               b->n_layouts, b->d_layout_id, (uint4*)d_state, (float4*)d_rewards, d_flags,
                               (float4*)d_ep_returns, b->n_envs, b->width, horizon, options, (uint32_t)seed,
                               (uint32_t)(seed >> 32), env_offset, t0, n_steps);
        else
            hipLaunchKernelGGL((k_rollout<NOBJ, false>), dim3(grid_for(b->n_envs)), dim3(BLOCK), 0, s, b->d_layouts,
                               b->n_layouts, b->d_layout_id, (uint4*)d_state, (float4*)d_rewards, d_flags,
                               (float4*)d_ep_returns, b->n_envs, b->width, horizon, options, (uint32_t)seed,
                               (uint32_t)(seed >> 32), env_offset, t0, n_steps);
    });
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
