// common.hpp — constants, the OcLayout accessor, per-env working registers, Philox, layout staging
// Part of liboc_amd.so: included by each translation unit (oc_amd.hip, rollout4.hip) inside its anonymous namespace after
// shared.hpp, in this order: common, host_util, reset, step_predicate, step_table, [step_one | step_lut4], ...
#pragma once

constexpr int BLOCK = 256;
constexpr int LDS_LAYOUT_MAX = 32;  // layout tables up to this many entries are staged in LDS (8 KiB)

// byte offsets inside OcLayout (include/oc_amd.h)
constexpr int L_NCELLS = 2, L_NPOTS = 3, L_NPLAYERS = 4, L_OLDDYN = 5, L_START_POS = 8, L_START_OR = 10, L_POT_CELL = 16,
              L_PCLASS = 24, L_REW = 32, L_COOK = 48, L_VALUE = 64, L_TERRAIN = 128;
static_assert(sizeof(OcLayout) == 256, "OcLayout must be 256 bytes");

// ------------------------------------------------------------------------------------------
// Workgroup id -> block of work.  The dispatcher deals workgroup b to XCD b % 8 (MI355X: 8 XCDs, each with its own L2).
// With block = workgroup id, neighbouring pieces of every output array belong to eight different L2s; measured with
// tools/store_front.hip / store_rollout.hip (profiles/r05_store_front.txt): a streaming write of 64 KiB pieces reaches 5.5 TB/s
// that way and 6.3 TB/s when each XCD owns one contiguous eighth of the buffer (16 KiB pieces: 6.0 -> 6.6; the rollout's rows at
// two workgroups per CU: 5.3 -> 5.6).  xcd_block() is that mapping: XCD x takes blocks [x * g / 8, (x + 1) * g / 8) of a grid of
// g workgroups (grids that are not a multiple of 8 keep the identity).  Envs are independent and every random stream is keyed
// by the env's global index, so which workgroup steps which envs changes no result.  (-DOC_NO_XCD_REMAP: the identity, for A/B.)
// ------------------------------------------------------------------------------------------
__device__ __forceinline__ uint32_t xcd_block() {
#ifdef OC_NO_XCD_REMAP
    return blockIdx.x;
#else
    const uint32_t b = blockIdx.x, g = gridDim.x;
    return (g & 7u) ? b : (b & 7u) * (g >> 3) + (b >> 3);
#endif
}

// A 16-byte store of output that nobody on this GPU reads again before the launch ends, `sc1 nt` = streaming, written through.
// Pays only where the output does not fit the 256 MB MALL AND leaves in large pieces: the rollout's reward rows beside
// [step][env] flags (+7 %), k_encode's f32 observations (616 MB: +9 %); it costs 28-31 % on outputs that are rewritten in place
// inside the MALL (u8 observations, the training step) and on k_rollout_encode's per-wavefront 28 KB pieces (profiles/
// r05_ab_streaming_stores.txt).  -DOC_PLAIN_STREAM: plain stores, for A/B.
__device__ __forceinline__ void stream_store16(uint4* p, const uint4 v) {
#ifdef OC_PLAIN_STREAM
    *p = v;
#else
    typedef uint32_t oc_stream_u32x4 __attribute__((ext_vector_type(4)));
    const oc_stream_u32x4 w = {v.x, v.y, v.z, v.w};
    asm volatile("global_store_dwordx4 %0, %1, off sc1 nt" : : "v"(p), "v"(w) : "memory");
#endif
}

using oc_detail::g_err;
using oc_detail::g_lds_refused;
using oc_detail::StartArgs;
using oc_detail::EvArgs;

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
                                             const uint16_t* layout_id, int64_t e, bool active,
                                             uint4* s_lay) {
    uint32_t lid = 0;
    if (layout_id != nullptr && active) lid = layout_id[e];
    if (LAY_LDS) {
        const uint4* src = reinterpret_cast<const uint4*>(g_layouts);
        for (int i = threadIdx.x; i < n_layouts * 16; i += BLOCK) s_lay[i] = src[i];
        __syncthreads();
        return Lay{reinterpret_cast<const uint8_t*>(s_lay) + lid * 256u};
    } else {
        // the callers stage their interact LUT cooperatively just before this call and read it right after: the
        // barrier belongs to the contract whether or not the layout table itself goes through LDS
        __syncthreads();
        return Lay{reinterpret_cast<const uint8_t*>(g_layouts) + (size_t)lid * 256u};
    }
}

__device__ __forceinline__ uint32_t make_delta4(int W) {
    // signed byte deltas of N, S, E, W for row-major cells (actions.py:12-16)
    return ((uint32_t)(-W) & 0xFFu) | (((uint32_t)W & 0xFFu) << 8) | (1u << 16) | (0xFFu << 24);
}

// post-transition bookkeeping shared by k_step and k_rollout (env.py:266-267, 321-325, 387-392)
