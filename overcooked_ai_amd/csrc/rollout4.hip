// rollout4.hip — the instances of k_rollout4 (step_lut4.hpp) and k_rollout5 (step_duo5.hpp), in three translation units: this
// file is compiled with -DOC_R4_PART=0 (joint move table + event logging), 1 (per-env terrain: k_rollout5's mover / interact
// workgroups and MODE 2, the pose one step ahead in one wavefront) and 2 (MODE 0: arithmetic movement), so that a clean build runs four hipcc processes side by side (overcooked_ai_amd/build.py) instead
// of one 80-second compile.  oc_amd.hip (oc_rollout_random) decides the family and calls the unit's launcher.
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "shared.hpp"

#ifndef OC_R4_PART
#error "compile with -DOC_R4_PART=0, 1 or 2"
#endif

namespace {

#include "common.hpp"
#include "host_util.hpp"
#include "reset.hpp"
#include "step_predicate.hpp"
#include "step_table.hpp"
#include "step_lut4.hpp"
#if OC_R4_PART == 1
#include "step_duo5.hpp"
#endif

// dynamic LDS of a k_rollout4 instance: its tables + the cell words of a workgroup's 256 envs
template <bool U, int MP, bool LL, int MODE, bool OUT, bool OLD, int NF = JOINT_MAX_FLOOR, bool EV = false, bool PIPE = true,
          bool RU = false, int CW = 2, bool NOCONF = false, bool FT8 = false>
constexpr size_t lds4_bytes(size_t cell_rows) {
    // (EV: + the per-episode event counters, [N_EVENT_TYPES][BLOCK] u32 behind the cell words)
    return (size_t)Lds4<U, LL, MODE, NF, U || RU, CW>::CELLS + cell_rows * BLOCK * CW + (EV ? (size_t)N_EVENT_TYPES * BLOCK * 4 : 0);
}

// (oc_rollout_plan: the instance is named instead of launched)
#define GO4(U, MP, LL, MODE, OUT, OLD, NF, ...)                                                                     \
    do {                                                                                                            \
        const size_t smem4 = lds4_bytes<U, MP, LL, MODE, OUT, OLD, NF, ##__VA_ARGS__>(cell_rows);                   \
        if (oc_detail::g_describe) {                                                                                \
            snprintf(oc_detail::g_describe, 256, "k_rollout4<UNIFORM=" #U ", MAXP=" #MP ", LAY_LDS=" #LL ", MODE=" #MODE ", OUT=" #OUT \
                     ", OLD=" #OLD ", NF=" #NF ", " #__VA_ARGS__ "> one wavefront per 64 envs, %zu B LDS", smem4);  \
            break;                                                                                                  \
        }                                                                                                           \
        if (!want_lds(k_rollout4<U, MP, LL, MODE, OUT, OLD, NF, ##__VA_ARGS__>, smem4)) break;                      \
        hipLaunchKernelGGL((k_rollout4<U, MP, LL, MODE, OUT, OLD, NF, ##__VA_ARGS__>), grid4, block4, smem4, c.stream, b->d_layouts, \
                           b->n_layouts, b->d_layout_id, (uint4*)c.d_state, (float4*)c.d_rewards, c.d_flags,        \
                           (float4*)c.d_ep_returns, b->n_envs, b->width, c.n_obj, c.horizon, c.options,             \
                           (uint32_t)c.seed, (uint32_t)(c.seed >> 32), c.env_offset, c.t0, c.n_steps, c.sa, c.ea);  \
    } while (0)

// k_rollout5 (step_duo5.hpp): the per-env-terrain mover / interact kernel of round 6; two spare cell rows per lane
#define GO5(LL, FT8F) do { if (c.old_dyn) GO5X(LL, FT8F, true, false, false); else GO5X(LL, FT8F, false, false, false); } while (0)
#define GO5BIG(FT8F) do { if (c.old_dyn) GO5X(true, FT8F, true, true, false); else GO5X(true, FT8F, false, true, false); } while (0)
#define GO5EV(FT8F) do { if (c.old_dyn) GO5X(true, FT8F, true, false, true); else GO5X(true, FT8F, false, false, true); } while (0)
// ... for launches without output arrays (NOOUT; the flags layout does not matter there)
#define GO5N(LL, BIGF, EVF) do { if (c.old_dyn) GO5X(LL, false, true, BIGF, EVF, true); else GO5X(LL, false, false, BIGF, EVF, true); } while (0)
#define GO5X(LL, FT8F, OLDF, BIGF, EVF, ...)                                                                           \
    do {                                                                                                            \
        const size_t smem5 = rollout5_lds_bytes(LL, BIGF, EVF, c.n_obj);                                            \
        if (oc_detail::g_describe) {                                                                                \
            snprintf(oc_detail::g_describe, 256, "k_rollout5<LAY_LDS=" #LL ", FT8=" #FT8F ", OLD=" #OLDF ", BIG=" #BIGF ", EV=" #EVF \
                     "%s> mover + interact wavefronts, %d round(s), %zu B LDS", sizeof(#__VA_ARGS__) > 1 ? ", NOOUT=" #__VA_ARGS__ : "",                                       \
                     (int)((b->n_envs + (simd_count() / 4) * BLOCK - 1) / ((simd_count() / 4) * BLOCK)), smem5);    \
            break;                                                                                                  \
        }                                                                                                           \
        if (!want_lds(k_rollout5<LL, FT8F, OLDF, BIGF, EVF, ##__VA_ARGS__>, smem5)) break;                          \
        hipLaunchKernelGGL((k_rollout5<LL, FT8F, OLDF, BIGF, EVF, ##__VA_ARGS__>), grid4, dim3(2 * BLOCK), smem5, c.stream, b->d_layouts, b->n_layouts, \
                           b->d_layout_id, (uint4*)c.d_state, (float4*)c.d_rewards, c.d_flags, (float4*)c.d_ep_returns, \
                           b->n_envs, b->width, c.n_obj, c.horizon, c.options, (uint32_t)c.seed, (uint32_t)(c.seed >> 32), \
                           c.env_offset, c.t0, c.n_steps, c.sa, c.ea);                                              \
    } while (0)

#define OC_R4_PROLOGUE                                                       \
    const OcBatch* b = c.b;                                                  \
    const size_t cell_rows = (size_t)c.n_obj * 16 + 2; /* + two spare words per lane (nopot_off) */ \
    const dim3 grid4(grid_for(b->n_envs)), block4(BLOCK)

}  // namespace

namespace oc_detail {

#if OC_R4_PART == 1
// dynamic LDS of a k_rollout5 workgroup: tables, ring, the cell words of 256 envs (+ two spare rows) and, with the event log,
// N_EVENT_TYPES rows of counters (9 x 5 grids: 163 120 of the CU's 163 840 bytes)
size_t rollout5_lds_bytes(bool lay_lds, bool big, bool ev, int n_obj) {
    return (size_t)(lay_lds ? Lds5<true>::CELLS : Lds5<false>::CELLS) + ((size_t)n_obj * 16 + 2) * BLOCK * (big ? 2 : 4) +
           (ev ? (size_t)N_EVENT_TYPES * BLOCK * 4 : 0);
}
#endif

#if OC_R4_PART == 0
void launch_rollout4_joint_events(const Rollout4Call& c) {
    OC_R4_PROLOGUE;
    if (c.events) {  // event logging: the general instances (arithmetic movement, either dynamics)
        if (c.uniform && c.small) GO4(true, 2, true, 0, false, true, 0, true);
        else if (c.small) GO4(false, 2, false, 0, false, true, 0, true);  // (mixed tables: the records are read through L2)
        else GO4(false, 8, false, 0, false, true, 0, true);
        return;
    }
    // c.joint: one wavefront per SIMD (or less) on a grid of at most 64 cells where no two players can face the same cell
    // (cramped_room): 32-bit cell words and the faced cells read a step ahead; else (big batches, shared faced cells, grids
    // above 64 cells) 16-bit words without the one-step-ahead reads (see PIPE in step_lut4.hpp)
    const bool noconf = (b->batch_flags & OC_BATCH_NO_SHARED_FACES) != 0;
    if (c.tiled8) {  // (oc_rollout_random has checked that this instance serves the batch and the launch)
        GO4(true, 1, true, 1, true, false, 6, false, true, false, 4, true, true);
        return;
    }
    if (c.pipe && b->width * b->height <= 64 && noconf) GO4(true, 1, true, 1, true, false, 6, false, true, false, 4, true);
    else GO4(true, 1, true, 1, true, false, 6, false, false);
}
#elif OC_R4_PART == 1
void launch_rollout4_mode2(const Rollout4Call& c) {
    OC_R4_PROLOGUE;
    // one wavefront per SIMD or less reads the faced cells a step ahead (32-bit cell words), more do not
#define GO4M2(U, MP, LL, RUF)                                                                             \
    do {                                                                                                  \
        if (c.pipe) GO4(U, MP, LL, 2, true, false, 0, false, true, RUF, 4); else GO4(U, MP, LL, 2, true, false, 0, false, false, RUF); \
    } while (0)
    if (c.duo) {  // whole workgroups of envs, whole 8-step blocks, at most one workgroup per CU: mover + interact wavefronts
        if (c.noout) {  // no output arrays: the same kernels without their two stores
            if (c.events) GO5N(true, false, true);
            else if (b->width * b->height > 64) GO5N(true, true, false);
            else if (c.lds) GO5N(true, false, false);
            else GO5N(false, false, false);
            return;
        }
        if (c.events) { if (c.tiled8) GO5EV(true); else GO5EV(false); }  // (event counters: tables in LDS, at most 64 cells)
        else if (b->width * b->height > 64) { if (c.tiled8) GO5BIG(true); else GO5BIG(false); }  // (65..128 cells: tables in LDS only)
        else if (c.lds) { if (c.tiled8) GO5(true, true); else GO5(true, false); }
        else { if (c.tiled8) GO5(false, true); else GO5(false, false); }
        return;
    }
    if (c.tiled8) {  // OC_OPT_FLAGS_TILED8: the instances BASELINE configs[3] / [4] run (oc_rollout_random has checked the conditions)
        if (c.lds) GO4(false, 2, true, 2, true, false, 0, false, true, true, 4, false, true);               // mixed table in LDS, pipelined
        else if (c.pipe) GO4(false, 1, false, 2, true, false, 0, false, true, true, 4, false, true);        // one-pot table in HBM
        else GO4(false, 1, false, 2, true, false, 0, false, false, true, 2, false, true);
        return;
    }
    if (c.uniform) { if (b->max_pots == 1 && c.pipe) GO4(true, 1, true, 2, true, false, 0, false, true, false, 4); else GO4M2(true, 2, true, false); }
    else if (c.lds) GO4M2(false, 2, true, true);
    else if (b->max_pots == 1) GO4M2(false, 1, false, true);
    else GO4M2(false, 2, false, true);
#undef GO4M2
}
#else
void launch_rollout4_mode0(const Rollout4Call& c) {
    OC_R4_PROLOGUE;
    if (c.uniform && !c.old_dyn && c.out && c.small) GO4(true, 2, true, 0, true, false, 0);
    else if (c.uniform && c.small) GO4(true, 2, true, 0, false, true, 0);
    else if (!c.old_dyn && c.out && c.small) {  // mixed table, new dynamics, both output arrays: no per-step NULL / old-dynamics tests
        if (c.lds) GO4(false, 2, true, 0, true, false, 0); else GO4(false, 2, false, 0, true, false, 0);
    }
    else if (c.small) GO4(false, 2, false, 0, false, true, 0);  // (old dynamics / no output arrays: the records through L2)
    else GO4(false, 8, false, 0, false, true, 0);               // more than two pots: one general instance
}
#endif

}  // namespace oc_detail
