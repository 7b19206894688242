// oc_amd.hip — MI355X (gfx950 / CDNA4) kernels and C-ABI for the batched Overcooked hot path.
//
// Layout of this translation unit (reference = HumanCompatibleAI/overcooked_ai, "mdp.py" =
// src/overcooked_ai_py/mdp/overcooked_mdp.py); the kernels live in the headers included below:
//   common.hpp          constants, OcLayout accessor, Philox4x32-10, layout staging
//   step_predicate.hpp  get_state_transition, mdp.py:1375 (interacts 1432 -> movement 1644 -> env effects 1691) with
//                       the predicate-network interact that also emits event_infos: k_step, k_rollout
//   step_table.hpp      the same transition with the table-driven interact: k_step3 (oc_step_many, grids above 64 cells)
//   step_one.hpp        one transition per launch on the wire format itself: k_step1 (oc_step)
//   step_lut4.hpp       the rollout path: key-byte cell words, 16-byte interact LUT, joint move table: k_rollout4 — compiled in
//                       rollout4.hip (three units, see shared.hpp), launched from here through oc_detail::launch_rollout4_*
//   rollout_pair.hpp    two lanes per env: k_rollout_pair
//   reset.hpp           get_standard_start_state mdp.py:1297, get_random_start_state_fn 1307: k_reset, k_reset_random
//   encode.hpp          lossless_state_encoding mdp.py:2385-2561: k_encode, k_encode_uniform
//   rollout_encode.hpp  K transitions with the observation of every step in one launch: k_rollout_encode
//   featurize.hpp       featurize_state mdp.py:2579-2898: k_featurize
//   potential.hpp       potential_function mdp.py:2920-3238: k_potential, k_potential2
//   shaping.hpp         OvercookedMultiAgent.step reward, rllib.py:306-329: k_shape_rewards
//   this file           launch dispatch and the extern "C" entry points declared in include/oc_amd.h
//
// Execution model: one lane per env, 64-lane wavefronts, 256-lane workgroups.  This is integer /
// indexing work (no MFMA).  Per-env state arrives as coalesced 16-byte planes (1 KiB per wavefront
// per plane), the object bytes of the grid are staged in LDS in [dword][lane] order — bank =
// lane % 32 whatever cell a lane touches, so divergent per-lane cell indices never conflict — and
// the compiled layout table (terrain tile, pot cells, recipe LUTs) is staged in LDS once per
// workgroup.  See DESIGN.md for the data layout and the roofline of each kernel.
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>

#include "shared.hpp"

namespace oc_detail {
__thread char g_err[256] = "";
__thread bool g_lds_refused = false;
__thread char* g_describe = nullptr;
}  // namespace oc_detail

namespace {

#include "common.hpp"
#include "host_util.hpp"
#include "reset.hpp"
#include "step_predicate.hpp"
#include "step_table.hpp"
#include "step_one.hpp"
#include "mailbox.hpp"
#include "step_server.hpp"
#include "rollout_pair.hpp"
#include "encode.hpp"
#include "rollout_encode.hpp"
#include "featurize.hpp"
#include "potential.hpp"
#include "shaping.hpp"
#include "train_obs.hpp"

// ------------------------------------------------------------------------------------------
// host side
// ------------------------------------------------------------------------------------------
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

// LDS bytes one encode workgroup may fill with output (40 KiB = 3 workgroups per CU).  The library reads no environment
// variable unless it is built with -DOC_AMD_TUNING (the knobs of the measurement scripts under tools/: OC_ENC_LDS,
// OC_STEP_NO_LEAN, OC_ROLLOUT_PIPE, OC_ROLLOUT_NO_MODE2, OC_ROLLOUT_ENCODE_WAVES).
inline int enc_lds_budget() {
#ifdef OC_AMD_TUNING
    static int v = []() { const char* e = getenv("OC_ENC_LDS"); return e ? atoi(e) : 40 * 1024; }();
    return v;
#else
    return 40 * 1024;
#endif
}
// ... for the persistent single-layout kernel: ~19 envs per group is the measured sweet spot (65 536 cramped_room envs,
// u8: 18.1 us with a 20 KB image, 20.6 us with 40 KB, 25.2 us with 12 KB; 9x5 grids: 40 KB = 17 envs is best)
inline size_t enc_uniform_budget(size_t env_bytes) {
#ifdef OC_AMD_TUNING
    static const bool forced = getenv("OC_ENC_LDS") != nullptr;
#else
    constexpr bool forced = false;
#endif
    const size_t cap = (size_t)enc_lds_budget();
    if (forced) return cap;
    const size_t want = 19 * env_bytes;
    return want < cap ? want : cap;
}

// OcStartSpec -> kernel argument; false when the spec is malformed (a message is left in g_err)
bool start_args(const OcStartSpec* sp, StartArgs* sa, const OcBatch* b = nullptr) {
    memset(sa, 0, sizeof(*sa));
    if (!sp) return true;
    if (!(sp->rnd_obj_prob_thresh >= 0.0 && sp->rnd_obj_prob_thresh <= 1.0)) return false;
    if (sp->regen_count) {  // per-episode layout re-draw: the ids must exist, be writable and in range
        if (!b || (uint64_t)sp->regen_first + sp->regen_count > (uint64_t)(b ? b->n_layouts : 0)) return false;
        if (b->n_layouts > 1) {
            if (!b->d_layout_id) return false;
            sa->regen_first = sp->regen_first;
            sa->regen_count = sp->regen_count;
            sa->layout_ids = const_cast<uint16_t*>(b->d_layout_id);
        }  // (one layout: nothing to draw)
    }
    sa->enabled = 1;
    sa->seed_lo = (uint32_t)sp->seed;
    sa->seed_hi = (uint32_t)(sp->seed >> 32);
    sa->epoch = sp->epoch;
    sa->env_offset = sp->env_offset;
    sa->thresh = (uint64_t)(sp->rnd_obj_prob_thresh * 4294967296.0);  // floor(thresh * 2^32); 1.0 -> 2^32: always
    sa->random_start_pos = sp->random_start_pos != 0;
    return true;
}

EvArgs ev_args(const OcEventSink* sink, uint64_t* d_events, uint32_t clear_on_done = 0) {
    EvArgs ea = {d_events, nullptr, nullptr, clear_on_done};
    if (sink) {
        if (sink->d_events) ea.events = sink->d_events;
        ea.counts = sink->d_counts;
        ea.counts_done = sink->d_counts_done;
    }
    return ea;
}
inline bool ev_on(const EvArgs& ea) { return ea.events || ea.counts; }

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
                 uint32_t options, hipStream_t s, const StartArgs& sa, const EvArgs& ea, int n_steps = 1) {
    const bool uniform = b->n_layouts == 1;
    const bool lds = b->n_layouts <= LDS_LAYOUT_MAX;
    const bool small = b->max_pots >= 1 && b->max_pots <= 2;
    const bool fast = (b->batch_flags & OC_BATCH_TWO_PLAYERS) != 0 && b->width * b->height <= 64;
    const size_t smem = (size_t)n_obj * 8 * BLOCK * sizeof(uint32_t);
    const dim3 grid(grid_for(b->n_envs)), block(BLOCK);
    if (!(options & OC_OPT_PREDICATE_INTERACT)) {
        // one step on a grid of at most 64 cells: the transition on the wire format itself (step_one.hpp) — in place or out of
        // place, with or without event logging
#ifdef OC_AMD_TUNING
        static const bool no_lean = getenv("OC_STEP_NO_LEAN") != nullptr;  // tuning builds: k_step3 for every oc_step
#else
        constexpr bool no_lean = false;
#endif
        if (n_steps == 1 && n_obj <= STEP1_MAX_PLANES && !no_lean) {
            const size_t smem1 = (size_t)n_obj * BLOCK * sizeof(uint4);
#define GO1(U, MP, LL)                                                                                                \
    hipLaunchKernelGGL((k_step1<U, MP, LL, EVENTS>), grid, block, smem1, s, b->d_layouts, b->n_layouts, b->d_layout_id, \
                       (uint4*)d_state_in, (uint4*)d_state_out, d_actions, (float4*)d_rewards, d_flags,               \
                       (float4*)d_ep_returns, b->n_envs, b->width, n_obj, horizon, options, sa, ea)
            if (uniform) { if (b->max_pots == 1) GO1(true, 1, true); else if (small) GO1(true, 2, true); else GO1(true, 8, true); }
            else if (lds && small) GO1(false, 2, true);
            else if (small) GO1(false, 2, false);
            else GO1(false, 8, false);  // (more than two pots on a mixed table: the general instance reads the table through L2)
#undef GO1
            return;
        }
#define GO3(U, MP, LL, F)                                                                                            \
    do {                                                                                                             \
        if (!want_lds(k_step3<U, MP, LL, F, EVENTS>, smem)) break;                                                   \
        hipLaunchKernelGGL((k_step3<U, MP, LL, F, EVENTS>), grid, block, smem, s, b->d_layouts, b->n_layouts, b->d_layout_id,   \
                           (const uint4*)d_state_in, (uint4*)d_state_out, d_actions, (float4*)d_rewards, d_flags,    \
                           (float4*)d_ep_returns, b->n_envs, b->width, n_obj, horizon, options, n_steps, sa, ea);    \
    } while (0)
        // (what is left for k_step3: oc_step_many's K transitions per launch and grids above 64 cells)
        if (uniform && fast && b->max_pots == 1) GO3(true, 1, true, true);
        else if (uniform && fast && small) GO3(true, 2, true, true);
        else if (lds && small) GO3(false, 2, true, false);
        else GO3(false, 8, false, false);
#undef GO3
        return;
    }
#define GO(U, MP, LL)                                                                                       \
    do {                                                                                                    \
        if (!want_lds(k_step<U, MP, LL, EVENTS>, smem)) break;                                              \
        hipLaunchKernelGGL((k_step<U, MP, LL, EVENTS>), grid, block, smem, s, b->d_layouts, b->n_layouts,   \
                           b->d_layout_id, (const uint4*)d_state_in, (uint4*)d_state_out, d_actions,        \
                           (float4*)d_rewards, d_flags, (float4*)d_ep_returns, ea.events, b->n_envs,        \
                           b->width, n_obj, horizon, options);                                              \
    } while (0)
    // (the predicate network is the independent second implementation: three instances cover every table)
    if (uniform) { if (small) GO(true, 2, true); else GO(true, 8, true); }
    else GO(false, 8, false);
#undef GO
}

// oc_output_stores_only: the output stores of a rollout and nothing else (include/oc_amd.h) — one store of each kind per step,
// in step order, through (row pointer of the step, lane offset) exactly as k_rollout4 addresses its rows
__global__ __launch_bounds__(BLOCK) void k_output_stores_only(float4* __restrict__ rewards, uint8_t* __restrict__ flags, int64_t n,
                                                              int n_steps) {
    const uint32_t blk = xcd_block();  // (as k_rollout4: each XCD owns a contiguous eighth of the envs)
    const int64_t e = (int64_t)blk * BLOCK + threadIdx.x;
    if (e >= n) return;
    const float4 zero4 = make_float4(0.f, 0.f, 0.f, 0.f);
    float4* rew_k = rewards + (int64_t)blk * BLOCK;                      // wave-uniform row pointers
    uint8_t* flg_k = flags ? flags + (int64_t)blk * BLOCK : nullptr;
#pragma unroll 1
    for (int k = 0; k < n_steps; ++k) {
        stream_store16(reinterpret_cast<uint4*>(rew_k + threadIdx.x), make_uint4(0u, 0u, 0u, 0u));  // (as k_rollout4 stores them beside [step][env] flags)
        if (flg_k) { flg_k[threadIdx.x] = 0; flg_k += n; }
        rew_k += n;
    }
}
// ... with the flags array tiled by 8 steps (OC_OPT_FLAGS_TILED8): per block of 8 steps eight reward rows and ONE 8-byte store
// per lane into the block's tile row, as the kernels that serve that layout write it
__global__ __launch_bounds__(BLOCK) void k_output_stores_only_tiled8(float4* __restrict__ rewards, uint2* __restrict__ flag_tiles,
                                                                     int64_t n, int n_blocks) {
    const uint32_t blk = xcd_block();
    const int64_t e = (int64_t)blk * BLOCK + threadIdx.x;
    if (e >= n) return;
    const float4 zero4 = make_float4(0.f, 0.f, 0.f, 0.f);
    float4* rew_k = rewards + (int64_t)blk * BLOCK;
    uint2* flg_k = flag_tiles + (int64_t)blk * BLOCK;
#pragma unroll 1
    for (int b = 0; b < n_blocks; ++b) {
#pragma unroll
        for (int k8 = 0; k8 < 8; ++k8) rew_k[(int64_t)k8 * n + threadIdx.x] = zero4;
        flg_k[threadIdx.x] = make_uint2(0u, 0u);
        rew_k += 8 * n;
        flg_k += n;
    }
}

}  // namespace

extern "C" {

int oc_abi_version(void) { return OC_ABI_VERSION; }
size_t oc_layout_size(void) { return sizeof(OcLayout); }
const char* oc_last_error(void) { return g_err; }
int oc_state_planes(int width, int height) { return 1 + (width * height + 15) / 16; }

int oc_batch_hints(const OcLayout* h_layouts, int n_layouts, OcBatch* batch) {
    if (!h_layouts || !batch || n_layouts < 1) return fail(OC_EINVAL, "oc_batch_hints: NULL table / batch or no layouts");
    int max_pots = 0;
    uint32_t max_free = 0;
    bool two = true, any_old = false, same_shaping = true, shared_faces = false;
    for (int i = 0; i < n_layouts; ++i) {
        const OcLayout& l = h_layouts[i];
        any_old = any_old || l.old_dynamics != 0;
        same_shaping = same_shaping && l.old_dynamics == h_layouts[0].old_dynamics &&
                       l.rew_placement_in_pot == h_layouts[0].rew_placement_in_pot &&
                       l.rew_dish_pickup == h_layouts[0].rew_dish_pickup && l.rew_soup_pickup == h_layouts[0].rew_soup_pickup;
        if (l.n_pots > OC_MAX_POTS || l.n_cells > OC_MAX_CELLS) return fail(OC_EINVAL, "oc_batch_hints: corrupt layout record");
        max_pots = l.n_pots > max_pots ? l.n_pots : max_pots;
        two = two && l.n_players == 2;
        uint32_t free_cells = 0;
        for (int c = 0; c < l.n_cells; ++c) free_cells += (l.terrain[c] & 7) == OC_T_FLOOR ? 1u : 0u;
        for (int c = 0; c < l.n_cells && l.width; ++c) {  // does some non-floor cell touch two floor cells (two players could face it)?
            if ((l.terrain[c] & 7) == OC_T_FLOOR) continue;
            const int x = c % l.width, y = c / l.width;
            int touching = 0;
            const int nb[4][2] = {{x, y - 1}, {x, y + 1}, {x + 1, y}, {x - 1, y}};
            for (int k = 0; k < 4; ++k) {
                const int nx = nb[k][0], ny = nb[k][1];
                if (nx < 0 || ny < 0 || nx >= l.width || ny >= l.height) continue;
                touching += (l.terrain[ny * l.width + nx] & 7) == OC_T_FLOOR ? 1 : 0;
            }
            shared_faces = shared_faces || touching >= 2;
        }
        max_free = free_cells > max_free ? free_cells : max_free;
    }
    batch->max_pots = max_pots;
    batch->batch_flags = (two ? OC_BATCH_TWO_PLAYERS : 0u) | (any_old ? 0u : OC_BATCH_NEW_DYNAMICS) |
                         (same_shaping ? OC_BATCH_UNIFORM_SHAPING : 0u) | (shared_faces ? 0u : OC_BATCH_NO_SHARED_FACES);
    batch->max_free_cells = max_free;
    return OC_OK;
}

int oc_step(const OcBatch* b, const void* d_state_in, void* d_state_out, const uint8_t* d_actions, float* d_rewards,
            uint8_t* d_flags, float* d_ep_returns, uint64_t* d_events, int horizon, uint32_t options,
            const OcStartSpec* start, const OcEventSink* events, void* stream) {
    int n_obj = 0;
    if (int rc = check_batch(b, &n_obj)) return rc;
    StartArgs sa;
    if (!start_args(start, &sa, b)) return fail(OC_EINVAL, "oc_step: start.rnd_obj_prob_thresh must be in [0, 1] and its regen range within the table");
    const EvArgs ea = ev_args(events, d_events);
    if (options & OC_OPT_PREDICATE_INTERACT) {
        if (start) return fail(OC_EINVAL, "oc_step: drawn start states need the table-driven kernel (no PREDICATE_INTERACT)");
        if (ea.counts) return fail(OC_EINVAL, "oc_step: event counters need the table-driven kernel (no PREDICATE_INTERACT)");
    }
    if (!d_state_in || !d_state_out || !d_actions || !d_rewards || !d_flags)
        return fail(OC_EINVAL, "oc_step: NULL state/actions/rewards/flags pointer");
    if (horizon < 1 || horizon > 65535) return fail(OC_EINVAL, "oc_step: horizon must be in 1..65535");
    if (b->n_envs == 0) return OC_OK;
    hipStream_t s = (hipStream_t)stream;
    if (ev_on(ea))
        launch_step<true>(b, n_obj, d_state_in, d_state_out, d_actions, d_rewards, d_flags, d_ep_returns, ea.events,
                          horizon, options, s, sa, ea);
    else
        launch_step<false>(b, n_obj, d_state_in, d_state_out, d_actions, d_rewards, d_flags, d_ep_returns, nullptr,
                           horizon, options, s, sa, ea);
    return check_launch("oc_step");
}

int oc_step_many(const OcBatch* b, void* d_state, const uint8_t* d_actions, float* d_rewards, uint8_t* d_flags,
                 float* d_ep_returns, int n_steps, int horizon, uint32_t options, const OcStartSpec* start,
                 const OcEventSink* events, void* stream) {
    if (n_steps < 0) return fail(OC_EINVAL, "oc_step_many: n_steps < 0");
    StartArgs sa;
    if (!start_args(start, &sa, b)) return fail(OC_EINVAL, "oc_step_many: start.rnd_obj_prob_thresh must be in [0, 1] and its regen range within the table");
    const EvArgs ea = ev_args(events, nullptr);
    if ((start || ev_on(ea)) && (options & OC_OPT_PREDICATE_INTERACT))
        return fail(OC_EINVAL, "oc_step_many: drawn start states / event logging need the table-driven kernel");
    int n_obj = 0;
    if (int rc = check_batch(b, &n_obj)) return rc;
    if (!d_state || !d_actions || !d_rewards || !d_flags)
        return fail(OC_EINVAL, "oc_step_many: NULL state/actions/rewards/flags pointer");
    if (horizon < 1 || horizon > 65535) return fail(OC_EINVAL, "oc_step_many: horizon must be in 1..65535");
    if (b->n_envs == 0 || n_steps == 0) return OC_OK;
    if (!(options & OC_OPT_PREDICATE_INTERACT)) {  // all K transitions in one launch, the envs stay on chip in between
        if (ev_on(ea))
            launch_step<true>(b, n_obj, d_state, d_state, d_actions, d_rewards, d_flags, d_ep_returns, nullptr, horizon,
                              options, (hipStream_t)stream, sa, ea, n_steps);
        else
            launch_step<false>(b, n_obj, d_state, d_state, d_actions, d_rewards, d_flags, d_ep_returns, nullptr, horizon,
                               options, (hipStream_t)stream, sa, ea, n_steps);
        return check_launch("oc_step_many");
    }
    for (int k = 0; k < n_steps; ++k) {
        const int64_t off = (int64_t)k * b->n_envs;
        if (int rc = oc_step(b, d_state, d_state, d_actions + 2 * off, d_rewards + 4 * off, d_flags + off, d_ep_returns,
                             nullptr, horizon, options, nullptr, nullptr, stream))
            return rc;
    }
    return OC_OK;
}

int oc_rollout_random(const OcBatch* b, void* d_state, float* d_rewards, uint8_t* d_flags, float* d_ep_returns,
                      int horizon, uint32_t options, uint64_t seed, int64_t env_offset, int64_t t0, int n_steps,
                      const OcStartSpec* start, const OcEventSink* events, void* stream) {
    int n_obj = 0;
    if (int rc = check_batch(b, &n_obj)) return rc;
    StartArgs sa;
    if (!start_args(start, &sa, b)) return fail(OC_EINVAL, "oc_rollout_random: start.rnd_obj_prob_thresh must be in [0, 1] and its regen range within the table");
    const EvArgs ea = ev_args(events, nullptr);
    if ((start || ev_on(ea)) && (options & (OC_OPT_LANE_PAIR | OC_OPT_PREDICATE_INTERACT)))
        return fail(OC_EINVAL, "oc_rollout_random: drawn start states / event logging need the default kernel (k_rollout4)");
    if (start && start->env_offset != env_offset) return fail(OC_EINVAL, "oc_rollout_random: start.env_offset differs from env_offset");
    if (!d_state) return fail(OC_EINVAL, "oc_rollout_random: NULL state pointer");
    if (horizon < 1 || horizon > 65535) return fail(OC_EINVAL, "oc_rollout_random: horizon must be in 1..65535");
    if (n_steps < 0 || n_steps > (1 << 30)) return fail(OC_EINVAL, "oc_rollout_random: n_steps must be in 0..2^30");
    const bool tiled8 = (options & OC_OPT_FLAGS_TILED8) != 0;
    if (tiled8) {  // the launch-shape half of the option's conditions (the batch half follows the kernel choice below)
        if (!d_rewards || !d_flags || ((uintptr_t)d_flags & 7u) != 0)
            return fail(OC_EINVAL, "oc_rollout_random: OC_OPT_FLAGS_TILED8 needs d_rewards and an 8-byte aligned d_flags");
        if ((t0 & 7) != 0 || (n_steps & 7) != 0)
            return fail(OC_EINVAL, "oc_rollout_random: OC_OPT_FLAGS_TILED8 needs t0 and n_steps to be multiples of 8");
        if (ea.events || (options & (OC_OPT_LANE_PAIR | OC_OPT_PREDICATE_INTERACT)))
            return fail(OC_EINVAL, "oc_rollout_random: OC_OPT_FLAGS_TILED8 goes with the default kernel and no per-step event masks "
                                   "(per-episode counters: the mover / interact kernel writes it)");
    }
    if (b->n_envs == 0 || n_steps == 0) return OC_OK;
    hipStream_t s = (hipStream_t)stream;
    const bool uniform = b->n_layouts == 1;
    const bool lds = b->n_layouts <= LDS_LAYOUT_MAX;
    const bool small = b->max_pots >= 1 && b->max_pots <= 2;
    // Lane pairs (two lanes per env) halve the per-wavefront instruction stream at ~1.5x the total VALU work.  They
    // used to win for batches that leave SIMDs without a wavefront (<= 32 768 envs); with the table-driven step built
    // for ILP the lane-per-env kernel is faster at every batch size (us per batched step, lane vs pair, cramped_room:
    // 0.55 vs 0.73 at 4 096 envs, 0.57 vs 0.76 at 32 768), so pairs run only when OC_OPT_LANE_PAIR asks for them.
    const bool pair_ok = small && (b->batch_flags & OC_BATCH_TWO_PLAYERS) != 0;
    const bool want_pair = (options & OC_OPT_LANE_PAIR) != 0;
    if (pair_ok && want_pair) {
        // two lanes per env (k_rollout_pair)
        if (oc_detail::g_describe) { snprintf(oc_detail::g_describe, 256, "k_rollout_pair (OC_OPT_LANE_PAIR: two lanes per env)"); return OC_OK; }
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
        // k_rollout4 (step_lut4.hpp; its instances are compiled in rollout4.hip, three units).  Which family runs:
        //   joint   one two-player, one-pot, new-dynamics layout with at most 6 free cells (cramped_room): the JOINT move table.
        //           Launches of a few steps cannot amortise the ~10 us the workgroups spend building it: they move arithmetically
        //   mode2   two players everywhere, at most two pots and 64 cells, one set of shaping rewards: per-env terrain with
        //           the pose one step ahead on a floor mask (BASELINE configs[3] / [4], single layouts with more free cells)
        //   else    arithmetic movement (MODE 0): any table, either dynamics, event logging
        oc_detail::Rollout4Call c;
        c.b = b; c.n_obj = n_obj; c.d_state = d_state; c.d_rewards = d_rewards; c.d_flags = d_flags; c.d_ep_returns = d_ep_returns;
        c.horizon = horizon; c.options = options; c.seed = seed; c.env_offset = env_offset; c.t0 = t0; c.n_steps = n_steps;
        c.sa = sa; c.ea = ea; c.stream = s;
        const bool two = (b->batch_flags & OC_BATCH_TWO_PLAYERS) != 0;
        c.uniform = uniform; c.lds = lds; c.small = small; c.events = ev_on(ea);
        c.old_dyn = (b->batch_flags & OC_BATCH_NEW_DYNAMICS) == 0;  // some layout may use old dynamics
        c.out = d_rewards != nullptr && d_flags != nullptr;
        c.noout = d_rewards == nullptr && d_flags == nullptr;  // (a rollout run for its final states / returns / event counters)
        // big batches (more than ~1.5 wavefronts per SIMD) hide latency with the other wavefronts: no one-step-ahead reads
#ifdef OC_AMD_TUNING
        static const int forced_pipe = []() { const char* e = getenv("OC_ROLLOUT_PIPE"); return e ? atoi(e) : -1; }();  // tuning builds
        static const bool no_mode2 = getenv("OC_ROLLOUT_NO_MODE2") != nullptr;
#else
        constexpr int forced_pipe = -1;
        constexpr bool no_mode2 = false;
#endif
        c.pipe = forced_pipe >= 0 ? forced_pipe != 0 : b->n_envs <= simd_count() * 64 * 3 / 2;
        c.joint = uniform && two && b->max_pots == 1 && b->max_free_cells >= 2 && b->max_free_cells <= 6u && c.out &&
                  !c.old_dyn && n_steps >= 8 && !c.events;
        const bool shaping_uniform = uniform || (b->batch_flags & OC_BATCH_UNIFORM_SHAPING) != 0;
        // what the per-env-terrain kernels serve: two players everywhere, at most two pots, 64 cells (k_rollout5: 128 with the table
        // in LDS), one set of shaping rewards, both output arrays (a one-pot joint-table layout is such a batch too)
        // (one dynamics flag for the whole table — OC_BATCH_UNIFORM_SHAPING —: old dynamics is served by k_rollout5, not by MODE 2)
        const int n_cells = b->width * b->height;
        // (an event log: per-episode counters only — no per-step masks —, table in LDS, <= 64 cells, and the counters must fit the
        //  CU's LDS beside the cell words: grids of up to 48 cells)
        const bool ev_ok = !c.events || (ea.events == nullptr && lds && n_cells <= 64 &&
                                         oc_detail::rollout5_lds_bytes(true, false, true, n_obj) <= (size_t)160 * 1024);
        const bool terrain_shape = two && small && shaping_uniform && (n_cells <= 64 || (n_cells <= 128 && lds)) && ev_ok && !no_mode2;
        const bool terrain_ok = terrain_shape && c.out;
        const bool mode2 = !c.joint && terrain_ok && !c.old_dyn && n_cells <= 64 && !c.events;
        // k_rollout5 (step_duo5.hpp): the step split between mover and interact wavefronts — whole workgroups of envs (every
        // wavefront meets every barrier) and whole 8-step blocks; a workgroup's 127-154 KB of LDS leave room for one per CU.
        // Bigger batches run these workgroups in ROUNDS, one per CU at a time — the next round's workgroups start as the first ones
        // finish their launch's steps — which keeps the one-workgroup-per-CU rate where the one-wavefront instances fall behind
        // (round 6, same box: the 5-layout mix at 65 536 / 131 072 / 262 144 envs 343 / 344 / 343 G env-steps/s; generated
        // terrains, table read through L2, 131 072 envs: 332 G in two rounds against 315 G with one wavefront per env group).
        // Beyond 8 rounds the one-wavefront instances win (round 5: 1 M cramped_room envs 356 G in 16 rounds against 384 G).
#ifdef OC_AMD_TUNING
        static const int forced_rounds = []() { const char* e = getenv("OC_DUO_ROUNDS"); return e ? atoi(e) : 0; }();  // tuning builds
#else
        constexpr int forced_rounds = 0;
#endif
        const int64_t per_round = (simd_count() / 4) * BLOCK;
        const int64_t max_rounds = forced_rounds > 0 ? forced_rounds : 8;
        const bool duo_batch = (terrain_ok || (terrain_shape && c.noout)) && !(options & OC_OPT_ONE_WAVEFRONT) && b->n_envs % BLOCK == 0 &&
                               b->n_envs <= per_round * max_rounds;
        // A long launch that is not made of whole 8-step blocks (t0 or n_steps not a multiple of 8 — e.g. every call after one
        // rollout of 150 steps): the steps up to the next block boundary and the last < 8 steps go through the one-wavefront
        // instances, the whole blocks between them through the mover / interact kernel — three launches on the stream, the same
        // results (the state lives in d_state between them, every random draw is keyed by the global step)
        if (duo_batch && !tiled8 && (((t0 & 7) != 0) || ((n_steps & 7) != 0))) {
            const int head = (int)((8 - (t0 & 7)) & 7);
            const int bulk = head < n_steps ? ((n_steps - head) & ~7) : 0;
            if (bulk >= 256) {
                const int lens[3] = {head, bulk, n_steps - head - bulk};
                int off = 0;
                for (int part = 0; part < 3; ++part) {
                    const int len = lens[part];
                    if (len > 0 && !(oc_detail::g_describe && part != 1)) {
                        OcStartSpec sk;
                        if (start) { sk = *start; sk.epoch = start->epoch + (uint32_t)off; }  // a restart at step k draws from epoch + k
                        const int rc = oc_rollout_random(b, d_state, d_rewards ? d_rewards + (int64_t)off * b->n_envs * 4 : nullptr,
                                                         d_flags ? d_flags + (int64_t)off * b->n_envs : nullptr, d_ep_returns, horizon,
                                                         options | (part == 1 ? 0u : (uint32_t)OC_OPT_ONE_WAVEFRONT), seed, env_offset, t0 + off, len,
                                                         start ? &sk : nullptr, events, stream);
                        if (rc) return rc;
                    }
                    off += len;
                }
                if (oc_detail::g_describe) {
                    const size_t used = strlen(oc_detail::g_describe);
                    snprintf(oc_detail::g_describe + used, 256 - used, "; %d + %d steps around the whole blocks: one-wavefront launches", lens[0], lens[2]);
                }
                return OC_OK;
            }
        }
        c.duo = duo_batch && n_steps >= 8 && (t0 & 7) == 0 && (n_steps & 7) == 0;
        c.tiled8 = tiled8;
        if (tiled8) {  // which instances write the tiled flags array: the pipelined joint-table one, the per-env-terrain ones of
                       // mixed tables in LDS (pipelined) and of one-pot tables in HBM
            const bool by_joint = c.joint && c.pipe && b->width * b->height <= 64 && (b->batch_flags & OC_BATCH_NO_SHARED_FACES) != 0;
            const bool by_mode2 = mode2 && !uniform && ((lds && c.pipe) || (!lds && b->max_pots == 1));
            if (!(c.duo || by_joint || by_mode2) || b->n_envs >= ((int64_t)1 << 24))
                return fail(OC_EINVAL, "oc_rollout_random: OC_OPT_FLAGS_TILED8 is served by the pipelined joint-table kernel (one two-player, "
                                       "one-pot layout with <= 6 free cells and no shared faced cells, <= ~98 000 envs) and by the per-env-"
                                       "terrain kernels of mixed tables (<= 32 layouts: <= ~98 000 envs; one-pot tables beyond that), or by the mover / interact kernel "
                                       "(whole 256-env workgroups, <= 524 288 envs)");
        }
        if (c.duo) oc_detail::launch_rollout4_mode2(c);
        else if (c.joint || c.events) oc_detail::launch_rollout4_joint_events(c);
        else if (mode2) oc_detail::launch_rollout4_mode2(c);
        else oc_detail::launch_rollout4_mode0(c);
        if (oc_detail::g_describe) return OC_OK;  // (oc_rollout_plan: nothing was launched)
        return check_launch("oc_rollout_random");
    }
    if (oc_detail::g_describe) { snprintf(oc_detail::g_describe, 256, "k_rollout (OC_OPT_PREDICATE_INTERACT: the predicate-network interact)"); return OC_OK; }
    const size_t smem = (size_t)n_obj * 8 * BLOCK * sizeof(uint32_t);
    const dim3 grid(grid_for(b->n_envs)), block(BLOCK);
#define GO(U, MP, LL)                                                                                       \
    do {                                                                                                    \
        if (!want_lds(k_rollout<U, MP, LL>, smem)) break;                                                   \
        hipLaunchKernelGGL((k_rollout<U, MP, LL>), grid, block, smem, s, b->d_layouts, b->n_layouts,        \
                           b->d_layout_id, (uint4*)d_state, (float4*)d_rewards, d_flags,                    \
                           (float4*)d_ep_returns, b->n_envs, b->width, n_obj, horizon, options,             \
                           (uint32_t)seed, (uint32_t)(seed >> 32), env_offset, t0, n_steps);                \
    } while (0)
    if (uniform) { if (small) GO(true, 2, true); else GO(true, 8, true); }
    else GO(false, 8, false);
#undef GO
    return check_launch("oc_rollout_random");
}

int oc_rollout_plan(const OcBatch* b, int horizon, uint32_t options, int64_t t0, int n_steps, int with_outputs, int event_sink,
                    const OcStartSpec* start, char* out, size_t out_size) {
    if (!out || out_size == 0) return fail(OC_EINVAL, "oc_rollout_plan: no output buffer");
    out[0] = 0;
    char buf[256] = "nothing to launch (no envs or no steps)";
    // the call is made with stand-in pointers (the launch sites name their instance and launch nothing): every check of
    // oc_rollout_random applies, every branch of its dispatch is the one a real call takes
    void* const fake = reinterpret_cast<void*>((uintptr_t)4096);
    OcEventSink sink;
    sink.d_events = event_sink == 2 ? reinterpret_cast<uint64_t*>(fake) : nullptr;
    sink.d_counts = event_sink >= 1 ? reinterpret_cast<uint32_t*>(fake) : nullptr;
    sink.d_counts_done = nullptr;
    oc_detail::g_describe = buf;
    const int rc = oc_rollout_random(b, fake, with_outputs ? reinterpret_cast<float*>(fake) : nullptr,
                                     with_outputs ? reinterpret_cast<uint8_t*>(fake) : nullptr, nullptr, horizon, options, 0, start ? start->env_offset : 0,
                                     t0, n_steps, start, event_sink ? &sink : nullptr, nullptr);
    oc_detail::g_describe = nullptr;
    if (rc != OC_OK) return rc;
    snprintf(out, out_size, "%s", buf);
    return OC_OK;
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
        if (!want_lds(k_featurize<true>, smem)) return check_launch("oc_featurize");
        hipLaunchKernelGGL((k_featurize<true>), grid, block, smem, s, b->d_layouts, b->n_layouts, b->d_layout_id, d_plan_blob,
                           d_plan_off, (const uint4*)d_state, d_features, b->n_envs, b->width, b->height, n_planes, num_pots);
    } else {
        if (!want_lds(k_featurize<false>, smem)) return check_launch("oc_featurize");
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

int oc_multi_agent_step(const OcBatch* b, void* d_state, const uint8_t* d_actions, float* d_rewards, uint8_t* d_flags,
                        float* d_ep_returns, float* d_ep_returns_out, const uint8_t* d_plan_blob,
                        const uint32_t* d_plan_off, const uint8_t* d_phi_tables, double* d_phi_next, double* d_phi_cur,
                        const double* d_phi_start, double reward_shaping_factor, double* d_shaped, uint8_t* d_done,
                        void* d_obs, int obs_dtype, int horizon, const OcStartSpec* start, const OcEventSink* events,
                        void* stream) {
    if (!d_done) return fail(OC_EINVAL, "oc_multi_agent_step: d_done is required (it is the reset mask)");
    StartArgs sa;
    if (!start_args(start, &sa, b)) return fail(OC_EINVAL, "oc_multi_agent_step: start.rnd_obj_prob_thresh must be in [0, 1] and its regen range within the table");
    const EvArgs ea = ev_args(events, nullptr, 1u);
    {
        int n_obj = 0;
        if (int rc = check_batch(b, &n_obj)) return rc;
        const bool fused = b->max_pots >= 1 && b->max_pots <= 2 && (b->batch_flags & OC_BATCH_TWO_PLAYERS) != 0;
        if (fused) {  // the whole step in one kernel (k_train_step), then the observation
            if (!d_state || !d_actions || !d_rewards || !d_flags || !d_shaped)
                return fail(OC_EINVAL, "oc_multi_agent_step: NULL state/actions/rewards/flags/shaped pointer");
            if (d_phi_tables && (!d_plan_blob || !d_plan_off || !d_phi_next || !d_phi_cur || !d_phi_start))
                return fail(OC_EINVAL, "oc_multi_agent_step: use_phi needs the plan tables and the three phi buffers");
            if (horizon < 1 || horizon > 65535) return fail(OC_EINVAL, "oc_multi_agent_step: horizon must be in 1..65535");
            if (((uintptr_t)d_shaped & 15u) != 0) return fail(OC_EINVAL, "oc_multi_agent_step: d_shaped must be 16-byte aligned");
            if (b->n_envs > 0) {
                const bool uniform = b->n_layouts == 1, lds = b->n_layouts <= LDS_LAYOUT_MAX;
                const bool fast = b->width * b->height <= 64;
                const size_t smem = (size_t)n_obj * 16 * BLOCK * sizeof(uint16_t);
                const dim3 grid(grid_for(b->n_envs)), block(BLOCK);
#define GOT(U, MP, LL, F)                                                                                             \
    do {                                                                                                              \
        if (ev_on(ea)) { GOT_(U, MP, LL, F, true); } else { GOT_(U, MP, LL, F, false); }                              \
    } while (0)
#define GOT_(U, MP, LL, F, EV)                                                                                        \
    do {                                                                                                              \
        if (!want_lds(k_train_step<U, MP, LL, F, EV>, smem)) break;                                                   \
        hipLaunchKernelGGL((k_train_step<U, MP, LL, F, EV>), grid, block, smem, (hipStream_t)stream, b->d_layouts,    \
                           b->n_layouts, b->d_layout_id, (uint4*)d_state, d_actions, (float4*)d_rewards, d_flags,     \
                           (float4*)d_ep_returns, (float4*)d_ep_returns_out, d_plan_blob, d_plan_off, d_phi_tables,   \
                           d_phi_next, d_phi_cur, d_phi_start, reward_shaping_factor, d_shaped, d_done, b->n_envs,    \
                           b->width, b->height, n_obj, horizon, sa, ea);                                              \
    } while (0)
#ifdef OC_AMD_TUNING
                static const bool no_lean = getenv("OC_STEP_NO_LEAN") != nullptr;  // tuning builds: k_train_step always
#else
                constexpr bool no_lean = false;
#endif
                // Round 5: the step AND its observation in one kernel (train_obs.hpp) for single-layout batches that give at
                // least half of the CUs a workgroup of 256 envs (smaller batches: the observation kernel below spreads over all CUs)
#ifdef OC_AMD_TUNING
                static const bool no_fused_obs = getenv("OC_TRAIN_NO_FUSED_OBS") != nullptr;
#else
                constexpr bool no_fused_obs = false;
#endif
                if (d_obs && uniform && fast && !ev_on(ea) && n_obj <= STEP1_MAX_PLANES && !no_lean && !no_fused_obs &&
                    (obs_dtype == OC_OBS_U8 || obs_dtype == OC_OBS_F32) && ((uintptr_t)d_obs & 15u) == 0 &&
                    b->n_envs >= (simd_count() / 8) * BLOCK) {
                    const size_t elem = obs_dtype == OC_OBS_U8 ? 1 : 4;
                    const size_t env_bytes = (size_t)2 * b->width * b->height * OC_NUM_LAYERS * elem;
                    int unit = 1;
                    while (((env_bytes * unit) & 15u) != 0) unit *= 2;  // 1, 2 or 4 envs per template
                    const size_t fixed = (size_t)n_obj * BLOCK * 16 + env_bytes * unit + (size_t)4 * BLOCK * 16;
                    const size_t budget = 150 * 1024;
                    // wavefronts per workgroup: 16 (4 owners, 4 helpers, 8 encoders; round 6) for u8 observations whose private images
                    // still hold >= 6 envs then (cramped_room-sized grids: the encode loop there is bound by what the wavefronts of a
                    // CU can issue, not by bytes), else 8
#ifdef OC_AMD_TUNING
                    static const int forced_w = []() { const char* e = getenv("OC_TRAIN_OBS_WAVES"); return e ? atoi(e) : 0; }();
                    static const int forced_g = []() { const char* e = getenv("OC_TRAIN_OBS_G"); return e ? atoi(e) : 0; }();
#else
                    constexpr int forced_w = 0, forced_g = 0;
#endif
                    int nwv = 0, gmax = 0;
                    for (int w : {16, 8}) {
                        if (forced_w && w != forced_w) continue;
                        if (!forced_w && w == 16 && obs_dtype != OC_OBS_U8) continue;
                        int g = fixed < budget ? (int)((budget - fixed) / ((size_t)w * env_bytes)) : 0;
                        if (g > 64) g = 64;
                        if (forced_g > 0 && forced_g < g) g = forced_g;
                        g -= g % unit;
                        if (w == 16 && g < 6 && !forced_w) continue;  // (measured: 7-env images 22.0 -> 19.5 us, 3-env images 31.8 -> 37.6)
                        if (g >= unit && g >= 2) { nwv = w; gmax = g; break; }
                    }
                    if (nwv) {  // (one env per image — 9x5 f32 — measured slower than the two kernels: 123-127 vs 116-122 us)
                        const size_t smem_o = fixed + (size_t)nwv * gmax * env_bytes;
#define GOTO(MP, T, NW)                                                                                                 \
    do {                                                                                                                \
        if (!want_lds(k_train_step_obs<MP, T, NW>, smem_o)) break;                                                      \
        hipLaunchKernelGGL((k_train_step_obs<MP, T, NW>), grid, dim3(NW * 64), smem_o, (hipStream_t)stream, b->d_layouts, \
                           (uint4*)d_state, d_actions, (float4*)d_rewards, d_flags, (float4*)d_ep_returns,              \
                           (float4*)d_ep_returns_out, d_plan_blob, d_plan_off, d_phi_tables, d_phi_next, d_phi_cur,     \
                           d_phi_start, reward_shaping_factor, d_shaped, d_done, (uint8_t*)d_obs, b->n_envs, b->width,  \
                           b->height, n_obj, horizon, unit, gmax, sa);                                                  \
    } while (0)
#define GOTOW(MP, T) do { if (nwv == 16) GOTO(MP, T, 16); else GOTO(MP, T, 8); } while (0)
                        if (obs_dtype == OC_OBS_U8) { if (b->max_pots == 1) GOTOW(1, uint8_t); else GOTOW(2, uint8_t); }
                        else { if (b->max_pots == 1) GOTOW(1, float); else GOTOW(2, float); }
#undef GOTOW
#undef GOTO
                        return check_launch("oc_multi_agent_step");
                    }
                }
                if (!ev_on(ea) && n_obj <= STEP1_MAX_PLANES && !no_lean) {  // the transition on the wire format itself (step_one.hpp)
                    const size_t smem1 = (size_t)n_obj * BLOCK * sizeof(uint4);
#define GOT1(U, MP, LL)                                                                                                \
    hipLaunchKernelGGL((k_train_step1<U, MP, LL>), grid, block, smem1, (hipStream_t)stream, b->d_layouts, b->n_layouts, \
                       b->d_layout_id, (uint4*)d_state, d_actions, (float4*)d_rewards, d_flags, (float4*)d_ep_returns,  \
                       (float4*)d_ep_returns_out, d_plan_blob, d_plan_off, d_phi_tables, d_phi_next, d_phi_cur,         \
                       d_phi_start, reward_shaping_factor, d_shaped, d_done, b->n_envs, b->width, b->height, n_obj,     \
                       horizon, sa)
                    if (uniform && b->max_pots == 1) GOT1(true, 1, true);
                    else if (uniform) GOT1(true, 2, true);
                    else if (lds) GOT1(false, 2, true);
                    else GOT1(false, 2, false);
#undef GOT1
                }
                // (k_train_step: event counters attached, or a grid above 64 cells)
                else if (uniform) GOT(true, 2, true, false);
                else GOT(false, 2, false, false);
#undef GOT
#undef GOT_
                if (int rc = check_launch("oc_multi_agent_step")) return rc;
            }
            if (d_obs) return oc_encode_lossless(b, d_state, d_obs, obs_dtype, horizon, stream);
            return OC_OK;
        }
    }
    {  // the general sequence: oc_step (finished envs are restarted below: their counters clear at the DONE step)
        int n_obj = 0;
        if (int rc = check_batch(b, &n_obj)) return rc;
        if (!d_state || !d_actions || !d_rewards || !d_flags) return fail(OC_EINVAL, "oc_multi_agent_step: NULL state/actions/rewards/flags pointer");
        if (horizon < 1 || horizon > 65535) return fail(OC_EINVAL, "oc_multi_agent_step: horizon must be in 1..65535");
        if (b->n_envs > 0) {
            StartArgs none;
            start_args(nullptr, &none);
            if (ev_on(ea))
                launch_step<true>(b, n_obj, d_state, d_state, d_actions, d_rewards, d_flags, d_ep_returns, nullptr, horizon, 0u,
                                  (hipStream_t)stream, none, ea);
            else
                launch_step<false>(b, n_obj, d_state, d_state, d_actions, d_rewards, d_flags, d_ep_returns, nullptr, horizon, 0u,
                                   (hipStream_t)stream, none, ea);
            if (int rc = check_launch("oc_multi_agent_step")) return rc;
        }
    }
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
    if (start) {  // finished envs restart from drawn states; d_phi_cur = the potential of what every env starts the next step from
        if (start->regen_count && b->n_layouts > 1) {  // ... on layouts drawn for their new episodes
            if (int rc = oc_regen_layouts(b, const_cast<uint16_t*>(b->d_layout_id), d_done, 0xFF, start, stream)) return rc;
        }
        if (int rc = oc_reset_random(b, d_state, d_done, d_ep_returns, start->seed, start->env_offset, start->epoch,
                                     start->random_start_pos, start->rnd_obj_prob_thresh, stream))
            return rc;
        if (d_phi_tables) {
            if (int rc = oc_potential(b, d_plan_blob, d_plan_off, d_phi_tables, d_state, d_phi_cur, stream)) return rc;
        }
    } else {
        if (int rc = oc_reset(b, d_state, d_done, d_ep_returns, stream)) return rc;
    }
    if (d_obs) return oc_encode_lossless(b, d_state, d_obs, obs_dtype, horizon, stream);
    return OC_OK;
}

int oc_regen_layouts(const OcBatch* b, uint16_t* d_layout_id, const uint8_t* d_mask, uint8_t mask_bits, const OcStartSpec* start,
                     void* stream) {
    int n_obj = 0;
    if (int rc = check_batch(b, &n_obj)) return rc;
    if (!d_layout_id || !start || !start->regen_count) return fail(OC_EINVAL, "oc_regen_layouts: needs d_layout_id and a start spec with regen_count > 0");
    if ((uint64_t)start->regen_first + start->regen_count > (uint64_t)b->n_layouts)
        return fail(OC_EINVAL, "oc_regen_layouts: regen_first + regen_count exceeds the layout table");
    if (b->n_envs == 0) return OC_OK;
    StartArgs sa;
    memset(&sa, 0, sizeof(sa));
    sa.enabled = 1; sa.seed_lo = (uint32_t)start->seed; sa.seed_hi = (uint32_t)(start->seed >> 32); sa.epoch = start->epoch;
    sa.env_offset = start->env_offset; sa.regen_first = start->regen_first; sa.regen_count = start->regen_count;
    sa.layout_ids = d_layout_id;
    hipLaunchKernelGGL(k_regen_layouts, dim3(grid_for(b->n_envs)), dim3(BLOCK), 0, (hipStream_t)stream, d_layout_id, d_mask,
                       mask_bits, b->n_envs, sa);
    return check_launch("oc_regen_layouts");
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
    // single layout + u8: the persistent template kernel (27.4 vs 32.4 us for the generic kernel on 65 536
    // asymmetric_advantages envs).  f32 is HBM-write bound either way: through the template kernel 5x4 grids gain when
    // encodes run back to back (43.5 vs 54.4 us) but not inside a training loop (43.5 vs 41.9 us), 9x5 is 112 us both ways
    if (b->n_layouts == 1 && obs_dtype == OC_OBS_U8) {
        int unit = 1;
        while (((env_bytes * unit) & 15u) != 0) unit *= 2;              // 1, 2 or 4 envs per template
        const size_t unit_bytes = env_bytes * unit;
        int upg = (int)(enc_uniform_budget(env_bytes) / unit_bytes);     // units per group
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
            if (!want_lds(k_encode_uniform<uint8_t>, smem_u)) return check_launch("oc_encode_lossless");
            hipLaunchKernelGGL((k_encode_uniform<uint8_t>), dim3((unsigned)grid_u), dim3(BLOCK), smem_u, s, b->d_layouts,
                               (const uint4*)d_state, (uint8_t*)d_obs, b->n_envs, b->width, b->height, n_planes, unit, upg, horizon);
            return check_launch("oc_encode_lossless");
        }
    }
    const unsigned grid = (unsigned)((b->n_envs + epb - 1) / epb);
    const bool lds = b->n_layouts <= LDS_LAYOUT_MAX;
#define LAUNCH_ENC(T, LDSFLAG)                                                                                        \
    do {                                                                                                              \
        if (!want_lds(k_encode<T, LDSFLAG>, smem)) break;                                                             \
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

int oc_step_encode(const OcBatch* b, void* d_state, const uint8_t* d_actions, float* d_rewards, uint8_t* d_flags,
                   float* d_ep_returns, void* d_obs, int obs_dtype, int horizon, uint32_t options,
                   const OcStartSpec* start, void* stream) {
    int n_obj = 0;
    if (int rc = check_batch(b, &n_obj)) return rc;
    if (!d_state || !d_actions || !d_rewards || !d_flags || !d_obs) return fail(OC_EINVAL, "oc_step_encode: NULL pointer");
    if (obs_dtype != OC_OBS_U8 && obs_dtype != OC_OBS_F32) return fail(OC_EINVAL, "oc_step_encode: bad obs_dtype");
    if (((uintptr_t)d_obs & 15u) != 0) return fail(OC_EINVAL, "oc_step_encode: d_obs must be 16-byte aligned");
    if (horizon < 1 || horizon > 65535) return fail(OC_EINVAL, "oc_step_encode: horizon must be in 1..65535");
    StartArgs sa;
    if (!start_args(start, &sa, b)) return fail(OC_EINVAL, "oc_step_encode: start.rnd_obj_prob_thresh must be in [0, 1] and its regen range within the table");
    if (b->n_envs == 0) return OC_OK;
    if (!start && !(options & ~(uint32_t)(OC_OPT_AUTO_RESET | OC_OPT_ONE_KERNEL)))  // one kernel where that applies
        return oc_rollout_encode(b, d_state, d_actions, d_rewards, d_flags, d_ep_returns, d_obs, obs_dtype, 0, horizon, options,
                                 0, 0, 0, 1, nullptr, stream);
    if (int rc = oc_step(b, d_state, d_state, d_actions, d_rewards, d_flags, d_ep_returns, nullptr, horizon,
                         options & ~(uint32_t)OC_OPT_ONE_KERNEL, start, nullptr, stream))
        return rc;
    return oc_encode_lossless(b, d_state, d_obs, obs_dtype, horizon, stream);
}

int oc_rollout_encode(const OcBatch* b, void* d_state, const uint8_t* d_actions, float* d_rewards, uint8_t* d_flags,
                      float* d_ep_returns, void* d_obs, int obs_dtype, int64_t obs_step_stride, int horizon,
                      uint32_t options, uint64_t seed, int64_t env_offset, int64_t t0, int n_steps,
                      const OcStartSpec* start, void* stream) {
    int n_obj = 0;
    if (int rc = check_batch(b, &n_obj)) return rc;
    StartArgs sa;
    if (!start_args(start, &sa, b)) return fail(OC_EINVAL, "oc_rollout_encode: start.rnd_obj_prob_thresh must be in [0, 1] and its regen range within the table");
    if (!d_state || !d_obs) return fail(OC_EINVAL, "oc_rollout_encode: NULL state / observation pointer");
    if (obs_dtype != OC_OBS_U8 && obs_dtype != OC_OBS_F32) return fail(OC_EINVAL, "oc_rollout_encode: bad obs_dtype");
    if (((uintptr_t)d_obs & 15u) != 0 || obs_step_stride < 0 || (obs_step_stride & 15) != 0)
        return fail(OC_EINVAL, "oc_rollout_encode: d_obs and obs_step_stride must be multiples of 16 bytes");
    if (horizon < 1 || horizon > 65535) return fail(OC_EINVAL, "oc_rollout_encode: horizon must be in 1..65535");
    if (n_steps < 0 || n_steps > (1 << 30)) return fail(OC_EINVAL, "oc_rollout_encode: n_steps must be in 0..2^30");
    if (options & ~(uint32_t)(OC_OPT_AUTO_RESET | OC_OPT_ONE_KERNEL))
        return fail(OC_EINVAL, "oc_rollout_encode: options other than OC_OPT_AUTO_RESET / OC_OPT_ONE_KERNEL");
    if (d_actions && (!d_rewards || !d_flags)) return fail(OC_EINVAL, "oc_rollout_encode: caller actions need the rewards and flags arrays");
    if (start && start->env_offset != env_offset)  // (both paths below: the one-step fallback would refuse it, the single kernel must too)
        return fail(OC_EINVAL, "oc_rollout_encode: start.env_offset differs from env_offset");
    if (b->n_envs == 0 || n_steps == 0) return OC_OK;
    hipStream_t s = (hipStream_t)stream;
    // one layout, u8 observations, at most two pots: the whole trajectory in one launch (k_rollout_encode).  The LDS of
    // a workgroup holds the cell words of its 256 envs, the template, the headers and one image per wavefront.
    const int cells = b->width * b->height;
    const size_t env_bytes = (size_t)2 * cells * OC_NUM_LAYERS * (obs_dtype == OC_OBS_U8 ? 1 : 4);
    // It keeps 256 envs per CU on chip and is bound by what one CU's four wavefronts can encode per step (~27 us for
    // 9x5), so it pays once every CU has a workgroup: 30 us vs 37 us per step at 65 536 envs, but 27 us vs 18 us at 16 384
    // (a single step is a wash against the two one-step kernels — 36.4 vs 37.2 us on 9x5, 25.1 vs 24.4 us on 5x4 — and
    // stays with them unless OC_OPT_ONE_KERNEL asks)
    const bool fills_gpu = b->n_envs >= (simd_count() / 4) * 192 && n_steps >= 2;
    const uint32_t step_options = options & (uint32_t)OC_OPT_AUTO_RESET;
    if ((fills_gpu || (options & OC_OPT_ONE_KERNEL)) && b->n_layouts == 1 && b->max_pots >= 1 && b->max_pots <= 2 && n_obj <= 3) {
        int unit = 1;
        while (((env_bytes * unit) & 15u) != 0) unit *= 2;  // 1, 2 or 4 envs per template
        const size_t cell_bytes = (size_t)n_obj * 16 * BLOCK * sizeof(uint16_t);
        const size_t fixed = cell_bytes + env_bytes * unit + (size_t)BLOCK * 16 + RE_LIST_BYTES;
        const bool fast = (b->batch_flags & OC_BATCH_TWO_PLAYERS) != 0 && cells <= 64;
        // what a workgroup may ask for on top of the kernel's static LDS (160 KiB per CU)
        auto dynamic_lds = [](const void* kernel) {
            hipFuncAttributes fa;
            if (hipFuncGetAttributes(&fa, kernel) != hipSuccess) { (void)hipGetLastError(); return (size_t)(144 * 1024); }
            return (size_t)160 * 1024 - (size_t)fa.sharedSizeBytes - 64;  // (the eight-wavefront instances keep 32 bytes more)
        };
        const size_t budget = obs_dtype == OC_OBS_U8
            ? (fast ? dynamic_lds((const void*)k_rollout_encode<2, 3, uint8_t, 4>) : dynamic_lds((const void*)k_rollout_encode<2, 0, uint8_t, 4>))
            : (fast ? dynamic_lds((const void*)k_rollout_encode<2, 3, float, 4>) : dynamic_lds((const void*)k_rollout_encode<2, 0, float, 4>));
        // eight wavefronts (four of them helpers that only encode) when eight images of at least 8 envs fit: small grids,
        // where four wavefronts cannot encode 256 envs in the time HBM takes them (5x4 u8: 14.4 vs 17.4 us per step); 9x5
        // is at the write ceiling either way (30.1 vs 30.4 us), f32 loses with one-env images (128 vs 117 us)
#ifdef OC_AMD_TUNING
        static const int forced_nw = []() { const char* e = getenv("OC_ROLLOUT_ENCODE_WAVES"); return e ? atoi(e) : 0; }();  // tuning builds
#else
        constexpr int forced_nw = 0;
#endif
        // round 6: u8 observations take eight wavefronts down to 4-env images — a wavefront that is issuing its image's stores into a
        // busy store path is not building the next one, and eight of them leave the path idle less often (65 536 envs, us per step,
        // four vs eight: 9x5 29.2 -> 27.6-28.3, 8x5 26.1 -> 24.2, 5x5 16.1 -> 15.4; profiles/r06_rollout_encode_ablation.txt)
        const int min_g8 = obs_dtype == OC_OBS_U8 ? 4 : 8;
        int nw = 8;
        int gmax = fixed < budget ? (int)((budget - fixed) / (nw * env_bytes)) : 0;
        if (((gmax < min_g8 || gmax < unit) && forced_nw != 8) || forced_nw == 4) {
            nw = 4;
            gmax = fixed < budget ? (int)((budget - fixed) / (nw * env_bytes)) : 0;
        }
        if (gmax > 64) gmax = 64;
        if (gmax >= unit) {
            const int span = nw == 8 ? 32 : 64;                  // envs one wavefront encodes per step
            const int parts = (span + gmax - 1) / gmax;          // its sub-groups, as even as the budget allows
            int g = (span + parts - 1) / parts;
            g = (g + unit - 1) / unit * unit;
            if (g > gmax) g = gmax / unit * unit;
#ifdef OC_AMD_TUNING
            static const int forced_reg = []() { const char* e = getenv("OC_ROLLOUT_ENCODE_G"); return e ? atoi(e) : 0; }();  // tuning builds
            if (forced_reg > 0 && forced_reg <= g && forced_reg % unit == 0) g = forced_reg;
#endif
            const size_t smem = fixed + (size_t)nw * g * env_bytes;
            const dim3 grid(grid_for(b->n_envs));
#define GORE(FAST, T, NW)                                                                                              \
    do {                                                                                                               \
        if (!want_lds(k_rollout_encode<2, FAST, T, NW>, smem)) break;                                                  \
        hipLaunchKernelGGL((k_rollout_encode<2, FAST, T, NW>), grid, dim3(NW * 64), smem, s, b->d_layouts,             \
                           (uint4*)d_state, d_actions, (float4*)d_rewards, d_flags, (float4*)d_ep_returns,             \
                           (uint8_t*)d_obs, obs_step_stride, b->n_envs, b->width, b->height, n_obj, horizon,           \
                           step_options, (uint32_t)seed, (uint32_t)(seed >> 32), env_offset, t0, n_steps, unit, g, sa); \
    } while (0)
#define GORE_T(FAST, NW)                                                                                               \
    do {                                                                                                               \
        if (obs_dtype == OC_OBS_U8) GORE(FAST, uint8_t, NW); else GORE(FAST, float, NW);                               \
    } while (0)
            if (nw == 8) { if (fast) GORE_T(3, 8); else GORE_T(0, 8); }
            else { if (fast) GORE_T(3, 4); else GORE_T(0, 4); }
#undef GORE_T
#undef GORE
            return check_launch("oc_rollout_encode");
        }
    }
    // every other table: the same result from the one-step kernels, step by step
    for (int k = 0; k < n_steps; ++k) {
        const int64_t off = (int64_t)k * b->n_envs;
        OcStartSpec sk;
        if (start) { sk = *start; sk.epoch = start->epoch + (uint32_t)k; }  // a restart at step k draws from epoch + k
        const OcStartSpec* spk = start ? &sk : nullptr;
        int rc;
        if (d_actions)
            rc = oc_step(b, d_state, d_state, d_actions + 2 * off, d_rewards + 4 * off, d_flags + off, d_ep_returns, nullptr,
                         horizon, step_options, spk, nullptr, stream);
        else
            rc = oc_rollout_random(b, d_state, d_rewards ? d_rewards + 4 * off : nullptr, d_flags ? d_flags + off : nullptr,
                                   d_ep_returns, horizon, step_options, seed, env_offset, t0 + k, 1, spk, nullptr, stream);
        if (rc) return rc;
        if ((rc = oc_encode_lossless(b, d_state, (uint8_t*)d_obs + (int64_t)k * obs_step_stride, obs_dtype, horizon, stream))) return rc;
    }
    return OC_OK;
}

// ---- the single-env mailbox (mailbox.hpp)
}  // extern "C"

// ---- the resident batched step (step_server.hpp)
struct OcStepServer {
    OcBatch b;
    int n_obj, horizon, device;
    uint32_t options;
    StartArgs sa;
    void* d_state;
    float* d_ep_returns;
    hipStream_t stream;      // the resident kernel's own stream
    hipStream_t ctl_stream;  // posts that must overtake it (SV_STOP)
    hipEvent_t ev0, ev1;
    uint64_t* d_req;         // [n_envs] request granules
    uint4* d_rsp;            // [n_envs][2] response granules
    uint32_t* h_ctl;         // [4 grid + SV_ERR_WORDS] pinned, GPU-mapped: 1 per serving wavefront, then the error / keep-alive words
    unsigned n_flags, n_serving;  // 4 grid; wavefronts with at least one env: ceil(n_envs / 64)
    uint32_t* d_ctl;         // its device address
    uint32_t* d_claims;      // [32] block claims per XCD: [0..8] the server's, [16..24] the client's (sv_claim_block)
    unsigned grid;
    uint32_t seq;            // the tag of the last request served (= steps served since the server was opened)
    uint64_t idle_ticks, life_ticks, client_ticks;
    double idle_s;
    struct timespec last_use;
    bool launched;
};

namespace {
static_assert(OC_SV_STOP == SV_STOP, "command bit of include/oc_amd.h");
// how the two ends wait: bits 0..7 naps (64 clk each) before the first look, 8..15 naps between looks, bit 16 light polls (the
// wavefront's first env alone until it shows the tag), bit 17 never look through the L2, bits 24..26 (client) which XCD's blocks
// a workgroup claims: its own + this.  Measured on MI355X, 65 536 envs, 1 000 dependent steps (gpurun_out -> profiles/
// r06_step_server.txt): naps and light polls change nothing (3.3 us either way, light polls +0.3: one more load round trip);
// both ends of every env on ONE XCD 4.17 us (5.37 without the looks through the L2: 32 pairs share one L2's channels), on
// neighbouring XCDs 3.2, four XCDs apart 3.02 — so the client claims the blocks of the XCD opposite its own; a single pair
// alone: 2.27 same XCD, 2.63 across.  (tuning builds: the named environment variable overrides the value)
uint32_t sv_knobs(const char* name, uint32_t dflt) {
#ifdef OC_AMD_TUNING
    if (const char* e = getenv(name)) return (uint32_t)strtoul(e, nullptr, 0);
#endif
    (void)name;
    return dflt;
}
double sv_since(const struct timespec& t) {
    struct timespec now;
    clock_gettime(CLOCK_MONOTONIC, &now);
    return (double)(now.tv_sec - t.tv_sec) + 1e-9 * (double)(now.tv_nsec - t.tv_nsec);
}
unsigned sv_resident(const OcStepServer* m) {  // wavefronts that say they serve
    unsigned alive = 0;
    for (unsigned i = 0; i < m->n_flags; ++i) alive += __atomic_load_n(m->h_ctl + i, __ATOMIC_ACQUIRE) != 0u;
    return alive;
}
void sv_mark(OcStepServer* m, uint32_t v) {
    for (unsigned i = 0; i < m->n_flags; ++i) __atomic_store_n(m->h_ctl + i, v, __ATOMIC_RELEASE);
}
template <typename K>
bool sv_fits(K kernel, const OcStepServer* m, size_t smem) {  // every workgroup must be resident at once: they all poll
    int per_cu = 0, cus = 0;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, kernel, BLOCK, smem) != hipSuccess ||
        hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, m->device) != hipSuccess) {
        (void)hipGetLastError();
        return false;
    }
    return (int64_t)m->grid <= (int64_t)per_cu * cus;
}
// the resident kernel leaves (writes the states back) and its stream drains; the host's step count follows the device's
int sv_stop(OcStepServer* m) {
    if (!m->launched) return OC_OK;
    if (sv_resident(m) != 0u)  // (on the control stream: behind the resident kernel on its own stream the post would never run)
        hipLaunchKernelGGL(k_step_server_post, dim3(m->grid), dim3(BLOCK), 0, m->ctl_stream, m->d_req, m->b.n_envs, SV_STOP, m->d_rsp, 1u);
    if (hipStreamSynchronize(m->ctl_stream) != hipSuccess || hipStreamSynchronize(m->stream) != hipSuccess) {
        (void)hipGetLastError();
        return fail(OC_ELAUNCH, "oc_step_server: the resident kernel did not leave");
    }
    m->launched = false;
    sv_mark(m, 0u);
    uint32_t tag = 0;  // env 0's last served tag (device-side callers may have advanced it without the host)
    if (hipMemcpy(&tag, reinterpret_cast<const uint8_t*>(m->d_rsp) + 28, 4, hipMemcpyDeviceToHost) != hipSuccess) {
        (void)hipGetLastError();
        return fail(OC_ELAUNCH, "oc_step_server: reading the step count back failed");
    }
    m->seq = tag;
    return OC_OK;
}
// (re)launch the resident kernel; the requests first lose any stale STOP (a no-op request: the tag already served)
int sv_launch(OcStepServer* m) {
    const OcBatch* b = &m->b;
    const size_t smem = (size_t)m->n_obj * 8 * BLOCK * sizeof(uint32_t);
    const bool uniform = b->n_layouts == 1, lds = b->n_layouts <= LDS_LAYOUT_MAX, small = b->max_pots >= 1 && b->max_pots <= 2;
    sv_mark(m, 0u);  // (every serving wavefront reports in by itself)
    hipLaunchKernelGGL(k_step_server_post, dim3(m->grid), dim3(BLOCK), 0, m->stream, m->d_req, b->n_envs, 0u, m->d_rsp, 0u);
    (void)hipMemsetAsync(m->d_claims, 0, 16 * sizeof(uint32_t), m->stream);
#define GOSV(U, MP, LL)                                                                                              \
    do {                                                                                                             \
        if (!want_lds(k_step_server<U, MP, LL>, smem)) break;                                                        \
        if (!sv_fits(k_step_server<U, MP, LL>, m, smem)) {                                                           \
            sv_mark(m, 0u);                                                                                          \
            return fail(OC_EINVAL, "oc_step_server: the batch needs more workgroups than the GPU keeps resident at once"); \
        }                                                                                                            \
        hipLaunchKernelGGL((k_step_server<U, MP, LL>), dim3(m->grid), dim3(BLOCK), smem, m->stream, b->d_layouts, b->n_layouts, \
                           b->d_layout_id, (uint4*)m->d_state, (float4*)m->d_ep_returns, m->d_req, m->d_rsp, m->d_ctl, m->d_claims, b->n_envs, \
                           b->width, m->n_obj, m->horizon, m->options, m->sa, m->idle_ticks, m->life_ticks, sv_knobs("OC_SV_SERVER", 0x0100u)); \
    } while (0)
    if (uniform && small) GOSV(true, 2, true);
    else if (lds && small) GOSV(false, 2, true);
    else GOSV(false, 8, false);
#undef GOSV
    const int rc = check_launch("oc_step_server");
    if (rc) { sv_mark(m, 0u); return rc; }
    m->launched = true;
    clock_gettime(CLOCK_MONOTONIC, &m->last_use);
    // the server is up when every wavefront that has envs has said so (its states loaded, its first look at the requests next)
    while (sv_resident(m) != m->n_serving) {
        if (sv_since(m->last_use) > 2.0) {
            (void)sv_stop(m);
            return fail(OC_ELAUNCH, "oc_step_server: the resident kernel did not come up within 2 s");
        }
        __builtin_ia32_pause();
    }
    clock_gettime(CLOCK_MONOTONIC, &m->last_use);
    return OC_OK;
}
// resident and fresh (no workgroup about to leave for idleness), or relaunched
// (caller_stream: the stream the caller's work on d_state runs on — a relaunch reads d_state on the server's own stream, so whatever the
//  caller enqueued there since the last sync must be complete first)
int sv_ensure(OcStepServer* m, const hipStream_t* caller_stream = nullptr) {
    if (m->launched) {
        // announce the caller FIRST (a workgroup about to leave for idleness or age looks at this word and stays), give a workgroup
        // that had already looked 20 us to say that it left, THEN count: whoever is counted is still there when the client arrives
        __atomic_fetch_add(m->h_ctl + m->n_flags + SV_KEEPALIVE, 1u, __ATOMIC_RELEASE);
        struct timespec t0;
        clock_gettime(CLOCK_MONOTONIC, &t0);
        while (sv_since(t0) < 20e-6) __builtin_ia32_pause();
    }
    if (m->launched && sv_resident(m) == m->n_serving) return OC_OK;
    if (int rc = sv_stop(m)) return rc;
    if (caller_stream && hipStreamSynchronize(*caller_stream) != hipSuccess) {
        (void)hipGetLastError();
        return fail(OC_ELAUNCH, "oc_step_server: the caller's stream failed");
    }
    return sv_launch(m);
}
struct SvDevice {  // the server's device current for the scope
    int prev = 0, want;
    explicit SvDevice(int d) : want(d) { (void)hipGetDevice(&prev); if (prev != want) (void)hipSetDevice(want); }
    ~SvDevice() { if (prev != want) (void)hipSetDevice(prev); }
};
}  // namespace

extern "C" {

int oc_step_server_open(const OcBatch* b, void* d_state, float* d_ep_returns, int horizon, uint32_t options,
                        const OcStartSpec* start, double idle_ms, double life_s, OcStepServer** out) {
    int n_obj = 0;
    if (!out) return fail(OC_EINVAL, "oc_step_server_open: NULL result pointer");
    *out = nullptr;
    if (int rc = check_batch(b, &n_obj)) return rc;
    StartArgs sa;
    if (!start_args(start, &sa, b)) return fail(OC_EINVAL, "oc_step_server_open: start.rnd_obj_prob_thresh must be in [0, 1] and its regen range within the table");
    if (!d_state) return fail(OC_EINVAL, "oc_step_server_open: NULL state pointer");
    if (horizon < 1 || horizon > 65535) return fail(OC_EINVAL, "oc_step_server_open: horizon must be in 1..65535");
    if (options & ~(uint32_t)OC_OPT_AUTO_RESET) return fail(OC_EINVAL, "oc_step_server_open: the only option is OC_OPT_AUTO_RESET");
    if (b->n_envs < 1) return fail(OC_EINVAL, "oc_step_server_open: no envs");
    if (!(idle_ms >= 0.0 && idle_ms <= 10000.0) || !(life_s >= 0.0 && life_s <= 86400.0))
        return fail(OC_EINVAL, "oc_step_server_open: idle_ms in 0..10 000 (0: 20 ms), life_s in 0..86 400 (0: 600 s)");
    OcStepServer* m = new OcStepServer();
    memset(m, 0, sizeof(*m));
    m->b = *b; m->n_obj = n_obj; m->horizon = horizon; m->options = options; m->sa = sa; m->d_state = d_state; m->d_ep_returns = d_ep_returns;
    m->grid = grid_for(b->n_envs);
    m->n_flags = m->grid * (BLOCK / 64);
    m->n_serving = (unsigned)((b->n_envs + 63) / 64);
    if (idle_ms == 0.0) idle_ms = 20.0;
    if (idle_ms < 0.2) idle_ms = 0.2;  // (a window the host's 20 us announcement always fits in)
    if (life_s == 0.0) life_s = 600.0;
    int khz = 0;
    bool ok = hipGetDevice(&m->device) == hipSuccess;
    if (ok && (hipDeviceGetAttribute(&khz, hipDeviceAttributeWallClockRate, m->device) != hipSuccess || khz <= 0)) khz = 100000;  // 100 MHz
    m->idle_s = idle_ms * 1e-3;
    m->idle_ticks = (uint64_t)((double)khz * idle_ms);
    m->life_ticks = (uint64_t)((double)khz * 1000.0 * life_s);
    m->client_ticks = (uint64_t)khz * 1000;  // a client wavefront gives up after 1 s without its responses
    const size_t ctl_bytes = ((size_t)m->n_flags + SV_ERR_WORDS) * sizeof(uint32_t);
    ok = ok && hipMalloc((void**)&m->d_req, (size_t)b->n_envs * 8) == hipSuccess;
    ok = ok && hipMalloc((void**)&m->d_rsp, (size_t)b->n_envs * 32) == hipSuccess;
    ok = ok && hipMalloc((void**)&m->d_claims, 32 * sizeof(uint32_t)) == hipSuccess;
    ok = ok && hipHostMalloc((void**)&m->h_ctl, ctl_bytes, hipHostMallocMapped | hipHostMallocCoherent) == hipSuccess;
    ok = ok && hipHostGetDevicePointer((void**)&m->d_ctl, m->h_ctl, 0) == hipSuccess;
    ok = ok && hipStreamCreateWithFlags(&m->stream, hipStreamNonBlocking) == hipSuccess;
    ok = ok && hipStreamCreateWithFlags(&m->ctl_stream, hipStreamNonBlocking) == hipSuccess;
    ok = ok && hipEventCreate(&m->ev0) == hipSuccess && hipEventCreate(&m->ev1) == hipSuccess;
    if (ok) {
        memset(m->h_ctl, 0, ctl_bytes);
        ok = hipMemsetAsync(m->d_req, 0, (size_t)b->n_envs * 8, m->stream) == hipSuccess &&
             hipMemsetAsync(m->d_rsp, 0, (size_t)b->n_envs * 32, m->stream) == hipSuccess &&
             hipMemsetAsync(m->d_claims, 0, 32 * sizeof(uint32_t), m->stream) == hipSuccess &&
             hipStreamSynchronize(m->stream) == hipSuccess;  // (before the resident kernel occupies the stream: clients run on other streams)
    }
    if (!ok) {
        (void)hipGetLastError();
        (void)oc_step_server_close(m);
        return fail(OC_ELAUNCH, "oc_step_server_open: device / pinned memory, streams or events refused");
    }
    if (int rc = sv_launch(m)) { (void)oc_step_server_close(m); return rc; }
    *out = m;
    return OC_OK;
}

void* oc_step_server_requests(OcStepServer* m) { return m ? m->d_req : nullptr; }
void* oc_step_server_responses(OcStepServer* m) { return m ? m->d_rsp : nullptr; }

int oc_step_server_resume(OcStepServer* m) {
    if (!m) return fail(OC_EINVAL, "oc_step_server_resume: NULL server");
    SvDevice dev(m->device);
    return sv_ensure(m);
}

int oc_step_server_play(OcStepServer* m, const uint8_t* d_actions, float* d_rewards, uint8_t* d_flags, int n_steps, void* stream,
                        float* elapsed_ms) {
    if (!m) return fail(OC_EINVAL, "oc_step_server_play: NULL server");
    if (!d_actions || !d_rewards || !d_flags) return fail(OC_EINVAL, "oc_step_server_play: NULL actions/rewards/flags pointer");
    if (n_steps < 0 || n_steps > (1 << 24)) return fail(OC_EINVAL, "oc_step_server_play: n_steps must be in 0..2^24");
    if (elapsed_ms) *elapsed_ms = 0.f;
    if (n_steps == 0) return OC_OK;
    SvDevice dev(m->device);
    hipStream_t s = (hipStream_t)stream;
    if (int rc = sv_ensure(m, &s)) return rc;
    __atomic_store_n(m->h_ctl + m->n_flags + SV_ERR_CLIENT, 0u, __ATOMIC_RELEASE);
    uint32_t* dbg = nullptr;
#ifdef OC_AMD_TUNING
    if (getenv("OC_SV_DEBUG")) (void)hipMalloc((void**)&dbg, (size_t)m->grid * 16);
#endif
    (void)hipEventRecord(m->ev0, s);
    hipLaunchKernelGGL(k_step_client, dim3(m->grid), dim3(BLOCK), 0, s, m->d_req, m->d_rsp, d_actions, (float4*)d_rewards, d_flags,
                       m->d_ctl + m->n_flags, m->d_claims + 16, m->b.n_envs, m->seq + 1u, n_steps, m->client_ticks, sv_knobs("OC_SV_CLIENT", 0x04000100u), dbg);
    (void)hipEventRecord(m->ev1, s);
    if (int rc = check_launch("oc_step_server_play")) return rc;
    if (hipEventSynchronize(m->ev1) != hipSuccess) {
        (void)hipGetLastError();
        return fail(OC_ELAUNCH, "oc_step_server_play: the client kernel failed");
    }
    clock_gettime(CLOCK_MONOTONIC, &m->last_use);
    if (__atomic_load_n(m->h_ctl + m->n_flags + SV_ERR_CLIENT, __ATOMIC_ACQUIRE) != 0u) {
        (void)sv_stop(m);  // (the host's step count follows whatever the device got to)
        (void)hipMemset(m->d_claims + 16, 0, 16 * sizeof(uint32_t));  // (a client that gave up did not hand its block claims back)
        return fail(OC_ELAUNCH, "oc_step_server_play: no answer from the resident kernel within 1 s");
    }
    m->seq += (uint32_t)n_steps;
    if (elapsed_ms) (void)hipEventElapsedTime(elapsed_ms, m->ev0, m->ev1);
#ifdef OC_AMD_TUNING
    if (dbg) {  // where a round trip goes, in 10 ns ticks: the client's post -> response seen, of which the server's seen -> sent
        uint32_t* h = (uint32_t*)malloc((size_t)m->grid * 16);
        (void)hipMemcpy(h, dbg, (size_t)m->grid * 16, hipMemcpyDeviceToHost);
        double rt = 0, sv = 0, worst = 0;
        int cross = 0;
        for (unsigned b = 0; b < m->grid; ++b) {
            rt += h[4 * b]; sv += h[4 * b + 1];
            if (h[4 * b] > worst) worst = h[4 * b];
            cross += (h[4 * b + 2] & 0xFu) != (h[4 * b + 3] & 0xFu);
        }
        fprintf(stderr, "[oc_step_server] %d steps x %u workgroups: round trip mean %.0f ns (in the server %.0f ns), slowest workgroup %.0f ns; %d pairs across XCDs\n",
                n_steps, m->grid, 10.0 * rt / n_steps / m->grid, 10.0 * sv / n_steps / m->grid, 10.0 * worst / n_steps, cross);
        if (getenv("OC_SV_DEBUG_ALL"))
            for (unsigned b = 0; b < m->grid; ++b)
                fprintf(stderr, "  wg %3u: rt %5.0f ns server %4.0f ns  server xcc %u  client xcc %u\n", b, 10.0 * h[4 * b] / n_steps, 10.0 * h[4 * b + 1] / n_steps,
                        h[4 * b + 2] & 0xFu, h[4 * b + 3] & 0xFu);
        free(h);
        (void)hipFree(dbg);
    }
#endif
    return OC_OK;
}

int oc_step_server_sync(OcStepServer* m) {
    if (!m) return fail(OC_EINVAL, "oc_step_server_sync: NULL server");
    SvDevice dev(m->device);
    return sv_stop(m);
}

int64_t oc_step_server_steps(OcStepServer* m) { return m ? (int64_t)m->seq : -1; }

int oc_step_server_close(OcStepServer* m) {
    if (!m) return OC_OK;
    SvDevice dev(m->device);
    int rc = OC_OK;
    if (m->stream && m->ctl_stream) rc = sv_stop(m);
    if (m->ev0) (void)hipEventDestroy(m->ev0);
    if (m->ev1) (void)hipEventDestroy(m->ev1);
    if (m->ctl_stream) (void)hipStreamDestroy(m->ctl_stream);
    if (m->stream) (void)hipStreamDestroy(m->stream);
    if (m->h_ctl) (void)hipHostFree(m->h_ctl);
    if (m->d_claims) (void)hipFree(m->d_claims);
    if (m->d_rsp) (void)hipFree(m->d_rsp);
    if (m->d_req) (void)hipFree(m->d_req);
    (void)hipGetLastError();
    delete m;
    return rc;
}

}  // extern "C"


struct OcMailbox {
    uint8_t* h;             // the mailbox (pinned host memory, mapped into the GPU's address space)
    uint8_t* d;             // its device address
    hipStream_t stream;     // the resident kernel's own stream
    const OcLayout* d_layout;
    int W, n_obj, horizon, device, max_pots;
    uint32_t seq;
    uint64_t idle_ticks, life_ticks;
};

namespace {
static_assert(OC_MB_STATE_IN == MB_IN && OC_MB_ACTIONS == MB_ACT && OC_MB_STATE_OUT == MB_OUT && OC_MB_REWARDS == MB_REW &&
              OC_MB_FLAGS == MB_FLAGS && OC_MB_EVENTS == MB_EV, "mailbox offsets of include/oc_amd.h");

// the request granules of `tag`: payload = n_state bytes of MB_IN + the two bytes of MB_ACT
void mailbox_post(OcMailbox* m, uint32_t tag, int n_state) {
    alignas(16) uint8_t pay[12 * MB_REQ_MAX + 4] = {0};
    memcpy(pay, m->h + MB_IN, (size_t)n_state);
    memcpy(pay + n_state, m->h + MB_ACT, 2);
    const int n_req = (n_state + 2 + 11) / 12;
    for (int g = 0; g < n_req; ++g) {
        alignas(16) uint32_t q[4];
        memcpy(q, pay + 12 * g, 12);
        q[3] = tag;
        typedef long long mb_i64x2 __attribute__((vector_size(16), aligned(16)));
        *reinterpret_cast<volatile mb_i64x2*>(m->h + MB_REQG + 16 * g) = *reinterpret_cast<const mb_i64x2*>(q);  // one movaps
    }
}

int mailbox_launch(OcMailbox* m) {
    *reinterpret_cast<volatile uint32_t*>(m->h + MB_ALIVE) = 1u;
    DISPATCH_NOBJ(m->n_obj, {
        constexpr int NO = NOBJ <= STEP1_MAX_PLANES ? NOBJ : 1;
        if (NOBJ <= STEP1_MAX_PLANES) {
            if (m->max_pots <= 2)
                hipLaunchKernelGGL((k_mailbox<NO, 2>), dim3(1), dim3(64), 0, m->stream, m->d_layout, m->d, m->W, m->horizon,
                                   m->idle_ticks, m->life_ticks);
            else
                hipLaunchKernelGGL((k_mailbox<NO, OC_MAX_POTS>), dim3(1), dim3(64), 0, m->stream, m->d_layout, m->d, m->W,
                                   m->horizon, m->idle_ticks, m->life_ticks);
        }
    });
    return check_launch("oc_mailbox: launch");
}
}  // namespace

extern "C" {

int oc_mailbox_open(const OcBatch* b, int horizon, OcMailbox** out) {
    int n_obj = 0;
    if (!out) return fail(OC_EINVAL, "oc_mailbox_open: NULL result pointer");
    *out = nullptr;
    if (int rc = check_batch(b, &n_obj)) return rc;
    if (b->n_layouts != 1 || n_obj > STEP1_MAX_PLANES) return fail(OC_EINVAL, "oc_mailbox_open: one layout, grids of at most 64 cells");
    if (horizon < 1 || horizon > 65535) return fail(OC_EINVAL, "oc_mailbox_open: horizon must be in 1..65535");
    OcMailbox* m = new OcMailbox();
    if (hipGetDevice(&m->device) != hipSuccess || hipHostMalloc((void**)&m->h, MB_BYTES, hipHostMallocMapped | hipHostMallocCoherent) != hipSuccess) {
        delete m;
        (void)hipGetLastError();
        return fail(OC_ELAUNCH, "oc_mailbox_open: pinned host memory refused");
    }
    memset(m->h, 0, MB_BYTES);
    int khz = 0;
    if (hipDeviceGetAttribute(&khz, hipDeviceAttributeWallClockRate, m->device) != hipSuccess || khz <= 0) khz = 100000;  // 100 MHz
    if (hipHostGetDevicePointer((void**)&m->d, m->h, 0) != hipSuccess ||
        hipStreamCreateWithFlags(&m->stream, hipStreamNonBlocking) != hipSuccess) {
        (void)hipHostFree(m->h);
        delete m;
        (void)hipGetLastError();
        return fail(OC_ELAUNCH, "oc_mailbox_open: device mapping / stream refused");
    }
    m->d_layout = b->d_layouts; m->W = b->width; m->n_obj = n_obj; m->horizon = horizon; m->seq = 0;
    m->max_pots = b->max_pots > 0 ? b->max_pots : OC_MAX_POTS;  // (hint of oc_batch_hints; unknown: the general instance)
    m->idle_ticks = (uint64_t)khz * 2;     // 2 ms without a request
    m->life_ticks = (uint64_t)khz * 2000;  // 2 s in any case
    if (int rc = mailbox_launch(m)) { (void)oc_mailbox_close(m); return rc; }
    *out = m;
    return OC_OK;
}

void* oc_mailbox_buffer(OcMailbox* m) { return m ? m->h : nullptr; }

int oc_mailbox_step(OcMailbox* m) {
    if (!m) return fail(OC_EINVAL, "oc_mailbox_step: NULL mailbox");
    uint32_t seq = m->seq + 1u;
    if (seq == MB_STOP || seq == 0u) seq = 1u;
    m->seq = seq;
    const int n_state = 16 * (1 + m->n_obj);
    const int n_words = (n_state + 28) / 4;  // payload dwords of the response
    // ---- the request: the caller's plain views (MB_IN, MB_ACT) as granules of {12 payload bytes, tag}, one aligned 16-byte
    //      store each (the kernel's lanes read one granule each with a single 16-byte load: payload and tag arrive together)
    mailbox_post(m, seq, n_state);
    // ---- the response: two 64-byte lines of {15 payload dwords, tag}
    volatile uint32_t* alive = reinterpret_cast<volatile uint32_t*>(m->h + MB_ALIVE);
    uint32_t spins = 0;
    struct timespec t0 = {0, 0};
    for (;;) {
        // the last dword of each response line carries the tag once the line has arrived (line 1 only when the payload needs it)
        int ok = __atomic_load_n(reinterpret_cast<uint32_t*>(m->h + MB_RSPG + 60), __ATOMIC_ACQUIRE) == seq;
        if (ok && n_words > 15) ok = __atomic_load_n(reinterpret_cast<uint32_t*>(m->h + MB_RSPG + 124), __ATOMIC_ACQUIRE) == seq;
        if (ok) break;
        __builtin_ia32_pause();
        if ((++spins & 0x3FFu) != 0u) continue;
        if (*alive == 0u) {  // the kernel has left (idle / lifetime): the next incarnation finds the request
            int dev = 0;
            (void)hipGetDevice(&dev);
            if (dev != m->device) (void)hipSetDevice(m->device);
            const int rc = mailbox_launch(m);
            if (dev != m->device) (void)hipSetDevice(dev);
            if (rc) return rc;
        }
        struct timespec now;
        clock_gettime(CLOCK_MONOTONIC, &now);
        if (t0.tv_sec == 0 && t0.tv_nsec == 0) t0 = now;
        else if ((now.tv_sec - t0.tv_sec) > 5) return fail(OC_ELAUNCH, "oc_mailbox_step: no answer from the resident kernel within 5 s");
    }
    uint8_t rsp[120];
    memcpy(rsp, m->h + MB_RSPG, 60);
    memcpy(rsp + 60, m->h + MB_RSPG + 64, 60);
    memcpy(m->h + MB_OUT, rsp, (size_t)n_state);
    memcpy(m->h + MB_REW, rsp + n_state, 16);
    memcpy(m->h + MB_FLAGS, rsp + n_state + 16, 4);
    memcpy(m->h + MB_EV, rsp + n_state + 20, 8);
    return OC_OK;
}

#ifdef OC_AMD_TUNING
// tuning builds: the phase stamps of the last k_train_step_obs launch (train_obs.hpp: g_obs_dbg), n_words u32
int oc_debug_train_obs(uint32_t* out, int n_words) {
    return hipMemcpyFromSymbol(out, HIP_SYMBOL(g_obs_dbg), (size_t)n_words * 4, 0, hipMemcpyDeviceToHost) == hipSuccess ? 0 : -1;
}
#endif
#ifdef OC_AMD_TUNING
// tuning builds: n back-to-back oc_mailbox_step calls from C (no Python / ctypes between them) -> microseconds per call
double oc_mailbox_bench(OcMailbox* m, int n) {
    struct timespec a, b;
    clock_gettime(CLOCK_MONOTONIC, &a);
    for (int i = 0; i < n; ++i)
        if (oc_mailbox_step(m)) return -1.0;
    clock_gettime(CLOCK_MONOTONIC, &b);
    return ((b.tv_sec - a.tv_sec) * 1e9 + (b.tv_nsec - a.tv_nsec)) / n * 1e-3;
}
#endif

int oc_output_stores_only(int64_t n_envs, int n_steps, float* d_rewards, uint8_t* d_flags, uint32_t options, void* stream) {
    if (n_envs < 0 || n_steps < 0 || !d_rewards) return fail(OC_EINVAL, "oc_output_stores_only: negative sizes or no rewards array");
    if (((uintptr_t)d_rewards & 15u) != 0) return fail(OC_EINVAL, "oc_output_stores_only: d_rewards must be 16-byte aligned");
    if (options & ~(uint32_t)OC_OPT_FLAGS_TILED8) return fail(OC_EINVAL, "oc_output_stores_only: the only option is OC_OPT_FLAGS_TILED8");
    if (options & OC_OPT_FLAGS_TILED8) {
        if (!d_flags || ((uintptr_t)d_flags & 7u) != 0 || (n_steps & 7) != 0)
            return fail(OC_EINVAL, "oc_output_stores_only: OC_OPT_FLAGS_TILED8 needs an 8-byte aligned d_flags and n_steps a multiple of 8");
        if (n_envs == 0 || n_steps == 0) return OC_OK;
        hipLaunchKernelGGL(k_output_stores_only_tiled8, dim3(grid_for(n_envs)), dim3(BLOCK), 0, (hipStream_t)stream, (float4*)d_rewards,
                           (uint2*)d_flags, n_envs, n_steps / 8);
        return check_launch("oc_output_stores_only");
    }
    if (n_envs == 0 || n_steps == 0) return OC_OK;
    hipLaunchKernelGGL(k_output_stores_only, dim3(grid_for(n_envs)), dim3(BLOCK), 0, (hipStream_t)stream, (float4*)d_rewards, d_flags,
                       n_envs, n_steps);
    return check_launch("oc_output_stores_only");
}

int oc_mailbox_close(OcMailbox* m) {
    if (!m) return OC_OK;
    mailbox_post(m, MB_STOP, 16 * (1 + m->n_obj));
    (void)hipStreamSynchronize(m->stream);
    (void)hipStreamDestroy(m->stream);
    (void)hipHostFree(m->h);
    (void)hipGetLastError();
    delete m;
    return OC_OK;
}

}  // extern "C"
