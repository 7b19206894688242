// shared.hpp — what the translation units of liboc_amd.so share (everything else is internal to a unit: the kernels and
// their helpers live in headers that each unit includes inside its own anonymous namespace).
// Units: oc_amd.hip (the C-ABI and every kernel family but k_rollout4) and rollout4.hip, compiled three times with
// -DOC_R4_PART=0/1/2 (k_rollout4's instances: joint-table + event-logging / per-env-terrain MODE 2 / arithmetic MODE 0), so
// that a clean build compiles them in parallel (overcooked_ai_amd/build.py).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/oc_amd.h"

#define OC_HIDDEN __attribute__((visibility("hidden")))

namespace oc_detail {

// (__thread, not thread_local: a C++ thread_local referenced from another translation unit goes through a "TLS init function"
//  hook that is weak-undefined for these constant-initialised variables; with hidden visibility the null test is folded away and
//  the first access from rollout4.hip jumped to the library's base address — found in round 6 by oc_rollout_plan, latent since
//  round 4 on the error paths of the rollout4 units)
extern __thread char g_err[256] OC_HIDDEN;       // oc_last_error()
extern __thread bool g_lds_refused OC_HIDDEN;    // a dynamic-LDS request was refused: nothing was launched
// oc_rollout_plan: when set, the launch sites of oc_rollout_random write the kernel instance they would launch here (256 bytes)
// and launch nothing
extern __thread char* g_describe OC_HIDDEN;

// Launch-time description of the start_state_fn (include/oc_amd.h, OcStartSpec), by value in kernel arguments.
struct StartArgs {
    uint32_t enabled, seed_lo, seed_hi, epoch;
    int64_t env_offset;
    uint64_t thresh;  // floor(rnd_obj_prob_thresh * 2^32)
    int32_t random_start_pos;
    uint32_t regen_first, regen_count;  // regen_count > 0: a restarted env moves to layout regen_first + draw % regen_count
    uint16_t* layout_ids;               // the batch's layout ids, writable (regen_count > 0)
};

// Where the event_infos of a launch go (include/oc_amd.h, OcEventSink), by value in kernel arguments.
struct EvArgs {
    uint64_t* events;       // [n_steps][n_envs] masks, or NULL
    uint32_t* counts;       // [n_envs][25] running counts of the current episode (player 0: bits 0..15, player 1: 16..31), or NULL
    uint32_t* counts_done;  // [n_envs][25] counts of the last finished episode, or NULL
    uint32_t clear_on_done; // the caller restarts finished envs itself right after this launch (oc_multi_agent_step)
};

// One oc_rollout_random call through k_rollout4, as oc_amd.hip hands it to the unit that holds the instance.
struct Rollout4Call {
    const OcBatch* b;
    int n_obj;
    void* d_state;
    float* d_rewards;
    uint8_t* d_flags;
    float* d_ep_returns;
    int horizon;
    uint32_t options;
    uint64_t seed;
    int64_t env_offset, t0;
    int n_steps;
    StartArgs sa;
    EvArgs ea;
    hipStream_t stream;
    // what oc_rollout_random derived from the batch and the call
    bool uniform, lds, small, joint, old_dyn, out, pipe, events;
    bool tiled8;  // OC_OPT_FLAGS_TILED8: d_flags is [n_steps / 8][n_envs][8]
    bool noout;   // neither d_rewards nor d_flags: k_rollout5 runs its store-free instances
    bool duo;     // MODE 3: the per-env-terrain step split between mover and interact wavefronts (step_lut4.hpp)
};
OC_HIDDEN void launch_rollout4_joint_events(const Rollout4Call& c);  // rollout4.hip, OC_R4_PART 0
OC_HIDDEN void launch_rollout4_mode2(const Rollout4Call& c);         // rollout4.hip, OC_R4_PART 1
OC_HIDDEN void launch_rollout4_mode0(const Rollout4Call& c);         // rollout4.hip, OC_R4_PART 2
OC_HIDDEN size_t rollout5_lds_bytes(bool lay_lds, bool big, bool ev, int n_obj);  // rollout4.hip, OC_R4_PART 1

}  // namespace oc_detail
