// shaping.hpp — training reward of the RLlib environment: k_shape_rewards
// Part of liboc_amd.so: included by oc_amd.hip inside its anonymous namespace, in this order:
//   common, step_predicate, step_table, rollout_pair, reset, encode, featurize, potential, shaping.
#pragma once

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
