// shaping.hpp — training reward of the RLlib environment: k_shape_rewards
// Part of liboc_amd.so: included by oc_amd.hip inside its anonymous namespace, in this order:
//   common, reset, step_predicate, step_table, step_one, step_lut4, rollout_pair, encode, rollout_encode, featurize, potential, shaping.
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

// ------------------------------------------------------------------------------------------
// k_train_step: the whole batched step of the RLlib environment for two-player tables with at most two pots, in
// one kernel: oc_step (table-driven, no auto-reset) -> phi(s') from the registers the step just produced
// (potential2_core) -> the shaped rewards of k_shape_rewards -> restart of finished envs -> state written once.
// Same outputs, bit for bit, as the sequence oc_step / oc_potential / oc_shape_rewards / copy / oc_reset that
// oc_multi_agent_step enqueues for every other table (tests/test_gpu_parity.py compares the two).
// ------------------------------------------------------------------------------------------
template <bool UNIFORM, int MAXP, bool LAY_LDS, bool FAST, bool EVENTS = false>
__global__ __launch_bounds__(BLOCK) void k_train_step(const OcLayout* __restrict__ g_layouts, int n_layouts,
                                                      const uint16_t* layout_id, uint4* st,
                                                      const uint8_t* __restrict__ actions, float4* __restrict__ rewards,
                                                      uint8_t* __restrict__ flags, float4* __restrict__ ep_returns,
                                                      float4* __restrict__ ep_out, const uint8_t* __restrict__ plan_blob,
                                                      const uint32_t* __restrict__ plan_off,
                                                      const uint8_t* __restrict__ phi_tables, double* __restrict__ phi_next,
                                                      double* __restrict__ phi_cur, const double* __restrict__ phi_start,
                                                      double factor, double* __restrict__ shaped, uint8_t* __restrict__ done,
                                                      int64_t n, int W, int H, int n_obj, int horizon, StartArgs sa,
                                                      EvArgs ea) {
#pragma clang fp contract(off)
    extern __shared__ __attribute__((aligned(16))) uint16_t s_cells3[];  // [n_obj * 16][BLOCK]
    __shared__ uint4 s_lay[LAY_LDS ? (UNIFORM ? 16 : LDS_LAYOUT_MAX * 16) : 1];  // one 256-byte record when the batch has one layout
    __shared__ uint2 s_lut[2 * LUT_ENTRIES];
    const int64_t e = (int64_t)blockIdx.x * BLOCK + threadIdx.x;
    const bool active = e < n;
    for (int i = threadIdx.x; i < 2 * LUT_ENTRIES; i += BLOCK) s_lut[i] = reinterpret_cast<const uint2*>(&g_lut)[i];
    Lay L = stage_layouts<LAY_LDS>(g_layouts, n_layouts, layout_id, e, active, s_lay);  // contains the barrier
    if (!active) return;
    uint16_t* cells = s_cells3 + threadIdx.x;
    LayC C = load_consts<UNIFORM>(L);
    const uint8_t* lut = reinterpret_cast<const uint8_t*>(s_lut) + (C.old_dyn ? LUT_ENTRIES * 8 : 0);
    const uint32_t delta4 = make_delta4(W);
    uint32_t lid = layout_id ? layout_id[e] : 0u;
    Env3<MAXP> s;
    load_env3<MAXP>(C, L, st, n, e, n_obj, s, cells);
    const uint32_t a01 = reinterpret_cast<const uint16_t*>(actions)[e];
    const uint32_t a0 = a01 & 0xFFu, a1 = a01 >> 8;
    float4 r = make_float4(0.f, 0.f, 0.f, 0.f);
    float4 ep = ep_returns ? ep_returns[e] : make_float4(0.f, 0.f, 0.f, 0.f);
    uint32_t fl;
    if (a0 > 5u || a1 > 5u) {
        fl = OC_F_BAD_ACTION;  // the env stays untouched (mdp.py:1394-1398 raises)
    } else {
        uint64_t ev = 0;
        env_step3<MAXP, FAST ? 2 : 0, EVENTS>(C, L, lut, cells, s, delta4, a0, a1, r,
                                              FAST ? make_floor_mask(L, (int)L.u8(L_NCELLS)) : 0ull, nullptr, &ev);
        fl = finish_step3<MAXP>(C, L, n_obj, cells, s, horizon, 0u, r, ep, sa, 0, 0);
        if (EVENTS) {
            if (ea.events) ea.events[e] = ev;
            count_events(ea, e, ev, (fl & OC_F_DONE) != 0u, true);  // finished envs restart below
        }
    }
    if (EVENTS && ea.events && (fl & OC_F_BAD_ACTION)) ea.events[e] = 0;
    const bool is_done = (fl & OC_F_DONE) != 0u;
    const double sparse = (double)r.x + (double)r.y;
    double d0 = (double)r.z, d1 = (double)r.w;
    if (phi_tables) {
        const Phi T{phi_tables + (size_t)lid * PHI_BYTES};
        const double pn = potential2_core(L, T, plan_blob + plan_off[lid], (uint32_t)(W * H), 2u, s.pos0, s.or0, s.held0,
                                          s.pos1, s.or1, s.held1, s.ps[0], MAXP > 1 ? s.ps[MAXP - 1] : 0u, s.tk[0],
                                          MAXP > 1 ? s.tk[MAXP - 1] : 0u);
        d0 = d1 = pn - phi_cur[e];
        phi_next[e] = pn;
        phi_cur[e] = is_done ? phi_start[lid] : pn;
    }
    reinterpret_cast<double2*>(shaped)[e] = make_double2(sparse + factor * d0, sparse + factor * d1);
    done[e] = is_done ? 1 : 0;
    if (ep_out) ep_out[e] = ep;
    if (is_done) {  // the next episode: the standard start state, or one drawn from the batch's start_state_fn
        if (sa.enabled) {
            // (... on a layout drawn for the new episode when the spec says so; phi_cur below is then that layout's)
            regen_layout<UNIFORM, LAY_LDS>(sa, (uint64_t)(sa.env_offset + e), sa.epoch, e, s_lay, g_layouts, L, C, &lid);
            env_reset3_draw<MAXP>(C, L, n_obj, s, cells, draw_start(L, (uint64_t)(sa.env_offset + e), sa.epoch, sa.seed_lo,
                                                                    sa.seed_hi, sa.random_start_pos, sa.thresh));
            if (phi_tables) {  // phi(s) of the next step is the potential of THAT state
                const Phi T{phi_tables + (size_t)lid * PHI_BYTES};
                phi_cur[e] = potential2_core(L, T, plan_blob + plan_off[lid], (uint32_t)(W * H), 2u, s.pos0, s.or0, s.held0,
                                             s.pos1, s.or1, s.held1, s.ps[0], MAXP > 1 ? s.ps[MAXP - 1] : 0u, s.tk[0],
                                             MAXP > 1 ? s.tk[MAXP - 1] : 0u);
            }
        } else {
            env_reset3<MAXP>(L, n_obj, s, cells);
        }
        ep = make_float4(0.f, 0.f, 0.f, 0.f);
    }
    store_env3<MAXP>(C, L, st, n, e, n_obj, s, cells);
    rewards[e] = r;
    flags[e] = (uint8_t)fl;
    if (ep_returns) ep_returns[e] = ep;
}

// ------------------------------------------------------------------------------------------
// k_train_step1: k_train_step with the transition computed on the wire format itself (step_one.hpp): the potential
// needs the players and the pots — registers the step has just produced — and nothing of the counters, so the env is never
// unpacked; header + changed object bytes are written back, the planes only for envs that restart.  No event logging
// (k_train_step then), grids of at most 64 cells.  Same outputs, bit for bit.
// ------------------------------------------------------------------------------------------
template <bool UNIFORM, int MAXP, bool LAY_LDS>
__global__ __launch_bounds__(BLOCK) void k_train_step1(const OcLayout* __restrict__ g_layouts, int n_layouts,
                                                       const uint16_t* layout_id, uint4* st,
                                                       const uint8_t* __restrict__ actions, float4* __restrict__ rewards,
                                                       uint8_t* __restrict__ flags, float4* ep_returns,
                                                       float4* __restrict__ ep_out, const uint8_t* __restrict__ plan_blob,
                                                       const uint32_t* __restrict__ plan_off,
                                                       const uint8_t* __restrict__ phi_tables, double* __restrict__ phi_next,
                                                       double* __restrict__ phi_cur, const double* __restrict__ phi_start,
                                                       double factor, double* __restrict__ shaped, uint8_t* __restrict__ done,
                                                       int64_t n, int W, int H, int n_obj, int horizon, StartArgs sa) {
#pragma clang fp contract(off)
    extern __shared__ __attribute__((aligned(16))) uint4 s_rows1[];  // [n_obj][BLOCK]: the object planes, one 16-byte row per lane
    __shared__ uint4 s_lay[LAY_LDS ? (UNIFORM ? 16 : LDS_LAYOUT_MAX * 16) : 1];
    __shared__ uint2 s_lut[2 * LUT_ENTRIES];
    const int64_t e = (int64_t)blockIdx.x * BLOCK + threadIdx.x;
    const bool active = e < n;
    const int64_t el = active ? e : n - 1;
    const OneIn in = one_load(st, actions, ep_returns, n, el, n_obj);
    const double phi_before = phi_tables ? phi_cur[el] : 0.0;
    for (int i = threadIdx.x; i < 2 * LUT_ENTRIES; i += BLOCK) s_lut[i] = reinterpret_cast<const uint2*>(&g_lut)[i];
    Lay L = stage_layouts<LAY_LDS>(g_layouts, n_layouts, layout_id, e, active, s_lay);  // contains the barrier
    if (!active) return;
#pragma unroll
    for (int p = 0; p < STEP1_MAX_PLANES; ++p)
        if (p < n_obj) s_rows1[p * BLOCK + threadIdx.x] = in.v[p];
    const uint8_t* row = reinterpret_cast<const uint8_t*>(s_rows1 + threadIdx.x);
    LayC C = load_consts<UNIFORM>(L);
    const uint8_t* lut = reinterpret_cast<const uint8_t*>(s_lut) + (C.old_dyn ? LUT_ENTRIES * 8 : 0);
    uint32_t lid = layout_id ? layout_id[e] : 0u;
    One<MAXP> q;
    one_decode<MAXP>(C, L, in.h, row, q);
    Env3<MAXP>& s = q.s;
    const uint32_t a0 = in.a01 & 0xFFu, a1 = in.a01 >> 8;
    float4 r = make_float4(0.f, 0.f, 0.f, 0.f), ep = in.ep;
    uint32_t fl = 0;
    if (a0 > 5u || a1 > 5u) {
        fl = OC_F_BAD_ACTION;  // the env stays untouched (mdp.py:1394-1398 raises)
    } else {
        one_transition<MAXP>(C, L, lut, make_delta4(W), a0, a1, in.v, n_obj, row, q, r);
        ep.x += r.x; ep.y += r.y; ep.z += r.z; ep.w += r.w;
        if ((int)s.t >= horizon) fl |= OC_F_DONE;
    }
    const bool is_done = (fl & OC_F_DONE) != 0u;
    const double sparse = (double)r.x + (double)r.y;
    double d0 = (double)r.z, d1 = (double)r.w;
    if (phi_tables) {
        const Phi T{phi_tables + (size_t)lid * PHI_BYTES};
        const double pn = potential2_core(L, T, plan_blob + plan_off[lid], (uint32_t)(W * H), 2u, s.pos0, s.or0, s.held0,
                                          s.pos1, s.or1, s.held1, s.ps[0], MAXP > 1 ? s.ps[MAXP - 1] : 0u, s.tk[0],
                                          MAXP > 1 ? s.tk[MAXP - 1] : 0u);
        d0 = d1 = pn - phi_before;
        phi_next[e] = pn;
        phi_cur[e] = is_done ? phi_start[lid] : pn;
    }
    reinterpret_cast<double2*>(shaped)[e] = make_double2(sparse + factor * d0, sparse + factor * d1);
    done[e] = is_done ? 1 : 0;
    if (ep_out) ep_out[e] = ep;
    if (is_done) {  // the next episode: the standard start state, or one drawn from the batch's start_state_fn
        const uint64_t g = (uint64_t)(sa.env_offset + e);
        if (sa.enabled) regen_layout<UNIFORM, LAY_LDS>(sa, g, sa.epoch, e, s_lay, g_layouts, L, C, &lid);  // (... on a layout drawn for it)
        one_restart<MAXP>(C, L, sa, g, s);
        if (sa.enabled && phi_tables) {  // phi(s) of the next step is the potential of THAT state
            const Phi T{phi_tables + (size_t)lid * PHI_BYTES};
            phi_cur[e] = potential2_core(L, T, plan_blob + plan_off[lid], (uint32_t)(W * H), 2u, s.pos0, s.or0, s.held0,
                                         s.pos1, s.or1, s.held1, s.ps[0], MAXP > 1 ? s.ps[MAXP - 1] : 0u, s.tk[0],
                                         MAXP > 1 ? s.tk[MAXP - 1] : 0u);
        }
        ep = make_float4(0.f, 0.f, 0.f, 0.f);
    }
    one_store<MAXP>(C, L, st, n, e, n_obj, q, is_done);
    rewards[e] = r;
    flags[e] = (uint8_t)fl;
    if (ep_returns) ep_returns[e] = ep;
}
