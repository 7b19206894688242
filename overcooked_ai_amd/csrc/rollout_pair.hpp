// rollout_pair.hpp — fused random-policy rollout with two lanes per env: k_rollout_pair
// Part of liboc_amd.so: included by oc_amd.hip inside its anonymous namespace, in this order:
//   common, reset, step_predicate, step_table, step_one, step_lut4, rollout_pair, encode, rollout_encode, featurize, potential, shaping.
#pragma once

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
    __shared__ uint4 s_lay[LAY_LDS ? (UNIFORM ? 16 : LDS_LAYOUT_MAX * 16) : 1];  // one 256-byte record when the batch has one layout
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
        h.y = ori_o | (held_o << 8) | (min(s.t, 0xFFFFu) << 16);  // the wire format's u16 timestep saturates
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
