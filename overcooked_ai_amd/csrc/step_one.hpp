// step_one.hpp — k_step1: ONE transition per launch, computed on the wire format itself (oc_step's in-place path)
// Part of liboc_amd.so: included by oc_amd.hip inside its anonymous namespace after step_table.hpp (interact3 and the
// 8-byte LUT, step3_env, regen_layout) and reset.hpp (draw_start).
#pragma once

// ==========================================================================================
// A launch that runs a single step executes every instruction once, cold, on the one wavefront a SIMD has: what it
// costs is the launch (2.3-2.8 us back to back for an empty kernel), one round trip to memory, and the length of the
// instruction stream.  k_step3 unpacks every cell of the env into LDS cell words, steps, and packs every cell again
// (~800 instructions executed, 5.9 us); a step reads two faced cells, two move targets and the pots, and changes at most
// two cells and the header.  k_step1 therefore
//   * requests everything it will read — header, object planes, actions, episode returns, the LUT and the layout
//     records — before its first wait (one round trip, coalesced: 16 B / plane / env);
//   * parks its lane's planes in LDS as they are (ds_write_b128 per plane) only to read single bytes at per-lane cell
//     indices, and keeps them in registers for the loose-dish count, which is taken only when some lane of the
//     wavefront takes a dish from a dispenser (is_dish_pickup_useful, mdp.py:2180-2204, is the only reader);
//   * writes back the header and the object bytes that changed (byte stores); the planes are rewritten only when the
//     env restarts.
// Same transition as env_step3 (get_state_transition, mdp.py:1375-1430): interact3 + the 8-byte LUT for
// resolve_interacts, the replay of player 1 when player 0 touched its cell or pot, resolve_movement on the layout's
// terrain bytes, step3_env.  In place only (state_out == state_in), no event logging, grids of at most 64 cells:
// oc_step falls back to k_step3 otherwise.
// ==========================================================================================
constexpr int STEP1_MAX_PLANES = 4;

// what a one-step kernel reads: requested together, before the first wait
struct OneIn {
    uint4 h;                       // header plane
    uint4 v[STEP1_MAX_PLANES];     // object planes (zeros past n_obj)
    uint32_t a01;                  // both actions
    float4 ep;                     // episode returns so far
};
__device__ __forceinline__ OneIn one_load(const uint4* st, const uint8_t* __restrict__ actions, const float4* ep_returns,
                                          int64_t n, int64_t el, int n_obj) {
    OneIn in;
    in.h = st[el];
    in.a01 = reinterpret_cast<const uint16_t*>(actions)[el];
    in.ep = ep_returns ? ep_returns[el] : make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
    for (int p = 0; p < STEP1_MAX_PLANES; ++p) in.v[p] = p < n_obj ? st[(int64_t)(1 + p) * n + el] : make_uint4(0u, 0u, 0u, 0u);
    return in;
}

// object byte of cell c in this lane's planes parked in LDS ([plane][BLOCK] rows of 16 bytes)
__device__ __forceinline__ uint32_t one_obj(const uint8_t* row, uint32_t c) {
    return (uint32_t)row[(c >> 4) * (uint32_t)(BLOCK * 16) + (c & 15u)];
}

// one env in flight: the registers of env_step3 plus what the step changed on the grid
template <int MAXP>
struct One {
    Env3<MAXP> s;
    uint32_t ps_in[MAXP];  // pot objects as loaded
    uint32_t f0, f1;       // faced cells
    uint32_t o0, o1;       // their object bytes after the interacts
    bool w0, w1;           // ... to be written back (counter pick-ups / drops)
};

template <int MAXP>
__device__ __forceinline__ void one_decode(const LayC& C, const Lay L, const uint4 h, const uint8_t* row, One<MAXP>& q) {
    Env3<MAXP>& s = q.s;
    s.pos0 = h.x & 0xFFu; s.or0 = (h.x >> 8) & 0xFFu; s.held0 = (h.x >> 16) & 0xFFu; s.pos1 = h.x >> 24;
    s.or1 = h.y & 0xFFu; s.held1 = (h.y >> 8) & 0xFFu; s.t = h.y >> 16;
    s.dcount = 0;
#pragma unroll
    for (int k = 0; k < MAXP; ++k) {
        s.ps[k] = 0; s.tk[k] = 0; s.pc[k] = PC_EMPTY;
        if ((uint32_t)k < C.n_pots) {
            s.ps[k] = one_obj(row, L.pot_cell(k));
            s.tk[k] = ((k < 4 ? h.z : h.w) >> (8 * (k & 3))) & 0xFFu;
            s.pc[k] = pot_class(C, s.ps[k], s.tk[k]);
        }
        q.ps_in[k] = s.ps[k];
    }
    q.f0 = q.f1 = q.o0 = q.o1 = 0;
    q.w0 = q.w1 = false;
}

// get_state_transition (mdp.py:1375-1430) for legal actions a0, a1: the sequencing of step3_main (both interacts against
// the pre-step pots / cells, player 1 again when player 0 changed what it faces), movement on the layout's terrain bytes,
// step3_env.  r = (sparse0, sparse1, shaped0, shaped1).
template <int MAXP, bool EVENTS = false>
__device__ __forceinline__ void one_transition(const LayC& C, const Lay L, const uint8_t* lut, uint32_t delta4, uint32_t a0,
                                               uint32_t a1, const uint4 (&v)[STEP1_MAX_PLANES], int n_obj,
                                               const uint8_t* row, One<MAXP>& q, float4& r, uint64_t* ev = nullptr) {
    Env3<MAXP>& s = q.s;
    const bool two = s.pos1 != 0xFFu;
    const bool mv0 = a0 < 4u, mv1 = two & (a1 < 4u);
    // the faced cells (pre-move pose, mdp.py:1452-1454) and the move targets
    const uint32_t f0 = step_cell(s.pos0, s.or0, delta4), f1 = two ? step_cell(s.pos1, s.or1, delta4) : f0;
    const uint32_t m0 = mv0 ? step_cell(s.pos0, a0, delta4) : s.pos0;
    const uint32_t m1 = mv1 ? step_cell(s.pos1, a1, delta4) : (two ? s.pos1 : s.pos0);
    const uint32_t c_f0 = one_obj(row, f0) | (L.terrain(f0) << 8), c_f1 = one_obj(row, f1) | (L.terrain(f1) << 8);  // interact3's cell words
    const uint32_t t_m0 = L.terrain(m0) & 7u, t_m1 = L.terrain(m1) & 7u;

    // ---- resolve_interacts (mdp.py:1432-1579)
    uint32_t useful_pots = 0, n_full = 0;  // pot_states before any interact (mdp.py:1439): ready / cooking / 1..2 idle items
    uint32_t ps_before[MAXP];
#pragma unroll
    for (int k = 0; k < MAXP; ++k) {
        useful_pots += ((s.pc[k] != PC_EMPTY) & (s.pc[k] != PC_IDLE3)) ? 1u : 0u;
        if (EVENTS) n_full += (((uint32_t)k < C.n_pots) & (s.pc[k] >= PC_IDLE3)) ? 1u : 0u;
        ps_before[k] = s.ps[k];
    }
    const bool act0 = a0 == OC_A_INTERACT, act1 = two & (a1 == OC_A_INTERACT);
    const uint32_t h0_before = s.held0, h1_before = s.held1;
    const IOut3 r0 = interact3<MAXP, false>(L, lut, act0, s.held0, c_f0, s.ps, s.tk, s.pc);
    IOut3 r1 = interact3<MAXP, false>(L, lut, act1, s.held1, c_f1, s.ps, s.tk, s.pc);
    // loose dishes on counters: only a dish taken from a dispenser asks (wave-uniform branch; the planes are still in registers)
    // (EVENTS: always — a dish picked up from a counter asks too, for its USEFUL_DISH_PICKUP event)
    if (EVENTS || __builtin_amdgcn_ballot_w64(((r0.flags | r1.flags) & LF_TAKE_DISH) != 0u) != 0ull) {
        uint32_t dishes = 0;
#pragma unroll
        for (int p = 0; p < STEP1_MAX_PLANES; ++p)
            if (p < n_obj) dishes += count_dish_bytes(v[p].x) + count_dish_bytes(v[p].y) + count_dish_bytes(v[p].z) + count_dish_bytes(v[p].w);
        s.dcount = (int32_t)dishes;
    }
    const bool du0 = two & (((s.held1 == OC_O_DISH) ? 1u : 0u) < useful_pots) & (s.dcount == 0);
    const float sh0 = ((r0.flags & LF_PLACE) ? C.rew_place : 0.f) + ((r0.flags & LF_PLATE) ? C.rew_soup : 0.f) +
                      ((((r0.flags & LF_TAKE_DISH) != 0u) & du0) ? C.rew_dish : 0.f);
    s.held0 = r0.new_h;
    s.dcount += r0.ddelta;
    apply_pot3<MAXP>(s, r0);
    const bool same_cell = f1 == f0;
    const bool swap0 = (r0.flags & LF_SWAP) != 0u;
    const uint32_t c_f1_live = (same_cell & swap0) ? ((c_f1 & 0xFF00u) | r0.cell_obj) : c_f1;
    const bool conflict = act1 & ((same_cell & swap0) | (((r0.flags & LF_POT_UPD) != 0u) &
                                                          (((c_f1 >> 8) & 7u) == OC_T_POT) & ((c_f1 >> 11) == r0.slot)));
    if (__builtin_expect(conflict, 0)) r1 = interact3<MAXP, false>(L, lut, act1, s.held1, c_f1_live, s.ps, s.tk, s.pc);
    const bool du1 = two & (((s.held0 == OC_O_DISH) ? 1u : 0u) < useful_pots) & (s.dcount == 0);
    const float sh1 = ((r1.flags & LF_PLACE) ? C.rew_place : 0.f) + ((r1.flags & LF_PLATE) ? C.rew_soup : 0.f) +
                      ((((r1.flags & LF_TAKE_DISH) != 0u) & du1) ? C.rew_dish : 0.f);
    if (EVENTS) {  // event_infos (mdp.py:2121-2308) from the two outcomes, as step3_main does
        auto faced = [&](uint32_t c16, const uint32_t (&ps)[MAXP]) {  // the object an interact sees: the soup for a pot
            uint32_t o = c16 & 0xFFu;
            if (((c16 >> 8) & 7u) == OC_T_POT) {
#pragma unroll
                for (int k = 0; k < MAXP; ++k) o = (c16 >> 11) == (uint32_t)k ? ps[k] : o;
            }
            return o;
        };
        const uint32_t t0_ = act0 ? (c_f0 >> 8) & 7u : 7u, t1_ = act1 ? (c_f1 >> 8) & 7u : 7u;
        const uint32_t o0 = faced(c_f0, ps_before), o1 = conflict ? faced(c_f1_live, s.ps) : faced(c_f1, ps_before);
        const bool disp0 = (t0_ == OC_T_ONION_DISP) | (t0_ == OC_T_TOMATO_DISP) | (t0_ == OC_T_DISH_DISP);
        const bool disp1 = (t1_ == OC_T_ONION_DISP) | (t1_ == OC_T_TOMATO_DISP) | (t1_ == OC_T_DISH_DISP);
        const int32_t dc_before = s.dcount - r0.ddelta;
        const bool du0e = two & (((h1_before == OC_O_DISH) ? 1u : 0u) < useful_pots) & (dc_before == 0);
        *ev = interact_events<0>(C, t0_, h0_before, o0, (r0.flags & LF_SWAP) != 0u, disp0 & (h0_before == 0u) & (r0.new_h != 0u),
                                 (r0.flags & LF_PLACE) != 0u, (r0.flags & LF_PLATE) != 0u, (r0.flags & LF_SERVE) != 0u,
                                 h1_before, du0e, n_full, two) |
              interact_events<1>(C, t1_, h1_before, o1, (r1.flags & LF_SWAP) != 0u, disp1 & (h1_before == 0u) & (r1.new_h != 0u),
                                 (r1.flags & LF_PLACE) != 0u, (r1.flags & LF_PLATE) != 0u, (r1.flags & LF_SERVE) != 0u,
                                 s.held0, du1, n_full, two);
    }
    s.held1 = r1.new_h;
    apply_pot3<MAXP>(s, r1);
    const bool swap1 = (r1.flags & LF_SWAP) != 0u;
    // (counter cells: the faced cell after a pick-up / drop; player 1's result stands when both changed the same cell)
    q.f0 = f0; q.f1 = f1; q.o0 = r0.cell_obj; q.o1 = r1.cell_obj;
    q.w0 = swap0 & !(same_cell & swap1);
    q.w1 = swap1;
    // deliver_soup (mdp.py:1631-1642): the recipe-value look-ups behind a wave-uniform branch
    float sp0 = 0.f, sp1 = 0.f;
    const bool serve0 = (r0.flags & LF_SERVE) != 0u, serve1 = (r1.flags & LF_SERVE) != 0u;
    if (__builtin_expect(__builtin_amdgcn_ballot_w64(serve0 | serve1) != 0ull, 0)) {
        sp0 = serve0 ? L.value(recipe_idx(h0_before) & 15u) : 0.f;
        sp1 = serve1 ? L.value(recipe_idx(h1_before) & 15u) : 0.f;
    }
    r = make_float4(sp0, sp1, sh0, sh1);

    // ---- resolve_movement (mdp.py:1644-1727)
    const uint32_t np0 = (mv0 & (t_m0 == OC_T_FLOOR)) ? m0 : s.pos0;
    const uint32_t np1 = (mv1 & (t_m1 == OC_T_FLOOR)) ? m1 : s.pos1;
    s.or0 = mv0 ? a0 : s.or0;
    s.or1 = mv1 ? a1 : s.or1;
    const bool collide = two & ((np0 == np1) | ((np0 == s.pos1) & (np1 == s.pos0)));
    s.pos0 = collide ? s.pos0 : np0;
    s.pos1 = collide ? s.pos1 : np1;
    step3_env<MAXP>(C, s);  // step_environment_effects (mdp.py:1691-1703)
}

// the env's next episode: the layout's standard start state, or one drawn by the batch's start_state_fn
// (get_random_start_state_fn, mdp.py:1307-1353) — players, hands, pots; nothing lies on the counters
template <int MAXP>
__device__ __forceinline__ void one_restart(const LayC& C, const Lay L, const StartArgs& sa, uint64_t g, Env3<MAXP>& s) {
    s.pos0 = L.u8(L_START_POS); s.pos1 = L.u8(L_START_POS + 1);
    s.or0 = L.u8(L_START_OR); s.or1 = s.pos1 == 0xFFu ? 0u : L.u8(L_START_OR + 1);
    s.held0 = s.held1 = 0; s.t = 0; s.dcount = 0;
#pragma unroll
    for (int k = 0; k < MAXP; ++k) { s.ps[k] = 0; s.tk[k] = 0; s.pc[k] = PC_EMPTY; }
    if (sa.enabled) {
        const StartDraw d = draw_start(L, g, sa.epoch, sa.seed_lo, sa.seed_hi, sa.random_start_pos, sa.thresh);
        s.pos0 = d.pos0; s.pos1 = d.pos1; s.held0 = d.held0; s.held1 = d.held1;
#pragma unroll
        for (int k = 0; k < MAXP; ++k) {
            if ((uint32_t)k < C.n_pots) {
                s.ps[k] = d.pot_obj((uint32_t)k);
                s.tk[k] = d.tick((uint32_t)k);
            }
        }
    }
}

// write-back: the header always; the planes from scratch for a restarted env, else only the object bytes that changed
// OOP (state_out != state_in): the changed bytes go into the lane's own LDS rows (`lrow`, where its planes are parked) and
// every plane of the new state is written from there.
template <int MAXP>
__device__ __forceinline__ void one_store(const LayC& C, const Lay L, uint4* st, int64_t n, int64_t e, int n_obj,
                                          const One<MAXP>& q, bool restarted, uint8_t* lrow = nullptr) {
    const Env3<MAXP>& s = q.s;
    uint8_t* gbytes = reinterpret_cast<uint8_t*>(st);
    if (lrow != nullptr) {  // out of place (wave-uniform: a property of the call)
        auto put = [&](uint32_t c, uint32_t o) __attribute__((always_inline)) {
            lrow[(c >> 4) * (uint32_t)(BLOCK * 16) + (c & 15u)] = (uint8_t)o;
        };
        uint4 ho;
        ho.x = s.pos0 | (s.or0 << 8) | (s.held0 << 16) | (s.pos1 << 24);
        ho.y = s.or1 | (s.held1 << 8) | (min(s.t, 0xFFFFu) << 16);
        ho.z = 0; ho.w = 0;
#pragma unroll
        for (int k = 0; k < MAXP; ++k) {
            if ((uint32_t)k < C.n_pots) {
                if (k < 4) ho.z |= s.tk[k] << (8 * (k & 3));
                else ho.w |= s.tk[k] << (8 * (k & 3));
            }
        }
        st[e] = ho;
        if (restarted) {
#pragma unroll
            for (int p = 0; p < STEP1_MAX_PLANES; ++p)
                if (p < n_obj) *reinterpret_cast<uint4*>(lrow + p * (BLOCK * 16)) = make_uint4(0u, 0u, 0u, 0u);
#pragma unroll
            for (int k = 0; k < MAXP; ++k)
                if ((uint32_t)k < C.n_pots && s.ps[k] != 0u) put(L.pot_cell(k), s.ps[k]);
        } else {
            if (q.w0) put(q.f0, q.o0);
            if (q.w1) put(q.f1, q.o1);
#pragma unroll
            for (int k = 0; k < MAXP; ++k)
                if ((uint32_t)k < C.n_pots && s.ps[k] != q.ps_in[k]) put(L.pot_cell(k), s.ps[k]);
        }
#pragma unroll
        for (int p = 0; p < STEP1_MAX_PLANES; ++p)  // (this lane's own rows: LDS operations of one wavefront execute in order)
            if (p < n_obj) st[(int64_t)(1 + p) * n + e] = *reinterpret_cast<const uint4*>(lrow + p * (BLOCK * 16));
        return;
    }
    auto store_obj = [&](uint32_t c, uint32_t o) __attribute__((always_inline)) {
        gbytes[((int64_t)(1 + (c >> 4)) * n + e) * 16 + (c & 15u)] = (uint8_t)o;
    };
    uint4 ho;
    ho.x = s.pos0 | (s.or0 << 8) | (s.held0 << 16) | (s.pos1 << 24);
    ho.y = s.or1 | (s.held1 << 8) | (min(s.t, 0xFFFFu) << 16);  // the wire format's u16 timestep saturates
    ho.z = 0; ho.w = 0;
#pragma unroll
    for (int k = 0; k < MAXP; ++k) {
        if ((uint32_t)k < C.n_pots) {
            if (k < 4) ho.z |= s.tk[k] << (8 * (k & 3));
            else ho.w |= s.tk[k] << (8 * (k & 3));
        }
    }
    st[e] = ho;
    if (restarted) {
        const uint4 z = make_uint4(0u, 0u, 0u, 0u);
#pragma unroll
        for (int p = 0; p < STEP1_MAX_PLANES; ++p)
            if (p < n_obj) st[(int64_t)(1 + p) * n + e] = z;
#pragma unroll
        for (int k = 0; k < MAXP; ++k)
            if ((uint32_t)k < C.n_pots && s.ps[k] != 0u) store_obj(L.pot_cell(k), s.ps[k]);  // (same lane, after the zeros)
    } else {
        if (q.w0) store_obj(q.f0, q.o0);
        if (q.w1) store_obj(q.f1, q.o1);
#pragma unroll
        for (int k = 0; k < MAXP; ++k)
            if ((uint32_t)k < C.n_pots && s.ps[k] != q.ps_in[k]) store_obj(L.pot_cell(k), s.ps[k]);
    }
}

template <bool UNIFORM, int MAXP, bool LAY_LDS, bool EVENTS = false>
__global__ __launch_bounds__(BLOCK) void k_step1(const OcLayout* __restrict__ g_layouts, int n_layouts,
                                                 const uint16_t* layout_id, uint4* st, uint4* st_out,
                                                 const uint8_t* __restrict__ actions, float4* __restrict__ rewards,
                                                 uint8_t* __restrict__ flags, float4* ep_returns, int64_t n, int W,
                                                 int n_obj, int horizon, uint32_t options, StartArgs sa, EvArgs ea) {
    extern __shared__ __attribute__((aligned(16))) uint4 s_rows1[];  // [n_obj][BLOCK]: the object planes, one 16-byte row per lane
    __shared__ uint4 s_lay[LAY_LDS ? (UNIFORM ? 16 : LDS_LAYOUT_MAX * 16) : 1];
    __shared__ uint2 s_lut[2 * LUT_ENTRIES];
    const int64_t e = (int64_t)blockIdx.x * BLOCK + threadIdx.x;
    const bool active = e < n;
    const int64_t el = active ? e : n - 1;  // (the lanes past the batch load valid addresses and leave after the barrier)
    const OneIn in = one_load(st, actions, ep_returns, n, el, n_obj);  // ---- one round trip: everything the step reads
    for (int i = threadIdx.x; i < 2 * LUT_ENTRIES; i += BLOCK) s_lut[i] = reinterpret_cast<const uint2*>(&g_lut)[i];
    Lay L = stage_layouts<LAY_LDS>(g_layouts, n_layouts, layout_id, e, active, s_lay);  // contains the barrier
    if (!active) return;
#pragma unroll
    for (int p = 0; p < STEP1_MAX_PLANES; ++p)
        if (p < n_obj) s_rows1[p * BLOCK + threadIdx.x] = in.v[p];  // read back by this lane only: no barrier
    const uint8_t* row = reinterpret_cast<const uint8_t*>(s_rows1 + threadIdx.x);
    LayC C = load_consts<UNIFORM>(L);
    const uint8_t* lut = reinterpret_cast<const uint8_t*>(s_lut) + (C.old_dyn ? LUT_ENTRIES * 8 : 0);
    One<MAXP> q;
    one_decode<MAXP>(C, L, in.h, row, q);
    const uint32_t a0 = in.a01 & 0xFFu, a1 = in.a01 >> 8;
    float4 r = make_float4(0.f, 0.f, 0.f, 0.f), ep = in.ep;
    if (__builtin_expect(a0 > 5u || a1 > 5u, 0)) {  // get_state_transition raises ValueError (mdp.py:1394-1398): the env stays as it is
        rewards[e] = r;
        flags[e] = (uint8_t)OC_F_BAD_ACTION;
        if (EVENTS && ea.events) ea.events[e] = 0;
        if (st_out != st) {
            st_out[e] = in.h;
#pragma unroll
            for (int p = 0; p < STEP1_MAX_PLANES; ++p)
                if (p < n_obj) st_out[(int64_t)(1 + p) * n + e] = in.v[p];
        }
        return;
    }
    uint64_t ev = 0;
    one_transition<MAXP, EVENTS>(C, L, lut, make_delta4(W), a0, a1, in.v, n_obj, row, q, r, &ev);
    // ---- OvercookedEnv.step bookkeeping (env.py:266-267, 321-325)
    ep.x += r.x; ep.y += r.y; ep.z += r.z; ep.w += r.w;
    uint32_t fl = 0;
    bool restarted = false;
    if (__builtin_expect((int)q.s.t >= horizon, 0)) {  // once per episode
        fl |= OC_F_DONE;
        if (options & OC_OPT_AUTO_RESET) {
            const uint64_t g = (uint64_t)(sa.env_offset + e);
            uint32_t lid;
            regen_layout<UNIFORM, LAY_LDS>(sa, g, sa.epoch, e, s_lay, g_layouts, L, C, &lid);  // (regen_mdp: the next episode's layout)
            one_restart<MAXP>(C, L, sa, g, q.s);
            ep = make_float4(0.f, 0.f, 0.f, 0.f);
            fl |= OC_F_RESET;
            restarted = true;
        }
    }
    if (EVENTS) {
        if (ea.events) ea.events[e] = ev;
        count_events(ea, e, ev, (fl & OC_F_DONE) != 0u, (fl & OC_F_RESET) != 0u);
    }
    one_store<MAXP>(C, L, st_out, n, e, n_obj, q, restarted, st_out != st ? reinterpret_cast<uint8_t*>(s_rows1 + threadIdx.x) : nullptr);
    rewards[e] = r;
    flags[e] = (uint8_t)fl;
    if (ep_returns) ep_returns[e] = ep;
}
