// step_server.hpp — k_step_server: a RESIDENT batched step (no launch per step), and k_step_client, its device-side caller
// Part of liboc_amd.so: included by oc_amd.hip inside its anonymous namespace after step_table.hpp (env_step3, finish_step3).
#pragma once

// ==========================================================================================
// OvercookedEnv.step (overcooked_env.py:244-274) for a whole batch, called once per policy decision: as a launch per step
// (oc_step, k_step1) it costs 4.5 us for 0.55 us of HBM work — a dependent kernel boundary (1.5-1.9 us on this chip) plus the
// kernel's own chain (state loads -> transition -> state stores).  k_step_server takes both away for callers that live on the
// GPU themselves (a persistent policy kernel, or the last kernel of a policy's forward pass): the envs stay with their lanes —
// registers + [cell][lane] words in LDS, exactly k_step3's loop — and every step is ONE hand-off each way through per-env
// mailboxes in HBM, the price list's cheapest primitive (MI355X_MICROARCH.md "handoff-1to1": data-tagged granules written by ONE
// write-through store, 0.8-1.0 us per hop idle; a separate flag behind a drained store costs 1.7-1.9 x that):
//   request  [n_envs] x 8 bytes   {u32 a0 | a1 << 8 | command << 16, u32 tag}      one `global_store_dwordx2 sc1` by the caller
//   response [n_envs] x 32 bytes  {r.x, r.y, r.z, tag} {r.w, flags, timestep, tag}  two `global_store_dwordx4 sc1` by the server
// tag = 1 + the number of steps the server has served since it was opened.  A lane polls ITS env's request with one `sc1` load
// (payload and tag arrive together), a wavefront steps when all its envs show the expected tag, and answers without waiting for
// anything: a granule that shows the tag is complete (an aligned 8- / 16-byte store is one write to one line).  No workgroup
// barrier, no counter, no fence in the loop; wavefronts serve their 64 envs independently of each other.
// The kernel never outlives its usefulness (as k_mailbox): it leaves on SV_STOP, after idle_ticks of wall_clock64 without a
// request, or after life_ticks in total — writing the states back, so that every other oc_* call finds them in d_state — and
// says so in a host-visible word per wavefront; the host side (oc_amd.hip: oc_step_server_*) relaunches it when needed.
// Not served: event sinks (OcEventSink), OC_OPT_PREDICATE_INTERACT.
// ==========================================================================================
constexpr uint32_t SV_STOP = 0x10000u;  // command bit of a request: write the states back and leave (no response)

typedef uint32_t sv_u32x2 __attribute__((ext_vector_type(2)));
typedef uint32_t sv_u32x4 __attribute__((ext_vector_type(4)));
// device-scope (sc1) accesses: past this XCD's L2 to the memory side, where every XCD sees them; complete on return
__device__ __forceinline__ sv_u32x2 sv_load8(const void* p) {
    sv_u32x2 v;
    asm volatile("global_load_dwordx2 %0, %1, off sc1\n\ts_waitcnt vmcnt(0)" : "=v"(v) : "v"(p) : "memory");
    return v;
}
__device__ __forceinline__ void sv_load16x2(const void* p, sv_u32x4& a, sv_u32x4& b) {
    asm volatile("global_load_dwordx4 %0, %2, off sc1\n\tglobal_load_dwordx4 %1, %2, off offset:16 sc1\n\ts_waitcnt vmcnt(0)"
                 : "=&v"(a), "=&v"(b) : "v"(p) : "memory");
}
__device__ __forceinline__ void sv_store8(void* p, sv_u32x2 v) {
    asm volatile("global_store_dwordx2 %0, %1, off sc1" : : "v"(p), "v"(v) : "memory");
}
__device__ __forceinline__ void sv_store16x2(void* p, sv_u32x4 a, sv_u32x4 b) {
    asm volatile("global_store_dwordx4 %0, %1, off sc1\n\tglobal_store_dwordx4 %0, %2, off offset:16 sc1" : : "v"(p), "v"(a), "v"(b) : "memory");
}

// the host-visible control words (pinned, GPU-mapped host memory): [0 .. 4 grid) 1 while WAVEFRONT w of workgroup b serves — each
// wavefront comes and goes by itself, so each reports by itself (round 6: with one word per workgroup a wavefront that had left
// for idleness hid behind three siblings that saw the caller's keep-alive and stayed; found by the stress test) —; then
// [grid + SV_ERR_CLIENT]: a client wavefront gave up waiting for its responses; [grid + SV_KEEPALIVE]: bumped by the host before it
// launches a client (a workgroup about to leave looks at it first)
enum { SV_ERR_CLIENT = 0, SV_KEEPALIVE = 1, SV_ERR_WORDS = 4 };

// L2-scope load (sc0: past this CU's vector cache, from the XCD's L2): sees a write-through store made on the SAME XCD ~3x sooner
// than a device-scope load, and never one made on another XCD — used only between looks at device scope
__device__ __forceinline__ sv_u32x2 sv_load8_near(const void* p) {
    sv_u32x2 v;
    asm volatile("global_load_dwordx2 %0, %1, off sc0\n\ts_waitcnt vmcnt(0)" : "=v"(v) : "v"(p) : "memory");
    return v;
}
__device__ __forceinline__ void sv_load16x2_near(const void* p, sv_u32x4& a, sv_u32x4& b) {
    asm volatile("global_load_dwordx4 %0, %2, off sc0\n\tglobal_load_dwordx4 %1, %2, off offset:16 sc0\n\ts_waitcnt vmcnt(0)"
                 : "=&v"(a), "=&v"(b) : "v"(p) : "memory");
}
__device__ __forceinline__ uint32_t sv_xcc() { return __builtin_amdgcn_s_getreg((3 << 11) | 20) & 0xFu; }  // HW_REG_XCC_ID, bits 0..3

// Which block of 256 envs a workgroup serves: one of the blocks of the XCD it RUNS on — XCD x owns blocks [x g / 8, (x + 1) g / 8)
// — claimed from a device-scope counter per XCD (claims[0..7]; a workgroup whose own XCD is full takes a free block of the next
// one).  The dispatcher deals consecutive workgroups to consecutive XCDs, but where a grid's first workgroup lands depends on
// what ran before (measured: every server / client pair of one launch sat on neighbouring XCDs with blockIdx-based blocks), so
// both kernels claim by the XCD they really run on: where the two ends of an env sit relative to each other is then a choice
// (`shift`: the client takes blocks of the XCD `shift` places from its own; oc_amd.hip sv_knobs has the measurements — at 256
// pairs the hand-offs are fastest with the ends four XCDs apart).  The whole workgroup gets the answer through LDS.
__device__ __forceinline__ uint32_t sv_claim_block(uint32_t* claims, uint32_t* s_blk, uint32_t shift = 0u) {
    if (threadIdx.x == 0) {
        const uint32_t g = gridDim.x, x0 = (sv_xcc() + shift) & 7u;
        uint32_t blk = 0xFFFFFFFFu;
        for (uint32_t i = 0; i < 8u && blk == 0xFFFFFFFFu; ++i) {
            const uint32_t x = (x0 + i) & 7u, lo = x * g / 8u, cap = (x + 1u) * g / 8u - lo;
            if (cap == 0u) continue;
            const uint32_t slot = __hip_atomic_fetch_add(claims + x, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (slot < cap) blk = lo + slot;
        }
        *s_blk = blk;
    }
    __syncthreads();
    return *s_blk;
}

// n x 64 clocks of sleep (s_sleep takes an immediate)
__device__ __forceinline__ void sv_nap(uint32_t n) {
    for (uint32_t i = 0; i < n; ++i) __builtin_amdgcn_s_sleep(1);
}

template <bool UNIFORM, int MAXP, bool LAY_LDS>
__global__ __launch_bounds__(BLOCK) void k_step_server(const OcLayout* __restrict__ g_layouts, int n_layouts,
                                                       const uint16_t* layout_id, uint4* st, float4* ep_returns,
                                                       const uint64_t* req, uint4* rsp, uint32_t* ctl, uint32_t* claims, int64_t n, int W,
                                                       int n_obj, int horizon, uint32_t options, StartArgs sa,
                                                       uint64_t idle_ticks, uint64_t life_ticks, uint32_t knobs) {
    extern __shared__ __attribute__((aligned(16))) uint16_t s_cells3[];  // [n_obj * 16][BLOCK]
    __shared__ uint4 s_lay[LAY_LDS ? (UNIFORM ? 16 : LDS_LAYOUT_MAX * 16) : 1];
    __shared__ uint2 s_lut[2 * LUT_ENTRIES];
    __shared__ uint32_t s_blk;
    const uint32_t blk = sv_claim_block(claims, &s_blk);  // (contains a barrier)
    const uint32_t my_xcc = sv_xcc();
    const int64_t e = (int64_t)blk * BLOCK + threadIdx.x;
    const bool active = e < n;
    const int64_t el = active ? e : n - 1;
    for (int i = threadIdx.x; i < 2 * LUT_ENTRIES; i += BLOCK) s_lut[i] = reinterpret_cast<const uint2*>(&g_lut)[i];
    Lay L = stage_layouts<LAY_LDS>(g_layouts, n_layouts, layout_id, el, true, s_lay);  // contains the barrier
    uint16_t* cells = s_cells3 + threadIdx.x;
    bool near = false;  // the caller's last request came from this XCD: look through the L2 between looks at device scope
    LayC C = load_consts<UNIFORM>(L);
    const uint8_t* lut = reinterpret_cast<const uint8_t*>(s_lut) + (C.old_dyn ? LUT_ENTRIES * 8 : 0);
    const uint32_t delta4 = make_delta4(W);
    Env3<MAXP> s;
    load_env3<MAXP>(C, L, st, n, el, n_obj, s, cells);  // (lanes beyond the batch carry a copy of the last env and never answer)
    const uint64_t g = (uint64_t)(sa.env_offset + el);
    float4 ep = ep_returns ? ep_returns[el] : make_float4(0.f, 0.f, 0.f, 0.f);
    const uint64_t* my_req = req + el;
    const uint64_t* first_req = req + (el & ~(int64_t)63);  // (wave-uniform: the wavefront's first env)
    uint4* my_rsp = rsp + 2 * el;
    // the last request this env was answered (by an earlier incarnation of the kernel): its second response granule's tag
    uint32_t expect;
    {
        sv_u32x4 a, b;
        sv_load16x2(my_rsp, a, b);
        expect = (uint32_t)__builtin_amdgcn_readfirstlane((int)b.w) + 1u;  // (a wavefront's envs are always served together)
    }
    const uint64_t born = wall_clock64();
    uint64_t last = born;
    uint32_t* const words = ctl + (size_t)gridDim.x * (BLOCK / 64u);  // behind the wavefronts' flags: the error / keep-alive words
    uint32_t keep = __hip_atomic_load(words + SV_KEEPALIVE, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    const bool wave_serves = __ballot(active) != 0ull;  // (a wavefront wholly beyond the batch has nobody to answer)
    uint32_t* const my_flag = ctl + (size_t)blockIdx.x * (BLOCK / 64u) + (threadIdx.x >> 6);
    if (wave_serves && (threadIdx.x & 63u) == 0u) __hip_atomic_store(my_flag, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);  // "serving"
    while (wave_serves) {
        // ---- the request: payload and tag in one load per lane
        sv_u32x2 q;
        bool leave = false;
        sv_nap(knobs & 0xFFu);  // (the caller cannot have answered yet)
        for (uint32_t look_i = 0;; ++look_i) {
            bool look = true;
            if (knobs & 0x10000u) {  // light poll: the wavefront's first env alone (one line), the others once it shows the tag
                const sv_u32x2 q0 = sv_load8(first_req);
                look = q0.y == expect;
            }
            if (look) {
                q = (near && (look_i & 3u) != 3u) ? sv_load8_near(my_req) : sv_load8(my_req);
                if (__ballot(!active || q.y == expect) == __ballot(true)) break;
            }
            const uint64_t now = wall_clock64();
            // leaving: idle for idle_ticks, or past life_ticks and not inside a burst of requests (200 us without one) — unless a
            // host-side caller has announced itself since the last look at the keep-alive word (oc_amd.hip: sv_ensure bumps it
            // BEFORE it checks who is resident, so a client launched after that check finds every workgroup it counted)
            if (now - last > idle_ticks || (now - born > life_ticks && now - last > 20000u)) {
                const uint32_t k = __hip_atomic_load(words + SV_KEEPALIVE, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
                if (k == keep) { leave = true; break; }
                keep = k;
                last = now;
            }
            sv_nap((knobs >> 8) & 0xFFu);
        }
        if (leave || __ballot(active && (q.x & SV_STOP) != 0u) != 0ull) break;
        {   // where the caller runs (bits 20..23 of the request, valid with bit 24): wave-uniform, from the first env served
            const uint32_t q0 = (uint32_t)__builtin_amdgcn_readfirstlane((int)q.x);
            near = !(knobs & 0x20000u) && (q0 & 0x1000000u) != 0u && ((q0 >> 20) & 0xFu) == my_xcc;
        }
#ifdef OC_AMD_TUNING
        const uint64_t tm_seen = mb_now();
#endif
        const uint32_t a0 = q.x & 0xFFu, a1 = (q.x >> 8) & 0xFFu;
        float4 r = make_float4(0.f, 0.f, 0.f, 0.f);
        uint32_t fl;
        if (__builtin_expect(a0 > 5u || a1 > 5u, 0)) {
            fl = OC_F_BAD_ACTION;  // get_state_transition raises ValueError (mdp.py:1394-1398): leave the env untouched
        } else {
            uint64_t ev = 0;
            const uint32_t k = expect - 1u;  // steps served since the server was opened: the epoch of a drawn restart
            env_step3<MAXP, 0, false>(C, L, lut, cells, s, delta4, a0, a1, r, 0ull, nullptr, &ev);
            fl = finish_step3<MAXP>(C, L, n_obj, cells, s, horizon, options, r, ep, sa, g, sa.epoch + k, [&]() {
                uint32_t lid;
                if (regen_layout<UNIFORM, LAY_LDS>(sa, g, sa.epoch + k, el, s_lay, g_layouts, L, C, &lid)) {
                    lut = reinterpret_cast<const uint8_t*>(s_lut) + (C.old_dyn ? LUT_ENTRIES * 8 : 0);
                    for (int c = 0; c < n_obj * 16; ++c) cells[c * BLOCK] = (uint16_t)(L.terrain((uint32_t)c) << 8);
                }
            });
        }
        // ---- the response: two self-validating 16-byte granules, written through
        if (active) {
            const sv_u32x4 ga = {__float_as_uint(r.x), __float_as_uint(r.y), __float_as_uint(r.z), expect};
#ifdef OC_AMD_TUNING
            // (tuning: ticks from request seen to response sent in the low half instead of the timestep)
            const sv_u32x4 gb = {__float_as_uint(r.w), fl, ((uint32_t)(mb_now() - tm_seen) & 0xFFFFu) | (my_xcc << 20) | 0x1000000u, expect};
#else
            const sv_u32x4 gb = {__float_as_uint(r.w), fl, min(s.t, 0xFFFFu) | (my_xcc << 20) | 0x1000000u, expect};
#endif
            sv_store16x2(my_rsp, ga, gb);
        }
        ++expect;
        last = wall_clock64();
    }
    // ---- leaving: the states go back to d_state (visible to everyone when the kernel ends)
    if (active) {
        store_env3<MAXP>(C, L, st, n, e, n_obj, s, cells);
        if (ep_returns) ep_returns[e] = ep;
    }
    if (wave_serves && (threadIdx.x & 63u) == 0u) __hip_atomic_store(my_flag, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);  // "gone"
}

// k_step_client: the caller's side of the protocol as a kernel — lane = env, K steps: post the request (the caller's action
// bytes of step k), poll the two response granules, leave rewards / flags of step k in the caller's [K][n] arrays.  With K = 1 it
// is oc_step_server_step; with K > 1 it measures the round trip and replays oc_step_many's inputs for the parity tests.  A policy
// that lives on the GPU does what this kernel does around its own arithmetic.
__global__ __launch_bounds__(BLOCK) void k_step_client(uint64_t* req, const uint4* rsp, const uint8_t* __restrict__ actions,
                                                       float4* __restrict__ rewards, uint8_t* __restrict__ flags, uint32_t* ctl_err, uint32_t* claims,
                                                       int64_t n, uint32_t tag0, int n_steps, uint64_t timeout_ticks, uint32_t knobs, uint32_t* dbg) {
    __shared__ uint32_t s_blk;
    const uint32_t blk = sv_claim_block(claims, &s_blk, (knobs >> 24) & 7u);  // (a block of the XCD opposite this workgroup's: see sv_knobs)
    const uint32_t my_xcc = sv_xcc();
    const int64_t e = (int64_t)blk * BLOCK + threadIdx.x;
    const bool active = e < n;
    const int64_t el = active ? e : n - 1;
    bool near = false;  // the server of these envs answered from this XCD
    const uint16_t* act = reinterpret_cast<const uint16_t*>(actions) + el;
    uint32_t a01 = n_steps > 0 ? act[0] : 0u;
    sv_u32x4 ga = {0u, 0u, 0u, 0u}, gb = {0u, 0u, 0u, 0u};
    for (int k = 0; k < n_steps; ++k) {
        const uint32_t tag = tag0 + (uint32_t)k;
#ifdef OC_AMD_TUNING
        const uint64_t tm_post = mb_now();
#endif
        if (active) sv_store8(req + e, sv_u32x2{a01 | (my_xcc << 20) | 0x1000000u, tag});
        // in the hand-off's shadow: the previous step's outputs go to the caller's arrays, the next step's action bytes arrive
        if (active && k > 0) {
            rewards[(int64_t)(k - 1) * n + e] = make_float4(__uint_as_float(ga.x), __uint_as_float(ga.y), __uint_as_float(ga.z), __uint_as_float(gb.x));
            flags[(int64_t)(k - 1) * n + e] = (uint8_t)gb.y;
        }
        if (k + 1 < n_steps) a01 = act[(int64_t)(k + 1) * n];
        const uint64_t t0 = wall_clock64();
        sv_nap(knobs & 0xFFu);  // (the server cannot have answered yet)
        for (uint32_t look_i = 0;; ++look_i) {
            bool look = true;
            if (knobs & 0x10000u) {  // light poll: the second granule of the wavefront's first env alone, the rest once it shows the tag
                sv_u32x4 g0;
                asm volatile("global_load_dwordx4 %0, %1, off offset:16 sc1\n\ts_waitcnt vmcnt(0)" : "=v"(g0) : "v"(rsp + 2 * (el & ~(int64_t)63)) : "memory");
                look = g0.w == tag;
            }
            if (look) {
                if (near && (look_i & 3u) != 3u) sv_load16x2_near(rsp + 2 * el, ga, gb); else sv_load16x2(rsp + 2 * el, ga, gb);
                if (__ballot(!active || (ga.w == tag && gb.w == tag)) == __ballot(true)) break;
            }
            if (wall_clock64() - t0 > timeout_ticks) {  // (the server has left or never came: say so instead of hanging the GPU)
                if ((threadIdx.x & 63u) == 0u) __hip_atomic_store(ctl_err + SV_ERR_CLIENT, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
                return;
            }
            sv_nap((knobs >> 8) & 0xFFu);
        }
        {
            const uint32_t z0 = (uint32_t)__builtin_amdgcn_readfirstlane((int)gb.z);
            near = !(knobs & 0x20000u) && (z0 & 0x1000000u) != 0u && ((z0 >> 20) & 0xFu) == my_xcc;
        }
#ifdef OC_AMD_TUNING
        if (dbg && threadIdx.x == 0) {  // per workgroup: the sums of both times, where its two ends ran
            if (k == 0) { dbg[4 * blockIdx.x] = 0; dbg[4 * blockIdx.x + 1] = 0; }
            dbg[4 * blockIdx.x] += (uint32_t)(mb_now() - tm_post);
            dbg[4 * blockIdx.x + 1] += gb.z & 0xFFFFu;
            dbg[4 * blockIdx.x + 2] = (gb.z >> 20) & 0xFu;
            dbg[4 * blockIdx.x + 3] = my_xcc;
        }
#endif
    }
    if (threadIdx.x == 0 && __hip_atomic_fetch_add(claims + 8, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) + 1u == gridDim.x)
        for (int i = 0; i < 9; ++i) __hip_atomic_store(claims + i, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);  // the last one out: for the next launch
    if (active && n_steps > 0) {
        rewards[(int64_t)(n_steps - 1) * n + e] = make_float4(__uint_as_float(ga.x), __uint_as_float(ga.y), __uint_as_float(ga.z), __uint_as_float(gb.x));
        flags[(int64_t)(n_steps - 1) * n + e] = (uint8_t)gb.y;
    }
}

// every env's request word <- {data, tag}: SV_STOP for the resident kernel, or a no-op request (the tag already served) that
// clears a stale STOP before the next incarnation starts
__global__ __launch_bounds__(BLOCK) void k_step_server_post(uint64_t* req, int64_t n, uint32_t data, const uint4* rsp, uint32_t tag_add) {
    const int64_t e = (int64_t)blockIdx.x * BLOCK + threadIdx.x;
    if (e >= n) return;
    sv_u32x4 a, b;
    sv_load16x2(rsp + 2 * e, a, b);  // the env's last served tag
    sv_store8(req + e, sv_u32x2{data, b.w + tag_add});
}
