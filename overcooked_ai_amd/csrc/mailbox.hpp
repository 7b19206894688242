// mailbox.hpp — k_mailbox: a resident one-wavefront kernel that serves single-env transitions from a pinned host mailbox
// Part of liboc_amd.so: included by oc_amd.hip inside its anonymous namespace after step_one.hpp (one_decode, one_transition).
#pragma once

// ==========================================================================================
// The reference's OvercookedEnv.step (overcooked_env.py:244) / get_state_transition (overcooked_mdp.py:1375) step ONE env
// per call.  Served by a kernel launch per call, that costs launch + stream wait (~17 us of a 27 us step) however few
// bytes move.  k_mailbox removes the launch from the call: one wavefront stays resident, polls a request word in pinned,
// GPU-mapped host memory, runs the same transition as k_step1 (interact3 + 8-byte LUT + replay rule + movement + env
// effects, with the event mask) on the state it finds there, writes next state / rewards / flags / events back into the
// mailbox and publishes a response word; the host (oc_mailbox_step) spins on that word.  A call then costs two PCIe
// round trips (the GPU's poll sees the request, the host's poll sees the response) instead of a launch.
//
// Protocol (4 KiB host mailbox; what each form cost is in docs/NOTEBOOK.md, round 4):
//   host:  REQUEST GRANULES at MB_REQG — 16 bytes each, {12 payload bytes, u32 tag = seq}, payload = the state planes the caller
//          left at MB_IN followed by the two action bytes of MB_ACT, one aligned 16-byte store per granule
//          ... spins until every RESPONSE GRANULE carries seq -> copies the outputs out
//   GPU :  lanes 0..7 poll with ONE system-scope 16-byte load each (lane i: request granule i; s_sleep between polls); when
//          every granule shows the same new tag the request is complete AND already in registers (round 5: the payload
//          rides on the poll — one PCIe read round trip per request instead of poll + fence + payload reads, which
//          profiles/r05_mailbox_phases.txt priced at 1.3 us of a 7.2 us call) -> lane 0 steps -> the RESPONSE: two 64-byte
//          lines of {15 payload dwords, tag} at MB_RSPG, sent by lanes 0..7 with one store instruction (16 bytes per lane).
//          An aligned 16-byte access is one PCIe transaction on one cache line and PCIe writes arrive in order, so a request
//          granule that shows the tag has its payload too, and so has a response line whose last dword shows it: no flag
//          behind a release fence, no wait for write acknowledgements.
// Measured on the box (oc_mailbox_step alone, cramped_room): request word + response word behind a release fence 7.9 us;
// request word + fence + payload loads, response granules 7.0-7.2 us (round 4); all-granule requests read by ONE lane with 8-byte
// system-scope atomic loads, one after the other, 7.6 us; 8-byte granules both ways 11.1 us (13 reads + 19 fabric writes per step).  Stores that are not write-through (plain, non-temporal) stay in the GPU's L2 until the kernel's final
// release: every step then took the 2 ms idle timeout and was answered by the NEXT incarnation of the kernel.
// The kernel never outlives its usefulness: it leaves when the request word is MB_STOP, after idle_ticks of wall_clock64
// without a request, or after life_ticks in total, and says so (alive = 0); oc_mailbox_step relaunches it when needed.  It
// serves one layout (the mailbox's batch holds one layout record), any number of pots, grids of at most 64 cells.
// ==========================================================================================
constexpr uint32_t MB_STOP = 0xFFFFFFFFu;
// byte offsets inside the mailbox.  IN / ACT / OUT / REW / FLAGS / EV are the caller's plain views (include/oc_amd.h);
// oc_mailbox_step unpacks the response granules into OUT .. EV.
constexpr int MB_REQG = 1024 /* request granules */, MB_REQ_MAX = (80 + 2 + 11) / 12 /* 7 */, MB_ALIVE = 128, MB_IN = 256 /* 5 planes x 16 B */, MB_ACT = MB_IN + 80, MB_OUT = 512 /* 5 planes */,
              MB_REW = MB_OUT + 80, MB_FLAGS = MB_REW + 16, MB_EV = MB_FLAGS + 8, MB_RSPG = 2048, MB_BYTES = 4096;
constexpr int MB_RSP_BYTES = 80 + 16 + 4 + 8;  // at most 27 dwords: two response lines of {15 payload dwords, tag}

typedef uint32_t mb_u32x4 __attribute__((ext_vector_type(4)));
#ifdef OC_AMD_TUNING
__device__ __forceinline__ uint64_t mb_now() {  // the 100 MHz clock, read where the program says (the builtin may be merged with its neighbours)
    uint64_t t;
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_memrealtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t) : : "memory");
    return t;
}
#endif
__device__ __forceinline__ uint32_t mb_load(const uint32_t* p) {
    return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
}
// A request granule: ONE 16-byte load at system scope (sc0 sc1: past the GPU's caches, from host memory), complete on return
__device__ __forceinline__ mb_u32x4 mb_load_granule(const uint8_t* p) {
    mb_u32x4 v;
    asm volatile("global_load_dwordx4 %0, %1, off sc0 sc1\n\ts_waitcnt vmcnt(0)" : "=v"(v) : "v"(p) : "memory");
    return v;
}
// NOBJ: object planes of the grid (1..4; every index into the granule words is a compile-time constant); MAXP: pot slots
template <int NOBJ, int MAXP>
__global__ __launch_bounds__(64) void k_mailbox(const OcLayout* __restrict__ g_layout, uint8_t* mb, int W, int horizon,
                                                uint64_t idle_ticks, uint64_t life_ticks) {
    constexpr int n_obj = NOBJ;
    __shared__ uint4 s_rows[STEP1_MAX_PLANES * BLOCK];  // the lane's planes, [plane][BLOCK] rows of 16 bytes (one_obj's layout)
    __shared__ uint4 s_out[1 + STEP1_MAX_PLANES];       // the new state: header + planes
    __shared__ uint4 s_lay[16];
    __shared__ uint2 s_lut[2 * LUT_ENTRIES];
    __shared__ uint4 s_rsp[8];                          // the response: two 64-byte lines
    for (int i = threadIdx.x; i < 2 * LUT_ENTRIES; i += 64) s_lut[i] = reinterpret_cast<const uint2*>(&g_lut)[i];
    if (threadIdx.x < 16) s_lay[threadIdx.x] = reinterpret_cast<const uint4*>(g_layout)[threadIdx.x];
    __syncthreads();
    if (threadIdx.x >= 8) return;  // lanes 0..7 poll the request granules; lane 0 steps the env (wave-uniform ballots inside the transition see only that lane)
    const Lay L{reinterpret_cast<const uint8_t*>(s_lay)};
    const LayC C = load_consts<false>(L);
    const uint8_t* lut = reinterpret_cast<const uint8_t*>(s_lut) + (C.old_dyn ? LUT_ENTRIES * 8 : 0);
    // granules a request / a response of this layout really needs (n_obj object planes + the header)
    constexpr int n_state = 16 * (1 + n_obj);
    constexpr int n_req = (n_state + 2 + 11) / 12;
    static_assert(n_req <= 8 && n_req <= MB_REQ_MAX, "one request granule per polling lane");
    const uint8_t* const my_granule = mb + MB_REQG + 16 * min((int)threadIdx.x, n_req - 1);
    uint32_t served = mb_load(reinterpret_cast<const uint32_t*>(mb + MB_RSPG + 60));  // the last request answered (by an earlier incarnation)
    const uint64_t born = wall_clock64();
    uint64_t last = born;
#ifdef OC_AMD_TUNING
    uint64_t tm_prev = 0;
#endif
    for (;;) {
        const mb_u32x4 gq = mb_load_granule(my_granule);
        const uint32_t tag = (uint32_t)__builtin_amdgcn_readfirstlane((int)gq.w);
        if (tag == MB_STOP) break;
        const bool whole = __ballot(gq.w == tag) == __ballot(true);  // every granule carries this tag: the request is complete
        if (tag == served || !whole) {  // nothing new (or a request half written)
            const uint64_t now = wall_clock64();
            if (now - last > idle_ticks || now - born > life_ticks) break;
            __builtin_amdgcn_s_sleep(4);
            continue;
        }
#ifdef OC_AMD_TUNING
        const uint64_t tm0 = mb_now();  // tuning builds: where a served request's time goes, in 10 ns ticks, left in the spare granule 8
#endif
        // the payload words, wave-uniform: word 3 i + j = payload dword j of granule i
        uint32_t pw[3 * n_req + 4];
#pragma unroll
        for (int i = 0; i < n_req; ++i) {
            pw[3 * i] = (uint32_t)__builtin_amdgcn_readlane((int)gq.x, i);
            pw[3 * i + 1] = (uint32_t)__builtin_amdgcn_readlane((int)gq.y, i);
            pw[3 * i + 2] = (uint32_t)__builtin_amdgcn_readlane((int)gq.z, i);
        }
        if (threadIdx.x == 0) {
        // ---- the request: header, object planes, the two action bytes
        OneIn q_in;
        q_in.h = make_uint4(pw[0], pw[1], pw[2], pw[3]);
#pragma unroll
        for (int p = 0; p < STEP1_MAX_PLANES; ++p)
            q_in.v[p] = p < n_obj ? make_uint4(pw[4 + 4 * p], pw[5 + 4 * p], pw[6 + 4 * p], pw[7 + 4 * p]) : make_uint4(0u, 0u, 0u, 0u);
        const uint32_t a01 = pw[n_state / 4] & 0xFFFFu;
        const uint32_t a0 = a01 & 0xFFu, a1 = a01 >> 8;
#pragma unroll
        for (int p = 0; p < STEP1_MAX_PLANES; ++p)
            if (p < n_obj) s_rows[p * BLOCK] = q_in.v[p];
#ifdef OC_AMD_TUNING
        const uint64_t tm1 = mb_now();
#endif
        const uint8_t* row = reinterpret_cast<const uint8_t*>(s_rows);
        One<MAXP> q;
        one_decode<MAXP>(C, L, q_in.h, row, q);
        float4 rw = make_float4(0.f, 0.f, 0.f, 0.f);
        uint32_t fl = 0;
        uint64_t ev = 0;
        uint4 o[1 + STEP1_MAX_PLANES];  // the new state
        if (a0 > 5u || a1 > 5u) {  // get_state_transition raises ValueError (mdp.py:1394-1398): the state comes back as it is
            fl = OC_F_BAD_ACTION;
            o[0] = q_in.h;
#pragma unroll
            for (int p = 0; p < STEP1_MAX_PLANES; ++p) o[1 + p] = q_in.v[p];
        } else {
            one_transition<MAXP, true>(C, L, lut, make_delta4(W), a0, a1, q_in.v, n_obj, row, q, rw, &ev);
            if ((int)q.s.t >= horizon) fl |= OC_F_DONE;
            // the new state, every plane, assembled in this lane's LDS rows (header in a row of its own)
            one_store<MAXP>(C, L, s_out, 1, 0, n_obj, q, false, reinterpret_cast<uint8_t*>(s_rows));
#pragma unroll
            for (int p = 0; p <= STEP1_MAX_PLANES; ++p) o[p] = s_out[p];
        }
        // ---- the response: payload = new state (n_state bytes), rewards (16), flags (4), events (8) = at most 27 dwords, laid
        //      out as TWO 64-byte lines of {15 payload dwords, tag}; lane 0 leaves the 32 dwords in LDS ...
        uint32_t r[32];
#pragma unroll
        for (int i = 0; i < 32; ++i) r[i] = 0u;
        int k = 0;
#pragma unroll
        for (int p = 0; p <= STEP1_MAX_PLANES; ++p)
            if (p <= n_obj) { r[k] = o[p].x; r[k + 1] = o[p].y; r[k + 2] = o[p].z; r[k + 3] = o[p].w; k += 4; }
        r[k] = __float_as_uint(rw.x); r[k + 1] = __float_as_uint(rw.y); r[k + 2] = __float_as_uint(rw.z); r[k + 3] = __float_as_uint(rw.w);
        r[k + 4] = fl; r[k + 5] = (uint32_t)ev; r[k + 6] = (uint32_t)(ev >> 32);
#ifdef OC_AMD_TUNING
        {
            const uint64_t tm2 = mb_now();
            r[27] = (uint32_t)(tm1 - tm0); r[28] = (uint32_t)(tm2 - tm1); r[29] = (uint32_t)(tm0 - tm_prev);  // (payload dwords 27..29 are spare)
            tm_prev = tm2;
        }
#endif
        // payload dword i sits at word i of line 0 (i < 15) or word i - 15 of line 1; word 15 of each line = the tag
        uint32_t w[32];
#pragma unroll
        for (int i = 0; i < 15; ++i) { w[i] = r[i]; w[16 + i] = r[15 + i]; }
        w[15] = tag; w[31] = tag;
#pragma unroll
        for (int g = 0; g < 8; ++g) s_rsp[g] = make_uint4(w[4 * g], w[4 * g + 1], w[4 * g + 2], w[4 * g + 3]);
        }  // (lane 0)
        // ... and lanes 0..7 send them with ONE store instruction: 16 bytes per lane, two whole 64-byte lines.  Measured
        // (tools/mailbox_latency.hip, profiles/r05_mailbox_latency.txt): nine 16-byte stores from one lane — one PCIe write each —
        // cost 1.6 us more per round trip than this.  PCIe writes arrive in order, so a line whose LAST dword shows the tag is complete.
        {
            const uint4 p4 = s_rsp[threadIdx.x];
            const mb_u32x4 part = {p4.x, p4.y, p4.z, p4.w};
            asm volatile("global_store_dwordx4 %0, %1, off sc0 sc1\n\ts_waitcnt vmcnt(0)" : : "v"(mb + MB_RSPG + 16 * threadIdx.x), "v"(part) : "memory");
        }
        served = tag;
        last = wall_clock64();
    }
    if (threadIdx.x != 0) return;
    __hip_atomic_store(reinterpret_cast<uint32_t*>(mb + MB_ALIVE), 0u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
}
