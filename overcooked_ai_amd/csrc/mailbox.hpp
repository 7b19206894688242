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
//   host:  payload (state planes at MB_IN, the two action bytes at MB_ACT) -> request word = seq (x86 stores stay in order)
//          ... spins until every RESPONSE GRANULE carries seq -> copies the outputs out
//   GPU :  polls the request word with a system-scope atomic load (s_sleep between polls) -> acquire fence (drops whatever
//          the GPU's caches hold of the mailbox) -> payload -> step -> response granules: 16 bytes each, {12 payload bytes,
//          u32 tag = seq}, one store per granule.  An aligned 16-byte store is one PCIe write on one cache line, so a granule
//          that shows the tag has its payload too: no flag behind a release fence, no wait for write acknowledgements.
// Measured on the box (oc_mailbox_step alone, cramped_room): request word + response word behind a release fence 7.9 us;
// this form 7.0 us; all-granule requests read with 8-byte system-scope atomic loads (one round trip on the request side
// too) 7.6 us — no better, so the request stays a word + fence; 8-byte granules both ways 11.1 us (13 reads + 19 fabric
// writes per step).  Stores that are not write-through (plain, non-temporal) stay in the GPU's L2 until the kernel's final
// release: every step then took the 2 ms idle timeout and was answered by the NEXT incarnation of the kernel.
// The kernel never outlives its usefulness: it leaves when the request word is MB_STOP, after idle_ticks of wall_clock64
// without a request, or after life_ticks in total, and says so (alive = 0); oc_mailbox_step relaunches it when needed.  It
// serves one layout (the mailbox's batch holds one layout record), any number of pots, grids of at most 64 cells.
// ==========================================================================================
constexpr uint32_t MB_STOP = 0xFFFFFFFFu;
// byte offsets inside the mailbox.  IN / ACT / OUT / REW / FLAGS / EV are the caller's plain views (include/oc_amd.h);
// oc_mailbox_step unpacks the response granules into OUT .. EV.
constexpr int MB_REQ = 0, MB_ALIVE = 128, MB_IN = 256 /* 5 planes x 16 B */, MB_ACT = MB_IN + 80, MB_OUT = 512 /* 5 planes */,
              MB_REW = MB_OUT + 80, MB_FLAGS = MB_REW + 16, MB_EV = MB_FLAGS + 8, MB_RSPG = 2048, MB_BYTES = 4096;
constexpr int MB_RSP_BYTES = 80 + 16 + 4 + 8, MB_RSP_GRANULES = (MB_RSP_BYTES + 11) / 12;  // 9

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
// A response granule: ONE 16-byte write-through store at system scope.  (`sc0 sc1` is what makes it leave the GPU's L2 now: a
// plain or non-temporal store stays there until the kernel's final release — measured: every step then took the idle
// timeout.)  Inline asm, because no builtin emits a 16-byte system-scope store (and the compiler does not see asm stores'
// hazards: NOTEBOOK 4.2c).
// ONE asm block for all nine granules and the wait: nothing the compiler schedules can land between a store and the moment it
// has read its data registers (granules past the response's length carry zeros: the host never looks at them).
__device__ __forceinline__ void mb_store_granules(uint8_t* base, const mb_u32x4 (&g)[MB_RSP_GRANULES]) {
    asm volatile(
        "global_store_dwordx4 %0, %9, off offset:128 sc0 sc1\n\t"
        "global_store_dwordx4 %0, %8, off offset:112 sc0 sc1\n\t"
        "global_store_dwordx4 %0, %7, off offset:96 sc0 sc1\n\t"
        "global_store_dwordx4 %0, %6, off offset:80 sc0 sc1\n\t"
        "global_store_dwordx4 %0, %5, off offset:64 sc0 sc1\n\t"
        "global_store_dwordx4 %0, %4, off offset:48 sc0 sc1\n\t"
        "global_store_dwordx4 %0, %3, off offset:32 sc0 sc1\n\t"
        "global_store_dwordx4 %0, %2, off offset:16 sc0 sc1\n\t"
        "global_store_dwordx4 %0, %1, off sc0 sc1\n\t"  // granule 0 last: the next incarnation of the kernel reads `served` from it
        "s_waitcnt vmcnt(0)"
        :
        : "v"(base), "v"(g[0]), "v"(g[1]), "v"(g[2]), "v"(g[3]), "v"(g[4]), "v"(g[5]), "v"(g[6]), "v"(g[7]), "v"(g[8])
        : "memory");
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
    for (int i = threadIdx.x; i < 2 * LUT_ENTRIES; i += 64) s_lut[i] = reinterpret_cast<const uint2*>(&g_lut)[i];
    if (threadIdx.x < 16) s_lay[threadIdx.x] = reinterpret_cast<const uint4*>(g_layout)[threadIdx.x];
    __syncthreads();
    if (threadIdx.x != 0) return;  // one env: one lane (wave-uniform ballots inside the transition see only this lane)
    const Lay L{reinterpret_cast<const uint8_t*>(s_lay)};
    const LayC C = load_consts<false>(L);
    const uint8_t* lut = reinterpret_cast<const uint8_t*>(s_lut) + (C.old_dyn ? LUT_ENTRIES * 8 : 0);
    // granules a request / a response of this layout really needs (n_obj object planes + the header)
    constexpr int n_state = 16 * (1 + n_obj);
    constexpr int n_rsp = (n_state + 28 + 11) / 12;
    const uint32_t* const req = reinterpret_cast<const uint32_t*>(mb + MB_REQ);
    const uint4* const in = reinterpret_cast<const uint4*>(mb + MB_IN);
    uint32_t served = mb_load(reinterpret_cast<const uint32_t*>(mb + MB_RSPG + 12));  // the last request answered (by an earlier incarnation)
    const uint64_t born = wall_clock64();
    uint64_t last = born;
#ifdef OC_AMD_TUNING
    uint64_t tm_prev = 0;
#endif
    for (;;) {
        const uint32_t tag = mb_load(req);
        if (tag == MB_STOP) break;
        if (tag == served) {  // nothing new
            const uint64_t now = wall_clock64();
            if (now - last > idle_ticks || now - born > life_ticks) break;
            __builtin_amdgcn_s_sleep(4);
            continue;
        }
#ifdef OC_AMD_TUNING
        const uint64_t tm0 = mb_now();  // tuning builds: where a served request's time goes, in 10 ns ticks, left in the spare granule 8
#endif
        __atomic_thread_fence(__ATOMIC_ACQUIRE);  // the payload was written before the request word; drop cached copies of it
        // ---- the request: header, object planes, the two action bytes
        OneIn q_in;
        q_in.h = in[0];
#pragma unroll
        for (int p = 0; p < STEP1_MAX_PLANES; ++p) q_in.v[p] = p < n_obj ? in[1 + p] : make_uint4(0u, 0u, 0u, 0u);
        const uint32_t a01 = *reinterpret_cast<const uint16_t*>(mb + MB_ACT);
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
        // ---- the response: payload = new state (n_state bytes), rewards (16), flags (4), events (8); tag = the request's
        uint32_t r[3 * MB_RSP_GRANULES + 4];
#pragma unroll
        for (int i = 0; i < 3 * MB_RSP_GRANULES + 4; ++i) r[i] = 0u;
        int k = 0;
#pragma unroll
        for (int p = 0; p <= STEP1_MAX_PLANES; ++p)
            if (p <= n_obj) { r[k] = o[p].x; r[k + 1] = o[p].y; r[k + 2] = o[p].z; r[k + 3] = o[p].w; k += 4; }
        r[k] = __float_as_uint(rw.x); r[k + 1] = __float_as_uint(rw.y); r[k + 2] = __float_as_uint(rw.z); r[k + 3] = __float_as_uint(rw.w);
        r[k + 4] = fl; r[k + 5] = (uint32_t)ev; r[k + 6] = (uint32_t)(ev >> 32);
        mb_u32x4 gr[MB_RSP_GRANULES];
#pragma unroll
        for (int g = 0; g < MB_RSP_GRANULES; ++g) gr[g] = g < n_rsp ? mb_u32x4{r[3 * g], r[3 * g + 1], r[3 * g + 2], tag} : mb_u32x4{0u, 0u, 0u, 0u};
#ifdef OC_AMD_TUNING
        if (n_rsp < MB_RSP_GRANULES) {
            const uint64_t tm2 = mb_now();
            gr[MB_RSP_GRANULES - 1] = mb_u32x4{(uint32_t)(tm1 - tm0), (uint32_t)(tm2 - tm1), (uint32_t)(tm0 - tm_prev), tag};
            tm_prev = tm2;
        }
#endif
        mb_store_granules(mb + MB_RSPG, gr);
        served = tag;
        last = wall_clock64();
    }
    __hip_atomic_store(reinterpret_cast<uint32_t*>(mb + MB_ALIVE), 0u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
}
