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
// Protocol (all words in the 4 KiB host mailbox, system-scope atomics on the GPU side, x86 TSO on the host side):
//   host:  payload (state planes, two action bytes) -> req = seq (release)           ... spin on rsp == seq -> read outputs
//   GPU :  poll req (relaxed, s_sleep between polls) -> acquire -> payload -> step -> outputs -> release -> rsp = seq
// The kernel never outlives its usefulness: it leaves when req == MB_STOP, after idle_ticks of wall_clock64 without a
// request, or after life_ticks in total, and says so (alive = 0); oc_mailbox_step relaunches it when needed.  It serves one
// layout (the mailbox's batch holds one layout record), any number of pots, grids of at most 64 cells.
// ==========================================================================================
constexpr uint32_t MB_STOP = 0xFFFFFFFFu;
// byte offsets inside the mailbox
constexpr int MB_REQ = 0, MB_RSP = 64, MB_ALIVE = 128, MB_IN = 256 /* 5 planes x 16 B + actions */, MB_ACT = MB_IN + 80,
              MB_OUT = 512 /* 5 planes */, MB_REW = MB_OUT + 80, MB_FLAGS = MB_REW + 16, MB_EV = MB_FLAGS + 8, MB_BYTES = 4096;

typedef uint32_t mb_u32x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ uint4 mb_load16(const uint4* p) {  // 16 bytes of the request, past every cache
    const mb_u32x4 v = __builtin_nontemporal_load(reinterpret_cast<const mb_u32x4*>(p));
    return make_uint4(v.x, v.y, v.z, v.w);
}
__device__ __forceinline__ uint32_t mb_load(const uint32_t* p) {
    return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
}
__device__ __forceinline__ void mb_store(uint32_t* p, uint32_t v) {
    __hip_atomic_store(p, v, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
}

__global__ __launch_bounds__(64) void k_mailbox(const OcLayout* __restrict__ g_layout, uint8_t* mb, int W, int n_obj, int horizon,
                                                uint64_t idle_ticks, uint64_t life_ticks) {
    constexpr int MAXP = OC_MAX_POTS;
    __shared__ uint4 s_rows[STEP1_MAX_PLANES * BLOCK];  // the lane's planes, [plane][BLOCK] rows of 16 bytes (one_obj's layout)
    __shared__ uint4 s_lay[16];
    __shared__ uint2 s_lut[2 * LUT_ENTRIES];
    for (int i = threadIdx.x; i < 2 * LUT_ENTRIES; i += 64) s_lut[i] = reinterpret_cast<const uint2*>(&g_lut)[i];
    if (threadIdx.x < 16) s_lay[threadIdx.x] = reinterpret_cast<const uint4*>(g_layout)[threadIdx.x];
    __syncthreads();
    if (threadIdx.x != 0) return;  // one env: one lane (wave-uniform ballots inside the transition see only this lane)
    const Lay L{reinterpret_cast<const uint8_t*>(s_lay)};
    const LayC C = load_consts<false>(L);
    const uint8_t* lut = reinterpret_cast<const uint8_t*>(s_lut) + (C.old_dyn ? LUT_ENTRIES * 8 : 0);
    uint32_t* const req = reinterpret_cast<uint32_t*>(mb + MB_REQ);
    uint32_t* const rsp = reinterpret_cast<uint32_t*>(mb + MB_RSP);
    uint32_t* const alive = reinterpret_cast<uint32_t*>(mb + MB_ALIVE);
    const uint4* in = reinterpret_cast<const uint4*>(mb + MB_IN);
    uint4* out = reinterpret_cast<uint4*>(mb + MB_OUT);
    uint32_t served = mb_load(rsp);  // the last request answered (by an earlier incarnation of this kernel)
    const uint64_t born = wall_clock64();
    uint64_t last = born;
    for (;;) {
        const uint32_t r = mb_load(req);
        if (r == served) {
            const uint64_t now = wall_clock64();
            if (now - last > idle_ticks || now - born > life_ticks) break;
            __builtin_amdgcn_s_sleep(8);
            continue;
        }
        if (r == MB_STOP) break;
        __atomic_thread_fence(__ATOMIC_ACQUIRE);  // the payload was written before the request word
        // ---- the request: header, object planes, both actions (fine-grained host memory: uncached on the GPU)
        OneIn q_in;
        q_in.h = mb_load16(in);
#pragma unroll
        for (int p = 0; p < STEP1_MAX_PLANES; ++p)
            q_in.v[p] = p < n_obj ? mb_load16(in + 1 + p) : make_uint4(0u, 0u, 0u, 0u);
        q_in.a01 = __builtin_nontemporal_load(reinterpret_cast<const uint16_t*>(mb + MB_ACT));
#pragma unroll
        for (int p = 0; p < STEP1_MAX_PLANES; ++p)
            if (p < n_obj) s_rows[p * BLOCK] = q_in.v[p];
        const uint8_t* row = reinterpret_cast<const uint8_t*>(s_rows);
        One<MAXP> q;
        one_decode<MAXP>(C, L, q_in.h, row, q);
        const uint32_t a0 = q_in.a01 & 0xFFu, a1 = q_in.a01 >> 8;
        float4 rw = make_float4(0.f, 0.f, 0.f, 0.f);
        uint32_t fl = 0;
        uint64_t ev = 0;
        if (a0 > 5u || a1 > 5u) {  // get_state_transition raises ValueError (mdp.py:1394-1398): the state comes back as it is
            fl = OC_F_BAD_ACTION;
            out[0] = q_in.h;
#pragma unroll
            for (int p = 0; p < STEP1_MAX_PLANES; ++p)
                if (p < n_obj) out[1 + p] = q_in.v[p];
        } else {
            one_transition<MAXP, true>(C, L, lut, make_delta4(W), a0, a1, q_in.v, n_obj, row, q, rw, &ev);
            if ((int)q.s.t >= horizon) fl |= OC_F_DONE;
            // the new state, every plane (n = 1, e = 0: plane p is out[p]); the changed bytes go through this lane's LDS rows
            one_store<MAXP>(C, L, out, 1, 0, n_obj, q, false, reinterpret_cast<uint8_t*>(s_rows));
        }
        *reinterpret_cast<float4*>(mb + MB_REW) = rw;
        *reinterpret_cast<uint32_t*>(mb + MB_FLAGS) = fl;
        *reinterpret_cast<uint64_t*>(mb + MB_EV) = ev;
        mb_store(rsp, r);  // release: the outputs are visible to the host before the response word
        served = r;
        last = wall_clock64();
    }
    mb_store(alive, 0u);
}
