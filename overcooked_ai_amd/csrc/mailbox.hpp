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
// Protocol: data-tagged 8-byte granules, {u32 payload, u32 tag = the request's sequence number}, in the 4 KiB host mailbox,
// read and written with ONE system-scope 8-byte access each (MI355X_MICROARCH.md, "handoff-1to1": whoever reads a granule with
// the expected tag has its payload too — no separate flag, no fence between payload and flag, and each direction costs ONE
// round trip).  The accesses are system-scope atomics: a plain or non-temporal load of host memory may be served from the
// GPU's caches and never see the host's next write (seen on the box: the first version polled with non-temporal loads and
// every step ran into the idle timeout).
//   host:  request granules (state planes + the two action bytes)   ... spins until every response granule carries the tag
//   GPU :  polls ALL request granules at once (s_sleep between polls) until they agree on a new tag -> step -> response
//          granules (next state, rewards, flags, events)
// The kernel never outlives its usefulness: it leaves when granule 0 carries MB_STOP, after idle_ticks of wall_clock64
// without a request, or after life_ticks in total, and says so (alive = 0); oc_mailbox_step relaunches it when needed.  It
// serves one layout (the mailbox's batch holds one layout record), any number of pots, grids of at most 64 cells.
// ==========================================================================================
constexpr uint32_t MB_STOP = 0xFFFFFFFFu;
// byte offsets inside the mailbox.  IN / ACT / OUT / REW / FLAGS / EV are the caller's plain views (include/oc_amd.h):
// oc_mailbox_step packs IN + ACT into the request granules and unpacks the response granules into OUT .. EV.
constexpr int MB_ALIVE = 128, MB_IN = 256 /* 5 planes x 16 B */, MB_ACT = MB_IN + 80, MB_OUT = 512 /* 5 planes */,
              MB_REW = MB_OUT + 80, MB_FLAGS = MB_REW + 16, MB_EV = MB_FLAGS + 8, MB_REQG = 1024, MB_RSPG = 2048, MB_BYTES = 4096;
constexpr int MB_REQ_WORDS = (80 + 2 + 3) / 4;       // 21 granules at most: header + 4 planes + the action bytes
constexpr int MB_RSP_WORDS = (80 + 16 + 4 + 8) / 4;  // 27: new state, rewards, flags, events

__device__ __forceinline__ uint64_t mb_load8(const uint8_t* p) {  // one granule, at system scope (past the GPU's caches)
    return __hip_atomic_load(reinterpret_cast<const uint64_t*>(p), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
}
__device__ __forceinline__ void mb_store8(uint8_t* p, uint32_t payload, uint32_t tag) {
    __hip_atomic_store(reinterpret_cast<uint64_t*>(p), (uint64_t)payload | ((uint64_t)tag << 32), __ATOMIC_RELAXED,
                       __HIP_MEMORY_SCOPE_SYSTEM);
}

template <int NOBJ>  // object planes of the grid (1..4): every index into the granule words is a compile-time constant
__global__ __launch_bounds__(64) void k_mailbox(const OcLayout* __restrict__ g_layout, uint8_t* mb, int W, int horizon,
                                                uint64_t idle_ticks, uint64_t life_ticks) {
    constexpr int MAXP = OC_MAX_POTS, n_obj = NOBJ;
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
    constexpr int n_req = (n_state + 2 + 3) / 4, n_rsp = (n_state + 28) / 4;
    uint32_t served = (uint32_t)(mb_load8(mb + MB_RSPG) >> 32);  // the last request answered (by an earlier incarnation of this kernel)
    const uint64_t born = wall_clock64();
    uint64_t last = born;
    for (;;) {
        uint32_t w[MB_REQ_WORDS];  // the request's payload words
        uint32_t tag = 0;
        bool same = true;
#pragma unroll
        for (int g = 0; g < MB_REQ_WORDS; ++g) {
            if (g < n_req) {
                const uint64_t v = mb_load8(mb + MB_REQG + 8 * g);
                w[g] = (uint32_t)v;
                if (g == 0) tag = (uint32_t)(v >> 32); else same &= (uint32_t)(v >> 32) == tag;
            } else {
                w[g] = 0u;
            }
        }
        if (tag == MB_STOP) break;
        if (tag == served || !same) {  // nothing new (or a request still being written)
            const uint64_t now = wall_clock64();
            if (now - last > idle_ticks || now - born > life_ticks) break;
            __builtin_amdgcn_s_sleep(4);
            continue;
        }
        // ---- the request: payload bytes 0 .. n_state - 1 = header + object planes, then the two action bytes
        auto byte_at = [&](int i) __attribute__((always_inline)) { return (w[i >> 2] >> (8 * (i & 3))) & 0xFFu; };
        OneIn q_in;
        q_in.h = make_uint4(w[0], w[1], w[2], w[3]);
#pragma unroll
        for (int p = 0; p < STEP1_MAX_PLANES; ++p)
            q_in.v[p] = p < n_obj ? make_uint4(w[4 + 4 * p], w[5 + 4 * p], w[6 + 4 * p], w[7 + 4 * p]) : make_uint4(0u, 0u, 0u, 0u);
        uint32_t a0 = 0, a1 = 0;
#pragma unroll
        for (int p = 0; p <= STEP1_MAX_PLANES; ++p)  // (the action bytes follow the last plane: n_state is 16 * (1 + n_obj))
            if (p == n_obj) { a0 = byte_at(16 * (1 + p)); a1 = byte_at(16 * (1 + p) + 1); }
#pragma unroll
        for (int p = 0; p < STEP1_MAX_PLANES; ++p)
            if (p < n_obj) s_rows[p * BLOCK] = q_in.v[p];
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
        uint32_t r[MB_RSP_WORDS];
#pragma unroll
        for (int i = 0; i < MB_RSP_WORDS; ++i) r[i] = 0u;
        int k = 0;
#pragma unroll
        for (int p = 0; p <= STEP1_MAX_PLANES; ++p)
            if (p <= n_obj) { r[k] = o[p].x; r[k + 1] = o[p].y; r[k + 2] = o[p].z; r[k + 3] = o[p].w; k += 4; }
        r[k] = __float_as_uint(rw.x); r[k + 1] = __float_as_uint(rw.y); r[k + 2] = __float_as_uint(rw.z); r[k + 3] = __float_as_uint(rw.w);
        r[k + 4] = fl; r[k + 5] = (uint32_t)ev; r[k + 6] = (uint32_t)(ev >> 32);
#pragma unroll
        for (int g = MB_RSP_WORDS - 1; g >= 0; --g)  // granule 0 last: the next incarnation reads `served` from it
            if (g < n_rsp) mb_store8(mb + MB_RSPG + 8 * g, r[g], tag);
        served = tag;
        last = wall_clock64();
    }
    __hip_atomic_store(reinterpret_cast<uint32_t*>(mb + MB_ALIVE), 0u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
}
