// Launch floor of a one-step kernel (hipcc --offload-arch=gfx950 -O3 -o launch_floor tools/launch_floor.hip): an empty kernel, and kernels that
// only move the bytes a one-step kernel moves behind the same barrier.  Most of the variants read the workgroup size from the
// dispatch packet (an alloca promoted to LDS by the compiler): that packet lives in host memory — 12 us per launch instead of 3.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>
__global__ __launch_bounds__(256) void k_empty(const uint4* st, int64_t n) {}
template <int NOBJ, bool LUT>
__global__ __launch_bounds__(256) void k_move(uint4* st, const uint16_t* act, float4* rew, uint8_t* flg, float4* ep, const uint2* lut, int64_t n) {
    __shared__ uint2 s_lut[480];
    __shared__ uint4 rows[NOBJ][256];
    const int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x;
    uint4 h = st[e];
    uint32_t a = act[e];
    float4 r = ep[e];
    uint4 v[NOBJ];
#pragma unroll
    for (int p = 0; p < NOBJ; ++p) v[p] = st[(int64_t)(1 + p) * n + e];
    if (LUT) {
        for (int i = threadIdx.x; i < 480; i += 256) s_lut[i] = lut[i];
    }
#pragma unroll
    for (int p = 0; p < NOBJ; ++p) rows[p][threadIdx.x] = v[p];
    if (LUT) __syncthreads();
    uint32_t c = (h.x + a) & 15u;
    uint32_t b = reinterpret_cast<uint8_t*>(&rows[(h.x >> 4) % NOBJ][threadIdx.x])[c];
    uint2 en = LUT ? s_lut[(b + h.y) % 480] : make_uint2(b, a);
    h.x += en.x; h.y ^= en.y;
    st[e] = h;
    r.x += (float)en.x;
    rew[e] = r; ep[e] = r; flg[e] = (uint8_t)b;
    if (en.y & 1) reinterpret_cast<uint8_t*>(st + (int64_t)(1 + (h.x >> 4) % NOBJ) * n + e)[c] = (uint8_t)a;
}
int main() {
    const int64_t n = 65536;
    uint4* st; uint16_t* act; float4 *rew, *ep; uint8_t* flg; uint2* lut;
    hipMalloc(&st, n * 16 * 5); hipMemset(st, 0, n * 16 * 5);
    hipMalloc(&act, n * 2); hipMemset(act, 0, n * 2);
    hipMalloc(&rew, n * 16); hipMalloc(&ep, n * 16); hipMemset(ep, 0, n * 16); hipMalloc(&flg, n);
    hipMalloc(&lut, 480 * 8); hipMemset(lut, 0, 480 * 8);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    auto timeit = [&](const char* name, auto launch) {
        for (int i = 0; i < 200; ++i) launch();
        hipDeviceSynchronize();
        float best = 1e9;
        for (int rep = 0; rep < 5; ++rep) {
            hipEventRecord(e0, 0);
            for (int i = 0; i < 2000; ++i) launch();
            hipEventRecord(e1, 0);
            hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1);
            if (ms < best) best = ms;
        }
        printf("%-28s %.2f us per launch\n", name, best / 2000 * 1e3);
    };
    for (int pass = 0; pass < 2; ++pass) {
    timeit("empty 256x256", [&] { hipLaunchKernelGGL(k_empty, dim3(256), dim3(256), 0, 0, st, n); });
    timeit("empty 1x64", [&] { hipLaunchKernelGGL(k_empty, dim3(1), dim3(64), 0, 0, st, n); });
    timeit("move nobj=3 lut", [&] { hipLaunchKernelGGL((k_move<3, true>), dim3(256), dim3(256), 0, 0, st, act, rew, flg, ep, lut, n); });
    timeit("move nobj=3 nolut", [&] { hipLaunchKernelGGL((k_move<3, false>), dim3(256), dim3(256), 0, 0, st, act, rew, flg, ep, lut, n); });
    timeit("move nobj=4 lut", [&] { hipLaunchKernelGGL((k_move<4, true>), dim3(256), dim3(256), 0, 0, st, act, rew, flg, ep, lut, n); });
    timeit("move nobj=4 nolut", [&] { hipLaunchKernelGGL((k_move<4, false>), dim3(256), dim3(256), 0, 0, st, act, rew, flg, ep, lut, n); });
    timeit("move nobj=3 lut again", [&] { hipLaunchKernelGGL((k_move<3, true>), dim3(256), dim3(256), 0, 0, st, act, rew, flg, ep, lut, n); });
    }
    return 0;
}
