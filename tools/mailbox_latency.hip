// mailbox_latency.hip — round trip of a host <-> resident-kernel ping-pong, piece by piece (what the single-env mailbox of
// csrc/mailbox.hpp is made of):  hipcc --offload-arch=gfx950 -O3 -o tools/mailbox_latency tools/mailbox_latency.hip
//   request in pinned host memory, polled with a 4-byte system-scope atomic load by one lane  /  with one 16-byte load per lane
//   by 8 lanes (request granules: payload and tag together); response of 1 or 9 granules; 0 or ~2 us of work before the answer;
//   request in fine-grained DEVICE memory that the host writes through the PCIe BAR (write-combined: needs sfence).
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
#include <stdlib.h>
#include <chrono>
#include <immintrin.h>

typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));

// MODE bit 0: granule poll (8 lanes x 16 B) instead of a 4-byte atomic; bit 1: 9 response granules instead of 1; bit 2: ~2 us of work;
// bit 3: no s_waitcnt vmcnt(0) behind the response stores; bit 4: the response as ONE store instruction of 8 lanes x 16 B = two 64-byte lines,
// tag in the last dword of each line
template <int MODE>
__global__ __launch_bounds__(64) void k_pong(const uint8_t* req, uint8_t* rsp, uint32_t n_iter, int n_req) {
    if (threadIdx.x >= 8) return;
    uint32_t served = 0;
    const uint64_t born = wall_clock64();
    const uint8_t* mine = req + 16 * min((int)threadIdx.x, n_req - 1);
    uint32_t acc = threadIdx.x;
    while (served < n_iter) {
        uint32_t tag;
        bool whole = true;
        if (MODE & 1) {
            u32x4 v;
            asm volatile("global_load_dwordx4 %0, %1, off sc0 sc1\n\ts_waitcnt vmcnt(0)" : "=v"(v) : "v"(mine) : "memory");
            tag = (uint32_t)__builtin_amdgcn_readfirstlane((int)v.w);
            whole = __ballot(v.w == tag) == __ballot(true);
            acc += v.x;
        } else {
            tag = __hip_atomic_load(reinterpret_cast<const uint32_t*>(req + 12), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
            tag = (uint32_t)__builtin_amdgcn_readfirstlane((int)tag);
        }
        if (tag == served || !whole) {
            if (wall_clock64() - born > 100000000ull * 4) break;  // ~4 s at 100 MHz: never hang the box
            __builtin_amdgcn_s_sleep(4);
            continue;
        }
        if (MODE & 4) {
#pragma unroll 1
            for (int j = 0; j < 600; ++j) acc = acc * 1664525u + 1013904223u;  // ~600 dependent multiply-adds: ~2 us on one wavefront
        }
        if (MODE & 16) {
            const u32x4 g = {acc, 1u, 2u, tag};
            asm volatile("global_store_dwordx4 %0, %1, off sc0 sc1" : : "v"(rsp + 16 * threadIdx.x), "v"(g) : "memory");
            if (!(MODE & 8)) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        } else if (threadIdx.x == 0) {
            const u32x4 g = {acc, 1u, 2u, tag};
            if (MODE & 2) {
#pragma unroll
                for (int i = 8; i >= 0; --i) asm volatile("global_store_dwordx4 %0, %1, off sc0 sc1" : : "v"(rsp + 16 * i), "v"(g) : "memory");
            } else {
                asm volatile("global_store_dwordx4 %0, %1, off sc0 sc1" : : "v"(rsp), "v"(g) : "memory");
            }
            if (!(MODE & 8)) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
        served = tag;
    }
}

template <int MODE>
static double pingpong(uint8_t* req_host_view, const uint8_t* req_dev, volatile uint8_t* rsp, uint32_t n_iter, int flush, int n_req) {
    for (int i = 0; i < 9; ++i) *(volatile uint32_t*)(rsp + 16 * i + 12) = 0;
    for (int i = 0; i < 8; ++i) *(volatile uint32_t*)(req_host_view + 16 * i + 12) = 0;
    if (flush) _mm_sfence();
    hipLaunchKernelGGL(k_pong<MODE>, dim3(1), dim3(64), 0, 0, req_dev, (uint8_t*)rsp, n_iter, n_req);
    const int n_rsp = (MODE & 16) ? 8 : (MODE & 2) ? 9 : 1;
    const auto t0 = std::chrono::steady_clock::now();
    for (uint32_t s = 1; s <= n_iter; ++s) {
        for (int g = 0; g < n_req; ++g) {
            const __m128i q = _mm_set_epi32((int)s, 3, 2, (int)(s + g));
            _mm_store_si128((__m128i*)(req_host_view + 16 * g), q);
        }
        if (flush) _mm_sfence();
        uint64_t spins = 0;
        for (;;) {
            int ok = 1;
            for (int g = n_rsp - 1; g >= 0 && ok; g -= ((MODE & 16) ? 4 : 1)) ok = *(volatile uint32_t*)(rsp + 16 * g + 12) == s;
            if (ok) break;
            if (++spins > 40000000ull) { printf("timeout at %u\n", s); (void)hipDeviceSynchronize(); return -1; }
        }
    }
    const auto t1 = std::chrono::steady_clock::now();
    (void)hipDeviceSynchronize();
    return std::chrono::duration<double, std::micro>(t1 - t0).count() / n_iter;
}

int main() {
    uint8_t *rsp, *req_pinned;
    (void)hipHostMalloc((void**)&rsp, 4096, hipHostMallocMapped | hipHostMallocCoherent);
    (void)hipHostMalloc((void**)&req_pinned, 4096, hipHostMallocMapped | hipHostMallocCoherent);
    const uint32_t N = 20000;
    printf("pinned request, 4-byte atomic poll, 1 response granule            : %.2f us\n", pingpong<0>(req_pinned, req_pinned, rsp, N, 0, 1));
    printf("pinned request, 8-lane 16-byte poll (5 granules), 1 response granule: %.2f us\n", pingpong<1>(req_pinned, req_pinned, rsp, N, 0, 5));
    printf("  ... 9 response granules                                          : %.2f us\n", pingpong<3>(req_pinned, req_pinned, rsp, N, 0, 5));
    printf("  ... 9 response granules, no wait behind the stores               : %.2f us\n", pingpong<11>(req_pinned, req_pinned, rsp, N, 0, 5));
    printf("  ... response = one 8-lane store (two 64-byte lines, 2 tags)      : %.2f us\n", pingpong<17>(req_pinned, req_pinned, rsp, N, 0, 5));
    printf("  ... the same, no wait behind the store                           : %.2f us\n", pingpong<25>(req_pinned, req_pinned, rsp, N, 0, 5));
    uint8_t* req_dev = nullptr;
    if (hipExtMallocWithFlags((void**)&req_dev, 4096, hipDeviceMallocFinegrained) == hipSuccess) {
        (void)hipMemset(req_dev, 0, 4096);
        (void)hipDeviceSynchronize();
        printf("request granules in fine-grained DEVICE memory (BAR writes + sfence), 1 response granule: %.2f us\n", pingpong<1>(req_dev, req_dev, rsp, N, 1, 5));
        printf("  ... response = one 8-lane store                                  : %.2f us\n", pingpong<17>(req_dev, req_dev, rsp, N, 1, 5));
    }
    return 0;
}
