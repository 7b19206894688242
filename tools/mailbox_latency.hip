// mailbox_latency.hip — round trip of a host <-> resident-kernel ping-pong, by where the REQUEST word lives:
//   (a) pinned, GPU-mapped host memory (the shipped mailbox: the GPU's poll is a PCIe read round trip)
//   (b) fine-grained DEVICE memory that the host writes through the PCIe BAR (the poll stays on the GPU; needs a large BAR)
// The response always goes to pinned host memory (a posted PCIe write).   hipcc --offload-arch=gfx950 -O3 -o tools/mailbox_latency tools/mailbox_latency.hip
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
#include <stdlib.h>
#include <chrono>
#include <signal.h>
#include <setjmp.h>

__global__ void k_pong(const uint32_t* req, uint32_t* rsp, uint32_t n_iter, int payload_dwords) {
    uint32_t served = 0;
    const uint64_t born = wall_clock64();
    while (served < n_iter) {
        const uint32_t tag = __hip_atomic_load(req, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        if (tag == served) {
            if (wall_clock64() - born > 100000000ull * 20) break;  // ~20 s at 100 MHz: never hang the box
            __builtin_amdgcn_s_sleep(2);
            continue;
        }
        __atomic_thread_fence(__ATOMIC_ACQUIRE);
        uint32_t acc = tag;
        for (int i = 1; i <= payload_dwords; ++i) acc += __hip_atomic_load(req + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM) & 0u;
        typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
        const u32x4 g = {acc, 1u, 2u, tag};
        asm volatile("global_store_dwordx4 %0, %1, off sc0 sc1\n\ts_waitcnt vmcnt(0)" : : "v"(rsp), "v"(g) : "memory");
        served = tag;
    }
}

static sigjmp_buf jb;
static void on_segv(int) { siglongjmp(jb, 1); }

static double pingpong(volatile uint32_t* req_host_view, const uint32_t* req_dev, volatile uint32_t* rsp, uint32_t n_iter, int payload) {
    rsp[3] = 0;
    for (int i = 0; i <= payload; ++i) req_host_view[i] = 0;
    hipLaunchKernelGGL(k_pong, dim3(1), dim3(1), 0, 0, req_dev, (uint32_t*)rsp, n_iter, payload);
    const auto t0 = std::chrono::steady_clock::now();
    for (uint32_t s = 1; s <= n_iter; ++s) {
        for (int i = 1; i <= payload; ++i) req_host_view[i] = s + i;
        __atomic_thread_fence(__ATOMIC_RELEASE);
        req_host_view[0] = s;
        uint64_t spins = 0;
        while (rsp[3] != s) { if (++spins > 400000000ull) { printf("timeout at %u\n", s); return -1; } }
    }
    const auto t1 = std::chrono::steady_clock::now();
    (void)hipDeviceSynchronize();
    return std::chrono::duration<double, std::micro>(t1 - t0).count() / n_iter;
}

int main() {
    uint32_t *rsp, *req_pinned;
    (void)hipHostMalloc((void**)&rsp, 4096, hipHostMallocMapped);
    (void)hipHostMalloc((void**)&req_pinned, 4096, hipHostMallocMapped);
    for (int payload : {0, 12}) {
        const double us = pingpong(req_pinned, req_pinned, rsp, 20000, payload);
        printf("request in pinned host memory, %2d payload dwords: %.2f us per round trip\n", payload, us);
    }
    uint32_t* req_dev = nullptr;
    hipError_t err = hipExtMallocWithFlags((void**)&req_dev, 4096, hipDeviceMallocFinegrained);
    printf("hipExtMallocWithFlags(finegrained): %s, ptr %p\n", hipGetErrorString(err), (void*)req_dev);
    if (err == hipSuccess) {
        (void)hipMemset(req_dev, 0, 4096);
        (void)hipDeviceSynchronize();
        signal(SIGSEGV, on_segv);
        signal(SIGBUS, on_segv);
        if (sigsetjmp(jb, 1) == 0) {
            volatile uint32_t* hv = (volatile uint32_t*)req_dev;
            const uint32_t probe = hv[0];  // faults when the BAR does not map this memory for the host
            printf("host read of device memory: %u (host-accessible)\n", probe);
            for (int payload : {0, 12}) {
                const double us = pingpong(hv, req_dev, rsp, 20000, payload);
                printf("request in fine-grained DEVICE memory (host writes over the BAR), %2d payload dwords: %.2f us per round trip\n", payload, us);
            }
        } else {
            printf("host access to fine-grained device memory faulted: no large BAR mapping for it\n");
        }
    }
    // managed memory with the device as preferred location
    uint32_t* req_m = nullptr;
    err = hipMallocManaged((void**)&req_m, 4096, hipMemAttachGlobal);
    printf("hipMallocManaged: %s\n", hipGetErrorString(err));
    return 0;
}
