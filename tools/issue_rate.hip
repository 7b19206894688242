// Issue rate of an integer VALU stream by wavefronts per SIMD (hipcc --offload-arch=gfx950 -O3 -o issue_rate tools/issue_rate.hip).
//
// The per-env-terrain rollout step (k_rollout4 MODE 2) executes ~125 instructions per env-step at ~6.3 wavefront clocks per
// instruction with ONE wavefront per SIMD (65 536 envs = 1 024 wavefronts on 1 024 SIMDs).  Question this tool answers before
// any kernel is restructured: when a SIMD hosts TWO wavefronts with half of that instruction stream each (a mover wavefront:
// Philox + resolve_movement, and an interact wavefront: look-ups + env effects + rewards), does the pair finish in about the
// time of ONE of them (their instructions interleave in the SIMD's issue slots) or in the sum?
//
// Kernels: a loop whose body is 64 integer VALU instructions in CHAINS independent chains (xor / add / shift / perm / mul24 —
// the instruction mix of the step), optionally with one ds_read_b32 + dependent use per 8 VALU.  Launched on 256 workgroups of
// 256 / 512 / 1 024 threads (1 / 2 / 4 wavefronts per SIMD); printed: clocks per instruction per wavefront (s_memtime) and per
// SIMD (= that / wavefronts per SIMD).  "per SIMD" halving from 1 to 2 wavefronts = the split pays.
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <vector>

template <int CHAINS, bool LDS>
__global__ __launch_bounds__(1024) void k_stream(uint32_t* out, uint64_t* clk, int iters, uint32_t seed) {
    __shared__ uint32_t tab[1024];
    tab[threadIdx.x & 1023] = threadIdx.x * 2654435761u + seed;
    __syncthreads();
    uint32_t x[CHAINS];
#pragma unroll
    for (int c = 0; c < CHAINS; ++c) x[c] = seed + threadIdx.x * (c + 3u);
    const uint64_t t0 = __builtin_readcyclecounter();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int k = 0; k < 64 / CHAINS; ++k) {
#pragma unroll
            for (int c = 0; c < CHAINS; ++c) {
                uint32_t v = x[c];
                switch ((k * CHAINS + c) & 7) {
                    case 0: v = v ^ (v >> 7); break;
                    case 1: v = v + 0x9E3779B9u; break;
                    case 2: v = __builtin_amdgcn_perm(v, seed, 0x02010003u); break;
                    case 3: v = (v & 0xFFFFu) * 37u + 11u; break;       // v_mad_u32_u24
                    case 4: v = v < 1000u ? v + 7u : v ^ seed; break;   // compare + select
                    case 5: v = (v << 3) + seed; break;                 // v_lshl_add_u32
                    case 6: v = min(v, 0x7FFFFFFFu) | 1u; break;
                    default: v = v - (v >> 3); break;
                }
                x[c] = v;
            }
            if (LDS && (k & 1) == 1) x[0] ^= tab[(x[1] >> 5) & 1023u];
        }
    }
    const uint64_t t1 = __builtin_readcyclecounter();
    uint32_t acc = 0;
#pragma unroll
    for (int c = 0; c < CHAINS; ++c) acc ^= x[c];
    out[(size_t)blockIdx.x * blockDim.x + threadIdx.x] = acc;
    if ((threadIdx.x & 63) == 0) clk[((size_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6] = t1 - t0;
}

template <int CHAINS, bool LDS>
static void run(const char* name, uint32_t* out, uint64_t* clk, int iters) {
    for (int threads : {256, 512, 1024}) {
        const int blocks = 256, waves = blocks * threads / 64;
        hipEvent_t a, b;
        hipEventCreate(&a); hipEventCreate(&b);
        hipLaunchKernelGGL((k_stream<CHAINS, LDS>), dim3(blocks), dim3(threads), 0, 0, out, clk, iters, 12345u);  // warm
        hipDeviceSynchronize();
        hipEventRecord(a);
        hipLaunchKernelGGL((k_stream<CHAINS, LDS>), dim3(blocks), dim3(threads), 0, 0, out, clk, iters, 12345u);
        hipEventRecord(b);
        hipEventSynchronize(b);
        float ms = 0;
        hipEventElapsedTime(&ms, a, b);
        std::vector<uint64_t> h(waves);
        hipMemcpy(h.data(), clk, waves * sizeof(uint64_t), hipMemcpyDeviceToHost);
        double mean = 0;
        for (uint64_t v : h) mean += (double)v;
        mean /= waves;
        const double per_instr = mean / ((double)iters * 64.0);  // (the loop's own counter / branch instructions not counted)
        printf("%-28s %4d threads/WG = %d waves/SIMD: %8.3f ms  s_memtime ticks per VALU instr: per wavefront %6.3f, per SIMD %6.3f\n",
               name, threads, threads / 256, ms, per_instr, per_instr / (threads / 256));
    }
}

int main() {
    uint32_t* out; uint64_t* clk;
    hipMalloc(&out, 256 * 1024 * sizeof(uint32_t));
    hipMalloc(&clk, 256 * 16 * sizeof(uint64_t));
    const int iters = 20000;
    // (s_memtime counts at the constant 100 MHz reference clock on gfx9; wall time from the events is the portable figure:
    //  ns per instruction per wavefront = ms * 1e6 / (iters * 64))
    run<1, false>("1 chain (fully dependent)", out, clk, iters);
    run<2, false>("2 chains", out, clk, iters);
    run<4, false>("4 chains", out, clk, iters);
    run<8, false>("8 chains", out, clk, iters);
    run<4, true>("4 chains + LDS read / 8", out, clk, iters);
    return 0;
}
