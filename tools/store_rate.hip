// store_rate.hip — what sets a CU's streaming-store rate on MI355X: hipcc --offload-arch=gfx950 -O3 -o /tmp/store_rate tools/store_rate.hip
// Every wavefront writes its own contiguous share of a 4 GiB buffer with global_store_dwordx4 (1 KiB per instruction), the way
// k_rollout_encode / k_encode stream their images; varied: wavefronts per CU, data from registers or from LDS, and pauses of
// ALU work between bursts of 28 stores (the build of the next image).
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>

template <int SRC_LDS, int PAUSE>
__global__ __launch_bounds__(256) void k_store(uint4* __restrict__ dst, size_t chunks_per_wave, int burst, size_t REGION) {
    __shared__ uint4 s_img[2048];  // 32 KB
    for (int i = threadIdx.x; i < 2048; i += 256) s_img[i] = make_uint4(i, 1, 2, 3);
    __syncthreads();
    const int lane = threadIdx.x & 63;
    const size_t wave = (size_t)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    // like the rollout: in "step" s wavefront w writes region s * n_waves + w of REGION KiB (146 KiB: not a power of two — with
    // one contiguous power-of-two share per wavefront every wavefront sits on the same HBM channel at any moment: 3.5 TB/s)
    const size_t n_waves = (size_t)gridDim.x * (blockDim.x >> 6);
    uint4 v = make_uint4((uint32_t)wave, lane, 7, 9);
    uint32_t acc = lane;
    for (size_t c = 0; c < chunks_per_wave; ++c) {
        if (SRC_LDS) v = s_img[((c & 31) << 6) + lane];
        const size_t sidx = c / REGION, i = c - sidx * REGION;
        dst[((sidx * n_waves + wave) * REGION + i) * 64 + lane] = v;
        if (PAUSE && (c % burst) == burst - 1) {  // dependent ALU chain: ~PAUSE x 8 clk
#pragma unroll 1
            for (int k = 0; k < PAUSE; ++k) acc = acc * 1664525u + 1013904223u;
            v.w = acc;
        }
    }
}

template <int SRC_LDS, int PAUSE>
void run(const char* name, uint4* d, size_t total_chunks, int cus, int wpc, int burst, size_t region = 146) {
    const int waves = cus * wpc;
    const size_t cpw = total_chunks / 64 / waves / region * region;  // whole regions
    const int wpb = wpc >= 4 ? 4 : wpc;
    dim3 grid(waves / wpb), block(wpb * 64);
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL((k_store<SRC_LDS, PAUSE>), grid, block, 0, 0, d, cpw, burst, region);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    for (int r = 0; r < 8; ++r) hipLaunchKernelGGL((k_store<SRC_LDS, PAUSE>), grid, block, 0, 0, d, cpw, burst, region);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms = 0; hipEventElapsedTime(&ms, e0, e1);
    const double bytes = 8.0 * (double)cpw * 64 * 16 * waves;
    printf("%-34s region %3zu KiB wpc %2d: %6.2f TB/s  = %5.2f B/clk/CU at 2.4 GHz (%.3f ms per 4 GiB)\n", name, region, wpc, bytes / (ms * 1e-3) / 1e12,
           bytes / (ms * 1e-3) / cus / 2.4e9, ms / 8);
}


// The rollout's own output pattern with nothing else in the loop: in step k wavefront w stores its 64 reward quads (1 KiB at
// rewards[k][64 w ...]) and its 64 flag bytes (flags[k][64 w ...]) — 17 bytes per env-step, rows of n = 64 x n_waves envs.
template <int PAUSE, int FLAGS = 1>
__global__ __launch_bounds__(256) void k_outputs(float4* __restrict__ rew, uint8_t* __restrict__ fl, int n_steps) {
    const int lane = threadIdx.x & 63;
    const size_t wave = (size_t)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    const size_t n = (size_t)gridDim.x * blockDim.x;
    float4 v = make_float4(0.f, 0.f, (float)lane, 1.f);
    uint32_t acc = lane;
    for (int k = 0; k < n_steps; ++k) {
        rew[(size_t)k * n + wave * 64 + lane] = v;
        if (FLAGS == 1) fl[(size_t)k * n + wave * 64 + lane] = (uint8_t)acc;
        if (FLAGS == 2 && (k & 15) == 15) {  // the flag bytes of 16 steps in one store: lane -> (row k - 15 + lane / 4, 16-byte segment lane % 4)
            *reinterpret_cast<uint4*>(fl + (size_t)(k - 15 + (lane >> 2)) * n + wave * 64 + (lane & 3) * 16) = make_uint4(acc, 1, 2, 3);
        }
        if (FLAGS == 4 && (k & 15) == 15 && (wave & 1) == 0) {  // even wavefronts store the pair's flag bytes of 16 steps: 16 rows x 128 B = FULL lines
#pragma unroll
            for (int h = 0; h < 2; ++h)  // lane -> (row k - 15 + 8 h + lane / 8, 16-byte segment lane % 8 of the pair's 128 bytes)
                *reinterpret_cast<uint4*>(fl + (size_t)(k - 15 + 8 * h + (lane >> 3)) * n + wave * 64 + (lane & 7) * 16) = make_uint4(acc, 1, 2, 3);
        }
        if (FLAGS == 3 && (k & 3) == 3) {  // four steps: lane -> (row k - 3 + lane / 16, dword lane % 16)
            *reinterpret_cast<uint32_t*>(fl + (size_t)(k - 3 + (lane >> 4)) * n + wave * 64 + (lane & 15) * 4) = acc;
        }
        if (PAUSE) {
#pragma unroll 1
            for (int j = 0; j < PAUSE; ++j) acc = acc * 1664525u + 1013904223u;
            v.w = (float)(acc & 1u);
        }
    }
}

template <int PAUSE, int FLAGS = 1>
void run_outputs(const char* name, void* d, int cus, int wpc, int n_steps, size_t fl_shift = 0) {
    const int waves = cus * wpc;
    const size_t n = (size_t)waves * 64;
    float4* rew = (float4*)d;
    uint8_t* fl = (uint8_t*)d + (size_t)n_steps * n * 16 + fl_shift;
    dim3 grid(waves / 4), block(256);
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL((k_outputs<PAUSE, FLAGS>), grid, block, 0, 0, rew, fl, n_steps);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    for (int r = 0; r < 8; ++r) hipLaunchKernelGGL((k_outputs<PAUSE, FLAGS>), grid, block, 0, 0, rew, fl, n_steps);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms = 0; hipEventElapsedTime(&ms, e0, e1);
    const double bytes = 8.0 * (FLAGS ? 17.0 : 16.0) * (double)n * n_steps;  // (n_steps: a multiple of 16)
    printf("%-44s (flags + %8zu B) %7zu envs x %d steps: %6.2f TB/s = %5.3f of 8 TB/s, %6.1f G env-steps/s, %.1f clk per step at 2.4 GHz\n", name, fl_shift, n, n_steps,
           bytes / (ms * 1e-3) / 1e12, bytes / (ms * 1e-3) / 8e12, 8.0 * n * n_steps / (ms * 1e-3) / 1e9, ms * 1e-3 / 8 / n_steps * 2.4e9);
}

int main() {
    int cus = 256;
    hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, 0);
    const size_t total = (size_t)4 << 30;
    uint4* d;
    hipMalloc(&d, total);
    const size_t chunks = total / 16;
    for (size_t region : {1, 2, 4, 8, 18, 28, 64, 146, 585}) run<1, 0>("LDS-sourced, no pause", d, chunks, cus, 4, 28, region);
    for (size_t region : {1, 18, 146}) run<1, 400>("LDS, 28-store bursts, ~3.2k clk pause", d, chunks, cus, 4, 28, region);
    for (size_t region : {1, 18, 146}) run<0, 0>("registers, no pause", d, chunks, cus, 16, 28, region);
    run_outputs<0>("rollout outputs only (quad + flag byte)", d, cus, 4, 3800);
    run_outputs<0>("rollout outputs only (quad + flag byte), again", d, cus, 4, 3800);
    run_outputs<0, 0>("reward quads only", d, cus, 4, 3808);
    run_outputs<0, 2>("quads + flags of 16 steps per dwordx4 store", d, cus, 4, 3808);
    run_outputs<0, 4>("quads + a wavefront PAIR's flags of 16 steps, full lines", d, cus, 4, 3808);
    run_outputs<0, 1>("quads + flag byte per step (as shipped)", d, cus, 4, 3808);
    run_outputs<0>("rollout outputs only, 2 wavefronts per SIMD", d, cus, 8, 1900);
    run_outputs<0>("rollout outputs only, 16 wavefronts per CU", d, cus, 16, 950);
    hipMemset(d, 0, 1 << 20);
    hipFree(d);
    return 0;
}
