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
    hipMemset(d, 0, 1 << 20);
    hipFree(d);
    return 0;
}
