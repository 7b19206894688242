// store_rollout.hip — the rollout's own output pattern with nothing else in the loop (persistent workgroups of 4 independent
// wavefronts, 256 envs each; per step a wavefront stores its 64 reward quads = 1 KiB of row k), varied: the flags format and
// which XCD (workgroup id % 8) owns which envs.   hipcc --offload-arch=gfx950 -O3 -o tools/store_rollout tools/store_rollout.hip
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
#include <string.h>

// FLAGS: 0 none; 1 one byte per env-step, [step][env]; 8 tiled by 8 steps ([steps/8][env][8]: 512 B per wavefront per 8 steps);
// 16 tiled by 16 steps ([steps/16][env][16]: 1 KiB per wavefront, an aligned 4 KiB per workgroup, per 16 steps)
// MODE: 0 as the rollout kernels store; 1 a workgroup barrier before every step's stores (the 4 KiB of a workgroup leave together);
// 2 quads tiled by 4 steps ([steps/4][env][4] quads: an aligned 4 KiB per wavefront per 4 steps, each lane 64 contiguous bytes);
// 3 = 2 with the four stores transposed through registers so that every store instruction writes 1 KiB of contiguous bytes
template <int FLAGS, int WAVES, int MODE = 0>
__global__ __launch_bounds__(WAVES * 64) void k_out(float4* __restrict__ rew, uint8_t* __restrict__ fl, int n_steps, int remap) {
    __shared__ uint32_t s_prog[4];
    if (threadIdx.x < 4) s_prog[threadIdx.x] = 0;
    __syncthreads();
    const int lane = threadIdx.x & 63;
    uint32_t blk = blockIdx.x;
    if (remap == 1) blk = (blockIdx.x & 7u) * (gridDim.x >> 3) + (blockIdx.x >> 3);  // XCD x owns the x-th contiguous eighth of the envs
    const size_t e = (size_t)blk * (WAVES * 64) + threadIdx.x;
    const size_t n = (size_t)gridDim.x * (WAVES * 64);
    float4 v = make_float4(0.f, 0.f, (float)lane, 1.f);
    uint32_t acc = lane;
    uint32_t wrng = (uint32_t)__builtin_amdgcn_readfirstlane((int)((blockIdx.x * 4 + (threadIdx.x >> 6)) * 2654435761u + 12345u));
    float4 q[4];
    for (int k = 0; k < n_steps; ++k) {
        if (MODE == 1) __syncthreads();
        if (MODE >= 10 && MODE < 100 && (k % (MODE - 10)) == 0) __syncthreads();
        if (MODE >= 200) {  // MODE 200 + P: ~7 % of the wavefront-steps run ~430 extra clocks (the rollout's rare branch); P > 0: a barrier every P steps
            wrng = wrng * 1664525u + 1013904223u;
            if (((wrng >> 10) % 100u) < 7u) {
#pragma unroll 1
                for (int j = 0; j < 54; ++j) acc = acc * 1664525u + 1013904223u;  // (a dependent multiply-add: ~8 clk each)
            }
            if (MODE > 200 && (k % (MODE - 200)) == 0) __syncthreads();
        }
        if (MODE >= 100 && MODE < 200 && (k & 7) == 0) {  // MODE 100 + D: no wavefront starts 8-step block b before all four have finished block b - D
            const uint32_t b = (uint32_t)k >> 3;
            if (lane == 0) *(volatile uint32_t*)&s_prog[threadIdx.x >> 6] = b;  // blocks finished
            for (;;) {
                const uint32_t m = min(min(*(volatile uint32_t*)&s_prog[0], *(volatile uint32_t*)&s_prog[1]), min(*(volatile uint32_t*)&s_prog[2], *(volatile uint32_t*)&s_prog[3]));
                if (m + (uint32_t)(MODE - 100) >= b) break;
                __builtin_amdgcn_s_sleep(1);
            }
        }  // MODE 10 + P: a workgroup barrier every P steps
        if (MODE == 0 || MODE == 1 || MODE >= 10) rew[(size_t)k * n + e] = v;
        if (MODE == 2 || MODE == 3) {
            q[k & 3] = v;
            if ((k & 3) == 3) {
                float4* dst = rew + ((size_t)(k >> 2) * n + e) * 4;
                if (MODE == 2) { dst[0] = q[0]; dst[1] = q[1]; dst[2] = q[2]; dst[3] = q[3]; }
                else {  // instruction i writes chunk (64 i + lane) of the wavefront's 256: env (64 i + lane) / 4, step (64 i + lane) % 4
                    float4* wdst = rew + ((size_t)(k >> 2) * n + (e - lane)) * 4;
#pragma unroll
                    for (int i = 0; i < 4; ++i) {
                        const int src_lane = (64 * i + lane) >> 2, st = lane & 3;
                        float4 o;
                        const float4 a = q[0], b = q[1], c = q[2], dd = q[3];
                        const float4 pick = st == 0 ? a : st == 1 ? b : st == 2 ? c : dd;  // (the lane-local pick is wrong data-wise, right cost-wise: 4 bpermutes below)
                        o.x = __shfl(pick.x, src_lane); o.y = __shfl(pick.y, src_lane); o.z = __shfl(pick.z, src_lane); o.w = __shfl(pick.w, src_lane);
                        wdst[64 * i + lane] = o;
                    }
                }
            }
        }
        if (FLAGS == 1) fl[(size_t)k * n + e] = (uint8_t)acc;
        if (FLAGS == 8 && (k & 7) == 7) *reinterpret_cast<uint2*>(fl + ((size_t)(k >> 3) * n + e) * 8) = make_uint2(acc, k);
        if (FLAGS == 16 && (k & 15) == 15) *reinterpret_cast<uint4*>(fl + ((size_t)(k >> 4) * n + e) * 16) = make_uint4(acc, k, 2, 3);
        acc = acc * 1664525u + 1013904223u;
        v.w = (float)(acc >> 31);
    }
}

// Wavefront tiles: rewards as [steps / TS][n / 64][TS][64] quads — every wavefront keeps the quads of TS steps in registers and
// then writes its own aligned TS KiB tile with TS back-to-back 1 KiB stores; flags as [steps / 64][n / 64][4][64][16] bytes (FL64:
// an aligned 4 KiB per wavefront per 64 steps) or tiled by 16 steps per env as above.
template <int TS, int FL64>
__global__ __launch_bounds__(256) void k_tiles(float4* __restrict__ rew, uint8_t* __restrict__ fl, int n_steps, int remap) {
    const int lane = threadIdx.x & 63;
    uint32_t blk = blockIdx.x;
    if (remap == 1) blk = (blockIdx.x & 7u) * (gridDim.x >> 3) + (blockIdx.x >> 3);
    const size_t wave = (size_t)blk * 4 + (threadIdx.x >> 6);
    const size_t n = (size_t)gridDim.x * 256, nw = n / 64;
    const size_t e = wave * 64 + lane;
    float4 q[TS];
    uint4 f[4];
    float4 v = make_float4(0.f, 0.f, (float)lane, 1.f);
    uint32_t acc = lane;
    for (int k = 0; k < n_steps; ++k) {
        q[k % TS] = v;
        if ((k % TS) == TS - 1) {
            float4* tile = rew + ((size_t)(k / TS) * nw + wave) * (TS * 64);
#pragma unroll
            for (int i = 0; i < TS; ++i) tile[i * 64 + lane] = q[i];
        }
        if (FL64) {
            if ((k & 15) == 15) f[(k >> 4) & 3] = make_uint4(acc, k, 2, 3);
            if ((k & 63) == 63) {
                uint4* tile = reinterpret_cast<uint4*>(fl) + ((size_t)(k >> 6) * nw + wave) * 256;
#pragma unroll
                for (int i = 0; i < 4; ++i) tile[i * 64 + lane] = f[i];
            }
        } else if ((k & 15) == 15) {
            *reinterpret_cast<uint4*>(fl + ((size_t)(k >> 4) * n + e) * 16) = make_uint4(acc, k, 2, 3);
        }
        acc = acc * 1664525u + 1013904223u;
        v.w = (float)(acc >> 31);
    }
}

// POL: 0 plain, 1 nt, 2 sc1, 3 sc0 sc1, 4 sc0 — the reward-quad store's cache policy bits (flags tiled by 8, plain)
template <int POL>
__global__ __launch_bounds__(256) void k_pol(float4* __restrict__ rew, uint8_t* __restrict__ fl, int n_steps) {
    const size_t e = (size_t)blockIdx.x * 256 + threadIdx.x;
    const size_t n = (size_t)gridDim.x * 256;
    typedef float f32x4 __attribute__((ext_vector_type(4)));
    f32x4 v = {0.f, 0.f, (float)(threadIdx.x & 63), 1.f};
    uint32_t acc = threadIdx.x;
    for (int k = 0; k < n_steps; ++k) {
        float4* p = rew + (size_t)k * n + e;
        if (POL == 0) asm volatile("global_store_dwordx4 %0, %1, off" : : "v"(p), "v"(v) : "memory");
        if (POL == 1) asm volatile("global_store_dwordx4 %0, %1, off nt" : : "v"(p), "v"(v) : "memory");
        if (POL == 2) asm volatile("global_store_dwordx4 %0, %1, off sc1" : : "v"(p), "v"(v) : "memory");
        if (POL == 3) asm volatile("global_store_dwordx4 %0, %1, off sc0 sc1" : : "v"(p), "v"(v) : "memory");
        if (POL == 4) asm volatile("global_store_dwordx4 %0, %1, off sc0" : : "v"(p), "v"(v) : "memory");
        if (POL == 5) asm volatile("global_store_dwordx4 %0, %1, off sc0 sc1 nt" : : "v"(p), "v"(v) : "memory");
        if (POL == 6) asm volatile("global_store_dwordx4 %0, %1, off sc1 nt" : : "v"(p), "v"(v) : "memory");
        if (POL == 7) asm volatile("global_store_dwordx4 %0, %1, off sc0 nt" : : "v"(p), "v"(v) : "memory");
        if (POL >= 8) asm volatile("global_store_dwordx4 %0, %1, off sc0 sc1 nt" : : "v"(p), "v"(v) : "memory");
        if ((k & 7) == 7) {
            typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));
            const u32x2 t = {acc, (uint32_t)k};
            uint8_t* q = fl + ((size_t)(k >> 3) * n + e) * 8;
            if (POL == 8) asm volatile("global_store_dwordx2 %0, %1, off nt" : : "v"(q), "v"(t) : "memory");
            else if (POL == 9) asm volatile("global_store_dwordx2 %0, %1, off sc0 sc1 nt" : : "v"(q), "v"(t) : "memory");
            else *reinterpret_cast<uint2*>(q) = make_uint2(acc, k);
        }
        acc = acc * 1664525u + 1013904223u;
        v.w = (float)(acc >> 31);
    }
}

// the [step][env] flags layout (one byte per env-step, 64 bytes per wavefront and row): cache-policy bits of the BYTE store
// BP: 0 plain, 1 nt, 2 sc1 nt, 3 sc0 sc1 nt, 4 sc1; QP: the quad store's policy (0 plain, 1 sc1 nt)
template <int BP, int QP>
__global__ __launch_bounds__(256) void k_bytes(float4* __restrict__ rew, uint8_t* __restrict__ fl, int n_steps) {
    const size_t e = (size_t)blockIdx.x * 256 + threadIdx.x;
    const size_t n = (size_t)gridDim.x * 256;
    typedef float f32x4 __attribute__((ext_vector_type(4)));
    f32x4 v = {0.f, 0.f, (float)(threadIdx.x & 63), 1.f};
    uint32_t acc = threadIdx.x;
    for (int k = 0; k < n_steps; ++k) {
        float4* p = rew + (size_t)k * n + e;
        uint8_t* q = fl + (size_t)k * n + e;
        if (QP == 0) asm volatile("global_store_dwordx4 %0, %1, off" : : "v"(p), "v"(v) : "memory");
        else asm volatile("global_store_dwordx4 %0, %1, off sc1 nt" : : "v"(p), "v"(v) : "memory");
        if (BP == 0) asm volatile("global_store_byte %0, %1, off" : : "v"(q), "v"(acc) : "memory");
        if (BP == 1) asm volatile("global_store_byte %0, %1, off nt" : : "v"(q), "v"(acc) : "memory");
        if (BP == 2) asm volatile("global_store_byte %0, %1, off sc1 nt" : : "v"(q), "v"(acc) : "memory");
        if (BP == 3) asm volatile("global_store_byte %0, %1, off sc0 sc1 nt" : : "v"(q), "v"(acc) : "memory");
        if (BP == 4) asm volatile("global_store_byte %0, %1, off sc1" : : "v"(q), "v"(acc) : "memory");
        acc = acc * 1664525u + 1013904223u;
        v.w = (float)(acc >> 31);
    }
}
template <int BP, int QP>
void run_bytes(const char* name, void* d, int n_wg, int n_steps);

static hipEvent_t e0, e1;
template <int BP, int QP>
void run_bytes(const char* name, void* d, int n_wg, int n_steps) {
    const size_t n = (size_t)n_wg * 256;
    float4* rew = (float4*)d;
    uint8_t* fl = (uint8_t*)d + (size_t)n_steps * n * 16;
    double sum = 0;
    for (int rep = 0; rep < 3; ++rep) {
        hipLaunchKernelGGL((k_bytes<BP, QP>), dim3(n_wg), dim3(256), 0, 0, rew, fl, n_steps);
        (void)hipDeviceSynchronize();
        (void)hipEventRecord(e0);
        for (int r = 0; r < 6; ++r) hipLaunchKernelGGL((k_bytes<BP, QP>), dim3(n_wg), dim3(256), 0, 0, rew, fl, n_steps);
        (void)hipEventRecord(e1);
        (void)hipEventSynchronize(e1);
        float ms = 0;
        (void)hipEventElapsedTime(&ms, e0, e1);
        sum += 6.0 * 17.0 * (double)n * n_steps / (ms * 1e-3) / 1e12;
    }
    printf("quads + flag byte [step][env], %-28s %7zu envs x %d steps: mean %5.2f TB/s = %.3f of 8; %6.1f G env-steps/s\n", name, n, n_steps, sum / 3, sum / 3 / 8, sum / 3 * 1e12 / 17.0 / 1e9);
}
template <int POL>
void run_pol(const char* name, void* d, int n_wg, int n_steps) {
    const size_t n = (size_t)n_wg * 256;
    float4* rew = (float4*)d;
    uint8_t* fl = (uint8_t*)d + (size_t)n_steps * n * 16;
    double sum = 0;
    for (int rep = 0; rep < 3; ++rep) {
        hipLaunchKernelGGL((k_pol<POL>), dim3(n_wg), dim3(256), 0, 0, rew, fl, n_steps);
        (void)hipDeviceSynchronize();
        (void)hipEventRecord(e0);
        for (int r = 0; r < 6; ++r) hipLaunchKernelGGL((k_pol<POL>), dim3(n_wg), dim3(256), 0, 0, rew, fl, n_steps);
        (void)hipEventRecord(e1);
        (void)hipEventSynchronize(e1);
        float ms = 0;
        (void)hipEventElapsedTime(&ms, e0, e1);
        sum += 6.0 * 17.0 * (double)n * n_steps / (ms * 1e-3) / 1e12;
    }
    printf("quads + flags tiled by 8, quad store %-12s %7zu envs x %d steps: mean %5.2f TB/s = %.3f of 8; %6.1f G env-steps/s\n", name, n, n_steps, sum / 3, sum / 3 / 8, sum / 3 * 1e12 / 17.0 / 1e9);
}

template <int TS, int FL64>
void run_tiles(const char* name, void* d, int n_wg, int n_steps, int remap) {
    const size_t n = (size_t)n_wg * 256;
    float4* rew = (float4*)d;
    uint8_t* fl = (uint8_t*)d + (size_t)n_steps * n * 16;
    double sum = 0;
    for (int rep = 0; rep < 3; ++rep) {
        hipLaunchKernelGGL((k_tiles<TS, FL64>), dim3(n_wg), dim3(256), 0, 0, rew, fl, n_steps, remap);
        (void)hipDeviceSynchronize();
        (void)hipEventRecord(e0);
        for (int r = 0; r < 6; ++r) hipLaunchKernelGGL((k_tiles<TS, FL64>), dim3(n_wg), dim3(256), 0, 0, rew, fl, n_steps, remap);
        (void)hipEventRecord(e1);
        (void)hipEventSynchronize(e1);
        float ms = 0;
        (void)hipEventElapsedTime(&ms, e0, e1);
        sum += 6.0 * 17.0 * (double)n * n_steps / (ms * 1e-3) / 1e12;
    }
    printf("%-44s remap %d %7zu envs x %d steps: mean %5.2f TB/s = %.3f of 8; %6.1f G env-steps/s\n", name, remap, n, n_steps, sum / 3, sum / 3 / 8, sum / 3 * 1e12 / 17.0 / 1e9);
}

template <int FLAGS, int WAVES, int MODE = 0>
void run(const char* name, void* d, int n_wg, int n_steps, int remap) {
    const size_t n = (size_t)n_wg * WAVES * 64;
    float4* rew = (float4*)d;
    uint8_t* fl = (uint8_t*)d + (size_t)n_steps * n * 16;
    double best = 0, sum = 0;
    for (int rep = 0; rep < 3; ++rep) {
        hipLaunchKernelGGL((k_out<FLAGS, WAVES, MODE>), dim3(n_wg), dim3(WAVES * 64), 0, 0, rew, fl, n_steps, remap);
        (void)hipDeviceSynchronize();
        (void)hipEventRecord(e0);
        for (int r = 0; r < 6; ++r) hipLaunchKernelGGL((k_out<FLAGS, WAVES, MODE>), dim3(n_wg), dim3(WAVES * 64), 0, 0, rew, fl, n_steps, remap);
        (void)hipEventRecord(e1);
        (void)hipEventSynchronize(e1);
        float ms = 0;
        (void)hipEventElapsedTime(&ms, e0, e1);
        const double bytes = 6.0 * (FLAGS ? 17.0 : 16.0) * (double)n * n_steps;
        const double tbs = bytes / (ms * 1e-3) / 1e12;
        sum += tbs;
        if (tbs > best) best = tbs;
    }
    printf("%-34s remap %d  %7zu envs x %d steps (%d wavefronts per workgroup): mean %5.2f TB/s = %.3f of 8, best %5.2f; %6.1f G env-steps/s\n", name, remap, n,
           n_steps, WAVES, sum / 3, sum / 3 / 8, best, sum / 3 * 1e12 / (FLAGS ? 17.0 : 16.0) / 1e9);
}

int main(int argc, char** argv) {
    // sections (any number of them as arguments; none = all): formats stride policy tiles barriers (barrier periods, jitter, drift bounds) barriers2 (a barrier per step x formats, quads tiled by 4, 131 072 envs)
    auto want = [&](const char* name) {
        if (argc < 2) return true;
        for (int i = 1; i < argc; ++i) if (!strcmp(argv[i], name)) return true;
        return false;
    };
    (void)hipEventCreate(&e0);
    (void)hipEventCreate(&e1);
    void* d;
    (void)hipMalloc(&d, (size_t)5 << 30);
    if (want("formats")) {
        for (int remap = 0; remap < 2; ++remap) {
            run<0, 4>("quads only", d, 256, 3808, remap);
            run<1, 4>("quads + flag byte [step][env]", d, 256, 3808, remap);
            run<8, 4>("quads + flags tiled by 8", d, 256, 3808, remap);
            run<16, 4>("quads + flags tiled by 16", d, 256, 3808, remap);
        }
    }
    if (want("stride")) {
        // row stride: 256 workgroups = rows of exactly 1 MiB (quads) — is the power of two special?
        for (int wg : {256, 255, 257, 248, 264, 240, 272}) {
            char nm[64];
            snprintf(nm, sizeof nm, "quads + flags tiled 8, %d workgroups", wg);
            run<8, 4>(nm, d, wg, 3808, 0);
        }
    }
    if (want("bytes")) {
        run_bytes<0, 0>("plain / plain", d, 256, 3808);
        run_bytes<1, 0>("byte nt", d, 256, 3808);
        run_bytes<2, 0>("byte sc1 nt", d, 256, 3808);
        run_bytes<3, 0>("byte sc0 sc1 nt", d, 256, 3808);
        run_bytes<4, 0>("byte sc1", d, 256, 3808);
        run_bytes<0, 1>("quad sc1 nt", d, 256, 3808);
        run_bytes<2, 1>("both sc1 nt", d, 256, 3808);
        run_bytes<0, 0>("plain / plain", d, 256, 3808);
    }
    if (want("policy")) {
        run_pol<0>("plain", d, 256, 3808);
        run_pol<1>("nt", d, 256, 3808);
        run_pol<2>("sc1", d, 256, 3808);
        run_pol<3>("sc0 sc1", d, 256, 3808);
        run_pol<4>("sc0", d, 256, 3808);
        run_pol<5>("sc0 sc1 nt", d, 256, 3808);
        run_pol<6>("sc1 nt", d, 256, 3808);
        run_pol<7>("sc0 nt", d, 256, 3808);
        run_pol<8>("sc0sc1nt, fl nt", d, 256, 3808);
        run_pol<9>("all sc0sc1nt", d, 256, 3808);
        run_pol<0>("plain", d, 256, 3808);
        run_pol<5>("sc0 sc1 nt", d, 512, 1904);
        run_pol<0>("plain", d, 512, 1904);
    }
    if (want("tiles")) {
        for (int remap = 0; remap < 2; ++remap) {
            run_tiles<4, 0>("wave tiles of 4 steps, flags tiled 16", d, 256, 3840, remap);
            run_tiles<4, 1>("wave tiles of 4 steps, flag tiles of 64 steps", d, 256, 3840, remap);
            run_tiles<8, 1>("wave tiles of 8 steps, flag tiles of 64 steps", d, 256, 3840, remap);
            run_tiles<2, 1>("wave tiles of 2 steps, flag tiles of 64 steps", d, 256, 3840, remap);
            run_tiles<1, 1>("wave rows (1 step), flag tiles of 64 steps", d, 256, 3840, remap);
            run_tiles<4, 1>("131 072 envs: tiles of 4, flag tiles of 64", d, 512, 1920, remap);
        }
        run<16, 4, 0>("tiled 16, no barrier (control)", d, 256, 3808, 0);
        run<8, 4, 0>("tiled 8, no barrier (control)", d, 256, 3808, 0);
    }
    if (want("barriers")) {
        run<16, 4, 1>("tiled 16, barrier every step", d, 256, 3808, 0);
        run<16, 4, 12>("tiled 16, barrier every 2 steps", d, 256, 3808, 0);
        run<16, 4, 14>("tiled 16, barrier every 4 steps", d, 256, 3808, 0);
        run<16, 4, 18>("tiled 16, barrier every 8 steps", d, 256, 3808, 0);
        run<16, 4, 26>("tiled 16, barrier every 16 steps", d, 256, 3808, 0);
        run<16, 4, 74>("tiled 16, barrier every 64 steps", d, 256, 3808, 0);
        run<16, 4, 0>("tiled 16, no barrier", d, 256, 3808, 0);
        run<16, 4, 200>("tiled 16, jitter, no barrier", d, 256, 3808, 0);
        run<16, 4, 201>("tiled 16, jitter, barrier every step", d, 256, 3808, 0);
        run<16, 4, 202>("tiled 16, jitter, barrier every 2", d, 256, 3808, 0);
        run<16, 4, 204>("tiled 16, jitter, barrier every 4", d, 256, 3808, 0);
        run<16, 4, 208>("tiled 16, jitter, barrier every 8", d, 256, 3808, 0);
        run<16, 4, 216>("tiled 16, jitter, barrier every 16", d, 256, 3808, 0);
        run<8, 4, 200>("tiled 8, jitter, no barrier", d, 256, 3808, 0);
        run<8, 4, 208>("tiled 8, jitter, barrier every 8", d, 256, 3808, 0);
        run<16, 4, 100>("tiled 16, drift bound 0 blocks", d, 256, 3808, 0);
        run<16, 4, 101>("tiled 16, drift bound 1 block", d, 256, 3808, 0);
        run<16, 4, 102>("tiled 16, drift bound 2 blocks", d, 256, 3808, 0);
        run<16, 4, 104>("tiled 16, drift bound 4 blocks", d, 256, 3808, 0);
        run<16, 4, 108>("tiled 16, drift bound 8 blocks", d, 256, 3808, 0);
        run<8, 4, 101>("tiled 8, drift bound 1 block", d, 256, 3808, 0);
        run<8, 4, 1>("tiled 8, barrier every step", d, 256, 3808, 0);
        run<8, 4, 18>("tiled 8, barrier every 8 steps", d, 256, 3808, 0);
        run<8, 4, 0>("tiled 8, no barrier", d, 256, 3808, 0);
        run<16, 4, 1>("131 072 envs tiled 16, barrier every step", d, 512, 1904, 1);
        run<16, 4, 18>("131 072 envs tiled 16, barrier every 8", d, 512, 1904, 1);
        run<16, 4, 0>("131 072 envs tiled 16, no barrier", d, 512, 1904, 1);
    }
    if (want("barriers2")) {
        for (int remap = 0; remap < 2; ++remap) {
            run<0, 4, 1>("quads only, barrier per step", d, 256, 3808, remap);
            run<8, 4, 1>("quads + tiled 8, barrier per step", d, 256, 3808, remap);
            run<16, 4, 1>("quads + tiled 16, barrier per step", d, 256, 3808, remap);
            run<0, 4, 2>("quads tiled by 4 steps, no flags", d, 256, 3808, remap);
            run<16, 4, 2>("quads tiled by 4 + flags tiled 16", d, 256, 3808, remap);
            run<16, 4, 3>("same, contiguous store instrs", d, 256, 3808, remap);
            run<0, 4, 0>("quads only (again)", d, 256, 3808, remap);
        }
        // two wavefronts per SIMD (BASELINE configs[4]'s 131 072 envs per GPU) and the 5-layout mix's shape
        for (int remap = 0; remap < 2; ++remap) {
            run<8, 4>("131 072 envs, flags tiled by 8", d, 512, 1904, remap);
            run<16, 4>("131 072 envs, flags tiled by 16", d, 512, 1904, remap);
        }
    }
    (void)hipFree(d);
    return 0;
}
