// store_front.hip — does a launch-ordered write front (short-lived workgroups, one contiguous piece each, dispatched in address
// order) stream faster than persistent workgroups that stride through the buffer?  VERDICT r4 item 4 / NOTEBOOK lead 4.
//   hipcc --offload-arch=gfx950 -O3 -o tools/store_front tools/store_front.hip
// Every variant writes the same 2 GiB with global_store_dwordx4; varied: bytes per workgroup, threads per workgroup, data from
// registers or from an LDS image that the workgroup fills first (the encode kernels' shape), persistence.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>

// one short-lived workgroup per contiguous piece of `chunks` 16-byte chunks
template <int THREADS, int SRC_LDS>
__global__ __launch_bounds__(THREADS) void k_front(uint4* __restrict__ dst, int chunks) {
    extern __shared__ uint4 s_img[];
    uint4* piece = dst + (size_t)blockIdx.x * chunks;
    if (SRC_LDS) {
        for (int i = threadIdx.x; i < chunks; i += THREADS) s_img[i] = make_uint4(i, blockIdx.x, 2, 3);
        __syncthreads();
        for (int i = threadIdx.x; i < chunks; i += THREADS) piece[i] = s_img[i];
    } else {
        const uint4 v = make_uint4(threadIdx.x, blockIdx.x, 2, 3);
        for (int i = threadIdx.x; i < chunks; i += THREADS) piece[i] = v;
    }
}

// persistent: gridDim.x workgroups stride over the pieces (the shipped encode kernels' shape)
template <int THREADS, int SRC_LDS>
__global__ __launch_bounds__(THREADS) void k_persist(uint4* __restrict__ dst, int chunks, int n_pieces) {
    extern __shared__ uint4 s_img[];
    for (int p = blockIdx.x; p < n_pieces; p += gridDim.x) {
        uint4* piece = dst + (size_t)p * chunks;
        if (SRC_LDS) {
            for (int i = threadIdx.x; i < chunks; i += THREADS) s_img[i] = make_uint4(i, p, 2, 3);
            __syncthreads();
            for (int i = threadIdx.x; i < chunks; i += THREADS) piece[i] = s_img[i];
            __syncthreads();
        } else {
            const uint4 v = make_uint4(threadIdx.x, p, 2, 3);
            for (int i = threadIdx.x; i < chunks; i += THREADS) piece[i] = v;
        }
    }
}

static hipEvent_t e0, e1;
template <typename F>
double timed(F launch, int reps = 8) {
    launch();
    hipDeviceSynchronize();
    hipEventRecord(e0);
    for (int r = 0; r < reps; ++r) launch();
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms = 0;
    hipEventElapsedTime(&ms, e0, e1);
    return ms * 1e-3 / reps;
}

template <int THREADS, int SRC_LDS>
void run_front(uint4* d, size_t total, int piece_bytes) {
    const int chunks = piece_bytes / 16;
    const int n_pieces = (int)(total / piece_bytes);
    const size_t lds = SRC_LDS ? (size_t)piece_bytes : 0;
    const double s = timed([&] { hipLaunchKernelGGL((k_front<THREADS, SRC_LDS>), dim3(n_pieces), dim3(THREADS), lds, 0, d, chunks); });
    printf("front   %4d thr %s piece %6d B x %7d: %6.2f TB/s (%.3f of 8)\n", THREADS, SRC_LDS ? "LDS image" : "registers", piece_bytes, n_pieces,
           (double)n_pieces * piece_bytes / s / 1e12, (double)n_pieces * piece_bytes / s / 8e12);
}

template <int THREADS, int SRC_LDS>
void run_persist(uint4* d, size_t total, int piece_bytes, int grid) {
    const int chunks = piece_bytes / 16;
    const int n_pieces = (int)(total / piece_bytes);
    const size_t lds = SRC_LDS ? (size_t)piece_bytes : 0;
    const double s = timed([&] { hipLaunchKernelGGL((k_persist<THREADS, SRC_LDS>), dim3(grid), dim3(THREADS), lds, 0, d, chunks, n_pieces); });
    printf("persist %4d thr %s piece %6d B x %7d grid %5d: %6.2f TB/s (%.3f of 8)\n", THREADS, SRC_LDS ? "LDS image" : "registers", piece_bytes, n_pieces,
           grid, (double)n_pieces * piece_bytes / s / 1e12, (double)n_pieces * piece_bytes / s / 8e12);
}

int main() {
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    const size_t total = (size_t)2 << 30;
    uint4* d;
    hipMalloc(&d, total + (1 << 20));
    hipMemset(d, 0, total);
    // (1) the fill shape: small pieces from registers
    for (int pb : {4096, 16384, 65536}) run_front<256, 0>(d, total, pb);
    // (2) pieces the size of an encode group (16 envs x 2 340 B = 37 440 B; 8 envs; 32 envs) through an LDS image
    for (int pb : {4096, 9360, 18720, 37440, 74880}) run_front<256, 1>(d, total, pb);
    for (int pb : {18720, 37440, 74880}) run_front<512, 1>(d, total, pb);
    for (int pb : {37440, 74880}) run_front<1024, 1>(d, total, pb);
    for (int pb : {9360, 18720}) run_front<128, 1>(d, total, pb);
    for (int pb : {4680, 9360}) run_front<64, 1>(d, total, pb);
    // (3) the shipped shape on the same box: persistent workgroups striding over the pieces
    for (int grid : {256, 512, 1024}) run_persist<256, 1>(d, total, 37440, grid);
    for (int grid : {256, 512}) run_persist<256, 0>(d, total, 37440, grid);
    run_persist<512, 1>(d, total, 37440, 512);
    // (4) small batch: 65 536 envs x 2 340 B = 153 MB (what one encode launch writes), front vs persistent
    const size_t small = (size_t)65536 * 2340;
    for (int pb : {9360, 18720, 37440}) run_front<256, 1>(d, small, pb);
    for (int grid : {512, 1024}) run_persist<256, 1>(d, small, 37440, grid);
    hipFree(d);
    return 0;
}
