// store_front.hip — what shape of streaming store reaches the HBM write rate of MI355X (torch's fill: 6.9 TB/s) and what
// shape stays at the 5.3-5.8 TB/s of the persistent encode / rollout kernels?  VERDICT r4 item 4 / NOTEBOOK lead 4.
//   hipcc --offload-arch=gfx950 -O3 -o tools/store_front tools/store_front.hip && tools/store_front [total_MiB]
// Every variant writes the same buffer with global_store_dwordx4; varied: bytes per workgroup (the "piece"), its alignment
// to 128-byte lines, threads per workgroup, data from registers or from an LDS image that the workgroup fills first (the
// encode kernels' shape), persistence, and which XCD (workgroup id % 8) writes which part of the buffer.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
#include <stdlib.h>

struct P {
    int chunks;      // 16-byte chunks per piece
    int n_pieces;
    int src_lds;     // 1: fill an LDS image of the piece first, then stream it
    int remap;       // 0: piece = workgroup id; 1: XCD x (= id % 8) writes the x-th contiguous eighth; 2: runs of 8 pieces per XCD
    int persistent;  // 1: gridDim.x workgroups stride over the pieces
    int aligned;     // 1: thread t stores the chunks whose ABSOLUTE index is t mod blockDim: every round of the loop writes one aligned 4 KiB (256 threads)
};

__global__ void k_store(uint4* __restrict__ dst, P p) {
    extern __shared__ uint4 s_img[];
    const int T = blockDim.x;
    for (size_t b = blockIdx.x; b < (size_t)p.n_pieces; b += gridDim.x) {
        size_t pidx = b;
        if (p.remap == 1) pidx = (b & 7) * (size_t)(p.n_pieces >> 3) + (b >> 3);
        if (p.remap == 2) pidx = ((b >> 6) << 6) + ((b & 7) << 3) + ((b >> 3) & 7);
        uint4* piece = dst + pidx * p.chunks;
        if (p.src_lds) {
            for (int i = threadIdx.x; i < p.chunks; i += T) s_img[i] = make_uint4(i, (uint32_t)b, 2, 3);
            __syncthreads();
            const int first = p.aligned ? (int)((threadIdx.x - (unsigned)((pidx * p.chunks) % T) + T) % T) : (int)threadIdx.x;
            for (int i = first; i < p.chunks; i += T) piece[i] = s_img[i];
            if (p.persistent) __syncthreads();
        } else {
            const uint4 v = make_uint4(threadIdx.x, (uint32_t)b, 2, 3);
            for (int i = threadIdx.x; i < p.chunks; i += T) piece[i] = v;
        }
        if (!p.persistent) break;
    }
}

static hipEvent_t e0, e1;
static uint4* d;

static void run(const char* what, size_t total, int piece_bytes, int threads, int src_lds, int remap, int grid /* 0 = one workgroup per piece */, int aligned = 0) {
    P p;
    p.chunks = piece_bytes / 16;
    p.n_pieces = (int)(total / piece_bytes) / 64 * 64;
    p.src_lds = src_lds; p.remap = remap; p.persistent = grid != 0; p.aligned = aligned;
    const size_t lds = src_lds ? (size_t)piece_bytes : 0;
    if (lds > 40000) (void)hipFuncSetAttribute((const void*)k_store, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    const int g = grid ? grid : p.n_pieces;
    hipLaunchKernelGGL(k_store, dim3(g), dim3(threads), lds, 0, d, p);
    (void)hipDeviceSynchronize();
    const int reps = total > ((size_t)1 << 30) ? 6 : 40;
    (void)hipEventRecord(e0);
    for (int r = 0; r < reps; ++r) hipLaunchKernelGGL(k_store, dim3(g), dim3(threads), lds, 0, d, p);
    (void)hipEventRecord(e1);
    (void)hipEventSynchronize(e1);
    float ms = 0;
    (void)hipEventElapsedTime(&ms, e0, e1);
    const hipError_t err = hipGetLastError();
    const double s = ms * 1e-3 / reps, bytes = (double)p.n_pieces * piece_bytes;
    printf("%-10s %4zu MiB  piece %6d B (%7.2f lines) %4d thr %-9s remap %d %-16s: %5.2f TB/s (%.3f)  %7.1f us%s\n", what, total >> 20, piece_bytes,
           piece_bytes / 128.0, threads, src_lds ? "LDS image" : "registers", remap, grid ? "persistent" : "one wg per piece", bytes / s / 1e12,
           bytes / s / 8e12, s * 1e6, err != hipSuccess ? "  !! launch error" : "");
}

int main(int argc, char** argv) {
    (void)hipEventCreate(&e0);
    (void)hipEventCreate(&e1);
    const size_t big = (size_t)(argc > 1 ? atoi(argv[1]) : 2048) << 20;
    (void)hipMalloc(&d, big + (1 << 20));
    (void)hipMemset(d, 0, big);
    for (int al = 0; al < 2; ++al) {
        printf("aligned bursts = %d\n", al);
        for (int pb : {37440, 74880, 28080}) run("burst", big, pb, 256, 1, 0, 0, al);
        for (int pb : {37440, 74880}) run("burst", big, pb, 256, 1, 1, 0, al);
        for (int pb : {37440, 74880}) run("burst", big, pb, 256, 1, 0, 512, al);
        for (int pb : {37440, 74880}) run("burst", big, pb, 512, 1, 1, 0, al);
    }
    return 0;
    // (1) piece size, line-aligned pieces, from registers and through an LDS image
    for (int src = 0; src < 2; ++src)
        for (int pb : {1024, 2048, 4096, 8192, 16384, 32768, 65536}) run("size", big, pb, 256, src, 0, 0);
    for (int pb : {2048, 4096, 8192, 16384}) run("size", big, pb, 128, 1, 0, 0);
    for (int pb : {1024, 2048, 4096, 8192}) run("size", big, pb, 64, 1, 0, 0);
    for (int pb : {8192, 16384, 32768, 65536}) run("size", big, pb, 512, 1, 0, 0);
    for (int pb : {16384, 32768, 65536}) run("size", big, pb, 1024, 1, 0, 0);
    // (2) alignment: pieces of whole lines that are not powers of two, and pieces that end inside a line
    for (int pb : {4224, 4160, 4112, 9344, 9360, 37376, 37440, 74880, 8320, 1040 * 4}) run("align", big, pb, 256, 1, 0, 0);
    // (3) which XCD writes what
    for (int rm : {1, 2})
        for (int pb : {4096, 8192, 16384, 65536, 74880}) run("xcd", big, pb, 256, 1, rm, 0);
    // (4) persistent workgroups striding over the pieces
    for (int grid : {256, 512, 1024, 2048})
        for (int pb : {4096, 16384, 74880}) run(grid == 256 ? "persist256" : grid == 512 ? "persist512" : grid == 1024 ? "persist1k" : "persist2k", big, pb, 256, 1, 0, grid);
    // (5) one encode launch's worth (65 536 envs x 2 340 B = 146 MiB, rewritten in place: the MALL absorbs part of it)
    const size_t small = (size_t)65536 * 2340;
    for (int pb : {4096, 8192, 16384, 37440, 74880}) run("small", small, pb, 256, 1, 0, 0);
    for (int pb : {37440, 74880}) run("small", small, pb, 256, 1, 0, 512);
    (void)hipFree(d);
    return 0;
}
