// Host cost of the one-step API without Python: a tight C loop over oc_step (include/oc_amd.h) on one layout.
//   hipcc -O2 -o step_loop tools/step_loop.cpp -ldl ;  ./step_loop overcooked_ai_amd/liboc_amd.so layout.bin [n_envs] [iters]
// layout.bin = one 256-byte OcLayout record (python -c "from overcooked_ai_amd.layouts import ...").
// Prints the wall time per call with the queue full (GPU- or launch-bound, whichever is larger) and the host time of a call
// alone (one call, then a sync, repeated).
#include <hip/hip_runtime.h>
#include <dlfcn.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include "../include/oc_amd.h"

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

int main(int argc, char** argv) {
    if (argc < 3) { fprintf(stderr, "usage: step_loop liboc_amd.so layout.bin [n_envs] [iters]\n"); return 2; }
    void* lib = dlopen(argv[1], RTLD_NOW);
    if (!lib) { fprintf(stderr, "%s\n", dlerror()); return 1; }
    auto p_hints = (decltype(&oc_batch_hints))dlsym(lib, "oc_batch_hints");
    auto p_reset = (decltype(&oc_reset))dlsym(lib, "oc_reset");
    auto p_step = (decltype(&oc_step))dlsym(lib, "oc_step");
    auto p_planes = (decltype(&oc_state_planes))dlsym(lib, "oc_state_planes");
    auto p_err = (decltype(&oc_last_error))dlsym(lib, "oc_last_error");
    OcLayout lay;
    FILE* f = fopen(argv[2], "rb");
    if (!f || fread(&lay, 1, sizeof lay, f) != sizeof lay) { fprintf(stderr, "cannot read %s\n", argv[2]); return 1; }
    fclose(f);
    const int64_t n = argc > 3 ? atoll(argv[3]) : 65536;
    const int iters = argc > 4 ? atoi(argv[4]) : 20000;
    OcBatch b = {};
    b.n_envs = n; b.n_layouts = 1; b.width = lay.width; b.height = lay.height;
    if (p_hints(&lay, 1, &b)) { fprintf(stderr, "hints: %s\n", p_err()); return 1; }
    const int planes = p_planes(lay.width, lay.height);
    OcLayout* d_lay; void* d_state; uint8_t* d_act; float *d_rew, *d_ep; uint8_t* d_fl;
    CK(hipMalloc(&d_lay, sizeof lay)); CK(hipMemcpy(d_lay, &lay, sizeof lay, hipMemcpyHostToDevice));
    CK(hipMalloc(&d_state, (size_t)planes * n * 16)); CK(hipMalloc(&d_act, 16 * n * 2));
    CK(hipMalloc(&d_rew, n * 16)); CK(hipMalloc(&d_ep, n * 16)); CK(hipMalloc(&d_fl, n));
    CK(hipMemset(d_ep, 0, n * 16));
    b.d_layouts = d_lay;
    std::vector<uint8_t> a(16 * n * 2);
    srand(1);
    for (auto& x : a) x = (uint8_t)(rand() % 6);
    CK(hipMemcpy(d_act, a.data(), a.size(), hipMemcpyHostToDevice));
    if (p_reset(&b, d_state, nullptr, d_ep, nullptr)) { fprintf(stderr, "reset: %s\n", p_err()); return 1; }
    auto call = [&](int i) { return p_step(&b, d_state, d_state, d_act + (size_t)(i & 15) * n * 2, d_rew, d_fl, d_ep, nullptr, 400, OC_OPT_AUTO_RESET, nullptr, nullptr, nullptr); };
    for (int i = 0; i < 500; ++i) if (call(i)) { fprintf(stderr, "step: %s\n", p_err()); return 1; }
    CK(hipDeviceSynchronize());
    for (int rep = 0; rep < 3; ++rep) {
        auto t0 = std::chrono::steady_clock::now();
        for (int i = 0; i < iters; ++i) call(i);
        auto t1 = std::chrono::steady_clock::now();
        CK(hipDeviceSynchronize());
        auto t2 = std::chrono::steady_clock::now();
        printf("queue full: %.2f us per call to enqueue, %.2f us per call to finish (%lld envs)\n",
               std::chrono::duration<double, std::micro>(t1 - t0).count() / iters,
               std::chrono::duration<double, std::micro>(t2 - t0).count() / iters, (long long)n);
    }
    double host = 0;
    for (int i = 0; i < 2000; ++i) {
        auto t0 = std::chrono::steady_clock::now();
        call(i);
        auto t1 = std::chrono::steady_clock::now();
        CK(hipDeviceSynchronize());
        host += std::chrono::duration<double, std::micro>(t1 - t0).count();
    }
    printf("queue empty: %.2f us of host time per call\n", host / 2000);
    return 0;
}
