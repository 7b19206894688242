// host_util.hpp — host-side launch helpers; included by every unit inside its anonymous namespace (after common.hpp).
#pragma once

int fail(int code, const char* msg) {
    snprintf(g_err, sizeof(g_err), "%s", msg);
    return code;
}

int check_launch(const char* what) {
    if (g_lds_refused) {  // want_lds already wrote the message; nothing was launched
        g_lds_refused = false;
        return OC_ELAUNCH;
    }
    hipError_t err = hipGetLastError();
    if (err != hipSuccess) {
        snprintf(g_err, sizeof(g_err), "%s: %s", what, hipGetErrorString(err));
        return OC_ELAUNCH;
    }
    return OC_OK;
}

// Raise a kernel's dynamic-LDS limit when a launch needs more than the default; a request above what the device
// offers is reported here (not as a later launch failure).
template <typename K>
bool want_lds(K kernel, size_t bytes, size_t dflt = 40 * 1024) {
    if (bytes <= dflt) return true;
    const hipError_t err = hipFuncSetAttribute((const void*)kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes);
    if (err == hipSuccess) return true;
    snprintf(g_err, sizeof(g_err), "dynamic LDS request of %zu bytes refused: %s", bytes, hipGetErrorString(err));
    (void)hipGetLastError();
    g_lds_refused = true;
    return false;
}

inline unsigned grid_for(int64_t n) { return (unsigned)((n + BLOCK - 1) / BLOCK); }

// SIMDs of the current device (4 per CU); cached per thread
inline int64_t simd_count() {
    thread_local int cached_dev = -1;
    thread_local int64_t cached = 1024;
    int dev = 0;
    if (hipGetDevice(&dev) == hipSuccess && dev != cached_dev) {
        int cus = 0;
        if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess && cus > 0) cached = 4 * (int64_t)cus;
        cached_dev = dev;
    }
    return cached;
}
