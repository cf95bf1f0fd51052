// Shared device/host helpers for the gfx950 NeRF-SOS kernels.  gfx950 only: wave = 64 lanes.
#pragma once
#include <hip/hip_runtime.h>
#include <cstdlib>
#include <stdint.h>

#include "nerf_sos_hip.h"

#define NSOS_WAVE 64

#define NSOS_REQUIRE(cond, code) \
    do {                         \
        if (!(cond)) return (code); \
    } while (0)

// Per-DEVICE one-time state.  hipFuncSetAttribute configures the current device's copy of a kernel, and CU counts are a
// property of the device: function-static flags would be per process, and a process that drives a second GPU would skip
// the attribute there (launches needing > 64 KiB of LDS then fail).  The usual deployment is one process per GPU; this keeps
// the library correct when it is not.
#define NSOS_MAX_DEVICES 64
static inline int nsos_current_device() {
    int d = 0;
    if (hipGetDevice(&d) != hipSuccess || d < 0 || d >= NSOS_MAX_DEVICES) d = 0;
    return d;
}
static inline int nsos_device_cus() {
    static int cus[NSOS_MAX_DEVICES];
    const int d = nsos_current_device();
    if (!cus[d]) {
        int n = 0;
        cus[d] = (hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, d) == hipSuccess && n > 0) ? n : 256;
    }
    return cus[d];
}
// diagnostic switches ("NSOS_..." = anything but empty / 0); read at call time, so a test can flip one inside a process
inline bool nsos_env_flag(const char* name) {
    const char* e = getenv(name);
    return e && e[0] && !(e[0] == '0' && !e[1]);
}

struct NsosPerDeviceFlag {   // zero-initialised static: "has this device been configured for this kernel instantiation"
    bool done[NSOS_MAX_DEVICES];
    bool& here() { return done[nsos_current_device()]; }
};

static inline int32_t nsos_launch_status() {
    hipError_t e = hipGetLastError();
    return e == hipSuccess ? NSOS_OK : (int32_t)e;
}

// torch.linspace(0, 1, n) fp32 exactly as ATen's CPU kernel evaluates it (symmetric fma form):
// step = 1/(n-1);  i < n/2 : fma(step, i, 0)  else  fma(-step, n-1-i, 1).
__device__ __forceinline__ float nsos_linspace01(int i, int n) {
    if (n == 1) return 0.0f;  // torch.linspace(0, 1, 1) == [0]
    const float step = 1.0f / (float)(n - 1);
    return (i < n / 2) ? __fmaf_rn(step, (float)i, 0.0f) : __fmaf_rn(-step, (float)(n - 1 - i), 1.0f);
}

// wave-wide sum of a double (all lanes receive the result)
__device__ __forceinline__ double nsos_wave_sum(double v) {
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) v += __shfl_xor(v, off, NSOS_WAVE);
    return v;
}
