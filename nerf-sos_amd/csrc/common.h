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

// ---- cross-lane traffic without the LDS crossbar (round 5).  __shfl_* compiles to ds_bpermute_b32: every step of a reduction is
// a round trip through the LDS pipe (two per double), and the one-wave-per-ray kernels are chains of such steps -- at 4096 rays their
// run time IS that latency (composite_importance_kernel: 23 us for work that moves 12 MB).  Within a 16-lane row DPP permutes ride on
// the VALU instruction itself; across rows gfx950 has v_permlane16_swap / v_permlane32_swap (VALU as well).
template <int CTRL>
__device__ __forceinline__ double nsos_dpp_perm(double v) {   // a permutation of the lanes inside every row: all sources valid
    const int lo = __builtin_amdgcn_update_dpp(0, __double2loint(v), CTRL, 0xf, 0xf, true);
    const int hi = __builtin_amdgcn_update_dpp(0, __double2hiint(v), CTRL, 0xf, 0xf, true);
    return __hiloint2double(hi, lo);
}
// `v` of the lane CTRL names, or `id` where that lane does not exist (row start) or the row is masked out
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ double nsos_dpp_or(double v, double id) {
    const int lo = __builtin_amdgcn_update_dpp(__double2loint(id), __double2loint(v), CTRL, ROW_MASK, 0xf, false);
    const int hi = __builtin_amdgcn_update_dpp(__double2hiint(id), __double2hiint(v), CTRL, ROW_MASK, 0xf, false);
    return __hiloint2double(hi, lo);
}

// wave-wide sum of a double (all lanes receive the same result: every lane adds the same pairs, and a + b == b + a)
__device__ __forceinline__ double nsos_wave_sum(double v) {
    v += nsos_dpp_perm<0xB1>(v);    // quad_perm [1,0,3,2]: lane ^ 1
    v += nsos_dpp_perm<0x4E>(v);    // quad_perm [2,3,0,1]: lane ^ 2
    v += nsos_dpp_perm<0x141>(v);   // row_half_mirror: the other quad of the 8
    v += nsos_dpp_perm<0x140>(v);   // row_mirror: the other half of the row
    {   // rows 0|1 and 2|3: v_permlane16_swap x, x -> {r0, r0, r2, r2}, {r1, r1, r3, r3}
        const unsigned lo = (unsigned)__double2loint(v), hi = (unsigned)__double2hiint(v);
        const auto l = __builtin_amdgcn_permlane16_swap(lo, lo, false, false);
        const auto h = __builtin_amdgcn_permlane16_swap(hi, hi, false, false);
        v = __hiloint2double((int)h[0], (int)l[0]) + __hiloint2double((int)h[1], (int)l[1]);
    }
    {   // halves: v_permlane32_swap x, x -> {lo half, lo half}, {hi half, hi half}
        const unsigned lo = (unsigned)__double2loint(v), hi = (unsigned)__double2hiint(v);
        const auto l = __builtin_amdgcn_permlane32_swap(lo, lo, false, false);
        const auto h = __builtin_amdgcn_permlane32_swap(hi, hi, false, false);
        v = __hiloint2double((int)h[0], (int)l[0]) + __hiloint2double((int)h[1], (int)l[1]);
    }
    return v;
}

// inclusive wave scan of a double (sum, or product with MUL): row_shr 1/2/4/8 inside the rows, then row_bcast 15 / 31 carry the
// rows' totals forward -- the sequence LLVM's atomic optimizer emits for gfx9.  Association differs from a Hillis-Steele shuffle
// scan by fp64 rounding only (every use here rounds the result to fp32 once).
template <bool MUL>
__device__ __forceinline__ double nsos_wave_scan_incl(double v) {
    const double id = MUL ? 1.0 : 0.0;
#define NSOS_SCAN_STEP(CTRL, MASK)                                      \
    {                                                                   \
        const double o = nsos_dpp_or<CTRL, MASK>(v, id);                \
        v = MUL ? v * o : v + o;                                        \
    }
    NSOS_SCAN_STEP(0x111, 0xf)   // row_shr:1
    NSOS_SCAN_STEP(0x112, 0xf)   // row_shr:2
    NSOS_SCAN_STEP(0x114, 0xf)   // row_shr:4
    NSOS_SCAN_STEP(0x118, 0xf)   // row_shr:8
    NSOS_SCAN_STEP(0x142, 0xa)   // row_bcast:15 into rows 1 and 3
    NSOS_SCAN_STEP(0x143, 0xc)   // row_bcast:31 into rows 2 and 3
#undef NSOS_SCAN_STEP
    return v;
}
// the value of lane - 1 (`id` in lane 0): wave_shr:1
__device__ __forceinline__ double nsos_wave_shr1(double v, double id) { return nsos_dpp_or<0x138, 0xf>(v, id); }
