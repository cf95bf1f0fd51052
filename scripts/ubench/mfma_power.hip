// Microbenchmark: what the 16-bit matrix pipe sustains on REAL (random, non-constant) operands.
// The nominal dense peak (2.5 PFLOP/s) assumes 2.4 GHz; the chip clocks to its power budget, and the power of an MFMA
// depends on how many operand bits toggle.  Pure MFMA streams, one wave per SIMD (256 WGs x 4 waves) or two (x 8 waves),
// with operands that are (a) lane-constant small numbers (the friendliest case), (b) N(0,1) random values, different per
// lane and rotating through 8 register sets, (c) random values of which half are exactly zero (what a ReLU network
// feeds the pipe).  Reports ticks/MFMA (issue efficiency), the effective shader clock and chip TFLOP/s.
//   hipcc -O3 --offload-arch=gfx950 scripts/ubench/mfma_power.hip -o /tmp/mfma_power && /tmp/mfma_power
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cmath>
#include <vector>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

// KIND 1: f16 32x32x16, 2: bf16 32x32x16
template <int KIND, int NW>
__global__ __launch_bounds__(64 * NW, 1) void k(const u32x4* __restrict__ ops, float* out, unsigned long long* ticks, int iters) {
    constexpr int NACC = 6;   // dependent MFMAs 6 issue slots apart (free from 4); few registers, so 1 or 2 waves per SIMD both fit
    f32x16 acc[NACC];
    for (int t = 0; t < NACC; ++t) for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;
    u32x4 a[8], b[8];
    for (int i = 0; i < 8; ++i) {   // 8 A and 8 B operand sets per lane, all different
        a[i] = ops[(size_t)(i * 2 + 0) * 4096 + blockIdx.x % 8 * 512 + threadIdx.x % 512];
        b[i] = ops[(size_t)(i * 2 + 1) * 4096 + blockIdx.x % 8 * 512 + threadIdx.x % 512];
    }
    const unsigned long long t0 = __builtin_readcyclecounter();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int g = 0; g < 48; ++g) {
            const int t = g % NACC;
            if (KIND == 1) acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, a[g % 8]), __builtin_bit_cast(f16x8, b[(g / 8 + g) % 8]), acc[t], 0, 0, 0);
            if (KIND == 2) acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a[g % 8]), __builtin_bit_cast(bf16x8, b[(g / 8 + g) % 8]), acc[t], 0, 0, 0);
        }
        if ((it & 63) == 63)   // keep the accumulators finite and O(1): halve them now and then (16 VALU per 3072 MFMAs)
            for (int t = 0; t < NACC; ++t) acc[t] *= 0.25f;
    }
    const unsigned long long t1 = __builtin_readcyclecounter();
    float s = 0;
    for (int t = 0; t < NACC; ++t) for (int r = 0; r < 16; ++r) s += acc[t][r];
    out[blockIdx.x * 64 * NW + threadIdx.x] = s;
    if (threadIdx.x == 0) ticks[blockIdx.x] = t1 - t0;
}

static unsigned short f2h(float x) { _Float16 h = (_Float16)x; unsigned short u; __builtin_memcpy(&u, &h, 2); return u; }
static unsigned short f2b(float x) { unsigned u; __builtin_memcpy(&u, &x, 4); u += 0x7fffu + ((u >> 16) & 1u); return (unsigned short)(u >> 16); }

template <int KIND, int NW>
void run(const char* name, const u32x4* ops, float* out, unsigned long long* ticks, int iters) {
    hipLaunchKernelGGL((k<KIND, NW>), dim3(256), dim3(64 * NW), 0, 0, ops, out, ticks, 100);
    hipDeviceSynchronize();
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipEventRecord(e0);
    hipLaunchKernelGGL((k<KIND, NW>), dim3(256), dim3(64 * NW), 0, 0, ops, out, ticks, iters);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    unsigned long long h[256]; hipMemcpy(h, ticks, sizeof h, hipMemcpyDeviceToHost);
    double avg = 0; for (int i = 0; i < 256; ++i) avg += h[i]; avg /= 256;
    const double n = 48.0 * iters;
    printf("%-44s %2d waves/CU iters %6d %8.3f ms  ticks/MFMA/wave %6.2f  clock %.3f GHz  chip %7.1f TFLOP/s\n", name, NW, iters, ms,
           avg / n, avg / (ms * 1e6), 2.0 * 32 * 32 * 16 * n * 256 * NW / (ms * 1e-3) / 1e12);
}

int main() {
    float* out; unsigned long long* ticks; u32x4* ops;
    hipMalloc(&out, 256 * 512 * 4); hipMalloc(&ticks, 256 * 8); hipMalloc(&ops, 16 * 4096 * 16);
    std::vector<unsigned short> h(16 * 4096 * 8);
    for (int kind = 1; kind <= 2; ++kind)
        for (int mode = 0; mode < 3; ++mode) {
            srand(1);
            for (size_t i = 0; i < h.size(); ++i) {
                float u1 = (rand() + 1.0f) / (RAND_MAX + 2.0f), u2 = rand() / (float)RAND_MAX;
                float v = sqrtf(-2.f * logf(u1)) * cosf(6.2831853f * u2);
                if (mode == 0) v = 0.5f + (i % 8) * 0.125f;            // the same small constants in every lane
                if (mode == 2 && (rand() & 1)) v = 0.0f;               // ReLU-like: half the values exactly zero
                h[i] = kind == 1 ? f2h(v) : f2b(v);
            }
            hipMemcpy(ops, h.data(), h.size() * 2, hipMemcpyHostToDevice);
            const char* dn = kind == 1 ? "f16" : "bf16";
            const char* mn[3] = {"lane-constant operands", "N(0,1) random operands", "random, half of them zero"};
            char name[96];
            snprintf(name, sizeof name, "%s 32x32x16, %s", dn, mn[mode]);
            for (int iters : {300, 20000}) {
                if (kind == 1) { run<1, 4>(name, ops, out, ticks, iters); run<1, 8>(name, ops, out, ticks, iters / 2); }
                else { run<2, 4>(name, ops, out, ticks, iters); run<2, 8>(name, ops, out, ticks, iters / 2); }
            }
        }
    return 0;
}
