// Microbenchmark: sustained MFMA rate of the whole chip (256 WGs x 4 waves, one wave per SIMD) for the three
// instructions the MLP kernels use, and what one VALU instruction between MFMAs costs in each case.
//   hipcc -O3 --offload-arch=gfx950 scripts/ubench/mfma_peak.hip -o /tmp/mfma_peak && /tmp/mfma_peak
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
#define PIN() __builtin_amdgcn_sched_barrier(0)

// KIND 0: f32 32x32x2 (64 cycles), 1: f16 32x32x16 (32 cycles), 2: bf16 32x32x16.  FILL = VALU ops per MFMA.
template <int KIND, int FILL>
__global__ __launch_bounds__(256, 1) void k(float* out, unsigned long long* ticks, int iters) {
    f32x16 acc[16];
    for (int t = 0; t < 16; ++t) for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;
    const float a = threadIdx.x * 1e-3f, b = 1.0f + threadIdx.x * 1e-4f;
    f16x8 ha, hb; bf16x8 ba, bb;
    for (int i = 0; i < 8; ++i) { ha[i] = (_Float16)(a + i); hb[i] = (_Float16)(b - i); ba[i] = (__bf16)(a + i); bb[i] = (__bf16)(b - i); }
    float v[8];
    for (int i = 0; i < 8; ++i) v[i] = a + i;
    const unsigned long long t0 = __builtin_readcyclecounter();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int g = 0; g < 16; ++g) {
            if (KIND == 0) acc[g] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[g], 0, 0, 0);
            if (KIND == 1) acc[g] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ha, hb, acc[g], 0, 0, 0);
            if (KIND == 2) acc[g] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ba, bb, acc[g], 0, 0, 0);
            if (FILL) {
                PIN();
                for (int q = 0; q < FILL; ++q) asm volatile("v_max_f32 %0, 0, %0" : "+v"(v[(g + q) & 7]));
                PIN();
            }
        }
    }
    const unsigned long long t1 = __builtin_readcyclecounter();
    float s = 0;
    for (int t = 0; t < 16; ++t) for (int r = 0; r < 16; ++r) s += acc[t][r];
    for (int i = 0; i < 8; ++i) s += v[i];
    out[blockIdx.x * 256 + threadIdx.x] = s;
    if (threadIdx.x == 0) ticks[blockIdx.x] = t1 - t0;
}

template <int KIND, int FILL>
void run(const char* name, float* out, unsigned long long* ticks, int iters) {
    hipLaunchKernelGGL((k<KIND, FILL>), dim3(256), dim3(256), 0, 0, out, ticks, 200);
    hipDeviceSynchronize();
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipEventRecord(e0);
    hipLaunchKernelGGL((k<KIND, FILL>), dim3(256), dim3(256), 0, 0, out, ticks, iters);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    unsigned long long h[256]; hipMemcpy(h, ticks, sizeof h, hipMemcpyDeviceToHost);
    double avg = 0; for (int i = 0; i < 256; ++i) avg += h[i]; avg /= 256;
    const double n = 16.0 * iters, flop = KIND == 0 ? 2.0 * 32 * 32 * 2 : 2.0 * 32 * 32 * 16;
    printf("%-26s iters %7d  %8.3f ms  ticks/MFMA %6.2f  tick clock %.3f GHz  chip %.1f TFLOP/s\n", name, iters, ms,
           avg / n, avg / (ms * 1e6), flop * n * 1024 / (ms * 1e-3) / 1e12);
}
int main() {
    float* out; unsigned long long* ticks;
    hipMalloc(&out, 256 * 256 * 4); hipMalloc(&ticks, 256 * 8);
    for (int iters : {2000, 200000}) {
        run<0, 0>("f32 32x32x2", out, ticks, iters);
        run<1, 0>("f16 32x32x16", out, ticks, iters);
        run<2, 0>("bf16 32x32x16", out, ticks, iters);
    }
    run<1, 1>("f16 + 1 VALU / MFMA", out, ticks, 20000);
    run<1, 2>("f16 + 2 VALU / MFMA", out, ticks, 20000);
    run<1, 4>("f16 + 4 VALU / MFMA", out, ticks, 20000);
    run<1, 8>("f16 + 8 VALU / MFMA", out, ticks, 20000);
    return 0;
}
