// Microbenchmark: what does a DEPENDENT chain of 16-bit MFMAs cost?  NACC accumulators are used round-robin, so an
// MFMA's SrcC is the result of the MFMA issued NACC slots earlier (NACC = 1: back to back on one accumulator).
//   hipcc -O3 --offload-arch=gfx950 scripts/ubench/mfma_chain.hip -o /tmp/mfma_chain && /tmp/mfma_chain
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));

template <int NACC>
__global__ __launch_bounds__(256, 1) void k(float* out, unsigned long long* ticks, int iters) {
    f32x16 acc[NACC];
    for (int t = 0; t < NACC; ++t) for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;
    const float a = threadIdx.x * 1e-3f, b = 1.0f + threadIdx.x * 1e-4f;
    f16x8 ha, hb;
    for (int i = 0; i < 8; ++i) { ha[i] = (_Float16)(a + i); hb[i] = (_Float16)(b - i); }
    const unsigned long long t0 = __builtin_readcyclecounter();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int g = 0; g < 24; ++g) acc[g % NACC] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ha, hb, acc[g % NACC], 0, 0, 0);
    }
    const unsigned long long t1 = __builtin_readcyclecounter();
    float s = 0;
    for (int t = 0; t < NACC; ++t) for (int r = 0; r < 16; ++r) s += acc[t][r];
    out[blockIdx.x * 256 + threadIdx.x] = s;
    if (threadIdx.x == 0) ticks[blockIdx.x] = t1 - t0;
}

template <int NACC>
void run(float* out, unsigned long long* ticks, int iters) {
    hipLaunchKernelGGL((k<NACC>), dim3(256), dim3(256), 0, 0, out, ticks, 200);
    hipDeviceSynchronize();
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipEventRecord(e0);
    hipLaunchKernelGGL((k<NACC>), dim3(256), dim3(256), 0, 0, out, ticks, iters);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    const double n = 24.0 * iters;
    printf("dependent distance %d: %8.3f ms  %.1f ns/MFMA  chip %.1f TFLOP/s\n", NACC, ms, ms * 1e6 / n,
           2.0 * 32 * 32 * 16 * n * 1024 / (ms * 1e-3) / 1e12);
}
int main() {
    float* out; unsigned long long* ticks;
    hipMalloc(&out, 256 * 256 * 4); hipMalloc(&ticks, 256 * 8);
    run<8>(out, ticks, 20000);
    run<1>(out, ticks, 20000); run<2>(out, ticks, 20000); run<3>(out, ticks, 20000); run<4>(out, ticks, 20000);
    run<6>(out, ticks, 20000); run<8>(out, ticks, 20000);
    return 0;
}
