// Microbenchmark: how many cycles does a stream of v_mfma_f32_32x32x2_f32 take per MFMA (one wave per SIMD)
// with various fillers between the MFMAs?  Prints ticks (s_memtime) per MFMA for each variant.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
#define PIN() __builtin_amdgcn_sched_barrier(0)

template <int VARIANT>
__global__ __launch_bounds__(256, 1) void k(float* out, unsigned long long* ticks, int iters) {
    __shared__ float lds[8192];
    for (int i = threadIdx.x; i < 8192; i += 256) lds[i] = i * 1e-6f;
    __syncthreads();
    f32x16 acc[8];
    for (int t = 0; t < 8; ++t) for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;
    float a = threadIdx.x * 1e-3f, b = 1.0f + threadIdx.x * 1e-4f;
    float v[16];
    for (int i = 0; i < 16; ++i) v[i] = a + i;
    unsigned addr = (unsigned)(size_t)(__attribute__((address_space(3))) float*)lds + (threadIdx.x & 63) * 16;
    f32x4 r4 = {a, a, a, a};
    unsigned long long t0 = __builtin_readcyclecounter();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int g = 0; g < 32; ++g) {
            if (VARIANT == 3 || VARIANT == 4) {
                asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(r4) : "v"(addr), "i"((g & 7) * 1024) : "memory");
                if (VARIANT == 4) asm volatile("s_waitcnt lgkmcnt(1)" ::: "memory");
            }
            PIN();
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                acc[(g & 1) * 4 + j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[(g & 1) * 4 + j], 0, 0, 0);
                if (VARIANT == 1) { PIN(); asm volatile("v_max_f32 %0, 0, %0" : "+v"(v[(g * 4 + j) & 15])); PIN(); }
                if (VARIANT == 2) { PIN(); for (int q = 0; q < 4; ++q) asm volatile("v_max_f32 %0, 0, %0" : "+v"(v[(g * 4 + j + q) & 15])); PIN(); }
                if (VARIANT == 5) { PIN(); asm volatile("s_add_u32 s20, s20, 1\n s_add_u32 s20, s20, 1\n s_add_u32 s20, s20, 1\n s_add_u32 s20, s20, 1\n s_add_u32 s20, s20, 1\n s_add_u32 s20, s20, 1\n s_add_u32 s20, s20, 1\n s_add_u32 s20, s20, 1" ::: "s20"); PIN(); }
                if (VARIANT == 6) { PIN(); for (int q = 0; q < 8; ++q) asm volatile("v_max_f32 %0, 0, %0" : "+v"(v[(g * 4 + j + q) & 15])); PIN(); }
            }
            PIN();
        }
    }
    unsigned long long t1 = __builtin_readcyclecounter();
    float s = r4[0];
    for (int t = 0; t < 8; ++t) for (int r = 0; r < 16; ++r) s += acc[t][r];
    for (int i = 0; i < 16; ++i) s += v[i];
    out[blockIdx.x * 256 + threadIdx.x] = s;
    if (threadIdx.x == 0) ticks[blockIdx.x] = t1 - t0;
}

template <int V>
void run(const char* name, float* out, unsigned long long* ticks) {
    const int iters = 200;
    hipLaunchKernelGGL(k<V>, dim3(256), dim3(256), 0, 0, out, ticks, iters);
    hipDeviceSynchronize();
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipEventRecord(e0);
    hipLaunchKernelGGL(k<V>, dim3(256), dim3(256), 0, 0, out, ticks, iters);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    unsigned long long h[256]; hipMemcpy(h, ticks, sizeof h, hipMemcpyDeviceToHost);
    double avg = 0; for (int i = 0; i < 256; ++i) avg += h[i]; avg /= 256;
    double n = 128.0 * iters;
    printf("%-34s ticks/MFMA %.2f   wall ns/MFMA %.2f  => tick clock %.3f GHz, TF %.1f\n", name, avg / n, ms * 1e6 / n,
           avg / (ms * 1e6), 1024.0 * n * 4096 / (ms * 1e-3) / 1e12);
}
int main() {
    float* out; unsigned long long* ticks;
    hipMalloc(&out, 256 * 256 * 4); hipMalloc(&ticks, 256 * 8);
    run<0>("pure MFMA", out, ticks);
    run<1>("+1 v_max per MFMA", out, ticks);
    run<2>("+4 v_max per MFMA", out, ticks);
    run<6>("+8 v_max per MFMA", out, ticks);
    run<5>("+8 s_add per MFMA", out, ticks);
    run<3>("+ds_read_b128 per 4 (no wait)", out, ticks);
    run<4>("+ds_read_b128 + lgkmcnt(1) per 4", out, ticks);
    return 0;
}
