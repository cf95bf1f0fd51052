// Microbenchmark for the reduced-precision kernel's next design step: can the activation pass (AGPR reads + pack +
// clamp, optionally + AGPR write of the packed word) ride inside a stream of 32-cycle MFMAs, and can MFMA take its B
// operand from the AGPR file?   hipcc -O3 --offload-arch=gfx950 scripts/ubench/mfma_act.hip -o /tmp/mfma_act
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
#define PIN() __builtin_amdgcn_sched_barrier(0)

// MODE 0: pure MFMA (B in VGPR)   1: B in AGPR   2: + per MFMA {2 agpr reads, cvt_pk, pk_max}
// 3: mode 2 + agpr write of the result   4: mode 3 with B in AGPR
template <int MODE>
__global__ __launch_bounds__(256, 1) void k(float* out, unsigned long long* ticks, int iters) {
    f32x16 acc[8], src[4];   // acc: written by the MFMAs; src: "finished" accumulators the activation reads
    for (int t = 0; t < 8; ++t) for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;
    for (int t = 0; t < 4; ++t) for (int r = 0; r < 16; ++r) src[t][r] = threadIdx.x * 0.01f + r;
    u32x4 a = {threadIdx.x * 3u + 0x3c003c00u, 0x3c003c00u, 0x38003800u, 0x3c003800u};
    u32x4 b = {0x3c003c00u, threadIdx.x + 0x38003800u, 0x3c003c00u, 0x34003400u};
    unsigned sink[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    f32x16 park = src[3];
    const unsigned long long t0 = __builtin_readcyclecounter();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int g = 0; g < 16; ++g) {
            if (MODE == 1 || MODE == 4)
                asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, %0" : "+a"(acc[g & 7]) : "v"(a), "a"(b));
            else
                asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, %0" : "+a"(acc[g & 7]) : "v"(a), "v"(b));
            if (MODE >= 2) {
                unsigned r, t;
                asm volatile("v_accvgpr_read_b32 %0, %2\n\tv_accvgpr_read_b32 %1, %3\n\tv_cvt_pk_f16_f32 %0, %0, %1\n\tv_pk_max_i16 %0, %0, 0"
                             : "=&v"(r), "=&v"(t) : "a"(src[(g >> 3) & 3][(2 * g) & 15]), "a"(src[(g >> 3) & 3][(2 * g + 1) & 15]));
                if (MODE >= 3) asm volatile("v_accvgpr_write_b32 %0, %1" : "=a"(park[g & 15]) : "v"(r));
                else sink[g & 7] ^= r;
            }
        }
    }
    const unsigned long long t1 = __builtin_readcyclecounter();
    float s = 0;
    for (int t = 0; t < 8; ++t) for (int r = 0; r < 16; ++r) s += acc[t][r];
    for (int r = 0; r < 16; ++r) s += park[r];
    for (int i = 0; i < 8; ++i) s += (float)sink[i];
    out[blockIdx.x * 256 + threadIdx.x] = s;
    if (threadIdx.x == 0) ticks[blockIdx.x] = t1 - t0;
}

template <int MODE>
void run(const char* name, float* out, unsigned long long* ticks) {
    const int iters = 20000;
    hipLaunchKernelGGL((k<MODE>), dim3(256), dim3(256), 0, 0, out, ticks, 200);
    (void)hipDeviceSynchronize();
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    (void)hipEventRecord(e0);
    hipLaunchKernelGGL((k<MODE>), dim3(256), dim3(256), 0, 0, out, ticks, iters);
    (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
    float ms; (void)hipEventElapsedTime(&ms, e0, e1);
    unsigned long long h[256]; (void)hipMemcpy(h, ticks, sizeof h, hipMemcpyDeviceToHost);
    double avg = 0; for (int i = 0; i < 256; ++i) avg += h[i]; avg /= 256;
    const double n = 16.0 * iters;
    printf("%-58s ticks/MFMA %6.2f   chip %.0f TFLOP/s\n", name, avg / n, 2.0 * 32 * 32 * 16 * n * 1024 / (ms * 1e-3) / 1e12);
}
int main() {
    float* out; unsigned long long* ticks;
    (void)hipMalloc(&out, 256 * 256 * 4); (void)hipMalloc(&ticks, 256 * 8);
    run<0>("pure MFMA f16 32x32x16, B in VGPR", out, ticks);
    run<1>("B operand in AGPR", out, ticks);
    run<2>("+ per MFMA: 2 accvgpr_read, cvt_pk, pk_max", out, ticks);
    run<3>("+ per MFMA: 2 reads, cvt_pk, pk_max, accvgpr_write", out, ticks);
    run<4>("same, B operand in AGPR", out, ticks);
    return 0;
}
