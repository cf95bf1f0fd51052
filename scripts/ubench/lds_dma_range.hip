// Does global_load_lds (LDS-DMA, destination base in M0) reach every byte of the 160 KiB LDS?
// DMA 1 KiB of a known pattern to LDS offset `off`, read it back with ds_read, compare; also report whether the
// data showed up at (off mod 64 KiB) or (off mod 128 KiB) instead (address wrap).
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void k(const unsigned* src, unsigned* out, unsigned off) {
    extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
    unsigned* l32 = reinterpret_cast<unsigned*>(lds);
    for (unsigned i = threadIdx.x; i < 160 * 256; i += 64) l32[i] = 0xdeadbeefu;
    __syncthreads();
    const unsigned base = (unsigned)(size_t)(__attribute__((address_space(3))) unsigned char*)lds + off;
    unsigned keep, voff = threadIdx.x * 16;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %2, %3\n\ts_mov_b32 m0, %0\n\ts_waitcnt vmcnt(0)"
                 : "=&s"(keep) : "s"(base), "v"(voff), "s"(src) : "memory");
    __syncthreads();
    int ok = 0, at64 = 0, at128 = 0;
    for (int w = 0; w < 4; ++w) {
        ok += l32[off / 4 + threadIdx.x * 4 + w] == src[threadIdx.x * 4 + w];
        at64 += l32[(off % 65536) / 4 + threadIdx.x * 4 + w] == src[threadIdx.x * 4 + w];
        at128 += l32[(off % 131072) / 4 + threadIdx.x * 4 + w] == src[threadIdx.x * 4 + w];
    }
    out[threadIdx.x * 3 + 0] = ok; out[threadIdx.x * 3 + 1] = at64; out[threadIdx.x * 3 + 2] = at128;
}
int main() {
    unsigned h[256]; for (int i = 0; i < 256; ++i) h[i] = 0x1000u + i;
    unsigned *src, *out; hipMalloc(&src, 1024); hipMalloc(&out, 64 * 12); hipMemcpy(src, h, 1024, hipMemcpyHostToDevice);
    hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    for (unsigned off : {0u, 32768u, 65536u - 1024, 65536u, 98304u, 131072u - 1024, 131072u, 140000u / 1024 * 1024, 147456u - 1024, 163840u - 1024}) {
        hipLaunchKernelGGL(k, dim3(1), dim3(64), 160 * 1024, 0, src, out, off);
        unsigned r[192]; hipMemcpy(r, out, sizeof r, hipMemcpyDeviceToHost);
        int ok = 0, a64 = 0, a128 = 0; for (int i = 0; i < 64; ++i) { ok += r[3 * i]; a64 += r[3 * i + 1]; a128 += r[3 * i + 2]; }
        printf("DMA to LDS offset %6u: landed-at-target %3d/256  at-(off mod 64K) %3d  at-(off mod 128K) %3d   [%s]\n", off, ok, a64, a128, hipGetErrorString(hipGetLastError()));
    }
    return 0;
}
