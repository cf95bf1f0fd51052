// What does ds_read_b64_tr_b16 return?  LDS holds the 16-bit value i at element i.  Each lane passes a byte address; the
// kernel prints, per address pattern, the four 16-bit elements every lane received.  Patterns:
//   0: addr = 8 * lane                       (every lane its own 8 contiguous bytes)
//   1: addr = 2*((l>>4)*64 + ((l&15)>>2)*16 + (l&3)*4)   (16-lane group = a [4 rows][16 cols] block, row stride 16 elements:
//      lane (r = (l&15)>>2, q = l&3) points at row r, columns 4q..4q+3)
//   2: same geometry with a row stride of 324 elements (648 B: a padded 320-channel row), rows = points 8*(l>>5) + r,
//      columns 16*((l>>4)&1) + 4q..: the sem_in operand fetch of a weight-gradient kernel
//   hipcc -O3 --offload-arch=gfx950 scripts/ubench/tr_read.hip -o /tmp/tr_read && /tmp/tr_read
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef unsigned u32x2 __attribute__((ext_vector_type(2)));

__global__ void k(int pattern, unsigned* out) {
    __shared__ unsigned short lds[16384];
    for (int i = threadIdx.x; i < 16384; i += 64) lds[i] = (unsigned short)i;
    __syncthreads();
    const int l = threadIdx.x;
    unsigned addr;
    if (pattern == 0) addr = 8u * l;
    else if (pattern == 1) addr = 2u * ((l >> 4) * 64 + ((l & 15) >> 2) * 16 + (l & 3) * 4);
    else addr = 2u * ((8 * (l >> 5) + ((l & 15) >> 2)) * 324 + 16 * ((l >> 4) & 1) + (l & 3) * 4);
    const unsigned base = (unsigned)(size_t)(__attribute__((address_space(3))) unsigned short*)lds;
    u32x2 v;
    asm volatile("ds_read_b64_tr_b16 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(v) : "v"(base + addr) : "memory");
    out[2 * l] = v[0];
    out[2 * l + 1] = v[1];
}

int main() {
    unsigned* d;
    hipMalloc(&d, 128 * sizeof(unsigned));
    std::vector<unsigned> h(128);
    for (int p = 0; p < 3; ++p) {
        hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, p, d);
        hipMemcpy(h.data(), d, 128 * sizeof(unsigned), hipMemcpyDeviceToHost);
        printf("pattern %d (element indices each lane received: e0 e1 e2 e3)\n", p);
        for (int l = 0; l < 64; ++l) {
            printf("  lane %2d: %5u %5u %5u %5u%s", l, h[2 * l] & 0xffff, h[2 * l] >> 16, h[2 * l + 1] & 0xffff, h[2 * l + 1] >> 16, (l & 1) ? "\n" : "   |");
        }
    }
    return 0;
}
