// Shared pieces of the split-fp16 ("x3") kernels: the forward network (mlp_x3.hip) and the input-gradient chain
// (mlp_x3_bwd.hip).  A value v is carried as hi = fp16(v), lo = fp16(v - hi); products run as three 16-bit MFMAs.
#pragma once
#include "mlp_common.h"

using namespace nsos;

typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));

namespace {

constexpr int kSlotGroups = 36;                          // LDS slot / stream stride per chunk, in 1 KiB A operands
constexpr int kSlotBytes = kSlotGroups * 1024;
constexpr int kSlots = 4;
constexpr int kDmaPieces = kSlotGroups / 4;              // 1 KiB pieces per wave per chunk
constexpr int kTilePts = 128;                            // 4 waves x 32 points
constexpr int kRing = 5, kPre = 8, kMid = 2;

constexpr unsigned kOnes = 0x3C003C00u;  // {1.0h, 1.0h}
constexpr float kLoScale = 2048.0f, kLoUnscale = 1.0f / 2048.0f;   // the weights' lo parts are stored x 2^11 (0x3a000000 = 2^-11)
__device__ __forceinline__ f32x16 mfma16(u32x4 a, u32x4 b, f32x16 c) {
    return __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, a), __builtin_bit_cast(f16x8, b), c, 0, 0, 0);
}
__host__ __device__ inline unsigned short f16_bits(float x) {   // round to nearest even
    const _Float16 h = (_Float16)x;
    return __builtin_bit_cast(unsigned short, h);
}
__host__ __device__ inline float f16_value(unsigned short b) { return (float)__builtin_bit_cast(_Float16, b); }

// (hi, lo) packed words of the fp32 pair (v0, v1):  hi = fp16(v), lo = fp16(v - hi)
__device__ __forceinline__ void split2(float v0, float v1, unsigned& hi, unsigned& lo) {
    float t0 = v0, t1 = v1;
    asm volatile("v_cvt_pk_f16_f32 %0, %2, %3\n\t"
                 "v_fma_mix_f32 %2, %0, -1.0, %2 op_sel_hi:[1,0,0]\n\t"
                 "v_fma_mix_f32 %3, %0, -1.0, %3 op_sel:[1,0,0] op_sel_hi:[1,0,0]\n\t"
                 "v_cvt_pk_f16_f32 %1, %2, %3"
                 : "=&v"(hi), "=&v"(lo), "+v"(t0), "+v"(t1));
}
// the same for the sum of two accumulator elements each (main + cross), straight from the AGPR file, optional ReLU
template <bool RELU>
__device__ __forceinline__ void split2_acc(const float& m0, const float& x0, const float& m1, const float& x1, unsigned& hi,
                                           unsigned& lo) {
    unsigned t0, t1, t2;
    if constexpr (RELU)
        asm volatile("v_accvgpr_read_b32 %2, %5\n\tv_accvgpr_read_b32 %4, %6\n\tv_accvgpr_read_b32 %3, %7\n\t"
                     "v_fmac_f32 %2, 0x3a000000, %4\n\tv_accvgpr_read_b32 %4, %8\n\tv_max_f32 %2, 0, %2\n\tv_fmac_f32 %3, 0x3a000000, %4\n\t"
                     "v_max_f32 %3, 0, %3\n\tv_cvt_pk_f16_f32 %0, %2, %3\n\t"
                     "v_fma_mix_f32 %2, %0, -1.0, %2 op_sel_hi:[1,0,0]\n\t"
                     "v_fma_mix_f32 %3, %0, -1.0, %3 op_sel:[1,0,0] op_sel_hi:[1,0,0]\n\t"
                     "v_cvt_pk_f16_f32 %1, %2, %3"
                     : "=&v"(hi), "=&v"(lo), "=&v"(t0), "=&v"(t1), "=&v"(t2) : "a"(m0), "a"(x0), "a"(m1), "a"(x1));
    else
        asm volatile("v_accvgpr_read_b32 %2, %5\n\tv_accvgpr_read_b32 %4, %6\n\tv_accvgpr_read_b32 %3, %7\n\t"
                     "v_fmac_f32 %2, 0x3a000000, %4\n\tv_accvgpr_read_b32 %4, %8\n\tv_fmac_f32 %3, 0x3a000000, %4\n\t"
                     "v_cvt_pk_f16_f32 %0, %2, %3\n\t"
                     "v_fma_mix_f32 %2, %0, -1.0, %2 op_sel_hi:[1,0,0]\n\t"
                     "v_fma_mix_f32 %3, %0, -1.0, %3 op_sel:[1,0,0] op_sel_hi:[1,0,0]\n\t"
                     "v_cvt_pk_f16_f32 %1, %2, %3"
                     : "=&v"(hi), "=&v"(lo), "=&v"(t0), "=&v"(t1), "=&v"(t2) : "a"(m0), "a"(x0), "a"(m1), "a"(x1));
}
// ReLU variant that also shifts the two "z > 0" bits into a per-lane bit mask (first element first): v_sub_co 0 - bits(z) borrows
// exactly when the clamped z is not +0, v_addc M + M + borrow appends the bit.  The masks replace fp32 reads in the backward.
__device__ __forceinline__ void split2_acc_bits(const float& m0, const float& x0, const float& m1, const float& x1, unsigned& hi,
                                                unsigned& lo, unsigned& mask) {
    unsigned t0, t1, t2;
    asm volatile("v_accvgpr_read_b32 %2, %6\n\tv_accvgpr_read_b32 %4, %7\n\tv_accvgpr_read_b32 %3, %8\n\t"
                 "v_fmac_f32 %2, 0x3a000000, %4\n\tv_accvgpr_read_b32 %4, %9\n\tv_max_f32 %2, 0, %2\n\tv_fmac_f32 %3, 0x3a000000, %4\n\t"
                 "v_max_f32 %3, 0, %3\n\t"
                 "v_sub_co_u32 %4, vcc, 0, %2\n\tv_addc_co_u32 %5, vcc, %5, %5, vcc\n\t"
                 "v_sub_co_u32 %4, vcc, 0, %3\n\tv_addc_co_u32 %5, vcc, %5, %5, vcc\n\t"
                 "v_cvt_pk_f16_f32 %0, %2, %3\n\t"
                 "v_fma_mix_f32 %2, %0, -1.0, %2 op_sel_hi:[1,0,0]\n\t"
                 "v_fma_mix_f32 %3, %0, -1.0, %3 op_sel:[1,0,0] op_sel_hi:[1,0,0]\n\t"
                 "v_cvt_pk_f16_f32 %1, %2, %3"
                 : "=&v"(hi), "=&v"(lo), "=&v"(t0), "=&v"(t1), "=&v"(t2), "+v"(mask) : "a"(m0), "a"(x0), "a"(m1), "a"(x1) : "vcc");
}
__device__ __forceinline__ float dot2(unsigned a, unsigned b, float acc) {
    asm volatile("v_dot2c_f32_f16 %0, %1, %2" : "+v"(acc) : "v"(a), "v"(b));
    return acc;
}

// fp32 value hi + lo of one half (SEL = 0: low, 1: high 16 bits) of a split pair of packed words
template <int SEL>
__device__ __forceinline__ float join(unsigned hw, unsigned lw) {
    float r;
    if constexpr (SEL == 0) asm volatile("v_fma_mix_f32 %0, %1, 1.0, %2 op_sel_hi:[1,0,1]" : "=v"(r) : "v"(hw), "v"(lw));
    else asm volatile("v_fma_mix_f32 %0, %1, 1.0, %2 op_sel:[1,0,1] op_sel_hi:[1,0,1]" : "=v"(r) : "v"(hw), "v"(lw));
    return r;
}

}  // namespace
