// Shared pieces of the 16-bit MLP kernels (mlp_lp.hip: 4 waves x 64 points; mlp_lp8.hip: 8 waves x 32 points):
// number formats, the accumulator -> packed-B-operand conversion, parameters, encoding slices, VALU heads.
#pragma once
#include "mlp_common.h"

namespace nsos {
namespace lp {

typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));


constexpr int kSlotGroups = 36;                          // LDS slot / stream stride per chunk, in 1 KiB A operands
constexpr int kSlotBytes = kSlotGroups * 1024;
constexpr int kSlots = 4;
constexpr int kDmaPieces = kSlotGroups / 4;              // 1 KiB pieces per wave per chunk
constexpr int kTilePts = 256;                            // 4 waves x 2 column tiles x 32 points
#ifdef NSOS_LP_RING
constexpr int kRing = NSOS_LP_RING, kPre = NSOS_LP_RING + 3, kMid = 2;
#else
constexpr int kRing = 5, kPre = 8, kMid = 2;
#endif

// aux stream, offsets in 4-byte words
constexpr int kAuxAlphaW = 0;     // 128 words: sigma-head weights packed 16-bit, [kg][slice 0..15][q 0..3]
constexpr int kAuxRgbW = 128;     // 3 x 128 fp32, accumulator layout [hi][t 0..3][r 0..15]
constexpr int kAuxSem2W = 512;    // 2 x 128 fp32
constexpr int kAuxScalars = 768;  // alpha_b, rgb_b[3], sem2_b[2]
constexpr int kAuxWords = 1024;

enum ChunkKind { kHid8 = 0, kEnc8 = 1, kHid4 = 2, kEnc4 = 3, kDir4 = 4,
                 kPair8 = 5 };   // mlp_lp8's tile-pair-major hidden chunk: [bias t0, bias t1, (s0,t0), (s0,t1), (s1,t0), ...], t = 2c + {0,1}

__host__ __device__ constexpr int lp_chunks(int sem) {
    // L0 enc(1) + 8 hidden layers x 4 + L5 enc(1) + [sem0 hid(2) (+enc 1)] + views hid(2) + dir(1)
    return 1 + 32 + 1 + (sem ? 2 + (sem == 2 ? 1 : 0) : 0) + 3;
}

struct F16 {
    static constexpr bool kIsF16 = true;
    __device__ static __forceinline__ float lo(unsigned w) { return (float)__builtin_bit_cast(_Float16, (unsigned short)(w & 0xffffu)); }
    __device__ static __forceinline__ float hi(unsigned w) { return (float)__builtin_bit_cast(_Float16, (unsigned short)(w >> 16)); }
    static constexpr unsigned kOnes = 0x3C003C00u;  // {1.0h, 1.0h}
    __device__ static __forceinline__ f32x16 mfma(u32x4 a, u32x4 b, f32x16 c) {
        return __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, a), __builtin_bit_cast(f16x8, b), c, 0, 0, 0);
    }
    __device__ static __forceinline__ f32x4 mfma_k32(u32x4 a, u32x4 b, f32x4 c) {   // 16x16x32: mlp_lp16.hip
        return __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, a), __builtin_bit_cast(f16x8, b), c, 0, 0, 0);
    }
    __device__ static __forceinline__ unsigned pack2(float lo, float hi) {
        unsigned r;
        asm volatile("v_cvt_pk_f16_f32 %0, %1, %2" : "=v"(r) : "v"(lo), "v"(hi));
        return r;
    }
    __device__ static __forceinline__ float dot2(unsigned a, unsigned b, float acc) {
        asm volatile("v_dot2c_f32_f16 %0, %1, %2" : "+v"(acc) : "v"(a), "v"(b));
        return acc;
    }
    __device__ static __forceinline__ unsigned short bits(float x) {
        const _Float16 h = (_Float16)x;  // round to nearest even
        return __builtin_bit_cast(unsigned short, h);
    }
};
struct BF16 {
    static constexpr bool kIsF16 = false;
    __device__ static __forceinline__ float lo(unsigned w) { return __builtin_bit_cast(float, w << 16); }
    __device__ static __forceinline__ float hi(unsigned w) { return __builtin_bit_cast(float, w & 0xffff0000u); }
    static constexpr unsigned kOnes = 0x3F803F80u;
    __device__ static __forceinline__ f32x16 mfma(u32x4 a, u32x4 b, f32x16 c) {
        return __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), c, 0, 0, 0);
    }
    __device__ static __forceinline__ f32x4 mfma_k32(u32x4 a, u32x4 b, f32x4 c) {   // 16x16x32: mlp_lp16.hip
        return __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), c, 0, 0, 0);
    }
    __device__ static __forceinline__ unsigned pack2(float lo, float hi) {
        unsigned r;
        asm volatile("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(r) : "v"(lo), "v"(hi));
        return r;
    }
    __device__ static __forceinline__ float dot2(unsigned a, unsigned b, float acc) {
        asm volatile("v_dot2c_f32_bf16 %0, %1, %2" : "+v"(acc) : "v"(a), "v"(b));
        return acc;
    }
    __device__ static __forceinline__ unsigned short bits(float x) {  // round to nearest even
        unsigned u = __builtin_bit_cast(unsigned, x);
        if ((u & 0x7fffffffu) > 0x7f800000u) return (unsigned short)((u >> 16) | 0x40);  // quiet NaN
        u += 0x7fffu + ((u >> 16) & 1u);
        return (unsigned short)(u >> 16);
    }
};

// four packed words (one u32x4 B operand) from eight accumulator elements, straight out of the AGPR file.  VALU
// cannot source AGPRs; doing the v_accvgpr_reads inside the asm keeps hipcc from hoisting hundreds of them ahead of
// the pass and spilling (a spilled "pending" ring register is a race, scripts/check_lds_ring.py).  Reads first,
// then converts, then clamps: no instruction depends on the one just before it.  Measured: ~5.8 cycles per VALU
// instruction, dominated by the AGPR reads.
#define NSOS_LP_QUAD(CVT, RELU_OPS)                                                                                  \
    asm volatile("v_accvgpr_read_b32 %0, %8\n\tv_accvgpr_read_b32 %4, %9\n\t"                                        \
                 "v_accvgpr_read_b32 %1, %10\n\tv_accvgpr_read_b32 %5, %11\n\t"                                      \
                 "v_accvgpr_read_b32 %2, %12\n\tv_accvgpr_read_b32 %6, %13\n\t"                                      \
                 "v_accvgpr_read_b32 %3, %14\n\tv_accvgpr_read_b32 %7, %15\n\t" CVT " %0, %0, %4\n\t" CVT          \
                 " %1, %1, %5\n\t" CVT " %2, %2, %6\n\t" CVT " %3, %3, %7" RELU_OPS                                 \
                 : "=&v"(r0), "=&v"(r1), "=&v"(r2), "=&v"(r3), "=&v"(t0), "=&v"(t1), "=&v"(t2), "=&v"(t3)            \
                 : "a"(z[0]), "a"(z[1]), "a"(z[2]), "a"(z[3]), "a"(z[4]), "a"(z[5]), "a"(z[6]), "a"(z[7]))
#define NSOS_LP_RELU4 "\n\tv_pk_max_i16 %0, %0, 0\n\tv_pk_max_i16 %1, %1, 0\n\tv_pk_max_i16 %2, %2, 0\n\tv_pk_max_i16 %3, %3, 0"
template <class T, bool RELU>
__device__ __forceinline__ u32x4 pack8_acc(const float (&z)[8]) {
    unsigned r0, r1, r2, r3, t0, t1, t2, t3;
    if constexpr (T::kIsF16) {
        if constexpr (RELU) NSOS_LP_QUAD("v_cvt_pk_f16_f32", NSOS_LP_RELU4); else NSOS_LP_QUAD("v_cvt_pk_f16_f32", "");
    } else {
        if constexpr (RELU) NSOS_LP_QUAD("v_cvt_pk_bf16_f32", NSOS_LP_RELU4); else NSOS_LP_QUAD("v_cvt_pk_bf16_f32", "");
    }
    return u32x4{r0, r1, r2, r3};
}

struct LpParams {
    const unsigned* aux;
    const unsigned char* chunks;
    const float* rays_o;
    const float* rays_d;
    const float* viewdirs;
    const float* z_vals;
    float* raw;
    long long n_pts;
    int n_samples;
    int n_tiles;
    unsigned long long* prof;  // diagnostics (nsos_mlp_profile_rays_lp): per-wave shader-clock stamps, or NULL
    float* sem_in;   // SAVE: [P,320] = [relu(h7) | x63 | 1.0] as the 16-bit values the semantic head consumed, widened to fp32
    unsigned* sem_in16;   // ... or (if not NULL) the same matrix kept in its 16-bit format T: [P,320] halves = 160 words per point
    float* sem_hid;  // SAVE: [P,128] = relu(semantic_linear.0(...)) (fp32 accumulators)
    unsigned* sem_hid16;  // ... or, with sem_in16 (the compact path), the same rounded to the 16-bit format T: [P,128] halves = 64 words per point
};
constexpr int kProfSlots = 64;

// encoded feature idx lives in half-wave (idx >> 3) & 1: a K-slice of 16 consecutive features, lane half kg
// supplying k-slots 8kg .. 8kg+7
struct SliceHalf {
    __host__ __device__ static constexpr int of(int idx) { return (idx >> 3) & 1; }
};

// one K-slice (4 packed words) of an encoding: word q of lane half kg = features 16s + 8kg + 2q, +1
template <class T, int L, int S, bool ONE_AT_63>
__device__ __forceinline__ u32x4 enc_slice(const Enc<L, SliceHalf>& e, const float (&x)[3], int kg) {
    u32x4 out;
    static_for<0, 4>([&](auto qc) {
        constexpr int q = decltype(qc)::value, f0 = 16 * S + 2 * q, f1 = 16 * S + 8 + 2 * q;
        const float lo_a = e.template feature<f0, 0>(x), lo_b = e.template feature<f0 + 1, 0>(x);
        const float hi_a = e.template feature<f1, 1>(x);
        const float hi_b = (ONE_AT_63 && f1 + 1 == 63) ? 1.0f : e.template feature<f1 + 1, 1>(x);
        out[q] = T::pack2(kg ? hi_a : lo_a, kg ? hi_b : lo_b);
    });
    return out;
}


// H[c][2t+u] = pack16(relu?(Z[c][t][8u .. 8u+7]))   -- one batched VALU pass per layer
template <class T, int NT, bool RELU>
__device__ __forceinline__ void activate(u32x4 (&H)[2][2 * NT], const f32x16 (&Z)[2][NT]) {
    // the asm below reads the accumulators behind hipcc's hazard tracking: an 8-pass MFMA result needs up to
    // 11 wait states before a VALU read (the compiler itself emits s_nop 6 after other filler here)
    asm volatile("s_nop 15\n\ts_nop 3" ::: "memory");
#pragma unroll
    for (int c = 0; c < 2; ++c)
#pragma unroll
        for (int t = 0; t < NT; ++t)
#pragma unroll
            for (int u = 0; u < 2; ++u) {
                float z[8];
#pragma unroll
                for (int i = 0; i < 8; ++i) z[i] = Z[c][t][8 * u + i];
                H[c][2 * t + u] = pack8_acc<T, RELU>(z);
            }
}

// fp32 vector-ALU heads on fp32 accumulators (rgb: NO = 3, semantics: NO = 2): this half-wave's partial chains
//   part[o] = fma(w[o][f], relu(h[f]), part[o]) over the lane's NT*16 features, weights from the LDS copy of aux.
// Each accumulator element is read from its AGPR and clamped ONCE (asm, see pack8_acc) and feeds all NO chains.
template <int NT, int NO>
__device__ __forceinline__ void heads_partial_f32(const f32x16 (&h)[NT], const float* w_lane, float (&part)[NO]) {
    asm volatile("s_nop 15\n\ts_nop 3" ::: "memory");  // MFMA result -> VALU read wait states
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            f32x4 w[NO];
#pragma unroll
            for (int o = 0; o < NO; ++o) w[o] = *reinterpret_cast<const f32x4*>(w_lane + o * 128 + t * 16 + q * 4);
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                float x;
                asm volatile("v_accvgpr_read_b32 %0, %1\n\tv_max_f32 %0, 0, %0" : "=v"(x) : "a"(h[t][q * 4 + j]));
#pragma unroll
                for (int o = 0; o < NO; ++o) part[o] = __fmaf_rn(w[o][j], x, part[o]);
            }
        }
}

// mlp_lp8.hip: the two-waves-per-SIMD kernel (8 waves x 32 points per 256-point tile), same packed stream and results
int32_t launch_lp8(const LpParams& p, int32_t sem_mode, bool is_f16, bool save, hipStream_t stream);

// mlp_lp16.hip: the same workgroup shape on v_mfma_f32_16x16x32 (its own packed stream: p.chunks points at it)
__host__ __device__ constexpr int lp16_chunks(int sem) {
    // L0 (1) + 7 quad layers x 4 + L5 h (4) + L5 x63 (1) + [sem0 h (2) + tail (1) | sigma (1)] + views (2)
    return 1 + 28 + 5 + (sem ? 3 : 1) + 2;
}
// behind the chunks: the four A operands of rgb_linear (rows 0..2 of the raw tile x 4 slices of the view branch's hidden
// activations), which the kernel keeps resident in LDS instead of streaming them as a chunk of their own
constexpr int kLp16TailBytes = 4096;
__host__ __device__ constexpr size_t lp16_stream_bytes(int sem) { return (size_t)lp16_chunks(sem) * kSlotBytes + kLp16TailBytes; }
int32_t launch_lp16(const LpParams& p, int32_t sem_mode, bool is_f16, bool save, hipStream_t stream);
int32_t pack_lp16(const void* tensors, int32_t sem_mode, bool is_f16, unsigned char* chunks, hipStream_t stream, bool heads_only = false);

}  // namespace lp
}  // namespace nsos
