// K2-X3: the fused positional-encoding + MLP kernel with SPLIT-fp16 operands -- fp32-grade results from the 16-bit
// matrix pipe.  Same network and replaced reference code as mlp_fused.hip (models/embedder.py:34-48,
// models/nerf_mlp.py:67-100,179-215).  Opt-in (NeRFNet.mlp_precision = "fp16x3"); the parity path and bench.py's
// headline stay on the exact-fp32 kernel.
//
// Every fp32 value v that enters a product is carried as two fp16 numbers, hi = fp16(v) and lo = fp16(v - hi)
// (together 22 mantissa bits), and every product W.h is evaluated as three 16-bit MFMAs accumulated in fp32:
//     W_hi.h_hi  +  W_hi.h_lo  +  W_lo.h_hi                       (the W_lo.h_lo term is below 2^-22 relative)
// The weights' lo parts are stored x 2^11: a weight of 0.05 has a residual of ~1e-5, an fp16 SUBNORMAL, and the
// quantisation of those residuals was the largest error term (CPU emulation: 2-5x the final error).  Scaled, they are
// normal numbers; their products therefore get their own accumulator, folded in as Zm + 2^-11 Zx by the activation.
// Measured on the GPU against an fp64 evaluation of the same network (scripts/accuracy_x3.py): at or below the error
// of plain fp32 arithmetic, 3 orders of magnitude inside the 1e-4 parity bar -- while the matrix work costs 3 x 32
// cycles per 16 k-slots instead of 8 x 64 for the exact-fp32 MFMA (5.3x less pipe time).
//
// Structure: the reduced-precision kernel's (mlp_lp.hip), with
//   * ONE 32-point column per wave (tile = 128 points): accumulators are 2 x 128 AGPRs -- Zm collects W_hi.(h_hi + h_lo),
//     Zx the scaled W_lo.h_hi -- and the activations are two packed files Hh, Hl of 64 VGPRs each;
//   * an "item" (output tile t, K-slice s or the bias) owns TWO A operands, hi and lo.  Their stream order is skewed,
//       hi_0, hi_1, lo_0, hi_2, lo_1, ..., hi_{n-1}, lo_{n-2}, lo_{n-1}
//       hi group k:  Zm[t_k] += A_hi.B_hi[s_k],  Zm[t_k] += A_hi.B_lo[s_k]        lo group k:  Zx[t_k] += A_lo.B_hi[s_k]
//     (the two MFMAs of a hi group are back to back on one accumulator: +3.5 cycles, scripts/ubench/mfma_chain.hip;
//     every other dependent pair is >= 4 issue slots apart, which is free);
//   * the activation pass also splits: z = Zm + 2^-11 Zx, relu, hi = cvt_pk(z), lo = cvt_pk(z - hi) with v_fma_mix_f32
//     reading the fp16 halves of hi directly (12 VALU per packed pair, all inside asm: see mlp_lp.hip on why);
//   * biases ride as leading items against B = 1.0 (hi and lo), layer 0's in the encoding's pad slot;
//   * sigma head: v_dot2c over the split activations and split weights (three dot products per word); rgb / semantic
//     output heads: fp32 VALU on the summed fp32 accumulators.
// Compiled with -ffp-contract=off (x = o + d*z stays a separately rounded multiply and add).
#include "x3_common.h"
#include "x316.h"
#include <cstdlib>
#include <cstring>

typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
namespace {

// aux stream, offsets in 4-byte words
constexpr int kAuxAlphaHi = 0;     // 128 words: sigma-head weights, fp16 hi parts, packed [kg][slice 0..15][q 0..3]
constexpr int kAuxAlphaLo = 128;   // 128 words: their lo parts
constexpr int kAuxRgbW = 256;      // 3 x 128 fp32, accumulator layout [hi][t 0..3][r 0..15]
constexpr int kAuxSem2W = 640;     // 2 x 128 fp32
constexpr int kAuxScalars = 896;   // alpha_b, rgb_b[3], sem2_b[2]
constexpr int kAuxWords = 1024;

enum ChunkKind { kHid8 = 0, kEnc8 = 1, kHid4 = 2, kEnc4 = 3, kDir4 = 4 };

__host__ __device__ constexpr int x3_chunks(int sem) {
    // L0 (2) + 8 layers x 8 + L5 x63 part (2) + [sem0 (4) (+ x63 part 1)] + views (4) + dir (1)
    return 2 + 64 + 2 + (sem ? 4 + (sem == 2 ? 1 : 0) : 0) + 5;
}

struct X3Params {
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
    float* sem_in;   // SAVE: [P,320] = [relu(h7) | x63 | 1.0], the fp32 values hi + lo the semantic head consumed
    float* sem_hid;  // SAVE: [P,128] = relu(semantic_linear.0(...)) (fp32 accumulators)
    float* acts;     // SAVE == 2 (full backward): [P, NSOS_ACTS_DIM] every layer's activations, see nerf_sos_hip.h
    unsigned long long* prof;  // diagnostics (nsos_mlp_profile_rays_x3): per-wave shader-clock stamps, or NULL
    unsigned* masks;  // SAVE == 2: ReLU bit masks of the 8 trunk layers, [tile][layer][256 threads][4 words] (activate_bits)
};
constexpr int kProfSlots = 64;

// encoded feature idx lives in half-wave (idx >> 3) & 1: a K-slice of 16 consecutive features, lane half kg
// supplying k-slots 8kg .. 8kg+7
struct SliceHalf {
    __host__ __device__ static constexpr int of(int idx) { return (idx >> 3) & 1; }
};
// one K-slice (4 packed words, hi and lo) of an encoding: word q of lane half kg = features 16s + 8kg + 2q, +1
template <int L, int S, bool ONE_AT_63>
__device__ __forceinline__ void enc_slice(const Enc<L, SliceHalf>& e, const float (&x)[3], int kg, u32x4& hi, u32x4& lo) {
    static_for<0, 4>([&](auto qc) {
        constexpr int q = decltype(qc)::value, f0 = 16 * S + 2 * q, f1 = 16 * S + 8 + 2 * q;
        const float lo_a = e.template feature<f0, 0>(x), lo_b = e.template feature<f0 + 1, 0>(x);
        const float hi_a = e.template feature<f1, 1>(x);
        const float hi_b = (ONE_AT_63 && f1 + 1 == 63) ? 1.0f : e.template feature<f1 + 1, 1>(x);
        unsigned h, l;
        split2(kg ? hi_a : lo_a, kg ? hi_b : lo_b, h, l);
        hi[q] = h;
        lo[q] = l;
    });
}

// Hh/Hl[2t+u] = split(relu?(Zm[t] + Zx[t])[8u .. 8u+7])   -- one batched VALU pass per layer
// ... and the same pass for the full-training variant: additionally collects the layer's ReLU pattern as a per-lane bit mask,
// bit 31 - ((t & 1) * 16 + r) of word t / 2 for accumulator element r of tile t (what mlp_x3_bwd.hip applies to the gradient)
template <int NT>
__device__ __forceinline__ void activate_bits(u32x4 (&Hh)[2 * NT], u32x4 (&Hl)[2 * NT], const f32x16 (&Zm)[NT], const f32x16 (&Zx)[NT],
                                              u32x4& mask) {
    asm volatile("s_nop 15\n\ts_nop 3" ::: "memory");
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
        for (int u = 0; u < 2; ++u)
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                unsigned h, l, m = mask[t >> 1];
                split2_acc_bits(Zm[t][8 * u + 2 * q], Zx[t][8 * u + 2 * q], Zm[t][8 * u + 2 * q + 1], Zx[t][8 * u + 2 * q + 1], h, l, m);
                mask[t >> 1] = m;
                Hh[2 * t + u][q] = h;
                Hl[2 * t + u][q] = l;
            }
}

template <int NT, bool RELU>
__device__ __forceinline__ void activate(u32x4 (&Hh)[2 * NT], u32x4 (&Hl)[2 * NT], const f32x16 (&Zm)[NT], const f32x16 (&Zx)[NT]) {
    asm volatile("s_nop 15\n\ts_nop 3" ::: "memory");   // MFMA result -> VALU read wait states (the reads are inside asm)
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
        for (int u = 0; u < 2; ++u)
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                unsigned h, l;
                split2_acc<RELU>(Zm[t][8 * u + 2 * q], Zx[t][8 * u + 2 * q], Zm[t][8 * u + 2 * q + 1], Zx[t][8 * u + 2 * q + 1], h, l);
                Hh[2 * t + u][q] = h;
                Hl[2 * t + u][q] = l;
            }
}

// fp32 vector-ALU heads on the fp32 accumulators of a 128-wide hidden layer (rgb: NO = 3, semantics: NO = 2)
template <int NO>
__device__ __forceinline__ void heads_partial_f32(const f32x16 (&hm)[4], const f32x16 (&hx)[4], const float* w_lane, float (&part)[NO]) {
    asm volatile("s_nop 15\n\ts_nop 3" ::: "memory");
#pragma unroll
    for (int t = 0; t < 4; ++t)
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            f32x4 w[NO];
#pragma unroll
            for (int o = 0; o < NO; ++o) w[o] = *reinterpret_cast<const f32x4*>(w_lane + o * 128 + t * 16 + q * 4);
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                float x, y;
                asm volatile("v_accvgpr_read_b32 %0, %2\n\tv_accvgpr_read_b32 %1, %3\n\tv_fmac_f32 %0, 0x3a000000, %1\n\tv_max_f32 %0, 0, %0"
                             : "=&v"(x), "=&v"(y) : "a"(hm[t][q * 4 + j]), "a"(hx[t][q * 4 + j]));
#pragma unroll
                for (int o = 0; o < NO; ++o) part[o] = __fmaf_rn(w[o][j], x, part[o]);
            }
        }
}

// ------------------------------------------------------------------------------------------ the kernel
// SAVE 1: training-mode variant that also stores what the semantic head's backward needs (K5, frozen backbone);
// SAVE 2: stores every layer's activations for the full backward (K7).  Stored values are the fp32 numbers hi + lo
// that the next layer consumed.
template <int SEM, int SAVE = 0>
__global__ __launch_bounds__(256, 1) void mlp_x3_kernel(const X3Params P) {
    extern __shared__ __attribute__((aligned(16))) unsigned char lds[];  // 4 x 36 KiB weight slots + 4 KiB head weights
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int pj = lane & 31, kg = lane >> 5;
    constexpr int NCH = x3_chunks(SEM);
    constexpr int C = SEM ? 6 : 4;

    // ---- weight stream: slots rotate (c0 = chunk cur, c1 = cur+1, c2 = cur+2, c3 = being filled with cur+3)
    const unsigned lds_base = (unsigned)(size_t)(__attribute__((address_space(3))) unsigned char*)lds;
    const int wave_s = __builtin_amdgcn_readfirstlane(wave);
    const unsigned voff = (unsigned)(wave * 1024 + lane * 16);
    auto lane_addr = [&](int b) { return lds_base + (unsigned)(b * kSlotBytes + lane * 16); };
    auto wave_addr = [&](int b) { return lds_base + (unsigned)(b * kSlotBytes + wave_s * 1024); };
    unsigned c0 = lane_addr(0), c1 = lane_addr(1), c2 = lane_addr(2), c3 = lane_addr(3);
    unsigned d0 = wave_addr(0), d1 = wave_addr(1), d2 = wave_addr(2), d3 = wave_addr(3);
    const unsigned char* const src_end = P.chunks + (size_t)NCH * kSlotBytes;
    const unsigned char* src3 = P.chunks + (size_t)(3 % NCH) * kSlotBytes;
    auto dma_piece = [&](const unsigned char* src_chunk, unsigned dst_wave, int i) {
        dma_1k(src_chunk + i * 4096, dst_wave + (unsigned)i * 4096u, voff);
    };
    auto side = [&](int i) { dma_piece(src3, d3, i); };
    auto mid = [&]() {
        // all DMA pieces except the newest kDmaPieces (chunk cur+2, issued one chunk ago) must have landed: that is chunk
        // cur+1, which the end of this chunk starts to read
        asm volatile("s_waitcnt vmcnt(%0)" ::"i"(kDmaPieces) : "memory");
        __builtin_amdgcn_s_barrier();
    };
    auto tail = [&]() {
        const unsigned tc = c0, td = d0;
        c0 = c1; c1 = c2; c2 = c3; c3 = tc;
        d0 = d1; d1 = d2; d2 = d3; d3 = td;
        src3 += kSlotBytes;
        if (src3 == src_end) src3 = P.chunks;
    };
    auto ctx = [&]() { return ChunkCtx{c0, c1}; };

    f32x4 ring[kRing];
#pragma unroll
    for (int k = 0; k < 3; ++k)
#pragma unroll
        for (int i = 0; i < kDmaPieces; ++i)
            dma_piece(P.chunks + (size_t)(k % NCH) * kSlotBytes, k == 0 ? d0 : (k == 1 ? d1 : d2), i);
    const unsigned* const aux_l = reinterpret_cast<const unsigned*>(lds + kSlots * kSlotBytes);
    *reinterpret_cast<u32x4*>(lds + kSlots * kSlotBytes + threadIdx.x * 16) = reinterpret_cast<const u32x4*>(P.aux)[threadIdx.x];
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    static_for<0, kRing>([&](auto ic) { lds_read_a<decltype(ic)::value * 1024>(ring[decltype(ic)::value], c0); });
    lgkm_wait<0>();
    NSOS_PIN();

#define IC(n) std::integral_constant<int, (n)> {}
    // One chunk = NI items of a part with NT output tiles and NB leading bias items; item i of the chunk is item I0 + i of
    // the part: a < NB the bias of tile a, else K-slice (a - NB) / NT of tile (a - NB) % NT.  2 NI groups (A operands) in
    // the skewed order  hi_0, [hi_k, lo_{k-1}] k = 1..NI-1, lo_{NI-1}.   ZF: the part starts the accumulation with its
    // slice 0 (C = 0) instead of with bias items.  NWORK: items that carry work (the rest is padding).
    // ride(g): extra work placed in the MFMA shadows after group g (the SAVE == 2 activation stores)
    auto run_chunk = [&](auto ni_c, auto nt_c, auto nb_c, auto i0_c, auto nwork_c, auto zf_c, auto& Zm, auto& Zx, auto&& bh, auto&& bl,
                         auto&& ride) __attribute__((always_inline)) {
        constexpr int NI = decltype(ni_c)::value, NT = decltype(nt_c)::value, NB = decltype(nb_c)::value;
        constexpr int I0 = decltype(i0_c)::value, NWORK = decltype(nwork_c)::value, NG = 2 * NI;
        constexpr bool ZF = decltype(zf_c)::value != 0;
        a_pipeline<NG, kRing, kPre, kMid>(ring, ctx(), [&](auto ic, const f32x4& a32) {
            constexpr int g = decltype(ic)::value;
            constexpr bool IS_HI = g == 0 || (g != NG - 1 && (g & 1));
            constexpr int item = g == 0 ? 0 : (g == NG - 1 ? NI - 1 : (IS_HI ? (g + 1) / 2 : (g - 2) / 2));
            constexpr int a = I0 + item;
            const u32x4 aop = __builtin_bit_cast(u32x4, a32);
            if constexpr (item < NWORK) {
                const f32x16 zero = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
                if constexpr (a < NB) {   // bias of tile a against B = 1.0: starts both accumulators
                    const u32x4 ones = {kOnes, kOnes, kOnes, kOnes};
                    if constexpr (IS_HI) Zm[a] = mfma16(aop, ones, zero);
                    else Zx[a] = mfma16(aop, ones, zero);
                } else {
                    constexpr int s = (a - NB) / NT, t = (a - NB) % NT;
                    constexpr bool FIRST = ZF && s == 0;
                    if constexpr (IS_HI) {
                        Zm[t] = mfma16(aop, bh(IC(s)), FIRST ? zero : Zm[t]);
                        Zm[t] = mfma16(aop, bl(IC(s)), Zm[t]);   // back to back on one accumulator: +3.5 cycles (mfma_chain.hip)
                    } else {
                        Zx[t] = mfma16(aop, bh(IC(s)), FIRST ? zero : Zx[t]);
                    }
                }
            }
            dma_slot<g - kMid, kDmaPieces>(side);
            ride(ic);
        }, mid, tail);
        // the accumulators are complete HERE: without a use at this point LLVM sinks whole chunks of MFMAs below later
        // branches (the encodings' range checks), keeping every A operand of the chunk alive in spill slots
#pragma unroll
        for (int t = 0; t < NT; ++t) asm volatile("" : "+a"(Zm[t]), "+a"(Zx[t]));
    };

    for (int tile = blockIdx.x; tile < P.n_tiles; tile += gridDim.x) {
        // ---- this lane's point: tile*128 + wave*32 + pj (both half-waves of a column hold the same point)
        int stamp_k = 0;
        auto stamp = [&]() {  // diagnostics only: one s_memtime per phase of the second tile of blocks 0..3 (steady state)
            if (P.prof && tile == (int)(blockIdx.x + gridDim.x) && blockIdx.x < 4) {
                const unsigned long long t = __builtin_readcyclecounter();
                if (lane == 0 && stamp_k < kProfSlots) P.prof[(blockIdx.x * 4 + wave) * kProfSlots + stamp_k] = t;
            }
            ++stamp_k;
        };
        stamp();  // 0: tile start
        const long long gp = (long long)tile * kTilePts + wave * 32 + pj;
        const long long gc = gp < P.n_pts ? gp : P.n_pts - 1;
        const int ray = (int)(gc / P.n_samples);
        float poison;
        u32x4 exh[4], exl[4];
        {
            const float z = P.z_vals[gc];
            float x[3];
#pragma unroll
            for (int k = 0; k < 3; ++k) {
                const float m = P.rays_d[3ll * ray + k] * z;  // models/sampler.py:70,166 (mul, then add)
                x[k] = P.rays_o[3ll * ray + k] + m;
            }
            poison = ((x[0] - x[0]) + (x[1] - x[1])) + (x[2] - x[2]);   // NaN iff an input is NaN / Inf (see mlp_fused.hip)
            Enc<NSOS_XYZ_FREQS, SliceHalf> e;
            e.evaluate(x, kg);
            enc_slice<NSOS_XYZ_FREQS, 0, true>(e, x, kg, exh[0], exl[0]);
            enc_slice<NSOS_XYZ_FREQS, 1, true>(e, x, kg, exh[1], exl[1]);
            enc_slice<NSOS_XYZ_FREQS, 2, true>(e, x, kg, exh[2], exl[2]);
            enc_slice<NSOS_XYZ_FREQS, 3, true>(e, x, kg, exh[3], exl[3]);  // feature 63 (pad) = 1.0: layer-0 bias
        }
        f32x16 Zm[8], Zx[8];
        u32x4 Hh[16], Hl[16];
        float sigma = 0.0f, sem_out[2] = {0.0f, 0.0f};
        auto ex_h = [&](auto sc) { return exh[decltype(sc)::value]; };
        auto ex_l = [&](auto sc) { return exl[decltype(sc)::value]; };
        auto h_h = [&](auto sc) { return Hh[decltype(sc)::value]; };
        auto h_l = [&](auto sc) { return Hl[decltype(sc)::value]; };
        auto store_H = [&](float* row) __attribute__((always_inline)) {   // 256 activations: word q of slice 2t+u = accumulator elements 8u+2q, +1 of tile t
#pragma unroll
            for (int t = 0; t < 8; ++t)
#pragma unroll
                for (int u = 0; u < 2; ++u)
#pragma unroll
                    for (int q = 0; q < 4; q += 2) {   // words q, q+1 = accumulator elements 8u+2q .. +3 = 4 consecutive features
                        const u32x4 hh = Hh[2 * t + u], hl = Hl[2 * t + u];
                        *reinterpret_cast<f32x4*>(row + 32 * t + 8 * (2 * u + (q >> 1)) + 4 * kg) =
                            f32x4{join<0>(hh[q], hl[q]), join<1>(hh[q], hl[q]), join<0>(hh[q + 1], hl[q + 1]), join<1>(hh[q + 1], hl[q + 1])};
                    }
        };
        // SAVE == 3: `acts` holds 16-bit floats; rows are formed as float pointers at ELEMENT offsets and re-based here
        auto acts16 = [&](const float* row) { return reinterpret_cast<unsigned short*>(P.acts) + (row - P.acts); };
        auto store_slices = [&](float* row, auto ns_c, const u32x4* sh, const u32x4* sl_) __attribute__((always_inline)) {   // encoded slices: features 16s + 8kg + 2q, +1
#pragma unroll
            for (int sl = 0; sl < decltype(ns_c)::value; ++sl)
#pragma unroll
                for (int q = 0; q < 4; q += 2) {
                    if constexpr (SAVE == 3)      // 16-bit activations: the hi words as they are (two packed words = four consecutive features)
                        *reinterpret_cast<u32x2*>(acts16(row) + 16 * sl + 8 * kg + 2 * q) = u32x2{sh[sl][q], sh[sl][q + 1]};
                    else
                        *reinterpret_cast<f32x4*>(row + 16 * sl + 8 * kg + 2 * q) =
                            f32x4{join<0>(sh[sl][q], sl_[sl][q]), join<1>(sh[sl][q], sl_[sl][q]), join<0>(sh[sl][q + 1], sl_[sl][q + 1]), join<1>(sh[sl][q + 1], sl_[sl][q + 1])};
                }
        };
        auto store_hidden128 = [&](float* hrow, const f32x16 (&am)[4], const f32x16 (&ax)[4]) __attribute__((always_inline)) {   // relu(Zm + 2^-11 Zx) of a 128-wide layer
            asm volatile("s_nop 15\n\ts_nop 3" ::: "memory");  // MFMA result -> VALU read wait states
            auto relu_acc = [](const float& m, const float& x) {   // AGPR reads inside asm: see split2_acc
                float r, y;
                asm volatile("v_accvgpr_read_b32 %0, %2\n\tv_accvgpr_read_b32 %1, %3\n\tv_fmac_f32 %0, 0x3a000000, %1\n\tv_max_f32 %0, 0, %0"
                             : "=&v"(r), "=&v"(y) : "a"(m), "a"(x));
                return r;
            };
#pragma unroll
            for (int t = 0; t < 4; ++t)
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const f32x4 v = f32x4{relu_acc(am[t][4 * q], ax[t][4 * q]), relu_acc(am[t][4 * q + 1], ax[t][4 * q + 1]),
                                          relu_acc(am[t][4 * q + 2], ax[t][4 * q + 2]), relu_acc(am[t][4 * q + 3], ax[t][4 * q + 3])};
                    if constexpr (SAVE == 3) {
                        unsigned w0, w1;
                        asm("v_cvt_pk_f16_f32 %0, %1, %2" : "=v"(w0) : "v"(v[0]), "v"(v[1]));
                        asm("v_cvt_pk_f16_f32 %0, %1, %2" : "=v"(w1) : "v"(v[2]), "v"(v[3]));
                        *reinterpret_cast<u32x2*>(acts16(hrow) + 32 * t + 8 * q + 4 * kg) = u32x2{w0, w1};
                    } else {
                        *reinterpret_cast<f32x4*>(hrow + 32 * t + 8 * q + 4 * kg) = v;
                    }
                }
        };
        const bool valid = gp < P.n_pts;
        float* const arow = SAVE >= 2 ? P.acts + gc * NSOS_ACTS_DIM : nullptr;
        auto no_ride = [](auto) {};
        // store number CC * PER + j (of 32 per layer: tile t, register half u, word pair) after group 12 + j * (20 / PER) of chunk CC
        auto ride_store = [&](auto gc_, auto cc_, auto per_, float* row) __attribute__((always_inline)) {
            constexpr int g = decltype(gc_)::value, CC = decltype(cc_)::value, PER = decltype(per_)::value, STRIDE = 20 / PER;
            if constexpr (SAVE >= 2 && g >= 12 && (g - 12) % STRIDE == 0 && (g - 12) / STRIDE < PER) {
                constexpr int k = CC * PER + (g - 12) / STRIDE, t = k >> 2, u = (k >> 1) & 1, q = 2 * (k & 1);
                const u32x4 hh = Hh[2 * t + u], hl = Hl[2 * t + u];
                if (valid) {
                    if constexpr (SAVE == 3)
                        *reinterpret_cast<u32x2*>(acts16(row) + 32 * t + 8 * (2 * u + (q >> 1)) + 4 * kg) = u32x2{hh[q], hh[q + 1]};
                    else
                        *reinterpret_cast<f32x4*>(row + 32 * t + 8 * (2 * u + (q >> 1)) + 4 * kg) =
                            f32x4{join<0>(hh[q], hl[q]), join<1>(hh[q], hl[q]), join<0>(hh[q + 1], hl[q + 1]), join<1>(hh[q + 1], hl[q + 1])};
                }
            }
        };
        if constexpr (SAVE >= 2) {
            if (valid) store_slices(arow + NSOS_ACTS_X, IC(4), exh, exl);   // slot 63 is the 1.0 pad
        }

        // pts_linears.0: 4 encoded slices x 8 tiles = 32 items in 2 chunks; slice 0 starts from C = 0
        stamp();  // 1: inputs + xyz encoding
        run_chunk(IC(16), IC(8), IC(0), IC(0), IC(16), IC(1), Zm, Zx, ex_h, ex_l, no_ride);
        run_chunk(IC(16), IC(8), IC(0), IC(16), IC(16), IC(1), Zm, Zx, ex_h, ex_l, no_ride);
        stamp();  // 2: L0 MFMAs
        // SAVE == 2: the trunk layers' ReLU patterns go to P.masks as bits (16 B per lane per layer, one coalesced store)
        u32x4 relu_bits = {0u, 0u, 0u, 0u};
        u32x4* const mrow = SAVE >= 2 ? reinterpret_cast<u32x4*>(P.masks) + (size_t)tile * 8 * 256 + threadIdx.x : nullptr;
        if constexpr (SAVE >= 2) { activate_bits<8>(Hh, Hl, Zm, Zx, relu_bits); mrow[0] = relu_bits; }
        else activate<8, true>(Hh, Hl, Zm, Zx);
        stamp();  // 3: L0 activation
        // pts_linears.1..7 (l = 1..7) and feature_linear (l = 8): 8 bias + 128 slice items = 8 chunks of 17
#pragma unroll 1
        for (int l = 1; l <= 8; ++l) {
            // SAVE == 2: the activations this layer consumes (H of layer l-1) are written to acts from inside its own MFMA stream,
            // 4 x 16 B per lane per chunk: stores issued in a burst after the activation pass made the next chunk's barrier
            // (a counted vmcnt) wait for HBM to take them
            float* const prow = SAVE >= 2 ? arow + 256 * (l - 1) : nullptr;
            static_for<0, 8>([&](auto cc) {
                run_chunk(IC(17), IC(8), IC(8), IC(17 * decltype(cc)::value), IC(17), IC(0), Zm, Zx, h_h, h_l,
                          [&](auto gc) { ride_store(gc, cc, IC(4), prow); });
            });
            if (l == 5) {   // skip connection: + W_x x63
                run_chunk(IC(16), IC(8), IC(0), IC(0), IC(16), IC(0), Zm, Zx, ex_h, ex_l, no_ride);
                run_chunk(IC(16), IC(8), IC(0), IC(16), IC(16), IC(0), Zm, Zx, ex_h, ex_l, no_ride);
            }
            stamp();  // 2 + 2l: MFMAs of layer l
            if (l < 8) {
                if constexpr (SAVE >= 2) { activate_bits<8>(Hh, Hl, Zm, Zx, relu_bits); mrow[256 * l] = relu_bits; }
                else activate<8, true>(Hh, Hl, Zm, Zx);
            } else {
                activate<8, false>(Hh, Hl, Zm, Zx);
            }
            stamp();  // 3 + 2l: activation pass
            if (l == 7) {
                // sigma head (models/nerf_mlp.py:77): three dot products of the split activations and split weights
                const unsigned* awh = aux_l + kAuxAlphaHi + kg * 64;
                const unsigned* awl = aux_l + kAuxAlphaLo + kg * 64;
                float pa = kg ? 0.0f : __builtin_bit_cast(float, aux_l[kAuxScalars]), px = 0.0f;
                // Two chains (main, scaled lo weights), each kept back to back, and an s_nop before anything else reads
                // them: a v_dot2c result read by a NON-dot VALU instruction needs 3 wait states (gfx940+ hazard), which
                // hipcc inserts for its own code but cannot for asm operands -- without the nop sigma came out wrong by 1e-2.
#pragma unroll
                for (int s = 0; s < 16; ++s) {
                    const u32x4 wh = *reinterpret_cast<const u32x4*>(awh + 4 * s);
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        pa = dot2(Hl[s][q], wh[q], pa);
                        pa = dot2(Hh[s][q], wh[q], pa);
                    }
                }
#pragma unroll
                for (int s = 0; s < 16; ++s) {
                    const u32x4 wl = *reinterpret_cast<const u32x4*>(awl + 4 * s);
#pragma unroll
                    for (int q = 0; q < 4; ++q) px = dot2(Hh[s][q], wl[q], px);   // the weights' lo parts are stored x 2^11
                }
                asm volatile("s_nop 3" ::: "memory");
                sigma = both_halves(__fmaf_rn(px, kLoUnscale, pa));
                if constexpr (SEM != 0) {  // semantic head (models/nerf_mlp.py:79-80): 4 bias + 64 slice items = 4 chunks of 17
                    f32x16 sm[4], sx[4];
                    static_for<0, 4>([&](auto cc) { run_chunk(IC(17), IC(4), IC(4), IC(17 * decltype(cc)::value), IC(17), IC(0), sm, sx, h_h, h_l, no_ride); });
                    if constexpr (SEM == 2) run_chunk(IC(16), IC(4), IC(0), IC(0), IC(16), IC(0), sm, sx, ex_h, ex_l, no_ride);
                    if constexpr (SAVE == 1) {
                        if (valid) {
                            float* row = P.sem_in + gp * 320;
                            store_H(row);
                            store_slices(row + 256, IC(4), exh, exl);
                            store_hidden128(P.sem_hid + gp * 128, sm, sx);
                        }
                    }
                    if constexpr (SAVE >= 2) {
                        if (valid) store_hidden128(arow + NSOS_ACTS_SEM, sm, sx);
                    }
                    float ps[2];
#pragma unroll
                    for (int o = 0; o < 2; ++o) ps[o] = kg ? 0.0f : __builtin_bit_cast(float, aux_l[kAuxScalars + 4 + o]);
                    heads_partial_f32<2>(sm, sx, reinterpret_cast<const float*>(aux_l) + kAuxSem2W + kg * 64, ps);
#pragma unroll
                    for (int o = 0; o < 2; ++o) sem_out[o] = both_halves(ps[o]);
                }
                stamp();  // 18 (l == 7 only; the later slots shift by one): sigma + semantic heads
            }
        }
        // view branch: cat([feature, dir27]) -> 128 -> rgb   (H = feature, no activation)
        f32x16 vm[4], vx[4];
        static_for<0, 4>([&](auto cc) {
            run_chunk(IC(17), IC(4), IC(4), IC(17 * decltype(cc)::value), IC(17), IC(0), vm, vx, h_h, h_l,
                      [&](auto gc) { ride_store(gc, cc, IC(8), SAVE >= 2 ? arow + NSOS_ACTS_FEAT : nullptr); });   // H = the feature vector
        });
        stamp();  // 21: view-branch MFMAs on the feature
        u32x4 edh[2], edl[2];   // the direction encoding is evaluated only now: its registers would not fit beside the trunk
        {
            float dv[3];
#pragma unroll
            for (int k = 0; k < 3; ++k) dv[k] = P.viewdirs[3ll * ray + k];
            poison += ((dv[0] - dv[0]) + (dv[1] - dv[1])) + (dv[2] - dv[2]);
            Enc<NSOS_DIR_FREQS, SliceHalf> e;
            e.evaluate(dv, kg);
            enc_slice<NSOS_DIR_FREQS, 0, false>(e, dv, kg, edh[0], edl[0]);
            enc_slice<NSOS_DIR_FREQS, 1, false>(e, dv, kg, edh[1], edl[1]);
        }
        stamp();  // 22: direction encoding
        run_chunk(IC(8), IC(4), IC(0), IC(0), IC(8), IC(0), vm, vx, [&](auto sc) { return edh[decltype(sc)::value & 1]; },
                  [&](auto sc) { return edl[decltype(sc)::value & 1]; }, no_ride);   // 2 slices x 4 tiles = 8 items (16 groups)
        stamp();  // 23: direction MFMAs
        if constexpr (SAVE >= 2) {
            if (valid) {
                store_slices(arow + NSOS_ACTS_D, IC(2), edh, edl);   // 27 features, zero pad
                store_hidden128(arow + NSOS_ACTS_VIEWS, vm, vx);
            }
        }
        float rgb[3];
#pragma unroll
        for (int o = 0; o < 3; ++o) rgb[o] = kg ? 0.0f : __builtin_bit_cast(float, aux_l[kAuxScalars + 1 + o]);
        heads_partial_f32<3>(vm, vx, reinterpret_cast<const float*>(aux_l) + kAuxRgbW + kg * 64, rgb);
#pragma unroll
        for (int o = 0; o < 3; ++o) rgb[o] = both_halves(rgb[o]);
        if (poison != poison) {
            const float qnan = __builtin_nanf("");
            rgb[0] = rgb[1] = rgb[2] = sigma = sem_out[0] = sem_out[1] = qnan;
        }
        if (gp < P.n_pts) {
            float* out = P.raw + gp * C;
            if constexpr (C == 4) {
                if (kg == 0) *reinterpret_cast<f32x4*>(out) = f32x4{rgb[0], rgb[1], rgb[2], sigma};
            } else {
                if (kg == 0) {
                    *reinterpret_cast<f32x2*>(out) = f32x2{rgb[0], rgb[1]};
                    *reinterpret_cast<f32x2*>(out + 2) = f32x2{rgb[2], sigma};
                } else {
                    *reinterpret_cast<f32x2*>(out + 4) = f32x2{sem_out[0], sem_out[1]};
                }
            }
        }
        stamp();  // 24: rgb head + stores
    }
#undef IC
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
}

// ------------------------------------------------------------------------------------------ packing
struct X3Chunk {
    const float* w;
    const float* bias;  // leading bias items of the part (a < NB), or the pad-slot bias (kEnc8 of layer 0), or NULL
    int in_dim, col_base, kind, i0, n_items;
};
struct X3PackParams {
    X3Chunk ch[84];
    int n_chunks;
    const float* alpha_w; const float* alpha_b;
    const float* rgb_w; const float* rgb_b;
    const float* sem2_w; const float* sem2_b;
    unsigned* aux;
    unsigned short* chunks;
};

__global__ __launch_bounds__(256) void x3_pack_kernel(const X3PackParams P) {
    const long long gid = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (gid < kAuxWords) {
        const int a = (int)gid;
        unsigned v = 0;
        auto feat128 = [](int rem) { return acc_feature((rem & 63) >> 4, rem & 15, rem >> 6); };
        if (a < kAuxRgbW) {  // sigma head weights (hi then lo parts), packed pairs in H order: word = [kg][s][q]
            const bool is_lo = a >= kAuxAlphaLo;
            const int b = a & 127, kgl = b >> 6, s = (b & 63) >> 2, q = b & 3;
            const int f0 = acc_feature(s >> 1, 8 * (s & 1) + 2 * q, kgl), f1 = acc_feature(s >> 1, 8 * (s & 1) + 2 * q + 1, kgl);
            const float w0 = P.alpha_w[f0], w1 = P.alpha_w[f1];
            unsigned short h0 = f16_bits(w0), h1 = f16_bits(w1);
            if (is_lo) { h0 = f16_bits((w0 - f16_value(h0)) * kLoScale); h1 = f16_bits((w1 - f16_value(h1)) * kLoScale); }
            v = (unsigned)h0 | ((unsigned)h1 << 16);
        } else if (a < kAuxSem2W) { const int rem = a - kAuxRgbW; v = __builtin_bit_cast(unsigned, P.rgb_w[(rem >> 7) * 128 + feat128(rem & 127)]); }
        else if (a < kAuxScalars) { const int rem = a - kAuxSem2W; v = P.sem2_w ? __builtin_bit_cast(unsigned, P.sem2_w[(rem >> 7) * 128 + feat128(rem & 127)]) : 0u; }
        else {
            const int i = a - kAuxScalars;
            float f = 0.0f;
            if (i == 0) f = P.alpha_b[0];
            else if (i < 4) f = P.rgb_b[i - 1];
            else if (i < 6) f = P.sem2_b ? P.sem2_b[i - 4] : 0.0f;
            v = __builtin_bit_cast(unsigned, f);
        }
        P.aux[a] = v;
    }
    const long long per_chunk = kSlotBytes / 2;  // 16-bit elements per slot
    if (gid >= (long long)P.n_chunks * per_chunk) return;
    const X3Chunk ck = P.ch[gid / per_chunk];
    const int within = (int)(gid % per_chunk);
    const int g = within >> 9, lane = (within >> 3) & 63, e = within & 7;  // 512 elements per A operand
    const int i = lane & 31, kgl = lane >> 5, m = 8 * kgl + e;             // output row i of the tile, k-slot m
    const int NG = 2 * ck.n_items;
    float v = 0.0f;
    bool is_hi = true;
    if (g < NG) {
        is_hi = g == 0 || (g != NG - 1 && (g & 1));
        const int item = g == 0 ? 0 : (g == NG - 1 ? ck.n_items - 1 : (is_hi ? (g + 1) / 2 : (g - 2) / 2));
        const int a = ck.i0 + item;
        const bool eight = ck.kind == kHid8 || ck.kind == kEnc8;
        const int nt = eight ? 8 : 4;
        const int nb = (ck.kind == kHid8 || ck.kind == kHid4) ? nt : 0;
        if (a < nb) {
            v = (m == 0) ? ck.bias[32 * a + i] : 0.0f;
        } else {
            const int s = (a - nb) / nt, t = (a - nb) % nt;
            int f = -1;
            switch (ck.kind) {
                case kHid8: case kHid4: f = acc_feature(s >> 1, 8 * (s & 1) + e, kgl); break;
                case kEnc8: case kEnc4: f = 16 * s + m; if (f >= NSOS_XYZ_DIM) f = (f == 63 && ck.bias) ? -2 : -1; break;
                case kDir4: f = 16 * s + m; if (f >= NSOS_DIR_DIM || s > 1) f = -1; break;
            }
            if (f >= 0) v = ck.w[(long long)(32 * t + i) * ck.in_dim + ck.col_base + f];
            else if (f == -2) v = ck.bias[32 * t + i];  // layer-0 bias rides in the encoding's pad slot (input 1.0)
        }
    }
    unsigned short h = f16_bits(v);
    if (!is_hi) h = f16_bits((v - f16_value(h)) * kLoScale);
    P.chunks[gid] = h;
}


constexpr int kLdsBytes = kSlots * kSlotBytes + kAuxWords * 4;

template <int SEM, int SAVE = 0>
int32_t launch_x3(const X3Params& p, hipStream_t stream) {
    static NsosPerDeviceFlag configured_on;
    bool& configured = configured_on.here();
    if (!configured) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&mlp_x3_kernel<SEM, SAVE>),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, kLdsBytes);
        if (e != hipSuccess) return (int32_t)e;
        configured = true;
    }
    const int cus = nsos_device_cus();
    const int grid = p.n_tiles < cus ? p.n_tiles : cus;
    hipLaunchKernelGGL((mlp_x3_kernel<SEM, SAVE>), dim3(grid), dim3(256), kLdsBytes, stream, p);
    return nsos_launch_status();
}

}  // namespace

// ------------------------------------------------------------------------------------------ C ABI
// One packed buffer, two forward kernels (round 6): [aux | mlp_x3_kernel's chunks | mlp_x316_kernel's stream (csrc/mlp_x316.hip)].
// The training variants (SAVE) and the backward chain read the first stream only; inference runs the selected kernel.
namespace {
size_t x3_first_stream_bytes(int sem_mode) { return (size_t)kAuxWords * 4 + (size_t)x3_chunks(sem_mode) * kSlotBytes; }
int g_x3_kernel = 0;      // 0: not chosen yet -> NSOS_X3_KERNEL (1 = mlp_x3_kernel, 32x32x16; 2 = mlp_x316_kernel, 16x16x32: the default)
int x3_kernel() {
    if (g_x3_kernel == 0) {
        const char* e = getenv("NSOS_X3_KERNEL");
        g_x3_kernel = (e && (e[0] == '1' || !strcmp(e, "x3"))) ? 1 : 2;
    }
    return g_x3_kernel;
}
}  // namespace

extern "C" size_t nsos_mlp_packed_bytes_x3(int32_t sem_mode) {
    if (sem_mode < 0 || sem_mode > 2) return 0;
    return x3_first_stream_bytes(sem_mode) + nsos::x316::stream_bytes(sem_mode);
}

extern "C" int32_t nsos_mlp_x3_select_kernel(int32_t kernel) {
    NSOS_REQUIRE(kernel == 1 || kernel == 2, NSOS_ERR_UNSUPPORTED);
    g_x3_kernel = kernel;
    return NSOS_OK;
}
extern "C" int32_t nsos_mlp_x3_selected_kernel(void) { return x3_kernel(); }

extern "C" int32_t nsos_mlp_pack_x3(const nsos_mlp_tensors* T_, int32_t sem_mode, void* packed, size_t packed_bytes, void* stream) {
    NSOS_REQUIRE(T_ && packed, NSOS_ERR_NULL_POINTER);
    NSOS_REQUIRE(sem_mode >= 0 && sem_mode <= 2, NSOS_ERR_UNSUPPORTED);
    NSOS_REQUIRE(packed_bytes >= nsos_mlp_packed_bytes_x3(sem_mode), NSOS_ERR_BUFFER_TOO_SMALL);
    NSOS_REQUIRE(((uintptr_t)packed & 15) == 0, NSOS_ERR_MISALIGNED);
    for (int l = 0; l < NSOS_NET_DEPTH; ++l) NSOS_REQUIRE(T_->pts_w[l] && T_->pts_b[l], NSOS_ERR_NULL_POINTER);
    NSOS_REQUIRE(T_->alpha_w && T_->alpha_b && T_->feature_w && T_->feature_b && T_->views_w && T_->views_b &&
                     T_->rgb_w && T_->rgb_b, NSOS_ERR_NULL_POINTER);
    if (sem_mode) NSOS_REQUIRE(T_->sem0_w && T_->sem0_b && T_->sem2_w && T_->sem2_b, NSOS_ERR_NULL_POINTER);

    X3PackParams P = {};
    int n = 0;
    auto add = [&](const float* w, const float* bias, int in_dim, int col, int kind, int i0, int ni) {
        P.ch[n++] = X3Chunk{w, bias, in_dim, col, kind, i0, ni};
    };
    auto hidden8 = [&](const float* w, const float* b, int in_dim, int col) {
        for (int c = 0; c < 8; ++c) add(w, b, in_dim, col, kHid8, 17 * c, 17);
    };
    auto hidden4 = [&](const float* w, const float* b, int in_dim, int col) {
        for (int c = 0; c < 4; ++c) add(w, b, in_dim, col, kHid4, 17 * c, 17);
    };
    auto enc8 = [&](const float* w, const float* pad_bias, int in_dim) {
        for (int c = 0; c < 2; ++c) add(w, pad_bias, in_dim, 0, kEnc8, 16 * c, 16);
    };
    const int X = NSOS_XYZ_DIM, W = NSOS_NET_WIDTH;
    enc8(T_->pts_w[0], T_->pts_b[0], X);                    // bias in the pad slot
    for (int l = 1; l <= 4; ++l) hidden8(T_->pts_w[l], T_->pts_b[l], W, 0);
    hidden8(T_->pts_w[5], T_->pts_b[5], X + W, X);          // skip layer: h part (with its bias items) ...
    enc8(T_->pts_w[5], nullptr, X + W);                     // ... then the x63 part
    hidden8(T_->pts_w[6], T_->pts_b[6], W, 0);
    hidden8(T_->pts_w[7], T_->pts_b[7], W, 0);
    if (sem_mode) {
        const int in_dim = sem_mode == NSOS_SEM_COORD ? W + X : W;
        hidden4(T_->sem0_w, T_->sem0_b, in_dim, 0);
        if (sem_mode == NSOS_SEM_COORD) add(T_->sem0_w, nullptr, in_dim, W, kEnc4, 0, 16);
    }
    hidden8(T_->feature_w, T_->feature_b, W, 0);
    hidden4(T_->views_w, T_->views_b, W + NSOS_DIR_DIM, 0);
    add(T_->views_w, nullptr, W + NSOS_DIR_DIM, W, kDir4, 0, 8);
    NSOS_REQUIRE(n == x3_chunks(sem_mode), NSOS_ERR_UNSUPPORTED);
    P.n_chunks = n;
    P.alpha_w = T_->alpha_w; P.alpha_b = T_->alpha_b;
    P.rgb_w = T_->rgb_w; P.rgb_b = T_->rgb_b;
    P.sem2_w = sem_mode ? T_->sem2_w : nullptr;
    P.sem2_b = sem_mode ? T_->sem2_b : nullptr;
    P.aux = static_cast<unsigned*>(packed);
    P.chunks = reinterpret_cast<unsigned short*>(P.aux + kAuxWords);
    const long long total = (long long)n * (kSlotBytes / 2);
    hipLaunchKernelGGL(x3_pack_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream, P);
    const int32_t rc = nsos_launch_status();
    if (rc != NSOS_OK) return rc;
    return nsos::x316::pack(T_, sem_mode, static_cast<unsigned char*>(packed) + x3_first_stream_bytes(sem_mode), (hipStream_t)stream);
}

namespace {
int32_t forward_x3(const void* packed, int32_t sem_mode, const float* rays_o, const float* rays_d, const float* viewdirs,
                   const float* z_vals, int64_t n_rays, int32_t n_samples, float* raw, float* sem_in, float* sem_hid,
                   float* acts, int save, void* stream, unsigned long long* prof = nullptr, unsigned* masks = nullptr) {
    if (n_rays == 0) return NSOS_OK;
    NSOS_REQUIRE(packed && rays_o && rays_d && viewdirs && z_vals && raw, NSOS_ERR_NULL_POINTER);
    NSOS_REQUIRE(n_rays > 0 && n_samples >= 1, NSOS_ERR_BAD_SHAPE);
    NSOS_REQUIRE(n_rays < (1ll << 31), NSOS_ERR_UNSUPPORTED);
    NSOS_REQUIRE(sem_mode >= 0 && sem_mode <= 2, NSOS_ERR_UNSUPPORTED);
    NSOS_REQUIRE(((uintptr_t)packed & 15) == 0 && ((uintptr_t)raw & 15) == 0, NSOS_ERR_MISALIGNED);
    if (save == 1) {
        NSOS_REQUIRE(sem_mode != 0, NSOS_ERR_UNSUPPORTED);
        NSOS_REQUIRE(sem_in && sem_hid, NSOS_ERR_NULL_POINTER);
        NSOS_REQUIRE(((uintptr_t)sem_in & 15) == 0 && ((uintptr_t)sem_hid & 15) == 0, NSOS_ERR_MISALIGNED);
    }
    if (save >= 2) {
        NSOS_REQUIRE(acts && masks, NSOS_ERR_NULL_POINTER);
        NSOS_REQUIRE(((uintptr_t)acts & 15) == 0 && ((uintptr_t)masks & 15) == 0, NSOS_ERR_MISALIGNED);
    }
    const long long n_pts = (long long)n_rays * n_samples;
    NSOS_REQUIRE((n_pts + kTilePts - 1) / kTilePts < (1ll << 31), NSOS_ERR_UNSUPPORTED);
    if (save == 0 && x3_kernel() == 2)          // inference: the 16x16x32 kernel on its own stream behind the first one
        return nsos::x316::launch(static_cast<const unsigned char*>(packed) + x3_first_stream_bytes(sem_mode), sem_mode, rays_o, rays_d, viewdirs,
                                  z_vals, n_pts, n_samples, raw, prof, (hipStream_t)stream);
    X3Params p = {};
    p.aux = static_cast<const unsigned*>(packed);
    p.chunks = reinterpret_cast<const unsigned char*>(p.aux + kAuxWords);
    p.rays_o = rays_o; p.rays_d = rays_d; p.viewdirs = viewdirs; p.z_vals = z_vals;
    p.raw = raw; p.n_pts = n_pts; p.n_samples = n_samples;
    p.n_tiles = (int)((n_pts + kTilePts - 1) / kTilePts);
    p.sem_in = sem_in; p.sem_hid = sem_hid; p.acts = acts; p.prof = prof; p.masks = masks;
    const hipStream_t st = (hipStream_t)stream;
    if (save == 1) return sem_mode == 1 ? launch_x3<1, 1>(p, st) : launch_x3<2, 1>(p, st);
    if (save == 2) return sem_mode == 0 ? launch_x3<0, 2>(p, st) : (sem_mode == 1 ? launch_x3<1, 2>(p, st) : launch_x3<2, 2>(p, st));
    if (save == 3) return sem_mode == 0 ? launch_x3<0, 3>(p, st) : (sem_mode == 1 ? launch_x3<1, 3>(p, st) : launch_x3<2, 3>(p, st));
    switch (sem_mode) {
        case 0: return launch_x3<0>(p, st);
        case 1: return launch_x3<1>(p, st);
        default: return launch_x3<2>(p, st);
    }
}
}  // namespace

extern "C" int32_t nsos_mlp_forward_rays_x3(const void* packed, int32_t sem_mode, const float* rays_o, const float* rays_d,
                                            const float* viewdirs, const float* z_vals, int64_t n_rays, int32_t n_samples,
                                            float* raw, void* stream) {
    return forward_x3(packed, sem_mode, rays_o, rays_d, viewdirs, z_vals, n_rays, n_samples, raw, nullptr, nullptr, nullptr, 0, stream);
}

extern "C" int32_t nsos_mlp_forward_rays_save_x3(const void* packed, int32_t sem_mode, const float* rays_o, const float* rays_d,
                                                 const float* viewdirs, const float* z_vals, int64_t n_rays, int32_t n_samples,
                                                 float* raw, float* sem_in, float* sem_hid, void* stream) {
    return forward_x3(packed, sem_mode, rays_o, rays_d, viewdirs, z_vals, n_rays, n_samples, raw, sem_in, sem_hid, nullptr, 1, stream);
}

extern "C" int32_t nsos_mlp_forward_rays_save_all_x3(const void* packed, int32_t sem_mode, const float* rays_o, const float* rays_d,
                                                     const float* viewdirs, const float* z_vals, int64_t n_rays, int32_t n_samples,
                                                     float* raw, float* acts, void* relu_masks, void* stream) {
    return forward_x3(packed, sem_mode, rays_o, rays_d, viewdirs, z_vals, n_rays, n_samples, raw, nullptr, nullptr, acts, 2, stream,
                      nullptr, static_cast<unsigned*>(relu_masks));
}

// the same with `acts` [P, NSOS_ACTS_DIM] in 16-bit floats: the hi parts of the split activations as the MFMAs consumed them (the
// 128-wide hidden layers rounded to nearest even): 5.3 KB per point instead of 10.6
extern "C" int32_t nsos_mlp_forward_rays_save_all16_x3(const void* packed, int32_t sem_mode, const float* rays_o, const float* rays_d,
                                                       const float* viewdirs, const float* z_vals, int64_t n_rays, int32_t n_samples,
                                                       float* raw, void* acts_f16, void* relu_masks, void* stream) {
    return forward_x3(packed, sem_mode, rays_o, rays_d, viewdirs, z_vals, n_rays, n_samples, raw, nullptr, nullptr,
                      static_cast<float*>(acts_f16), 3, stream, nullptr, static_cast<unsigned*>(relu_masks));
}

extern "C" size_t nsos_mlp_relu_masks_bytes_x3(int64_t n_pts) {   // 8 trunk layers x 256 threads x 16 B per 128-point tile
    return n_pts <= 0 ? 0 : (size_t)((n_pts + kTilePts - 1) / kTilePts) * 8 * 256 * 16;
}

extern "C" int32_t nsos_mlp_profile_rays_x3(const void* packed, int32_t sem_mode, const float* rays_o, const float* rays_d,
                                            const float* viewdirs, const float* z_vals, int64_t n_rays, int32_t n_samples,
                                            float* raw, uint64_t* stamps, void* stream) {
    if (n_rays == 0) return NSOS_OK;
    NSOS_REQUIRE(stamps, NSOS_ERR_NULL_POINTER);
    return forward_x3(packed, sem_mode, rays_o, rays_d, viewdirs, z_vals, n_rays, n_samples, raw, nullptr, nullptr, nullptr, 0, stream,
                      reinterpret_cast<unsigned long long*>(stamps));
}
