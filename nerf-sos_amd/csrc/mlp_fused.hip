// K2: positional encoding + the whole NeRF-SOS MLP, fused, on gfx950's exact-fp32 matrix cores.
//
// Replaces PositionEncoder.forward x2 (models/embedder.py:34-48), the encoder join and point-chunk loop
// (models/nerf_mlp.py:179-215) and MLP.forward (models/nerf_mlp.py:67-100).
//
// Mapping (DESIGN.md "K2"):
//   * MFMA-bound (593k-634k MAC per point against 16-28 B of HBM I/O): everything is organised to keep
//     v_mfma_f32_32x32x2_f32 issuing back to back (64 cycles each, 64 FLOP/clk/SIMD = the fp32 peak).
//   * A workgroup = 4 waves = one wave per SIMD; wave w owns 32 points; a workgroup tile = 128 points.
//     The products are computed TRANSPOSED: D[out feature i][point j] = sum_k W[i][k] * h[k][j], i.e.
//     the weights are the MFMA A operand and the activations the B operand.  With that orientation the
//     accumulator layout of layer L (lane (j, hi) holds features 32t + (r&3) + 8(r>>2) + 4hi of point j)
//     IS the B-operand layout of layer L+1 (lane (j, hi) supplies k = hi of a k-pair), so activations
//     never leave the register file: no LDS round trip, no shuffles, 128 live registers per lane.
//     The price is a fixed, non-monotone k order of every contraction (0,4,1,5,2,6,3,7 per 8 features);
//     since the fp32 MFMA is bitwise an fmaf chain, the oracle simply follows the same order.
//   * The weights (2.3-2.4 MiB per net, L2-resident) are pre-packed by nsos_mlp_pack into the exact
//     order the MFMAs consume them and streamed through LDS in 32 KiB chunks (128 A-operands each) by
//     direct global->LDS DMA, double buffered: one s_barrier per 128 MFMAs (8192 cycles).
//     Each lane fetches the A operands of 4 MFMAs with one conflict-free ds_read_b128.
//   * Small heads (sigma 256->1, rgb 128->3, semantics 128->2) would waste 31/32 of an MFMA tile, so they
//     run on the vector ALU from the same registers (two half-wave partial sums + one cross-lane add).
//   * Persistent grid (one workgroup per CU); the chunk stream is cyclic so the prefetch of the next
//     tile's first chunk overlaps the current tile's tail.
// Compiled with -ffp-contract=off (x = o + d*z must stay a separately rounded multiply and add).
#include "common.h"

#include <type_traits>

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x2 __attribute__((ext_vector_type(2)));

namespace {

constexpr int kChunkFloats = 8192;  // 32 KiB = 128 MFMA A-operands (64 lanes x 4 B each)
constexpr int kTilePts = 128;

// aux stream (biases + vector-ALU head weights), offsets in floats.  256-wide vectors are stored in the
// accumulator layout [hi][tile 0..7][reg 0..15], 128-wide ones as [hi][tile 0..3][reg 0..15].
constexpr int kAuxBias = 0;         // 9 x 256: pts_linears.0..7, feature_linear
constexpr int kAuxViewsB = 2304;    // 128
constexpr int kAuxSem0B = 2432;     // 128
constexpr int kAuxAlphaW = 2560;    // 256
constexpr int kAuxRgbW = 2816;      // 3 x 128
constexpr int kAuxSem2W = 3200;     // 2 x 128
constexpr int kAuxScalars = 3456;   // alpha_b, rgb_b[3], sem2_b[2], 0, 0
constexpr int kAuxFloats = 3584;

enum SegKind { kHid8 = 0, kEnc8 = 1, kHid4 = 2, kEnc4 = 3, kDir4 = 4 };

__host__ __device__ constexpr int chunks_per_net(int sem) {
    // L0 enc(2) + L1-4 (4x8) + L5 enc(2)+hid(8) + L6,L7 (2x8) + [sem0 hid(4) (+enc 1)] + feature(8) + views hid(4)+dir(1)
    return 2 + 32 + 10 + 16 + (sem ? 4 + (sem == 2 ? 1 : 0) : 0) + 8 + 5;
}

// feature index held by (tile t, reg r, half hi) in the 32x32 accumulator layout
__host__ __device__ constexpr int acc_feature(int t, int r, int hi) { return 32 * t + (r & 3) + 8 * (r >> 2) + 4 * hi; }

struct MlpParams {
    const float* aux;
    const float* chunks;
    const float* rays_o;
    const float* rays_d;
    const float* viewdirs;
    const float* z_vals;
    const float* pts;
    const float* dirs;
    float* raw;
    long long n_pts;
    int n_samples;
    int n_tiles;
};

// ------------------------------------------------------------------------------------------ device helpers
__device__ __forceinline__ f32x16 mfma(float a, float b, f32x16 c) {
    return __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, c, 0, 0, 0);
}

// ---- A-operand pipeline -------------------------------------------------------------------------
// One ds_read_b128 feeds a group of 4 MFMAs (256 cycles of matrix pipe).  hipcc, left alone, issues each
// read right before its use and waits lgkmcnt(0) (measured: 131 TF of the 157 TF peak), and with
// source-level prefetching it still waits for the YOUNGEST read.  So the reads are inline asm, invisible to
// the compiler's wait-count pass, with hand-counted waits (cdna_hip_programming.md section 5.7, form iii):
// a ring of kRing slots; slot g%kRing is re-loaded for group g+kRing right after group g's MFMAs were
// issued, so while group g computes, the reads of groups g+1 .. g+kRing-1 are in flight (LDS returns in
// order: "lgkmcnt(kRing-1)" == "group g has landed").  Every chunk starts right after a __syncthreads()
// (lgkmcnt(0)), and no other LGKM-counted op (ds_*, s_load) is issued inside a chunk, so the counts hold.
constexpr int kRing = 3;
#define NSOS_PIN() __builtin_amdgcn_sched_barrier(0)

template <int OFF_BYTES>
__device__ __forceinline__ void lds_read_a(f32x4& dst, unsigned lds_addr) {
    asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(dst) : "v"(lds_addr), "i"(OFF_BYTES) : "memory");
}
template <int N>
__device__ __forceinline__ void lgkm_wait() {
    asm volatile("s_waitcnt lgkmcnt(%0)" ::"i"(N) : "memory");
}
template <int I, int N, class F>
__device__ __forceinline__ void static_for(F&& f) {
    if constexpr (I < N) {
        f(std::integral_constant<int, I>{});
        static_for<I + 1, N>(f);
    }
}

// NG groups of 4 MFMAs; group g's A operands are the f32x4 at byte offset g*1024 from wl (which already
// includes lane*16).  mfma4(g, a) issues the 4 MFMAs of group g.
template <int NG, class M>
__device__ __forceinline__ void a_pipeline(unsigned wl, M&& mfma4) {
    f32x4 ring[kRing];
    static_for<0, (kRing < NG ? kRing : NG)>([&](auto ic) {
        constexpr int g = decltype(ic)::value;
        lds_read_a<g * 1024>(ring[g], wl);
    });
    static_for<0, NG>([&](auto ic) {
        constexpr int g = decltype(ic)::value;
        constexpr int in_flight_after = (NG - 1 - g) < (kRing - 1) ? (NG - 1 - g) : (kRing - 1);
        lgkm_wait<in_flight_after>();
        NSOS_PIN();
        mfma4(ic, ring[g % kRing]);
        NSOS_PIN();
        if constexpr (g + kRing < NG) lds_read_a<(g + kRing) * 1024>(ring[g % kRing], wl);
    });
}

// The DMA of the NEXT chunk (8 x 1 KiB pieces per wave) is issued from inside the current chunk: piece i
// right after the i-th MFMA of groups 1-2, so its address arithmetic and issue slots are covered by the
// 64-cycle MFMAs instead of idling the matrix pipe after every barrier.  side(i) issues piece i.
template <int G, int J, class S>
__device__ __forceinline__ void dma_slot(S&& side) {
    if constexpr (G == 1 || G == 2) {
        NSOS_PIN();
        side((G - 1) * 4 + J);
        NSOS_PIN();
    }
}

// 16 k-steps x 8 output tiles = 128 MFMAs; group g = (k-step g>>1, tile quad g&1).
template <class S>
__device__ __forceinline__ void chunk8(f32x16 (&acc)[8], unsigned wl, const f32x16& b, S&& side) {
    a_pipeline<32>(wl, [&](auto ic, const f32x4& a) {
        constexpr int g = decltype(ic)::value, q = g & 1;
        static_for<0, 4>([&](auto jc) {
            constexpr int j = decltype(jc)::value;
            acc[4 * q + j] = mfma(a[j], b[g >> 1], acc[4 * q + j]);
            dma_slot<g, j>(side);
        });
    });
}

// NKS k-steps x 4 output tiles; group g = k-step g.  b0 covers k-steps 0..15, b1 16..31.
template <int NKS, class S>
__device__ __forceinline__ void chunk4(f32x16 (&acc)[4], unsigned wl, const f32x16& b0, const f32x16& b1, S&& side) {
    a_pipeline<NKS>(wl, [&](auto ic, const f32x4& a) {
        constexpr int g = decltype(ic)::value;
        const float b = g < 16 ? b0[g & 15] : b1[g & 15];
        static_for<0, 4>([&](auto jc) {
            constexpr int j = decltype(jc)::value;
            acc[j] = mfma(a[j], b, acc[j]);
            dma_slot<g, j>(side);
        });
    });
}

template <int NT>
__device__ __forceinline__ void load_bias(f32x16 (&acc)[NT], const float* aux_lane /* base + hi*NT*16 */) {
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const f32x4 v = *reinterpret_cast<const f32x4*>(aux_lane + t * 16 + q * 4);
#pragma unroll
            for (int j = 0; j < 4; ++j) acc[t][q * 4 + j] = v[j];
        }
}

// vector-ALU head: this half-wave's partial fmaf chain over its NT*16 features (oracle: dot_halves)
template <int NT>
__device__ __forceinline__ float head_partial(const f32x16 (&h)[NT], const float* w_lane, float init) {
    float part = init;
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const f32x4 w = *reinterpret_cast<const f32x4*>(w_lane + t * 16 + q * 4);
#pragma unroll
            for (int j = 0; j < 4; ++j) part = __fmaf_rn(w[j], h[t][q * 4 + j], part);
        }
    return part;
}

__device__ __forceinline__ float both_halves(float part) { return part + __shfl_xor(part, 32, NSOS_WAVE); }

// positional-encoding feature idx of a 3-vector with L octaves (models/embedder.py:34-48):
//   [x y z | sin(2^0 x..z) cos(2^0 x..z) | sin(2^1 ..) ...];   idx >= 3+6L is zero padding.
struct EncSlot {
    int coord;   // 0..2
    int octave;  // 0..L-1, or -1 = raw coordinate, -2 = pad
    bool is_cos;
};
__host__ __device__ constexpr EncSlot enc_slot(int idx, int L) {
    if (idx < 3) return {idx, -1, false};
    if (idx >= 3 + 6 * L) return {0, -2, false};
    const int k = (idx - 3) / 6, j = (idx - 3) % 6;
    return {j % 3, k, j >= 3};
}

// Feature pair (2s, 2s+1) of an encoding: the lo half-wave needs 2s, the hi half-wave 2s+1.  One
// sincos per lane: select the argument by half, evaluate, select sin or cos by half.
template <int L, int S0>
__device__ __forceinline__ float enc_pair(const float (&x)[3], int hi) {
    constexpr EncSlot e0 = enc_slot(2 * S0, L), e1 = enc_slot(2 * S0 + 1, L);
    float v0 = 0.0f, v1 = 0.0f;
    constexpr bool t0 = e0.octave >= 0, t1 = e1.octave >= 0;
    if constexpr (e0.octave == -1) v0 = x[e0.coord];
    if constexpr (e1.octave == -1) v1 = x[e1.coord];
    if constexpr (t0 || t1) {
        const float a0 = t0 ? x[e0.coord] * (float)(1 << (t0 ? e0.octave : 0)) : 0.0f;
        const float a1 = t1 ? x[e1.coord] * (float)(1 << (t1 ? e1.octave : 0)) : 0.0f;
        const float arg = hi ? a1 : a0;
        float sn, cs;
        sincosf(arg, &sn, &cs);
        if constexpr (t0) v0 = e0.is_cos ? cs : sn;
        if constexpr (t1) v1 = e1.is_cos ? cs : sn;
    }
    return hi ? v1 : v0;
}

template <int L, int S0, int N>
struct EncFill {
    __device__ __forceinline__ static void run(f32x16& out, const float (&x)[3], int hi) {
        out[S0 & 15] = enc_pair<L, S0>(x, hi);
        if constexpr ((S0 & 15) + 1 < N) EncFill<L, S0 + 1, N>::run(out, x, hi);
    }
};

// ------------------------------------------------------------------------------------------ the kernel
template <int SEM, bool RAYS>
__global__ __launch_bounds__(256, 1) void mlp_fused_kernel(const MlpParams P) {
    extern __shared__ __attribute__((aligned(16))) float lds[];  // 2 x 32 KiB weight buffers
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int pj = lane & 31, hi = lane >> 5;
    constexpr int NCH = chunks_per_net(SEM);
    constexpr int C = SEM ? 6 : 4;

    // ---- weight stream: chunk `cur` is consumed from buffer `par`; chunk cur+1 is DMA'd into par^1 meanwhile
    int cur = 0, par = 0;
    const float* const src_lane = P.chunks + wave * 256 + lane * 4;  // + chunk*8192 + i*1024 floats
    auto dma_piece = [&](int chunk, int buf, int i) {  // 32 pieces of 1 KiB per chunk; this wave copies 8
        const float* src = src_lane + (size_t)chunk * kChunkFloats + i * 1024;
        float* dst = lds + buf * kChunkFloats + (i * 4 + wave) * 256;  // wave-uniform; HW adds lane*16 B
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                         (__attribute__((address_space(3))) void*)dst, 16, 0, 0);
    };
    const unsigned lds_base = (unsigned)(size_t)(__attribute__((address_space(3))) float*)lds;
    // stage protocol: the barrier proves (a) chunk `cur` has landed (every wave drained its own DMA before
    // arriving) and (b) every wave is done reading buffer par^1, which this stage's DMA pieces overwrite.
    auto stage_begin = [&]() -> unsigned {  // LDS byte address of this lane's first A operand
        __syncthreads();
        return lds_base + (unsigned)(par * kChunkFloats + lane * 4) * 4u;
    };
    auto side = [&](int i) { dma_piece(cur + 1 == NCH ? 0 : cur + 1, par ^ 1, i); };
    auto stage_end = [&]() {
        cur = (cur + 1 == NCH) ? 0 : cur + 1;
        par ^= 1;
    };
#pragma unroll
    for (int i = 0; i < 8; ++i) dma_piece(0, 0, i);

    for (int tile = blockIdx.x; tile < P.n_tiles; tile += gridDim.x) {
        // ---- this lane's point (both half-waves of a column hold the same point)
        const long long gp = (long long)tile * kTilePts + wave * 32 + pj;
        const bool valid = gp < P.n_pts;
        const long long gc = valid ? gp : P.n_pts - 1;
        float x[3], dv[3];
        if constexpr (RAYS) {
            const long long ray = gc / P.n_samples;
            const float z = P.z_vals[gc];
#pragma unroll
            for (int c = 0; c < 3; ++c) {
                const float m = P.rays_d[3 * ray + c] * z;  // models/sampler.py:70,166 (mul, then add)
                x[c] = P.rays_o[3 * ray + c] + m;
                dv[c] = P.viewdirs[3 * ray + c];
            }
        } else {
#pragma unroll
            for (int c = 0; c < 3; ++c) {
                x[c] = P.pts[3 * gc + c];
                dv[c] = P.dirs[3 * gc + c];
            }
        }
        // ---- encodings, directly in B-operand form: k-step s of lane (j,hi) = feature 2s+hi
        f32x16 ex[2], ed;
        EncFill<NSOS_XYZ_FREQS, 0, 16>::run(ex[0], x, hi);
        EncFill<NSOS_XYZ_FREQS, 16, 16>::run(ex[1], x, hi);
        EncFill<NSOS_DIR_FREQS, 0, 16>::run(ed, dv, hi);

        f32x16 acc[8], hin[8];
        float sigma = 0.0f, sem_out[2] = {0.0f, 0.0f};

        // ---- trunk: pts_linears.0..7 (l = 0..7) and feature_linear (l = 8)
#pragma unroll 1
        for (int l = 0; l <= 8; ++l) {
            load_bias<8>(acc, P.aux + kAuxBias + l * 256 + hi * 128);
            if (l == 0 || l == 5) {  // encoded xyz enters (layer 0; skip connection, input-first cat)
#pragma unroll
                for (int c = 0; c < 2; ++c) {
                    const unsigned wl = stage_begin();
                    chunk8(acc, wl, ex[c], side);
                    stage_end();
                }
            }
            if (l != 0) {
#pragma unroll
                for (int c = 0; c < 8; ++c) {
                    const unsigned wl = stage_begin();
                    chunk8(acc, wl, hin[c], side);
                    stage_end();
                }
            }
            if (l < 8) {
#pragma unroll
                for (int t = 0; t < 8; ++t)
#pragma unroll
                    for (int r = 0; r < 16; ++r) hin[t][r] = fmaxf(acc[t][r], 0.0f);
            } else {
#pragma unroll
                for (int t = 0; t < 8; ++t) hin[t] = acc[t];  // feature_linear has no activation
            }
            if (l == 7) {
                // sigma head (models/nerf_mlp.py:77), on the vector ALU
                const float pa = head_partial<8>(hin, P.aux + kAuxAlphaW + hi * 128, hi ? 0.0f : P.aux[kAuxScalars]);
                sigma = both_halves(pa);
                if constexpr (SEM != 0) {  // semantic head (models/nerf_mlp.py:79-80), cat([h, x63]): h first
                    f32x16 sacc[4];
                    load_bias<4>(sacc, P.aux + kAuxSem0B + hi * 64);
#pragma unroll
                    for (int c = 0; c < 4; ++c) {
                        const unsigned wl = stage_begin();
                        chunk4<32>(sacc, wl, hin[2 * c], hin[2 * c + 1], side);
                        stage_end();
                    }
                    if constexpr (SEM == 2) {
                        const unsigned wl = stage_begin();
                        chunk4<32>(sacc, wl, ex[0], ex[1], side);
                        stage_end();
                    }
#pragma unroll
                    for (int t = 0; t < 4; ++t)
#pragma unroll
                        for (int r = 0; r < 16; ++r) sacc[t][r] = fmaxf(sacc[t][r], 0.0f);
#pragma unroll
                    for (int o = 0; o < 2; ++o) {
                        const float ps = head_partial<4>(sacc, P.aux + kAuxSem2W + o * 128 + hi * 64,
                                                         hi ? 0.0f : P.aux[kAuxScalars + 4 + o]);
                        sem_out[o] = both_halves(ps);
                    }
                }
            }
        }
        // ---- view branch (models/nerf_mlp.py:87-92): cat([feature, dir27]) -> 128 -> rgb
        f32x16 vacc[4];
        load_bias<4>(vacc, P.aux + kAuxViewsB + hi * 64);
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            const unsigned wl = stage_begin();
            chunk4<32>(vacc, wl, hin[2 * c], hin[2 * c + 1], side);
            stage_end();
        }
        {
            const unsigned wl = stage_begin();
            chunk4<16>(vacc, wl, ed, ed, side);
            stage_end();
        }
#pragma unroll
        for (int t = 0; t < 4; ++t)
#pragma unroll
            for (int r = 0; r < 16; ++r) vacc[t][r] = fmaxf(vacc[t][r], 0.0f);
        float rgb[3];
#pragma unroll
        for (int o = 0; o < 3; ++o) {
            const float pr = head_partial<4>(vacc, P.aux + kAuxRgbW + o * 128 + hi * 64,
                                             hi ? 0.0f : P.aux[kAuxScalars + 1 + o]);
            rgb[o] = both_halves(pr);
        }
        // ---- raw = [r, g, b, sigma, (sem0, sem1)]   (models/nerf_mlp.py:93-96)
        if (valid) {
            float* out = P.raw + gp * C;
            if constexpr (C == 4) {
                if (hi == 0) *reinterpret_cast<f32x4*>(out) = f32x4{rgb[0], rgb[1], rgb[2], sigma};
            } else {
                if (hi == 0) {
                    *reinterpret_cast<f32x2*>(out) = f32x2{rgb[0], rgb[1]};
                    *reinterpret_cast<f32x2*>(out + 2) = f32x2{rgb[2], sigma};
                } else {
                    *reinterpret_cast<f32x2*>(out + 4) = f32x2{sem_out[0], sem_out[1]};
                }
            }
        }
    }
    // the cyclic prefetch leaves one DMA in flight: drain it before the LDS allocation is released
    __syncthreads();
}

// ------------------------------------------------------------------------------------------ packing
struct PackSeg {
    const float* w;
    int in_dim;
    int col_base;
    int kind;
    int n_chunks;
};
struct PackParams {
    PackSeg seg[16];
    int n_seg;
    int n_chunks;
    const float* bias256[9];  // pts_linears.0..7 bias, feature bias
    const float* views_b;
    const float* sem0_b;
    const float* alpha_w;
    const float* alpha_b;
    const float* rgb_w;
    const float* rgb_b;
    const float* sem2_w;
    const float* sem2_b;
    float* aux;
    float* chunks;
};

__global__ __launch_bounds__(256) void pack_kernel(const PackParams P) {
    const long long gid = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (gid < kAuxFloats) {
        const int a = (int)gid;
        float v = 0.0f;
        auto feat256 = [](int rem) { return acc_feature((rem & 127) >> 4, rem & 15, rem >> 7); };
        auto feat128 = [](int rem) { return acc_feature((rem & 63) >> 4, rem & 15, rem >> 6); };
        if (a < kAuxViewsB) v = P.bias256[a >> 8][feat256(a & 255)];
        else if (a < kAuxSem0B) v = P.views_b[feat128(a - kAuxViewsB)];
        else if (a < kAuxAlphaW) v = P.sem0_b ? P.sem0_b[feat128(a - kAuxSem0B)] : 0.0f;
        else if (a < kAuxRgbW) v = P.alpha_w[feat256(a - kAuxAlphaW)];
        else if (a < kAuxSem2W) { const int rem = a - kAuxRgbW; v = P.rgb_w[(rem >> 7) * 128 + feat128(rem & 127)]; }
        else if (a < kAuxScalars) { const int rem = a - kAuxSem2W; v = P.sem2_w ? P.sem2_w[(rem >> 7) * 128 + feat128(rem & 127)] : 0.0f; }
        else {
            const int i = a - kAuxScalars;
            if (i == 0) v = P.alpha_b[0];
            else if (i < 4) v = P.rgb_b[i - 1];
            else if (i < 6) v = P.sem2_b ? P.sem2_b[i - 4] : 0.0f;
        }
        P.aux[a] = v;
    }
    if (gid >= (long long)P.n_chunks * kChunkFloats) return;
    int chunk = (int)(gid >> 13);
    int s = 0;
    while (chunk >= P.seg[s].n_chunks) { chunk -= P.seg[s].n_chunks; ++s; }
    const PackSeg sg = P.seg[s];
    const int within = (int)(gid & (kChunkFloats - 1));
    const int g = within >> 8, lane = (within >> 2) & 63, j = within & 3;
    const int hi = lane >> 5, i = lane & 31;
    int t, ks, f = -1;
    if (sg.kind == kHid8 || sg.kind == kEnc8) { ks = chunk * 16 + (g >> 1); t = 4 * (g & 1) + j; }
    else { ks = chunk * 32 + g; t = j; }
    switch (sg.kind) {
        case kHid8: case kHid4: f = acc_feature(ks >> 4, ks & 15, hi); break;
        case kEnc8: case kEnc4: f = 2 * ks + hi; if (f >= NSOS_XYZ_DIM) f = -1; break;
        case kDir4: f = (ks < 16) ? 2 * ks + hi : -1; if (f >= NSOS_DIR_DIM) f = -1; break;
    }
    const int out = 32 * t + i;
    P.chunks[gid] = (f >= 0) ? sg.w[(long long)out * sg.in_dim + sg.col_base + f] : 0.0f;
}

int g_num_cus = 0;
int num_cus() {
    if (g_num_cus == 0) {
        int dev = 0, n = 0;
        if (hipGetDevice(&dev) == hipSuccess &&
            hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess && n > 0)
            g_num_cus = n;
        else
            return 256;
    }
    return g_num_cus;
}

template <int SEM, bool RAYS>
int32_t launch_mlp(const MlpParams& p, hipStream_t stream) {
    static bool configured = false;
    if (!configured) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&mlp_fused_kernel<SEM, RAYS>),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, 2 * kChunkFloats * 4);
        if (e != hipSuccess) return (int32_t)e;
        configured = true;
    }
    const int grid = p.n_tiles < num_cus() ? p.n_tiles : num_cus();
    hipLaunchKernelGGL((mlp_fused_kernel<SEM, RAYS>), dim3(grid), dim3(256), 2 * kChunkFloats * 4, stream, p);
    return nsos_launch_status();
}

template <bool RAYS>
int32_t dispatch_mlp(int sem_mode, const MlpParams& p, hipStream_t stream) {
    switch (sem_mode) {
        case NSOS_SEM_NONE: return launch_mlp<0, RAYS>(p, stream);
        case NSOS_SEM_PLAIN: return launch_mlp<1, RAYS>(p, stream);
        case NSOS_SEM_COORD: return launch_mlp<2, RAYS>(p, stream);
    }
    return NSOS_ERR_UNSUPPORTED;
}

}  // namespace

// ------------------------------------------------------------------------------------------ C ABI
extern "C" size_t nsos_mlp_packed_bytes(int32_t sem_mode) {
    if (sem_mode < 0 || sem_mode > 2) return 0;
    return sizeof(float) * ((size_t)kAuxFloats + (size_t)chunks_per_net(sem_mode) * kChunkFloats);
}

extern "C" int32_t nsos_mlp_pack(const nsos_mlp_tensors* T, int32_t sem_mode, void* packed, size_t packed_bytes,
                                 void* stream) {
    NSOS_REQUIRE(T && packed, NSOS_ERR_NULL_POINTER);
    NSOS_REQUIRE(sem_mode >= 0 && sem_mode <= 2, NSOS_ERR_UNSUPPORTED);
    NSOS_REQUIRE(packed_bytes >= nsos_mlp_packed_bytes(sem_mode), NSOS_ERR_BUFFER_TOO_SMALL);
    NSOS_REQUIRE(((uintptr_t)packed & 15) == 0, NSOS_ERR_MISALIGNED);
    for (int l = 0; l < NSOS_NET_DEPTH; ++l) NSOS_REQUIRE(T->pts_w[l] && T->pts_b[l], NSOS_ERR_NULL_POINTER);
    NSOS_REQUIRE(T->alpha_w && T->alpha_b && T->feature_w && T->feature_b && T->views_w && T->views_b &&
                     T->rgb_w && T->rgb_b, NSOS_ERR_NULL_POINTER);
    if (sem_mode) NSOS_REQUIRE(T->sem0_w && T->sem0_b && T->sem2_w && T->sem2_b, NSOS_ERR_NULL_POINTER);

    PackParams P = {};
    int n = 0;
    auto add = [&](const float* w, int in_dim, int col, int kind, int nch) { P.seg[n++] = PackSeg{w, in_dim, col, kind, nch}; };
    const int X = NSOS_XYZ_DIM, W = NSOS_NET_WIDTH;
    add(T->pts_w[0], X, 0, kEnc8, 2);
    for (int l = 1; l <= 4; ++l) add(T->pts_w[l], W, 0, kHid8, 8);
    add(T->pts_w[5], X + W, 0, kEnc8, 2);  // skip layer input = cat([x63, h]) (models/nerf_mlp.py:73-74)
    add(T->pts_w[5], X + W, X, kHid8, 8);
    add(T->pts_w[6], W, 0, kHid8, 8);
    add(T->pts_w[7], W, 0, kHid8, 8);
    if (sem_mode) {
        const int in_dim = sem_mode == NSOS_SEM_COORD ? W + X : W;  // cat([h, x63]) (models/nerf_mlp.py:79)
        add(T->sem0_w, in_dim, 0, kHid4, 4);
        if (sem_mode == NSOS_SEM_COORD) add(T->sem0_w, in_dim, W, kEnc4, 1);
    }
    add(T->feature_w, W, 0, kHid8, 8);
    add(T->views_w, W + NSOS_DIR_DIM, 0, kHid4, 4);  // cat([feature, dir27]) (models/nerf_mlp.py:87)
    add(T->views_w, W + NSOS_DIR_DIM, W, kDir4, 1);
    P.n_seg = n;
    P.n_chunks = chunks_per_net(sem_mode);
    for (int l = 0; l < 8; ++l) P.bias256[l] = T->pts_b[l];
    P.bias256[8] = T->feature_b;
    P.views_b = T->views_b;
    P.sem0_b = sem_mode ? T->sem0_b : nullptr;
    P.alpha_w = T->alpha_w; P.alpha_b = T->alpha_b;
    P.rgb_w = T->rgb_w; P.rgb_b = T->rgb_b;
    P.sem2_w = sem_mode ? T->sem2_w : nullptr;
    P.sem2_b = sem_mode ? T->sem2_b : nullptr;
    P.aux = static_cast<float*>(packed);
    P.chunks = P.aux + kAuxFloats;
    const long long total = (long long)P.n_chunks * kChunkFloats;
    hipLaunchKernelGGL(pack_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream, P);
    return nsos_launch_status();
}

extern "C" int32_t nsos_mlp_forward_rays(const void* packed, int32_t sem_mode, const float* rays_o,
                                         const float* rays_d, const float* viewdirs, const float* z_vals,
                                         int64_t n_rays, int32_t n_samples, float* raw, void* stream) {
    if (n_rays == 0) return NSOS_OK;  // empty batch: nothing to launch (empty tensors have NULL data pointers)
    NSOS_REQUIRE(packed && rays_o && rays_d && viewdirs && z_vals && raw, NSOS_ERR_NULL_POINTER);
    NSOS_REQUIRE(n_rays >= 0 && n_samples >= 1, NSOS_ERR_BAD_SHAPE);
    NSOS_REQUIRE(((uintptr_t)packed & 15) == 0 && ((uintptr_t)raw & 15) == 0, NSOS_ERR_MISALIGNED);
    if (n_rays == 0) return NSOS_OK;
    const long long n_pts = (long long)n_rays * n_samples;
    NSOS_REQUIRE((n_pts + kTilePts - 1) / kTilePts < (1ll << 31), NSOS_ERR_UNSUPPORTED);
    MlpParams p = {};
    p.aux = static_cast<const float*>(packed);
    p.chunks = p.aux + kAuxFloats;
    p.rays_o = rays_o; p.rays_d = rays_d; p.viewdirs = viewdirs; p.z_vals = z_vals;
    p.raw = raw; p.n_pts = n_pts; p.n_samples = n_samples;
    p.n_tiles = (int)((n_pts + kTilePts - 1) / kTilePts);
    return dispatch_mlp<true>(sem_mode, p, (hipStream_t)stream);
}

extern "C" int32_t nsos_mlp_forward_points(const void* packed, int32_t sem_mode, const float* pts,
                                           const float* dirs, int64_t n_pts, float* raw, void* stream) {
    if (n_pts == 0) return NSOS_OK;  // empty batch: nothing to launch (empty tensors have NULL data pointers)
    NSOS_REQUIRE(packed && pts && dirs && raw, NSOS_ERR_NULL_POINTER);
    NSOS_REQUIRE(n_pts >= 0, NSOS_ERR_BAD_SHAPE);
    NSOS_REQUIRE(((uintptr_t)packed & 15) == 0 && ((uintptr_t)raw & 15) == 0, NSOS_ERR_MISALIGNED);
    if (n_pts == 0) return NSOS_OK;
    NSOS_REQUIRE((n_pts + kTilePts - 1) / kTilePts < (1ll << 31), NSOS_ERR_UNSUPPORTED);
    MlpParams p = {};
    p.aux = static_cast<const float*>(packed);
    p.chunks = p.aux + kAuxFloats;
    p.pts = pts; p.dirs = dirs; p.raw = raw; p.n_pts = n_pts; p.n_samples = 1;
    p.n_tiles = (int)((n_pts + kTilePts - 1) / kTilePts);
    return dispatch_mlp<false>(sem_mode, p, (hipStream_t)stream);
}
