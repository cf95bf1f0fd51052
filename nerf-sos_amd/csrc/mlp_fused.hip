// K2: positional encoding + the whole NeRF-SOS MLP, fused, on gfx950's exact-fp32 matrix cores.
//
// Replaces PositionEncoder.forward x2 (models/embedder.py:34-48), the encoder join and point-chunk loop
// (models/nerf_mlp.py:179-215) and MLP.forward (models/nerf_mlp.py:67-100).
//
// Mapping (DESIGN.md "K2"):
//   * MFMA-bound (593k-634k MAC per point against 16-28 B of HBM I/O): everything is organised to keep
//     v_mfma_f32_32x32x2_f32 issuing back to back (64 cycles each, 64 FLOP/clk/SIMD = the fp32 peak).
//   * A workgroup = 4 waves = one wave per SIMD (all 512 registers per lane); wave w owns 32 points; a
//     workgroup tile = 128 points.  Products are computed TRANSPOSED: D[out feature i][point j] =
//     sum_k W[i][k] * h[k][j]: weights are the MFMA A operand, activations the B operand.  With that
//     orientation the accumulator layout of layer L (lane (j, hi) holds features 32t + (r&3) + 8(r>>2) + 4hi
//     of point j) IS the B-operand layout of layer L+1 (lane (j, hi) supplies k = hi of a k-pair), so
//     activations never leave the register file: no LDS round trip, no shuffles.  Accumulators Z live in
//     the AGPR half of the file; after each layer ONE batched pass H = max(Z, 0) (v_accvgpr_read + v_max_f32
//     per element) moves them to 128 VGPRs that the next layer's MFMAs name directly as srcB.
//     Why batched: measured on this chip (scripts/ubench/mfma_gap.hip), a wave's own VALU instructions do
//     NOT hide under its MFMAs -- an isolated VALU op between two MFMAs costs +12 cycles, a run of n costs
//     ~8+4n -- while SALU, ds_read and s_waitcnt are free.  So chunks contain no VALU work at all.
//     The price is a fixed, non-monotone k order of every contraction (0,4,1,5,2,6,3,7 per 8 features);
//     since the fp32 MFMA is bitwise an fmaf chain, the oracle simply follows the same order.
//   * Bias enters as a leading k-step with A = bias (k=0) / 0 (k=1), B = 1.0 and C = 0: fma(bias,1,0) = bias
//     exactly, so the chain is bitwise "acc = bias; acc = fma(w, x, acc) ...", without bias loads, without
//     accumulator initialisation, for 8 extra MFMAs per layer.
//   * The weights (2.6-2.8 MiB per net incl. padding, L2-resident) are pre-packed by nsos_mlp_pack into the
//     exact order the MFMAs consume them and streamed through LDS in 36 KiB slots by direct global->LDS DMA,
//     three slots deep, ONE s_barrier per chunk (~8200 cycles).  Each lane fetches the A operands of 4 MFMAs
//     with one conflict-free ds_read_b128 through a hand-counted register ring that never drains.
//   * Small heads (sigma 256->1, rgb 128->3, semantics 128->2) would waste 31/32 of an MFMA tile, so they
//     run on the vector ALU from the same registers (two half-wave partial sums + one cross-lane add).
//   * Persistent grid (one workgroup per CU); the chunk stream is cyclic so the DMA of the next tile's
//     first chunks overlaps the current tile's tail.
// Compiled with -ffp-contract=off (x = o + d*z must stay a separately rounded multiply and add).
#include "mlp_common.h"

using namespace nsos;

namespace {

constexpr int kSlotGroups = 36;                   // LDS slot / packed-stream stride per chunk (34 used at most)
constexpr int kSlotFloats = kSlotGroups * kGroupFloats;  // 36 KiB
constexpr int kDmaPieces = kSlotGroups / 4;       // 1 KiB pieces per wave per chunk
constexpr int kTilePts = 128;

// aux stream (vector-ALU head weights), offsets in floats.  256-wide vectors are stored in the accumulator
// layout [hi][tile 0..7][reg 0..15], 128-wide ones as [hi][tile 0..3][reg 0..15].
constexpr int kAuxAlphaW = 0;      // 256
constexpr int kAuxRgbW = 256;      // 3 x 128
constexpr int kAuxSem2W = 640;     // 2 x 128
constexpr int kAuxScalars = 896;   // alpha_b, rgb_b[3], sem2_b[2], 0, 0
constexpr int kAuxFloats = 1024;

// Segment kinds of the packed stream.  "8"/"4" = output tiles; Hid = 256 hidden features in accumulator
// order; Enc = the 63 encoded-xyz features (+1 pad) in natural order; Dir = the 27 encoded-direction features.
enum SegKind { kHid8 = 0, kEnc8 = 1, kHid4 = 2, kEnc4 = 3, kDir4 = 4 };

__host__ __device__ constexpr int chunks_per_net(int sem) {
    // L0 enc(2) + L1-4 (4x8) + L5 hid(8)+enc(2) + L6,L7 (2x8) + [sem0 hid(4) (+enc 1)] + feature(8) + views hid(4)+dir(1)
    return 2 + 32 + 10 + 16 + (sem ? 4 + (sem == 2 ? 1 : 0) : 0) + 8 + 5;
}

constexpr int kProfSlots = 64;
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
    unsigned long long* prof;  // diagnostics (nsos_mlp_profile_rays): per-wave shader-clock stamps, or NULL
    float* sem_in;   // SAVE: [P,320] = [h7 (256) | x63 (63) | 1.0]  inputs of semantic_linear.0 (+ ones column for the bias grad)
    float* sem_hid;  // SAVE: [P,128] = relu(semantic_linear.0(...))   inputs of semantic_linear.2
    float* acts;     // SAVE == 2 (full backward): [P, NSOS_ACTS_DIM] every layer's activations, see nerf_sos_hip.h
};

// ------------------------------------------------------------------------------------------ device helpers
__device__ __forceinline__ f32x16 mfma(float a, float b, f32x16 c) {
    return __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, c, 0, 0, 0);
}

// A-operand pipeline geometry of this kernel (machinery + rationale: mlp_common.h): one group = 4 MFMAs
// (256 cycles); ring of 3; next chunk's first groups pre-read during groups NG-6..NG-4; three 36 KiB LDS
// slots (chunk c consumed, c+1 resident, c+2 in flight); barrier before group 4, vmcnt(0) at it.
constexpr int kRing = 3, kPre = 6, kMidGroup = 4;
template <int G, int J, class S>
__device__ __forceinline__ void dma_after_mfma(S&& side) {  // piece i rides behind the i-th MFMA after the barrier
    dma_slot<(G - kMidGroup) * 4 + J, kDmaPieces>(side);
}

// One chunk of an 8-tile layer: 16 k-steps x 8 tiles (+ a leading bias k-step if BIAS).
//   group g (after the bias groups) = (k-step g>>1, tile quad g&1);  b: this chunk's 16 B operands, VGPRs.
template <bool BIAS, class S, class B, class T>
__device__ __forceinline__ void chunk8(f32x16 (&acc)[8], f32x4 (&ring)[kRing], const ChunkCtx ctx, const f32x16& b,
                                       S&& side, B&& mid, T&& tail) {
    constexpr int NB = BIAS ? 2 : 0;
    a_pipeline<32 + NB, kRing, kPre, kMidGroup>(ring, ctx, [&](auto ic, const f32x4& a) {
        constexpr int g = decltype(ic)::value;
        static_for<0, 4>([&](auto jc) {
            constexpr int j = decltype(jc)::value;
            if constexpr (g < NB) {  // acc = bias * 1 + 0
                const f32x16 zero = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
                acc[4 * g + j] = mfma(a[j], 1.0f, zero);
            } else {
                constexpr int gr = g - NB, q = gr & 1;
                acc[4 * q + j] = mfma(a[j], b[gr >> 1], acc[4 * q + j]);
            }
            dma_after_mfma<g, j>(side);
        });
    }, mid, tail);
}

// One chunk of a 4-tile layer: NKS k-steps x 4 tiles (+ a leading bias k-step if BIAS); group = k-step.
// b0 covers k-steps 0..15, b1 16..31.
template <int NKS, bool BIAS, class S, class B, class T>
__device__ __forceinline__ void chunk4(f32x16 (&acc)[4], f32x4 (&ring)[kRing], const ChunkCtx ctx, const f32x16& b0,
                                       const f32x16& b1, S&& side, B&& mid, T&& tail) {
    constexpr int NB = BIAS ? 1 : 0;
    a_pipeline<NKS + NB, kRing, kPre, kMidGroup>(ring, ctx, [&](auto ic, const f32x4& a) {
        constexpr int g = decltype(ic)::value;
        static_for<0, 4>([&](auto jc) {
            constexpr int j = decltype(jc)::value;
            if constexpr (g < NB) {
                const f32x16 zero = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
                acc[j] = mfma(a[j], 1.0f, zero);
            } else {
                constexpr int ks = g - NB;
                acc[j] = mfma(a[j], ks < 16 ? b0[ks & 15] : b1[ks & 15], acc[j]);
            }
            dma_after_mfma<g, j>(side);
        });
    }, mid, tail);
}

// H = max(Z, floor) for one layer, as ONE batch of v_max_f32 (gfx950 has no packed f32 max).  floor = 0 is
// the ReLU; -inf copies (feature_linear has no activation).  The asm keeps hipcc from CSE-ing / re-spreading
// the pass (and from adding a NaN-canonicalising second v_max per element), and "=v" pins H into arch VGPRs,
// where MFMAs can name it as srcB.  (v_max(0, NaN) = 0 where torch.relu propagates NaN: out of scope.)
__device__ __forceinline__ void activate(f32x16 (&H)[8], const f32x16 (&Z)[8], float floor) {
#pragma unroll
    for (int t = 0; t < 8; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            float out;
            asm volatile("v_max_f32 %0, %1, %2" : "=v"(out) : "v"(Z[t][r]), "v"(floor));
            H[t][r] = out;
        }
}

// vector-ALU head: this half-wave's partial fmaf chain over its NT*16 features (oracle: dot_halves)
template <int NT, bool RELU>
__device__ __forceinline__ float head_partial(const f32x16 (&h)[NT], const float* w_lane, float init) {
    float part = init;
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const f32x4 w = *reinterpret_cast<const f32x4*>(w_lane + t * 16 + q * 4);
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const float x = h[t][q * 4 + j];
                part = __fmaf_rn(w[j], RELU ? fmaxf(x, 0.0f) : x, part);
            }
        }
    return part;
}

// encoded feature idx lives in half-wave idx & 1 (k-step s of lane (j, hi) = feature 2s+hi)
struct ParityHalf {
    __host__ __device__ static constexpr int of(int idx) { return idx & 1; }
};
template <int L, int S0, int N>
__device__ __forceinline__ void enc_fill(f32x16& out, const Enc<L, ParityHalf>& e, const float (&x)[3], int hi) {
    const float lo_v = e.template feature<2 * S0, 0>(x), hi_v = e.template feature<2 * S0 + 1, 1>(x);
    out[S0 & 15] = hi ? hi_v : lo_v;
    if constexpr ((S0 & 15) + 1 < N) enc_fill<L, S0 + 1, N>(out, e, x, hi);
}

// ------------------------------------------------------------------------------------------ the kernel
// SAVE: training-mode variants that additionally store what a backward pass needs --
//   1: the semantic head's operands only (frozen backbone, K5);  2: every layer's activations (full backward, K7).
template <int SEM, bool RAYS, int SAVE = 0>
__global__ __launch_bounds__(256, 1) void mlp_fused_kernel(const MlpParams P) {
    extern __shared__ __attribute__((aligned(16))) float lds[];  // 3 x 36 KiB weight slots
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int pj = lane & 31, hi = lane >> 5;
    constexpr int NCH = chunks_per_net(SEM);
    constexpr int C = SEM ? 6 : 4;

    // ---- weight stream (see "A-operand pipeline" above).  State, all advanced by register rotation in tail():
    //   c0/c1/c2 : this lane's LDS byte address of the slot holding chunk cur / cur+1 / cur+2 (being filled)
    //   d2..     : wave-uniform LDS byte address (slot base + wave*1 KiB) of the same slots, for the DMA
    //   src2     : global address of chunk cur+2 in the packed stream (wraps at the end of the net)
    const unsigned lds_base = (unsigned)(size_t)(__attribute__((address_space(3))) float*)lds;
    const int wave_s = __builtin_amdgcn_readfirstlane(wave);        // provably wave-uniform copy
    const unsigned voff = (unsigned)(wave * kGroupFloats + lane * 4) * 4u;  // lane's byte offset in a 4 KiB piece row
    auto lane_addr = [&](int b) { return lds_base + (unsigned)(b * kSlotFloats + lane * 4) * 4u; };
    auto wave_addr = [&](int b) { return lds_base + (unsigned)(b * kSlotFloats + wave_s * kGroupFloats) * 4u; };
    unsigned c0 = lane_addr(0), c1 = lane_addr(1), c2 = lane_addr(2);
    unsigned d0 = wave_addr(0), d1 = wave_addr(1), d2 = wave_addr(2);
    const float* const src_end = P.chunks + (size_t)NCH * kSlotFloats;
    const float* src2 = P.chunks + (size_t)(2 % NCH) * kSlotFloats;
    // one 1 KiB piece: this wave copies pieces i*4+wave of a chunk, i < kDmaPieces.  Inline asm so that the
    // whole issue is a few SALU + 1 VMEM instruction with wave-uniform (SGPR) addressing (through the builtin
    // hipcc spends ~10 VALU/readfirstlane instructions per piece, ~110 cycles, of which only 64 hide under an
    // MFMA).  M0 (LDS destination) is saved/restored inside the statement.
    auto dma_piece = [&](const float* src_chunk, unsigned dst_wave, int i) {
        const float* src = src_chunk + i * 4 * kGroupFloats;  // uniform
        const unsigned dst = dst_wave + (unsigned)i * 4096u;   // uniform
        dma_1k(src, dst, voff);
    };
    auto side = [&](int i) { dma_piece(src2, d2, i); };
    auto mid = [&]() {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
    };
    auto tail = [&]() {  // chunk cur -> cur+1
        const unsigned tc = c0, td = d0;
        c0 = c1; c1 = c2; c2 = tc;
        d0 = d1; d1 = d2; d2 = td;
        src2 += kSlotFloats;
        if (src2 == src_end) src2 = P.chunks;
    };
    auto stage_begin = [&]() { return ChunkCtx{c0, c1}; };
    // prologue: chunks 0 and 1 resident, ring = first groups of chunk 0
    f32x4 ring[kRing];
#pragma unroll
    for (int i = 0; i < kDmaPieces; ++i) dma_piece(P.chunks, d0, i);
#pragma unroll
    for (int i = 0; i < kDmaPieces; ++i) dma_piece(P.chunks + (size_t)(1 % NCH) * kSlotFloats, d1, i);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // the asm DMAs are invisible to hipcc's own counting
    __syncthreads();
    static_for<0, kRing>([&](auto ic) { lds_read_a<decltype(ic)::value * 1024>(ring[decltype(ic)::value], lane_addr(0)); });
    lgkm_wait<0>();
    NSOS_PIN();

    for (int tile = blockIdx.x; tile < P.n_tiles; tile += gridDim.x) {
        int stamp_k = 0;
        auto stamp = [&]() {  // diagnostics only: one s_memtime per phase of the first tile of blocks 0..3
            if (P.prof && tile == (int)blockIdx.x && blockIdx.x < 4) {
                const unsigned long long t = __builtin_readcyclecounter();
                if (lane == 0 && stamp_k < kProfSlots) P.prof[(blockIdx.x * 4 + wave) * kProfSlots + stamp_k] = t;
            }
            ++stamp_k;
        };
        stamp();  // 0: tile start
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
        // NaN / Inf in a point's inputs must come out as NaN (the reference propagates them silently through sin, the
        // Linears and relu; v_max(0, NaN) = 0 here would quietly launder them at the first ReLU): 0 if all six are finite
        const float poison = ((x[0] - x[0]) + (x[1] - x[1])) + ((x[2] - x[2]) + (dv[0] - dv[0])) + ((dv[1] - dv[1]) + (dv[2] - dv[2]));
        // ---- encodings, directly in B-operand form: k-step s of lane (j,hi) = feature 2s+hi
        f32x16 ex[2], ed;
        {
            Enc<NSOS_XYZ_FREQS, ParityHalf> e;
            e.evaluate(x, hi);
            enc_fill<NSOS_XYZ_FREQS, 0, 16>(ex[0], e, x, hi);
            enc_fill<NSOS_XYZ_FREQS, 16, 16>(ex[1], e, x, hi);
        }
        {
            Enc<NSOS_DIR_FREQS, ParityHalf> e;
            e.evaluate(dv, hi);
            enc_fill<NSOS_DIR_FREQS, 0, 16>(ed, e, dv, hi);
        }

        // Z: accumulators of the layer being computed; H: previous layer's activations (VGPRs, MFMA srcB)
        f32x16 Z[8], H[8];
        auto save_layer = [&](int col0) {   // SAVE == 2: H (4 consecutive features per (tile, register quad)) -> acts
            if (valid) {
                float* row = P.acts + gp * NSOS_ACTS_DIM + col0;
#pragma unroll
                for (int t = 0; t < 8; ++t)
#pragma unroll
                    for (int q = 0; q < 4; ++q)
                        *reinterpret_cast<f32x4*>(row + 32 * t + 8 * q + 4 * hi) =
                            f32x4{H[t][4 * q], H[t][4 * q + 1], H[t][4 * q + 2], H[t][4 * q + 3]};
            }
        };
        if constexpr (SAVE == 2) {
            if (valid) {
                float* row = P.acts + gp * NSOS_ACTS_DIM;
#pragma unroll
                for (int k = 0; k < 32; ++k)   // x63 feature 2k+hi (slot 63 = 1.0), dir27 feature 2k+hi (k < 16)
                    row[NSOS_ACTS_X + 2 * k + hi] = (k == 31 && hi == 1) ? 1.0f : ex[k >> 4][k & 15];
#pragma unroll
                for (int k = 0; k < 16; ++k) row[NSOS_ACTS_D + 2 * k + hi] = (2 * k + hi < NSOS_DIR_DIM) ? ed[k] : 0.0f;
            }
        }
        float sigma = 0.0f, sem_out[2] = {0.0f, 0.0f};
        stamp();  // 1: inputs loaded + encoded

        // the 63 encoded-xyz features enter (layer 0; skip connection into layer 5): 2 chunks
        auto enc_part = [&](auto with_bias) {
            constexpr bool WB = decltype(with_bias)::value;
            chunk8<WB>(Z, ring, stage_begin(), ex[0], side, mid, tail);
            chunk8<false>(Z, ring, stage_begin(), ex[1], side, mid, tail);
        };
        enc_part(std::true_type{});  // pts_linears.0
        const float relu = 0.0f, pass = -__builtin_inff();
        activate(H, Z, relu);
        if constexpr (SAVE == 2) save_layer(0);
        stamp();  // 2
        // pts_linears.1..7 (l = 1..7) and feature_linear (l = 8): Z = bias + W * H
#pragma unroll 1
        for (int l = 1; l <= 8; ++l) {
            static_for<0, 8>([&](auto cc) {
                constexpr int c = decltype(cc)::value;
                chunk8<(c == 0)>(Z, ring, stage_begin(), H[c], side, mid, tail);
            });
            if (l == 5) enc_part(std::false_type{});  // skip: layer 5 = bias + W_h h4 + W_x x63
            activate(H, Z, l < 8 ? relu : pass);          // feature_linear's output is used without activation
            if constexpr (SAVE == 2) save_layer(256 * l);  // l == 8: the (linear) feature vector at NSOS_ACTS_FEAT
            stamp();                                      // 2 + l
            if (l == 7) {
                // H = h7.  sigma head (models/nerf_mlp.py:77), on the vector ALU
                const float pa = head_partial<8, false>(H, P.aux + kAuxAlphaW + hi * 128, hi ? 0.0f : P.aux[kAuxScalars]);
                sigma = both_halves(pa);
                if constexpr (SEM != 0) {  // semantic head (models/nerf_mlp.py:79-80), cat([h, x63]): h first
                    f32x16 sacc[4];
                    static_for<0, 4>([&](auto cc) {
                        constexpr int c = decltype(cc)::value;
                        chunk4<32, (c == 0)>(sacc, ring, stage_begin(), H[2 * c], H[2 * c + 1], side, mid, tail);
                    });
                    if constexpr (SEM == 2) chunk4<32, false>(sacc, ring, stage_begin(), ex[0], ex[1], side, mid, tail);
                    if constexpr (SAVE == 2) {
                        if (valid) {
                            float* hrow = P.acts + gp * NSOS_ACTS_DIM + NSOS_ACTS_SEM;
#pragma unroll
                            for (int t = 0; t < 4; ++t)
#pragma unroll
                                for (int q = 0; q < 4; ++q)
                                    *reinterpret_cast<f32x4*>(hrow + 32 * t + 8 * q + 4 * hi) =
                                        f32x4{fmaxf(sacc[t][4 * q], 0.0f), fmaxf(sacc[t][4 * q + 1], 0.0f),
                                              fmaxf(sacc[t][4 * q + 2], 0.0f), fmaxf(sacc[t][4 * q + 3], 0.0f)};
                        }
                    }
                    if constexpr (SAVE == 1) {
                        if (valid) {
                            float* row = P.sem_in + gp * 320;   // H = relu(h7): 4 consecutive features per (tile, reg quad)
                            float* hrow = P.sem_hid + gp * 128;
#pragma unroll
                            for (int t = 0; t < 8; ++t)
#pragma unroll
                                for (int q = 0; q < 4; ++q)
                                    *reinterpret_cast<f32x4*>(row + 32 * t + 8 * q + 4 * hi) =
                                        f32x4{H[t][4 * q], H[t][4 * q + 1], H[t][4 * q + 2], H[t][4 * q + 3]};
#pragma unroll
                            for (int k = 0; k < 32; ++k)  // x63 feature 2k+hi; the pad slot (feature 63) carries the 1.0
                                row[256 + 2 * k + hi] = (k == 31 && hi == 1) ? 1.0f : ex[k >> 4][k & 15];
#pragma unroll
                            for (int t = 0; t < 4; ++t)
#pragma unroll
                                for (int q = 0; q < 4; ++q)
                                    *reinterpret_cast<f32x4*>(hrow + 32 * t + 8 * q + 4 * hi) =
                                        f32x4{fmaxf(sacc[t][4 * q], 0.0f), fmaxf(sacc[t][4 * q + 1], 0.0f),
                                              fmaxf(sacc[t][4 * q + 2], 0.0f), fmaxf(sacc[t][4 * q + 3], 0.0f)};
                        }
                    }
#pragma unroll
                    for (int o = 0; o < 2; ++o) {
                        const float ps = head_partial<4, true>(sacc, P.aux + kAuxSem2W + o * 128 + hi * 64,
                                                               hi ? 0.0f : P.aux[kAuxScalars + 4 + o]);
                        sem_out[o] = both_halves(ps);
                    }
                }
            }
        }
        // ---- view branch (models/nerf_mlp.py:87-92): cat([feature, dir27]) -> 128 -> rgb.  H = feature.
        f32x16 vacc[4];
        static_for<0, 4>([&](auto cc) {
            constexpr int c = decltype(cc)::value;
            chunk4<32, (c == 0)>(vacc, ring, stage_begin(), H[2 * c], H[2 * c + 1], side, mid, tail);
        });
        chunk4<16, false>(vacc, ring, stage_begin(), ed, ed, side, mid, tail);
        if constexpr (SAVE == 2) {
            if (valid) {
                float* vrow = P.acts + gp * NSOS_ACTS_DIM + NSOS_ACTS_VIEWS;
#pragma unroll
                for (int t = 0; t < 4; ++t)
#pragma unroll
                    for (int q = 0; q < 4; ++q)
                        *reinterpret_cast<f32x4*>(vrow + 32 * t + 8 * q + 4 * hi) =
                            f32x4{fmaxf(vacc[t][4 * q], 0.0f), fmaxf(vacc[t][4 * q + 1], 0.0f),
                                  fmaxf(vacc[t][4 * q + 2], 0.0f), fmaxf(vacc[t][4 * q + 3], 0.0f)};
            }
        }
        float rgb[3];
#pragma unroll
        for (int o = 0; o < 3; ++o) {
            const float pr = head_partial<4, true>(vacc, P.aux + kAuxRgbW + o * 128 + hi * 64,
                                                   hi ? 0.0f : P.aux[kAuxScalars + 1 + o]);
            rgb[o] = both_halves(pr);
        }
        stamp();  // 11: view branch + rgb head done
        // ---- raw = [r, g, b, sigma, (sem0, sem1)]   (models/nerf_mlp.py:93-96)
        if (poison != poison) {
            const float qnan = __builtin_nanf("");
            rgb[0] = rgb[1] = rgb[2] = sigma = sem_out[0] = sem_out[1] = qnan;
        }
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
        stamp();  // 12: outputs stored
    }
    // the cyclic prefetch leaves DMAs in flight: drain them before the LDS allocation is released
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
}

// ------------------------------------------------------------------------------------------ packing
struct PackSeg {
    const float* w;
    const float* bias;  // leading bias k-step of the segment's first chunk, or NULL
    int in_dim;
    int col_base;
    int kind;
    int n_chunks;
};
struct PackParams {
    PackSeg seg[16];
    int n_seg;
    int n_chunks;
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
        if (a < kAuxRgbW) v = P.alpha_w[feat256(a - kAuxAlphaW)];
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
    if (gid >= (long long)P.n_chunks * kSlotFloats) return;
    int chunk = (int)(gid / kSlotFloats);
    int s = 0;
    while (chunk >= P.seg[s].n_chunks) { chunk -= P.seg[s].n_chunks; ++s; }
    const PackSeg sg = P.seg[s];
    const int within = (int)(gid % kSlotFloats);
    int g = within >> 8;
    const int lane = (within >> 2) & 63, j = within & 3;
    const int hi = lane >> 5, i = lane & 31;
    const bool eight = (sg.kind == kHid8 || sg.kind == kEnc8);
    float v = 0.0f;
    const int nb = (sg.bias && chunk == 0) ? (eight ? 2 : 1) : 0;  // leading bias groups of this chunk
    if (g < nb) {
        const int t = eight ? 4 * g + j : j;
        v = hi == 0 ? sg.bias[32 * t + i] : 0.0f;  // k = 0: bias * 1.0 ;  k = 1: 0 * 1.0
    } else {
        g -= nb;
        const int groups = sg.kind == kDir4 ? 16 : 32;
        if (g < groups) {
            int t, ks, f = -1;
            if (eight) { ks = chunk * 16 + (g >> 1); t = 4 * (g & 1) + j; }
            else { ks = chunk * 32 + g; t = j; }
            switch (sg.kind) {
                case kHid8: case kHid4: f = acc_feature(ks >> 4, ks & 15, hi); break;
                case kEnc8: case kEnc4: f = 2 * ks + hi; if (f >= NSOS_XYZ_DIM) f = -1; break;
                case kDir4: f = 2 * ks + hi; if (f >= NSOS_DIR_DIM) f = -1; break;
            }
            if (f >= 0) v = sg.w[(long long)(32 * t + i) * sg.in_dim + sg.col_base + f];
        }
    }
    P.chunks[gid] = v;
}

int num_cus() { return nsos_device_cus(); }

constexpr int kLdsBytes = 3 * kSlotFloats * 4;

template <int SEM, bool RAYS, int SAVE = 0>
int32_t launch_mlp(const MlpParams& p, hipStream_t stream) {
    static NsosPerDeviceFlag configured_on;
    bool& configured = configured_on.here();
    if (!configured) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&mlp_fused_kernel<SEM, RAYS, SAVE>),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, kLdsBytes);
        if (e != hipSuccess) return (int32_t)e;
        configured = true;
    }
    const int grid = p.n_tiles < num_cus() ? p.n_tiles : num_cus();
    hipLaunchKernelGGL((mlp_fused_kernel<SEM, RAYS, SAVE>), dim3(grid), dim3(256), kLdsBytes, stream, p);
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

int32_t fill_ray_params(MlpParams& p, const void* packed, const float* rays_o, const float* rays_d,
                        const float* viewdirs, const float* z_vals, int64_t n_rays, int32_t n_samples, float* raw) {
    NSOS_REQUIRE(packed && rays_o && rays_d && viewdirs && z_vals && raw, NSOS_ERR_NULL_POINTER);
    NSOS_REQUIRE(n_rays >= 0 && n_samples >= 1, NSOS_ERR_BAD_SHAPE);
    NSOS_REQUIRE(((uintptr_t)packed & 15) == 0 && ((uintptr_t)raw & 15) == 0, NSOS_ERR_MISALIGNED);
    const long long n_pts = (long long)n_rays * n_samples;
    NSOS_REQUIRE((n_pts + kTilePts - 1) / kTilePts < (1ll << 31), NSOS_ERR_UNSUPPORTED);
    p = MlpParams{};
    p.aux = static_cast<const float*>(packed);
    p.chunks = p.aux + kAuxFloats;
    p.rays_o = rays_o; p.rays_d = rays_d; p.viewdirs = viewdirs; p.z_vals = z_vals;
    p.raw = raw; p.n_pts = n_pts; p.n_samples = n_samples;
    p.n_tiles = (int)((n_pts + kTilePts - 1) / kTilePts);
    return NSOS_OK;
}

}  // namespace

// ------------------------------------------------------------------------------------------ C ABI
extern "C" size_t nsos_mlp_packed_bytes(int32_t sem_mode) {
    if (sem_mode < 0 || sem_mode > 2) return 0;
    return sizeof(float) * ((size_t)kAuxFloats + (size_t)chunks_per_net(sem_mode) * kSlotFloats);
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
    auto add = [&](const float* w, const float* bias, int in_dim, int col, int kind, int nch) {
        P.seg[n++] = PackSeg{w, bias, in_dim, col, kind, nch};
    };
    const int X = NSOS_XYZ_DIM, W = NSOS_NET_WIDTH;
    add(T->pts_w[0], T->pts_b[0], X, 0, kEnc8, 2);
    for (int l = 1; l <= 4; ++l) add(T->pts_w[l], T->pts_b[l], W, 0, kHid8, 8);
    // skip layer: input = cat([x63, h]) (models/nerf_mlp.py:73-74); the kernel contracts h first, then x63
    add(T->pts_w[5], T->pts_b[5], X + W, X, kHid8, 8);
    add(T->pts_w[5], nullptr, X + W, 0, kEnc8, 2);
    add(T->pts_w[6], T->pts_b[6], W, 0, kHid8, 8);
    add(T->pts_w[7], T->pts_b[7], W, 0, kHid8, 8);
    if (sem_mode) {
        const int in_dim = sem_mode == NSOS_SEM_COORD ? W + X : W;  // cat([h, x63]) (models/nerf_mlp.py:79)
        add(T->sem0_w, T->sem0_b, in_dim, 0, kHid4, 4);
        if (sem_mode == NSOS_SEM_COORD) add(T->sem0_w, nullptr, in_dim, W, kEnc4, 1);
    }
    add(T->feature_w, T->feature_b, W, 0, kHid8, 8);
    add(T->views_w, T->views_b, W + NSOS_DIR_DIM, 0, kHid4, 4);  // cat([feature, dir27]) (models/nerf_mlp.py:87)
    add(T->views_w, nullptr, W + NSOS_DIR_DIM, W, kDir4, 1);
    P.n_seg = n;
    P.n_chunks = chunks_per_net(sem_mode);
    P.alpha_w = T->alpha_w; P.alpha_b = T->alpha_b;
    P.rgb_w = T->rgb_w; P.rgb_b = T->rgb_b;
    P.sem2_w = sem_mode ? T->sem2_w : nullptr;
    P.sem2_b = sem_mode ? T->sem2_b : nullptr;
    P.aux = static_cast<float*>(packed);
    P.chunks = P.aux + kAuxFloats;
    const long long total = (long long)P.n_chunks * kSlotFloats;
    hipLaunchKernelGGL(pack_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream, P);
    return nsos_launch_status();
}

extern "C" int32_t nsos_mlp_forward_rays(const void* packed, int32_t sem_mode, const float* rays_o,
                                         const float* rays_d, const float* viewdirs, const float* z_vals,
                                         int64_t n_rays, int32_t n_samples, float* raw, void* stream) {
    if (n_rays == 0) return NSOS_OK;  // empty batch: nothing to launch (empty tensors have NULL data pointers)
    MlpParams p;
    const int32_t rc = fill_ray_params(p, packed, rays_o, rays_d, viewdirs, z_vals, n_rays, n_samples, raw);
    if (rc != NSOS_OK) return rc;
    return dispatch_mlp<true>(sem_mode, p, (hipStream_t)stream);
}

extern "C" int32_t nsos_mlp_forward_rays_save(const void* packed, int32_t sem_mode, const float* rays_o,
                                              const float* rays_d, const float* viewdirs, const float* z_vals,
                                              int64_t n_rays, int32_t n_samples, float* raw, float* sem_in,
                                              float* sem_hid, void* stream) {
    if (n_rays == 0) return NSOS_OK;
    NSOS_REQUIRE(sem_in && sem_hid, NSOS_ERR_NULL_POINTER);
    NSOS_REQUIRE(sem_mode == NSOS_SEM_PLAIN || sem_mode == NSOS_SEM_COORD, NSOS_ERR_UNSUPPORTED);
    NSOS_REQUIRE(((uintptr_t)sem_in & 15) == 0 && ((uintptr_t)sem_hid & 15) == 0, NSOS_ERR_MISALIGNED);
    MlpParams p;
    const int32_t rc = fill_ray_params(p, packed, rays_o, rays_d, viewdirs, z_vals, n_rays, n_samples, raw);
    if (rc != NSOS_OK) return rc;
    p.sem_in = sem_in;
    p.sem_hid = sem_hid;
    return sem_mode == NSOS_SEM_COORD ? launch_mlp<2, true, true>(p, (hipStream_t)stream)
                                      : launch_mlp<1, true, true>(p, (hipStream_t)stream);
}

extern "C" int32_t nsos_mlp_forward_rays_save_all(const void* packed, int32_t sem_mode, const float* rays_o,
                                                  const float* rays_d, const float* viewdirs, const float* z_vals,
                                                  int64_t n_rays, int32_t n_samples, float* raw, float* acts, void* stream) {
    if (n_rays == 0) return NSOS_OK;
    NSOS_REQUIRE(acts, NSOS_ERR_NULL_POINTER);
    NSOS_REQUIRE(((uintptr_t)acts & 15) == 0, NSOS_ERR_MISALIGNED);
    MlpParams p;
    const int32_t rc = fill_ray_params(p, packed, rays_o, rays_d, viewdirs, z_vals, n_rays, n_samples, raw);
    if (rc != NSOS_OK) return rc;
    p.acts = acts;
    switch (sem_mode) {
        case NSOS_SEM_NONE: return launch_mlp<0, true, 2>(p, (hipStream_t)stream);
        case NSOS_SEM_PLAIN: return launch_mlp<1, true, 2>(p, (hipStream_t)stream);
        case NSOS_SEM_COORD: return launch_mlp<2, true, 2>(p, (hipStream_t)stream);
    }
    return NSOS_ERR_UNSUPPORTED;
}

extern "C" int32_t nsos_mlp_profile_rays(const void* packed, int32_t sem_mode, const float* rays_o,
                                         const float* rays_d, const float* viewdirs, const float* z_vals,
                                         int64_t n_rays, int32_t n_samples, float* raw, uint64_t* stamps,
                                         void* stream) {
    if (n_rays == 0) return NSOS_OK;
    NSOS_REQUIRE(stamps, NSOS_ERR_NULL_POINTER);
    MlpParams p;
    const int32_t rc = fill_ray_params(p, packed, rays_o, rays_d, viewdirs, z_vals, n_rays, n_samples, raw);
    if (rc != NSOS_OK) return rc;
    p.prof = reinterpret_cast<unsigned long long*>(stamps);
    return dispatch_mlp<true>(sem_mode, p, (hipStream_t)stream);
}

extern "C" int32_t nsos_mlp_forward_points(const void* packed, int32_t sem_mode, const float* pts,
                                           const float* dirs, int64_t n_pts, float* raw, void* stream) {
    if (n_pts == 0) return NSOS_OK;  // empty batch: nothing to launch (empty tensors have NULL data pointers)
    NSOS_REQUIRE(packed && pts && dirs && raw, NSOS_ERR_NULL_POINTER);
    NSOS_REQUIRE(n_pts >= 0, NSOS_ERR_BAD_SHAPE);
    NSOS_REQUIRE(((uintptr_t)packed & 15) == 0 && ((uintptr_t)raw & 15) == 0, NSOS_ERR_MISALIGNED);
    NSOS_REQUIRE((n_pts + kTilePts - 1) / kTilePts < (1ll << 31), NSOS_ERR_UNSUPPORTED);
    MlpParams p = {};
    p.aux = static_cast<const float*>(packed);
    p.chunks = p.aux + kAuxFloats;
    p.pts = pts; p.dirs = dirs; p.raw = raw; p.n_pts = n_pts; p.n_samples = 1;
    p.n_tiles = (int)((n_pts + kTilePts - 1) / kTilePts);
    return dispatch_mlp<false>(sem_mode, p, (hipStream_t)stream);
}
