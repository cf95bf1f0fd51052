// K7-X3: the input-gradient chain of one NeRF MLP as ONE fused kernel on the 16-bit matrix pipe with split-fp16 operands.
// Replaces, for the full backward (every parameter trainable, models/nerf_mlp.py:67-100 under autograd), the nine library
// GEMMs  g_in = g_out @ W  and the ReLU-mask passes between them (nerf-sos_amd/backward.py): what autograd computes as
//     g_v    = (g_rgb @ W_rgb)            * (views > 0)                       [P,128]
//     g_feat =  g_v  @ W_views[:, :256]                                       [P,256]
//     g_hs   = (g_sem @ W_sem2)           * (sem_hidden > 0)                  [P,128]   (semantic head only)
//     g_z7   = (g_feat @ W_feature + g_sigma * w_alpha + g_hs @ W_sem0[:, :256]) * (h7 > 0)
//     g_z(l-1) = (g_z(l) @ W_l[:, h part]) * (h(l-1) > 0),   l = 7 .. 1
// is evaluated per 128-point tile with the weight stream / LDS ring / accumulator layout of mlp_x3.hip (transposed weights,
// no bias items) and written to gbuf [P, NSOS_GBUF_DIM] in the column map of the saved activations, ready for the weight-
// gradient kernel (nsos_wgrad).  Masks come from the activations nsos_mlp_forward_rays_save_all[_x3] stored.
// Gradients have no natural scale, fp16 has 5 exponent bits: the caller passes powers of two, scale[0..2] (device memory):
//   scale[0]  brings max |g_raw| over ALL channels to ~2^4: the trunk's gradients (gbuf columns 256 l) carry it;
//   scale[1]  >= 1, extra factor of the COLOUR branch: g_rgb -> g_v -> g_feat (gbuf blocks VIEWS and FEAT carry scale[0] scale[1]);
//   scale[2]  >= 1, extra factor of the SEMANTIC branch: g_sem -> g_hs (gbuf block SEM carries scale[0] scale[2]).
// Round 5: one common scale is not enough.  With the reference's own loss shapes (img2mse's mean over the batch next to per-logit
// gradients from the correlation losses) the colour channels of g_raw sit 3-4 decades under the largest one; scaled by scale[0]
// alone their hi parts are barely normal fp16 numbers and their lo parts subnormal: 11-12 significant bits, 2-5e-4 of scale in
// feature_linear's and views_linears' gradients on a TRAINED field (tests/test_gpu_trained.py; random-init goldens with N(0,1)
// upstream gradients on every map never showed it).  Each branch runs at its own scale and is brought back to the trunk's --
// an exact multiplication by a power of two -- where it enters the trunk product as the B operand (there its rounding is relative
// to a sum it contributes 1e-3 of).  The caller divides the weight gradients by the scale their gbuf block carries.
// Products: g_hi.W_hi + g_lo.W_hi + g_hi.(2^11 W_lo), fp32 accumulation, as in the forward kernel.
#include "x3_common.h"

typedef unsigned u32x2_b __attribute__((ext_vector_type(2)));
namespace {

// aux stream (fp32 words): head weights in accumulator layout
constexpr int kBAuxRgbW = 0;      // 3 x 128: [o][kg][t 0..3][r 0..15] -> rgb_linear.weight[o][acc_feature(t, r, kg)]
constexpr int kBAuxSem2W = 384;   // 2 x 128
constexpr int kBAuxAlphaW = 640;  // 256: [kg][t 0..7][r 0..15] -> alpha_linear.weight[0][acc_feature(t, r, kg)]
constexpr int kBAuxWords = 1024;

#ifndef NSOS_X3B_RING
#define NSOS_X3B_RING 16
#endif
constexpr int kMaskRing = NSOS_X3B_RING;   // ReLU-mask quads (16 B per lane) in flight ahead of the pass that uses them
static_assert(kMaskRing >= 4 && kMaskRing <= 16 && kDmaPieces + 32 <= 63, "vmcnt is a 6-bit counter");

__host__ __device__ constexpr int x3_bwd_chunks(int sem) { return 4 + 8 + (sem ? 4 : 0) + 56; }   // 16 items (32 A operands) each

struct X3BwdParams {
    const unsigned* aux;
    const unsigned char* chunks;
    const float* g_raw;    // [P, n_ch]
    const float* acts;     // [P, NSOS_ACTS_DIM]
    float* gbuf;           // [P, NSOS_GBUF_DIM]
    const float* scale;    // device memory, three powers of two: trunk scale, colour-branch factor, semantic-branch factor
    const unsigned* masks; // BITS: ReLU bit masks of the trunk layers from nsos_mlp_forward_rays_save_all_x3, [tile][layer][256][4]
    long long n_pts;
    int n_tiles;
};

// BITS: the trunk layers' ReLU masks come as bits (one 16-byte load per lane per layer) instead of as the fp32 activations
// A16 (with BITS): `acts` holds 16-bit floats (nsos_mlp_forward_rays_save_all16_x3); only the two 128-wide heads' masks are read from it
template <int SEM, bool BITS, bool A16 = false>
__global__ __launch_bounds__(256, 1) void mlp_x3_bwd_kernel(const X3BwdParams P) {
    static_assert(BITS || !A16, "16-bit activations come with the forward's bit masks");
    extern __shared__ __attribute__((aligned(16))) unsigned char lds[];  // 4 x 36 KiB weight slots + 4 KiB head weights
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int pj = lane & 31, kg = lane >> 5;
    constexpr int NCH = x3_bwd_chunks(SEM);
    constexpr int C = SEM ? 6 : 4;

    // ---- weight stream (as mlp_x3.hip): slots rotate (c0 = chunk cur, c1 = cur+1, c2 = cur+2, c3 = being filled with cur+3)
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
        asm volatile("s_waitcnt vmcnt(%0)" ::"i"(kDmaPieces) : "memory");   // counted: the newest chunk's pieces may still fly
        __builtin_amdgcn_s_barrier();
    };
    // The first two chunks after a pass: the mask loads of that pass's ring refill and the kMaskRing preloads of the next pass
    // were issued AFTER the pieces this barrier needs, and loads retire in order, so that many more operations may stay in
    // flight (stores are NOT counted: nothing here relies on their order relative to loads).  The loads are unconditional
    // (lanes past the end redo the last point), so the count is exact.
    auto mid_extra = [&](auto n_c) {
        asm volatile("s_waitcnt vmcnt(%0)" ::"i"(kDmaPieces + decltype(n_c)::value) : "memory");
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
    const float* const aux_l = reinterpret_cast<const float*>(lds + kSlots * kSlotBytes);
    *reinterpret_cast<u32x4*>(lds + kSlots * kSlotBytes + threadIdx.x * 16) = reinterpret_cast<const u32x4*>(P.aux)[threadIdx.x];
    const float scale = P.scale[0], s_rgb = P.scale[1], s_sem = P.scale[2];
    const float inv_rgb = 1.0f / s_rgb, inv_sem = 1.0f / s_sem;          // exact: powers of two
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    static_for<0, kRing>([&](auto ic) { lds_read_a<decltype(ic)::value * 1024>(ring[decltype(ic)::value], c0); });
    lgkm_wait<0>();
    NSOS_PIN();

#define IC(n) std::integral_constant<int, (n)> {}
    // One chunk = 16 items of a product with 8 output tiles; item I0 + i = (K-slice s = a / 8, tile t = a % 8).  32 A operands in
    // the skewed order hi_0, [hi_k, lo_{k-1}], lo_15 (mlp_x3.hip).  ZF: slice 0 starts the accumulation (C = 0).
    auto run_chunk = [&](auto i0_c, auto zf_c, auto relaxed_c, auto& Zm, auto& Zx, auto&& bh, auto&& bl) __attribute__((always_inline)) {
        constexpr int NI = 16, NG = 32, I0 = decltype(i0_c)::value;
        constexpr bool ZF = decltype(zf_c)::value != 0;
        constexpr int EXTRA = decltype(relaxed_c)::value;   // loads known to have been issued after the pieces this chunk's barrier needs
        auto mid_sel = [&]() { if constexpr (EXTRA > 0) mid_extra(std::integral_constant<int, EXTRA>{}); else mid(); };
        a_pipeline<NG, kRing, kPre, kMid>(ring, ctx(), [&](auto ic, const f32x4& a32) {
            constexpr int g = decltype(ic)::value;
            constexpr bool IS_HI = g == 0 || (g != NG - 1 && (g & 1));
            constexpr int item = g == 0 ? 0 : (g == NG - 1 ? NI - 1 : (IS_HI ? (g + 1) / 2 : (g - 2) / 2));
            constexpr int a = I0 + item, s = a / 8, t = a % 8;
            constexpr bool FIRST = ZF && s == 0;
            const u32x4 aop = __builtin_bit_cast(u32x4, a32);
            const f32x16 zero = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
            if constexpr (IS_HI) {
                Zm[t] = mfma16(aop, bh(IC(s)), FIRST ? zero : Zm[t]);
                Zm[t] = mfma16(aop, bl(IC(s)), Zm[t]);
            } else {
                Zx[t] = mfma16(aop, bh(IC(s)), FIRST ? zero : Zx[t]);
            }
            dma_slot<g - kMid, kDmaPieces>(side);
        }, mid_sel, tail);
#pragma unroll
        for (int t = 0; t < 8; ++t) asm volatile("" : "+a"(Zm[t]), "+a"(Zx[t]));   // see mlp_x3.hip: keeps LLVM from sinking the MFMAs
    };

    for (int tile = blockIdx.x; tile < P.n_tiles; tile += gridDim.x) {
        const long long gp = (long long)tile * kTilePts + wave * 32 + pj;
        const long long gc = gp < P.n_pts ? gp : P.n_pts - 1;
        const bool valid = gp < P.n_pts;
        const float* const arow = P.acts + gc * NSOS_ACTS_DIM;
        float* const grow = P.gbuf + gc * NSOS_GBUF_DIM;
        float gr[C];
#pragma unroll
        for (int c = 0; c < C; ++c) gr[c] = P.g_raw[gc * C + c] * scale * (c < 3 ? s_rgb : (c > 3 ? s_sem : 1.0f));

        f32x16 Zm[8], Zx[8];
        u32x4 Hh[16], Hl[16];
        auto h_h = [&](auto sc) { return Hh[decltype(sc)::value]; };
        auto h_l = [&](auto sc) { return Hl[decltype(sc)::value]; };

        // 128-wide head gradient on the vector ALU: v[f] = (sum_o w[o][f] g[o]) * (act[f] > 0) -> gbuf, and split into K-slices 0..7
        // `opf`: factor (a power of two) on the values that become the NEXT product's B operands -- gbuf keeps the branch's own scale
        auto head_grad = [&](auto no_c, const float* act, float* out, const float* w_lane, const float* g, const float opf) __attribute__((always_inline)) {
            constexpr int NO = decltype(no_c)::value;
            f32x4 mk[16];
            if constexpr (A16) {      // `act` was formed as a float pointer at the column's ELEMENT offset: redo it in 2-byte elements
                const unsigned short* a16 = reinterpret_cast<const unsigned short*>(P.acts) + (act - P.acts);
#pragma unroll
                for (int i = 0; i < 16; ++i) {
                    const u32x2_b w = *reinterpret_cast<const u32x2_b*>(a16 + 32 * (i >> 2) + 8 * (i & 3) + 4 * kg);
                    // stored behind a ReLU: > 0 <=> a non-zero magnitude (as a float for the comparison below)
                    mk[i] = f32x4{(float)(w[0] & 0x7fffu), (float)((w[0] >> 16) & 0x7fffu), (float)(w[1] & 0x7fffu), (float)((w[1] >> 16) & 0x7fffu)};
                }
            } else {
#pragma unroll
                for (int i = 0; i < 16; ++i) mk[i] = *reinterpret_cast<const f32x4*>(act + 32 * (i >> 2) + 8 * (i & 3) + 4 * kg);
            }
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                const int t = i >> 2, q = i & 3;
                f32x4 v;
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    float acc = w_lane[t * 16 + 4 * q + j] * g[0];
#pragma unroll
                    for (int o = 1; o < NO; ++o) acc = __fmaf_rn(w_lane[o * 128 + t * 16 + 4 * q + j], g[o], acc);
                    v[j] = mk[i][j] > 0.0f ? acc : 0.0f;
                }
                if (valid) *reinterpret_cast<f32x4*>(out + 32 * t + 8 * q + 4 * kg) = v;
                unsigned h0, l0, h1, l1;
                split2(v[0] * opf, v[1] * opf, h0, l0);
                split2(v[2] * opf, v[3] * opf, h1, l1);
                Hh[2 * t + (q >> 1)][2 * (q & 1)] = h0; Hl[2 * t + (q >> 1)][2 * (q & 1)] = l0;
                Hh[2 * t + (q >> 1)][2 * (q & 1) + 1] = h1; Hl[2 * t + (q >> 1)][2 * (q & 1) + 1] = l1;
            }
        };
        // ReLU masks of the NEXT pass: the first kMaskRing quads are requested before that layer's MFMA chunks start
        f32x4 mk[BITS ? 1 : kMaskRing];
        u32x4 bits = {0u, 0u, 0u, 0u};
        const u32x4* const mrow = BITS ? reinterpret_cast<const u32x4*>(P.masks) + (size_t)tile * 8 * 256 + threadIdx.x : nullptr;
        auto preload = [&](const float* act, int layer) __attribute__((always_inline)) {
            if constexpr (BITS) {
                bits = mrow[256 * layer];
            } else {
#pragma unroll
                for (int i = 0; i < kMaskRing; ++i) mk[i] = *reinterpret_cast<const f32x4*>(act + 32 * (i >> 2) + 8 * (i & 3) + 4 * kg);
            }
        };
        // accumulators -> gbuf and the next product's split B operands: z = Zm + 2^-11 Zx [+ w_alpha g_sigma], [* (act > 0)]
        auto pass = [&](auto mask_c, auto alpha_c, const float* act, float* out, const float opf) __attribute__((always_inline)) {
            constexpr bool MASK = decltype(mask_c)::value != 0, ALPHA = decltype(alpha_c)::value != 0;
            constexpr int RING = kMaskRing;
            asm volatile("s_nop 15\n\ts_nop 3" ::: "memory");   // MFMA result -> VALU read wait states (the reads are inside asm)
#pragma unroll
            for (int i = 0; i < 32; ++i) {
                const int t = i >> 2, q = i & 3;
                f32x4 z;
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    float r, y;
                    asm volatile("v_accvgpr_read_b32 %0, %2\n\tv_accvgpr_read_b32 %1, %3\n\tv_fmac_f32 %0, 0x3a000000, %1"
                                 : "=&v"(r), "=&v"(y) : "a"(Zm[t][4 * q + j]), "a"(Zx[t][4 * q + j]));
                    if constexpr (ALPHA) r = __fmaf_rn(aux_l[kBAuxAlphaW + kg * 128 + t * 16 + 4 * q + j], gr[3], r);
                    if constexpr (MASK && BITS) r = (bits[t >> 1] >> (31 - ((t & 1) * 16 + 4 * q + j))) & 1u ? r : 0.0f;   // activate_bits' order
                    if constexpr (MASK && !BITS) r = mk[i % RING][j] > 0.0f ? r : 0.0f;
                    z[j] = r;
                }
                if constexpr (MASK && !BITS) {
                    if (i + RING < 32)
                        mk[i % RING] = *reinterpret_cast<const f32x4*>(act + 32 * ((i + RING) >> 2) + 8 * ((i + RING) & 3) + 4 * kg);
                }
                if (valid) *reinterpret_cast<f32x4*>(out + 32 * t + 8 * q + 4 * kg) = z;
                unsigned h0, l0, h1, l1;
                split2(z[0] * opf, z[1] * opf, h0, l0);
                split2(z[2] * opf, z[3] * opf, h1, l1);
                Hh[2 * t + (q >> 1)][2 * (q & 1)] = h0; Hl[2 * t + (q >> 1)][2 * (q & 1)] = l0;
                Hh[2 * t + (q >> 1)][2 * (q & 1) + 1] = h1; Hl[2 * t + (q >> 1)][2 * (q & 1) + 1] = l1;
            }
        };

        // view branch: g_v (VALU) -> g_feat = g_v @ W_views[:, :256]   (K = 128: 64 items)
        head_grad(IC(3), arow + NSOS_ACTS_VIEWS, grow + NSOS_ACTS_VIEWS, aux_l + kBAuxRgbW + kg * 64, gr, 1.0f);
        static_for<0, 4>([&](auto cc) { run_chunk(IC(16 * decltype(cc)::value), IC(1), IC(0), Zm, Zx, h_h, h_l); });
        pass(IC(0), IC(0), arow, grow + NSOS_ACTS_FEAT, inv_rgb);     // g_feat: gbuf at the colour scale, the trunk's operand at the trunk's
        // d/d h7 = g_feat @ W_feature (+ g_hs @ W_sem0[:, :256]) (+ g_sigma w_alpha, in the pass)
        preload(arow + 256 * 7, 7);
        static_for<0, 8>([&](auto cc) { run_chunk(IC(16 * decltype(cc)::value), IC(1), IC(decltype(cc)::value < 2 && !BITS ? kMaskRing : 0), Zm, Zx, h_h, h_l); });   // pass(feat) loads nothing
        if constexpr (SEM != 0) {
            head_grad(IC(2), arow + NSOS_ACTS_SEM, grow + NSOS_ACTS_SEM, aux_l + kBAuxSem2W + kg * 64, gr + 4, inv_sem);
            static_for<0, 4>([&](auto cc) { run_chunk(IC(16 * decltype(cc)::value), IC(0), IC(0), Zm, Zx, h_h, h_l); });
        }
        pass(IC(1), IC(1), arow + 256 * 7, grow + 256 * 7, 1.0f);
        // trunk: g_z(l-1) = (g_z(l) @ W_l[:, h part]) * (h(l-1) > 0)
#pragma unroll 1
        for (int l = 7; l >= 1; --l) {
            preload(arow + 256 * (l - 1), l - 1);
            static_for<0, 8>([&](auto cc) { run_chunk(IC(16 * decltype(cc)::value), IC(1), IC(decltype(cc)::value < 2 && !BITS ? 32 : 0), Zm, Zx, h_h, h_l); });   // 32 - ring refills + ring preloads
            pass(IC(1), IC(0), arow + 256 * (l - 1), grow + 256 * (l - 1), 1.0f);
        }
    }
#undef IC
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
}

// ------------------------------------------------------------------------------------------ packing
struct X3BwdChunk {
    const float* w;   // [out, in_dim] row-major; the chunk's A operands are W^T tiles: rows = input features col_base + 32t + i
    int in_dim, col_base, i0;
};
struct X3BwdPackParams {
    X3BwdChunk ch[72];
    int n_chunks;
    const float* rgb_w; const float* sem2_w; const float* alpha_w;
    unsigned* aux;
    unsigned short* chunks;
};

__global__ __launch_bounds__(256) void x3_bwd_pack_kernel(const X3BwdPackParams P) {
    const long long gid = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (gid < kBAuxWords) {
        const int a = (int)gid;
        float f = 0.0f;
        auto feat128 = [](int rem) { return acc_feature((rem & 63) >> 4, rem & 15, rem >> 6); };
        if (a < kBAuxSem2W) f = P.rgb_w[(a >> 7) * 128 + feat128(a & 127)];
        else if (a < kBAuxAlphaW) { const int rem = a - kBAuxSem2W; f = P.sem2_w ? P.sem2_w[(rem >> 7) * 128 + feat128(rem & 127)] : 0.0f; }
        else if (a < kBAuxAlphaW + 256) { const int rem = a - kBAuxAlphaW; f = P.alpha_w[acc_feature((rem & 127) >> 4, rem & 15, rem >> 7)]; }
        P.aux[a] = __builtin_bit_cast(unsigned, f);
    }
    const long long per_chunk = kSlotBytes / 2;  // 16-bit elements per slot
    if (gid >= (long long)P.n_chunks * per_chunk) return;
    const X3BwdChunk ck = P.ch[gid / per_chunk];
    const int within = (int)(gid % per_chunk);
    const int g = within >> 9, lane = (within >> 3) & 63, e = within & 7;  // 512 elements per A operand
    const int i = lane & 31, kgl = lane >> 5;
    constexpr int NG = 32, NI = 16;
    float v = 0.0f;
    bool is_hi = true;
    if (g < NG) {
        is_hi = g == 0 || (g != NG - 1 && (g & 1));
        const int item = g == 0 ? 0 : (g == NG - 1 ? NI - 1 : (is_hi ? (g + 1) / 2 : (g - 2) / 2));
        const int a = ck.i0 + item, s = a / 8, t = a % 8;
        const int k = acc_feature(s >> 1, 8 * (s & 1) + e, kgl);   // contraction index = output feature of the layer (row of W)
        v = ck.w[(long long)k * ck.in_dim + ck.col_base + 32 * t + i];
    }
    unsigned short h = f16_bits(v);
    if (!is_hi) h = f16_bits((v - f16_value(h)) * kLoScale);
    P.chunks[gid] = h;
}


constexpr int kLdsBytes = kSlots * kSlotBytes + kBAuxWords * 4;

template <int SEM, bool BITS, bool A16 = false>
int32_t launch_x3_bwd(const X3BwdParams& p, hipStream_t stream) {
    static NsosPerDeviceFlag configured_on;
    bool& configured = configured_on.here();
    if (!configured) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&mlp_x3_bwd_kernel<SEM, BITS, A16>),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, kLdsBytes);
        if (e != hipSuccess) return (int32_t)e;
        configured = true;
    }
    const int cus = nsos_device_cus();
    const int grid = p.n_tiles < cus ? p.n_tiles : cus;
    hipLaunchKernelGGL((mlp_x3_bwd_kernel<SEM, BITS, A16>), dim3(grid), dim3(256), kLdsBytes, stream, p);
    return nsos_launch_status();
}

}  // namespace

// ------------------------------------------------------------------------------------------ C ABI
extern "C" size_t nsos_mlp_bwd_packed_bytes_x3(int32_t sem_mode) {
    if (sem_mode < 0 || sem_mode > 2) return 0;
    return (size_t)kBAuxWords * 4 + (size_t)x3_bwd_chunks(sem_mode) * kSlotBytes;
}

extern "C" int32_t nsos_mlp_bwd_pack_x3(const nsos_mlp_tensors* T_, int32_t sem_mode, void* packed, size_t packed_bytes, void* stream) {
    NSOS_REQUIRE(T_ && packed, NSOS_ERR_NULL_POINTER);
    NSOS_REQUIRE(sem_mode >= 0 && sem_mode <= 2, NSOS_ERR_UNSUPPORTED);
    NSOS_REQUIRE(packed_bytes >= nsos_mlp_bwd_packed_bytes_x3(sem_mode), NSOS_ERR_BUFFER_TOO_SMALL);
    NSOS_REQUIRE(((uintptr_t)packed & 15) == 0, NSOS_ERR_MISALIGNED);
    for (int l = 1; l < NSOS_NET_DEPTH; ++l) NSOS_REQUIRE(T_->pts_w[l], NSOS_ERR_NULL_POINTER);
    NSOS_REQUIRE(T_->alpha_w && T_->feature_w && T_->views_w && T_->rgb_w, NSOS_ERR_NULL_POINTER);
    if (sem_mode) NSOS_REQUIRE(T_->sem0_w && T_->sem2_w, NSOS_ERR_NULL_POINTER);

    X3BwdPackParams P = {};
    int n = 0;
    const int X = NSOS_XYZ_DIM, W = NSOS_NET_WIDTH;
    auto product = [&](const float* w, int in_dim, int col, int k_dim) {   // k_dim / 16 K-slices x 8 tiles, 16 items per chunk
        for (int c = 0; c < k_dim / 32; ++c) P.ch[n++] = X3BwdChunk{w, in_dim, col, 16 * c};
    };
    product(T_->views_w, W + NSOS_DIR_DIM, 0, 128);
    product(T_->feature_w, W, 0, 256);
    if (sem_mode) product(T_->sem0_w, sem_mode == NSOS_SEM_COORD ? W + X : W, 0, 128);
    for (int l = 7; l >= 1; --l) product(T_->pts_w[l], l == 5 ? X + W : W, l == 5 ? X : 0, 256);
    NSOS_REQUIRE(n == x3_bwd_chunks(sem_mode), NSOS_ERR_UNSUPPORTED);
    P.n_chunks = n;
    P.rgb_w = T_->rgb_w; P.sem2_w = sem_mode ? T_->sem2_w : nullptr; P.alpha_w = T_->alpha_w;
    P.aux = static_cast<unsigned*>(packed);
    P.chunks = reinterpret_cast<unsigned short*>(P.aux + kBAuxWords);
    const long long total = (long long)n * (kSlotBytes / 2);
    hipLaunchKernelGGL(x3_bwd_pack_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream, P);
    return nsos_launch_status();
}

static int32_t input_grads_x3(const void* packed, int32_t sem_mode, const float* g_raw, const float* acts, bool acts16,
                              const void* relu_masks, int64_t n_pts, const float* scale, float* gbuf, void* stream) {
    if (n_pts == 0) return NSOS_OK;
    NSOS_REQUIRE(packed && g_raw && acts && scale && gbuf, NSOS_ERR_NULL_POINTER);
    NSOS_REQUIRE(n_pts > 0, NSOS_ERR_BAD_SHAPE);
    NSOS_REQUIRE(sem_mode >= 0 && sem_mode <= 2, NSOS_ERR_UNSUPPORTED);
    NSOS_REQUIRE(((uintptr_t)packed & 15) == 0 && ((uintptr_t)acts & 15) == 0 && ((uintptr_t)gbuf & 15) == 0, NSOS_ERR_MISALIGNED);
    NSOS_REQUIRE((n_pts + kTilePts - 1) / kTilePts < (1ll << 31), NSOS_ERR_UNSUPPORTED);
    X3BwdParams p = {};
    p.aux = static_cast<const unsigned*>(packed);
    p.chunks = reinterpret_cast<const unsigned char*>(p.aux + kBAuxWords);
    NSOS_REQUIRE(((uintptr_t)relu_masks & 15) == 0, NSOS_ERR_MISALIGNED);
    p.g_raw = g_raw; p.acts = acts; p.gbuf = gbuf; p.scale = scale; p.masks = static_cast<const unsigned*>(relu_masks);
    p.n_pts = n_pts;
    p.n_tiles = (int)((n_pts + kTilePts - 1) / kTilePts);
    const hipStream_t st = (hipStream_t)stream;
    if (acts16) {
        NSOS_REQUIRE(relu_masks, NSOS_ERR_NULL_POINTER);      // 16-bit activations carry no usable trunk masks of their own
        return sem_mode == 0 ? launch_x3_bwd<0, true, true>(p, st) : launch_x3_bwd<1, true, true>(p, st);
    }
    if (relu_masks) return sem_mode == 0 ? launch_x3_bwd<0, true>(p, st) : launch_x3_bwd<1, true>(p, st);
    return sem_mode == 0 ? launch_x3_bwd<0, false>(p, st) : launch_x3_bwd<1, false>(p, st);
}

extern "C" int32_t nsos_mlp_input_grads_x3(const void* packed, int32_t sem_mode, const float* g_raw, const float* acts,
                                           const void* relu_masks, int64_t n_pts, const float* scale, float* gbuf, void* stream) {
    return input_grads_x3(packed, sem_mode, g_raw, acts, false, relu_masks, n_pts, scale, gbuf, stream);
}
extern "C" int32_t nsos_mlp_input_grads_x3_a16(const void* packed, int32_t sem_mode, const float* g_raw, const void* acts_f16,
                                               const void* relu_masks, int64_t n_pts, const float* scale, float* gbuf, void* stream) {
    return input_grads_x3(packed, sem_mode, g_raw, static_cast<const float*>(acts_f16), true, relu_masks, n_pts, scale, gbuf, stream);
}
