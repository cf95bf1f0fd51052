// K2-X3/16: the split-fp16 ("fp16x3", fp32-grade) fused positional-encoding + MLP kernel on v_mfma_f32_16x16x32 (round 6;
// "re-tile mlp_x3_kernel onto 16x16x32 by factoring mlp_lp16's chunk schedule").  Same network and the
// same replaced reference code as mlp_x3.hip (models/embedder.py:34-48, models/nerf_mlp.py:67-100,179-215), the same operand
// arithmetic -- every value entering a product is hi = fp16(v), lo = fp16(v - hi); every product is three MFMAs
//     W_hi.h_hi + W_hi.h_lo   (main accumulator)          W_lo.h_hi   (its own accumulator, the weights' lo parts stored x 2^11)
// folded as main + 2^-11 lo by the activation -- on the OTHER 16-bit MFMA shape and with mlp_lp16.hip's workgroup: 8 waves = two
// per SIMD at <= 256 registers, four 36 KiB weight slots fed by global->LDS DMA, one barrier per chunk, the 4-deep A-operand ring
// with compile-time LDS waits (lp16_sched.h: Sched<>, pipeline16x -- shared with mlp_lp16.hip, instantiated unchanged).
//
// Mapping.  mlp_lp16_kernel gives a wave TWO 16-point column blocks c = 0, 1: accumulator Z[t][c], operand H[s][c], one A operand
// feeding two MFMAs.  Here a wave has ONE 16-point block (tile = 128 points) and the index that was the block is the PART:
//     Z[t][0] = main accumulator, Z[t][1] = scaled-lo accumulator;   Hh[s] / Hl[s] = hi / lo operand of k-slice s
// so the register budget is lp16's (128 accumulator registers per 256-wide layer, 64 for the operands).  An A operand is the hi
// or the lo part of (tile t, slice s): the hi group runs two MFMAs (x Hh, x Hl) into Z[t][0], the lo group one (x Hh) into Z[t][1]:
// 3 MFMAs per 2 KiB of LDS reads (lp16: 2 per 1 KiB; mlp_x3_kernel's 32x32x16 tiles: 3 per 2 KiB at twice the FLOPs per MFMA).
// Why it still pays: scripts/ubench/mfma_mix.hip's x3 rows (profiles/r05/c2_mfma_mix_with_x3_rows.txt) -- the 16x16x32 shape
// sustains a higher clock at the chip's power limit and two waves per SIMD cover each other's LDS and barrier latency.
//   * Hidden layers are tile-PAIR-major: chunk c = output tiles 2c, 2c+1 over all 8 slices, hi + lo = 32 A operands, 48 MFMAs, and
//     the pair's fp32 biases as a 128-byte block behind them, read by the PREVIOUS chunk straight into the main accumulators (no bias
//     MFMAs; the bias enters in full fp32 -- mlp_x3_kernel carries it as hi + lo items against B = 1).  The previous pair's
//     activation -- z = main + 2^-11 lo, ReLU, split into (hi, lo): 8 VALU per packed word pair -- rides behind the MFMAs.
//   * sigma, rgb and the semantic logits are split MFMAs into one 16-row "raw" tile (two accumulators), as in mlp_lp16_kernel.
//   * Encodings: the accurate branch-free Cody-Waite + Cephes sincos of the exact kernel (mlp_common.h sincos_fast), evaluated for
//     the lane's 16 (+ 8) feature slots from lp16's per-(q, slot) table -- NOT v_sin_f32: this path promises fp32-grade results.
// Inference only: the SAVE variants (frozen-backbone and full training) stay on mlp_x3_kernel, whose packed stream sits in front
// of this kernel's in the same buffer (nsos_mlp_pack_x3 writes both; nsos_mlp_x3_select_kernel / NSOS_X3_KERNEL pick the forward).
// Results are NOT bit-identical to mlp_x3_kernel (other contraction order inside the MFMAs, fp32 biases, MFMA heads): both are held
// to the same bars against the reference goldens and the exact kernel (tests/test_gpu_parity.py, test_gpu_trained.py).
// Compiled with -ffp-contract=off (x = o + d*z stays a separately rounded multiply and add).
#include "lp_common.h"
#include "lp16_sched.h"
#include "x316.h"

using namespace nsos;
using namespace nsos::lp;
using nsos::x316::x316_chunks;
using nsos::x316::kX316TailBytes;

namespace {

static_assert(kW16 == 8, "mlp_x316_kernel: 8 waves per workgroup");
constexpr int kTileX = 16 * kW16;                 // points per tile: one 16-point block per wave

struct HiLo { unsigned hi, lo; };
// (hi, lo) packed words of the fp32 pair (v0, v1): hi = fp16(v), lo = fp16(v - hi)     [x3_common.h split2]
__device__ __forceinline__ HiLo split2x(float v0, float v1) {
    HiLo r;
    asm volatile("v_cvt_pk_f16_f32 %0, %2, %3\n\t"
                 "v_fma_mix_f32 %2, %0, -1.0, %2 op_sel_hi:[1,0,0]\n\t"
                 "v_fma_mix_f32 %3, %0, -1.0, %3 op_sel:[1,0,0] op_sel_hi:[1,0,0]\n\t"
                 "v_cvt_pk_f16_f32 %1, %2, %3"
                 : "=&v"(r.hi), "=&v"(r.lo), "+v"(v0), "+v"(v1));
    return r;
}
// the same for z = m + 2^-11 x of two accumulator elements, clamped from below at `floor` (0: ReLU; -inf: none).  (The main
// accumulator registers are dead afterwards -- the next chunk that uses them starts from a bias read -- so the in-place v_fmac on the
// by-value copies costs no move.)
__device__ __forceinline__ HiLo split2_acc(float m0, float x0, float m1, float x1, float floor) {
    HiLo r;
    asm volatile("v_fmac_f32 %2, 0x3a000000, %4\n\tv_fmac_f32 %3, 0x3a000000, %5\n\t"
                 "v_max_f32 %2, %6, %2\n\tv_max_f32 %3, %6, %3\n\t"
                 "v_cvt_pk_f16_f32 %0, %2, %3\n\t"
                 "v_fma_mix_f32 %2, %0, -1.0, %2 op_sel_hi:[1,0,0]\n\t"
                 "v_fma_mix_f32 %3, %0, -1.0, %3 op_sel:[1,0,0] op_sel_hi:[1,0,0]\n\t"
                 "v_cvt_pk_f16_f32 %1, %2, %3"
                 : "=&v"(r.hi), "=&v"(r.lo), "+v"(m0), "+v"(m1) : "v"(x0), "v"(x1), "v"(floor));
    return r;
}

__device__ __forceinline__ f32x4 mfma_x(u32x4 a, u32x4 b, f32x4 c) {
    return __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, a), __builtin_bit_cast(f16x8, b), c, 0, 0, 0);
}

constexpr int kStageX = 384;                                               // output stage per wave: 16 points x 24 B
constexpr int kRgbLdsX = kSlots * kSlotBytes + kAuxWords * 4 + kW16 * kStageX;   // LDS offset of rgb_linear's eight resident A operands
constexpr int kLdsBytesX = kRgbLdsX + kX316TailBytes;
static_assert(kLdsBytesX <= 160 * 1024, "LDS");

// NS slices of one point's encoding as split B operands: word w of slice s = slots (8 s + 2 w, 8 s + 2 w + 1) of the lane group's
// table (lp16_sched.h enc_table_entry: slot j = 8 s + e of lane group q carries feature 32 s + 8 q + e).
template <int NS, int L, int ONE_AT>
__device__ __forceinline__ void encode_x(u32x4 (&hi)[NS], u32x4 (&lo)[NS], const float (&x)[3], const unsigned char* lds_tab, int q) {
    float val[8 * NS];
    const f32x4* tab = reinterpret_cast<const f32x4*>(lds_tab) + q * (8 * NS);
    const float amax = fmaxf(fmaxf(fabsf(x[0]), fabsf(x[1])), fabsf(x[2])) * (float)(1 << (L - 1));
    const bool big = !(amax < 32768.0f);                // also true for NaN / Inf inputs (the cold path propagates them)
    static_for<0, 8 * NS>([&](auto jc) {
        constexpr int j = decltype(jc)::value;
        const f32x4 t = tab[j];
        // exact: one scale is a power of two, the others are zero
        const float a = __fmaf_rn(x[2], t[2], __fmaf_rn(x[1], t[1], x[0] * t[0]));
        float sn, cs;
        sincos_fast(a, sn, cs);
        val[j] = t[3] != 0.0f ? cs : sn;
    });
    if (__builtin_expect(big, 0)) {                     // arguments >= 2^15 (never produced by a scene-normalised NeRF): ocml
        static_for<0, 8 * NS>([&](auto jc) {
            constexpr int j = decltype(jc)::value;
            const f32x4 t = tab[j];
            const float a = __fmaf_rn(x[2], t[2], __fmaf_rn(x[1], t[1], x[0] * t[0]));
            float sn, cs;
            sincosf(a, &sn, &cs);
            val[j] = t[3] != 0.0f ? cs : sn;
        });
    }
    // pad slots have all-zero table entries: sin(0) = 0; raw coordinates: slots 0..2 of lane group 0; the constant 1.0: lane group 3
#pragma unroll
    for (int e = 0; e < 3; ++e) val[e] = q == 0 ? x[e] : val[e];
    constexpr int one_slot = 8 * (NS - 1) + (ONE_AT - 32 * (NS - 1) - 24);
    static_assert(one_slot >= 8 * (NS - 1) && one_slot < 8 * NS, "the constant input lives in lane group 3 of the last slice");
    val[one_slot] = q == 3 ? 1.0f : val[one_slot];
#pragma unroll
    for (int s = 0; s < NS; ++s)
#pragma unroll
        for (int w = 0; w < 4; ++w) {
            const HiLo r = split2x(val[8 * s + 2 * w], val[8 * s + 2 * w + 1]);
            hi[s][w] = r.hi;
            lo[s][w] = r.lo;
        }
}

struct X316Params {
    const unsigned char* chunks;     // the 16x16x32 stream (x316_chunks(sem) slots + the rgb operands)
    const float* rays_o;
    const float* rays_d;
    const float* viewdirs;
    const float* z_vals;
    float* raw;
    long long n_pts;
    int n_samples;
    int n_tiles;
    unsigned long long* prof;
};

template <int SEM, bool PROF = false>
__global__ __launch_bounds__(64 * kW16, 1) void mlp_x316_kernel(const X316Params P) {
    extern __shared__ __attribute__((aligned(16))) unsigned char lds[];  // 4 x 36 KiB weight slots + 4 KiB tables + 3 KiB output stage + 8 KiB rgb operands
    const int lane0 = threadIdx.x & 63, wave = threadIdx.x >> 6;
    constexpr int NCH = x316_chunks(SEM);
    constexpr int C = SEM ? 6 : 4;

    // ---- weight stream (as mlp_lp16_kernel): slots rotate (c0 = chunk cur, c1 = cur+1, c2 = cur+2, c3 = the slot that becomes free at the next barrier)
    const unsigned lds_base = (unsigned)(size_t)(__attribute__((address_space(3))) unsigned char*)lds;
    const int wave_s = __builtin_amdgcn_readfirstlane(wave);
    const unsigned voff = (unsigned)(lane0 * 16);
    auto lane_addr = [&](int b) { return lds_base + (unsigned)(b * kSlotBytes + lane0 * 16); };
    auto slot_addr = [&](int b) { return lds_base + (unsigned)(b * kSlotBytes); };
    unsigned c0 = lane_addr(0), c1 = lane_addr(1), c2 = lane_addr(2), c3 = lane_addr(3);
    unsigned d0 = slot_addr(0), d1 = slot_addr(1), d2 = slot_addr(2), d3 = slot_addr(3);
    const unsigned woff = (unsigned)wave_s * 1024u;
    const unsigned wlast = wave_s + kW16 * kFull16 < kSlotGroups ? woff + (unsigned)(kW16 * kFull16) * 1024u : (unsigned)(kSlotGroups - 1) * 1024u;
    auto poff = [&](int i) { return i < kFull16 ? woff + (unsigned)(kW16 * 1024) * (unsigned)i : wlast; };
    const unsigned char* const src_end = P.chunks + (size_t)NCH * kSlotBytes;
    const unsigned char* srcf = P.chunks + (size_t)(2 % NCH) * kSlotBytes;
    auto dma_piece = [&](const unsigned char* src_chunk, unsigned dst_slot, int i) {
        const unsigned long long sp = (unsigned long long)(src_chunk + poff(i));
        const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)sp), hi = __builtin_amdgcn_readfirstlane((unsigned)(sp >> 32));
        const unsigned dst = __builtin_amdgcn_readfirstlane(dst_slot + poff(i));
        dma_1k(reinterpret_cast<const void*>(((unsigned long long)hi << 32) | lo), dst, voff);
    };
    unsigned fill_lo = 0, fill_hi = 0, fill_dst = 0, fill_w = 0;
    auto side = [&](int i, int nfill) {     // this wave's i-th piece of the chunk being fetched, if the chunk holds it (see mlp_lp16.hip)
        const unsigned off = fill_w + (unsigned)(kW16 * 1024) * (unsigned)i;
        const unsigned long long sp = (((unsigned long long)fill_hi << 32) | fill_lo) + off;
        if (__builtin_constant_p(nfill) && kW16 * i + kW16 - 1 < nfill) {
            dma_1k(reinterpret_cast<const void*>(sp), fill_dst + off, voff);
        } else {
            unsigned keep;
            unsigned long long saved;
            asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_mov_b64 %1, exec\n\ts_cmp_lt_i32 %5, %6\n\ts_cselect_b64 exec, %1, 0\n\ts_nop 2\n\t"
                         "global_load_lds_dwordx4 %3, %4\n\ts_mov_b64 exec, %1\n\ts_mov_b32 m0, %0"
                         : "=&s"(keep), "=&s"(saved) : "s"(fill_dst + off), "v"(voff), "s"(reinterpret_cast<const void*>(sp)), "s"(wave_s + kW16 * i), "s"(nfill)
                         : "memory", "scc");
        }
    };
    auto mid = [&]() {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        const unsigned long long sp = (unsigned long long)srcf;
        fill_lo = __builtin_amdgcn_readfirstlane((unsigned)sp);
        fill_hi = __builtin_amdgcn_readfirstlane((unsigned)(sp >> 32));
        fill_dst = __builtin_amdgcn_readfirstlane(d2);
        fill_w = __builtin_amdgcn_readfirstlane(woff);
    };
    auto tail = [&]() {
        const unsigned tc = c0, td = d0;
        c0 = c1; c1 = c2; c2 = c3; c3 = tc;
        d0 = d1; d1 = d2; d2 = d3; d3 = td;
        srcf += kSlotBytes;
        if (srcf == src_end) srcf = P.chunks;
    };
    auto ctx = [&]() { return ChunkCtx{c0, c1}; };

    f32x4 ring[kRing16];
#pragma unroll
    for (int k = 0; k < 3; ++k)
#pragma unroll
        for (int i = 0; i < kDma16; ++i)
            dma_piece(P.chunks + (size_t)(k % NCH) * kSlotBytes, k == 0 ? d0 : (k == 1 ? d1 : d2), i);
    {   // rgb_linear's eight A operands (4 slices x hi, lo) stay resident in LDS: wave w fetches operand w
        const unsigned long long sp = (unsigned long long)(P.chunks + (size_t)NCH * kSlotBytes + (size_t)wave_s * 1024);
        const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)sp), hi = __builtin_amdgcn_readfirstlane((unsigned)(sp >> 32));
        const unsigned dst = __builtin_amdgcn_readfirstlane(lds_base + (unsigned)(kRgbLdsX + wave_s * 1024));
        dma_1k(reinterpret_cast<const void*>(((unsigned long long)hi << 32) | lo), dst, voff);
    }
    unsigned char* const tabs = lds + kSlots * kSlotBytes;
    if (threadIdx.x < 64) {            // xyz table: entry (q, j) of feature 32 (j >> 3) + 8 q + (j & 7)
        const int q = threadIdx.x >> 4, j = threadIdx.x & 15;
        reinterpret_cast<f32x4*>(tabs + kTabXyz)[threadIdx.x] = enc_table_entry(32 * (j >> 3) + 8 * q + (j & 7), NSOS_XYZ_FREQS);
    } else if (threadIdx.x < 96) {     // direction table: feature 8 q + j
        const int i = threadIdx.x - 64, q = i >> 3, j = i & 7;
        reinterpret_cast<f32x4*>(tabs + kTabDir)[i] = enc_table_entry(8 * q + j, NSOS_DIR_FREQS);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    static_for<0, kRing16>([&](auto ic) { lds_read_a<decltype(ic)::value * 1024>(ring[decltype(ic)::value], c0); });
    NSOS_PIN();

#define IC(n) std::integral_constant<int, (n)> {}
    const f32x4 zero4 = {0.0f, 0.0f, 0.0f, 0.0f};

    if constexpr (PROF)
        if (P.prof && blockIdx.x < 2 && lane0 == 0 && wave_s < 8) P.prof[(blockIdx.x * 8 + wave_s) * kProfSlots + kProfSlots - 2] = __builtin_readcyclecounter();
    for (int tile = blockIdx.x; tile < P.n_tiles; tile += gridDim.x) {
        int lane = lane0;
        asm volatile("" : "+v"(lane));
        const int n = lane & 15, q = lane >> 4;
        int stamp_k = 0;
        auto stamp = [&]() {
            if constexpr (PROF) {
                if (P.prof && tile == (int)(blockIdx.x + gridDim.x) && blockIdx.x < 2) {
                    const unsigned long long t = __builtin_readcyclecounter();
                    if (lane == 0 && stamp_k < kProfSlots && wave_s < 8) P.prof[(blockIdx.x * 8 + wave_s) * kProfSlots + stamp_k] = t;
                }
                ++stamp_k;
            }
        };
        stamp();  // 0: tile start
        // ---- this lane's point: tile*128 + wave*16 + n (the four lane groups q hold the same point)
        const long long gp = (long long)tile * kTileX + wave_s * 16 + n;
        const bool exists = gp < P.n_pts;
        const long long gc = exists ? gp : P.n_pts - 1;
        const int ray = (int)(gc / P.n_samples);
        float poison;
        u32x4 exh[2], exl[2];
        {
            const float z = P.z_vals[gc];
            float x[3];
#pragma unroll
            for (int k = 0; k < 3; ++k) {
                const float m = P.rays_d[3ll * ray + k] * z;  // models/sampler.py:70,166 (mul, then add)
                x[k] = P.rays_o[3ll * ray + k] + m;
            }
            poison = ((x[0] - x[0]) + (x[1] - x[1])) + (x[2] - x[2]);   // NaN iff an input is NaN / Inf (see mlp_fused.hip)
            encode_x<2, NSOS_XYZ_FREQS, 63>(exh, exl, x, tabs + kTabXyz, q);   // feature 63 (pad) = 1.0: the bias input of layers 0 and 5
        }

        u32x4 Hh[8], Hl[8];
        f32x4 R[2] = {zero4, zero4};      // the "raw" tile (main, scaled lo): rows 0..2 rgb, 3 sigma, 4..5 semantics

        // ---- chunk runners ----------------------------------------------------------------------------------------
        // hi group: Zm (+)= A x Bh, Zm += A x Bl;  lo group: Zx (+)= A x Bh        (FIRST: the accumulators start here)
        auto part_work = [&](auto p_c, auto first_c, const u32x4 aop, f32x4& zm, f32x4& zx, const u32x4& bh, const u32x4& bl) {
            constexpr int p = decltype(p_c)::value;
            constexpr bool FIRST = decltype(first_c)::value != 0;
            if constexpr (p == 0) {
                zm = mfma_x(aop, bh, FIRST ? zero4 : zm);
                zm = mfma_x(aop, bl, zm);
            } else {
                zx = mfma_x(aop, bh, FIRST ? zero4 : zx);
            }
        };
        // slice-major chunk over NT tiles and NSL slices: group g = (slice g / (2 NT), tile (g % (2 NT)) / 2, part g & 1).  ZF: 0 accumulate;
        // 1: slice 0 starts both accumulators from zero; 2: slice 0 starts the lo accumulator only (the main one holds the bias)
        auto slice_chunk_r = [&](auto nt_c, auto nsl_c, auto zf_c, auto& acc, auto&& bh, auto&& bl, auto&& ride, const int nfill) {
            constexpr int NT = decltype(nt_c)::value, NSL = decltype(nsl_c)::value, ZF = decltype(zf_c)::value;
            pipeline16<2 * NT * NSL, 2 * NT * NSL, 0ull>(ring, ctx(), [&](auto ic, const f32x4& a32, const f32x4&) {
                constexpr int g = decltype(ic)::value, sl = g / (2 * NT), t = (g % (2 * NT)) >> 1, p = g & 1;
                const u32x4 aop = __builtin_bit_cast(u32x4, a32);
                if constexpr (sl == 0 && ZF == 1) part_work(IC(p), IC(1), aop, acc[t][0], acc[t][1], bh(IC(sl)), bl(IC(sl)));
                else if constexpr (sl == 0 && ZF == 2 && p == 1) part_work(IC(1), IC(1), aop, acc[t][0], acc[t][1], bh(IC(sl)), bl(IC(sl)));
                else part_work(IC(p), IC(0), aop, acc[t][0], acc[t][1], bh(IC(sl)), bl(IC(sl)));
                ride(ic);
            }, mid, tail, side, nfill);
#pragma unroll
            for (int t = 0; t < NT; ++t) asm volatile("" : "+v"(acc[t][0]), "+v"(acc[t][1]));   // (keeps LLVM from sinking the chunk: see mlp_lp8.hip)
        };
        auto slice_chunk = [&](auto nt_c, auto nsl_c, auto zf_c, auto& acc, auto&& bh, auto&& bl, const int nfill) {
            slice_chunk_r(nt_c, nsl_c, zf_c, acc, bh, bl, [](auto) {}, nfill);
        };
        // tile-pair chunk of a hidden layer: 32 A operands, group g = (slice g >> 2, tile (g >> 1) & 1, part g & 1); zq[t][0] holds the
        // tile's bias when the chunk starts (read from LDS by the PREVIOUS chunk, or by bias_now).  EXTRA / extra: this chunk's own
        // reads of the NEXT pair chunk's bias block (counted by Sched)
        auto pair_chunk = [&](auto extra_c, auto& zq, auto&& ride, auto&& extra, const int nfill) {
            constexpr unsigned long long EXTRA = (unsigned long long)decltype(extra_c)::value;
            pipeline16x<32, 32, 0ull, EXTRA>(ring, ctx(), [&](auto ic, const f32x4& a32, const f32x4&) {
                constexpr int g = decltype(ic)::value, s = g >> 2, t = (g >> 1) & 1, p = g & 1;
                const u32x4 aop = __builtin_bit_cast(u32x4, a32);
                if constexpr (p == 0) {
                    zq[t][0] = mfma_x(aop, Hh[s], zq[t][0]);
                    zq[t][0] = mfma_x(aop, Hl[s], zq[t][0]);
                } else {
                    zq[t][1] = mfma_x(aop, Hh[s], s == 0 ? zero4 : zq[t][1]);
                }
                ride(ic);
            }, mid, tail, side, extra, nfill);
#pragma unroll
            for (int t = 0; t < 2; ++t) asm volatile("" : "+v"(zq[t][0]), "+v"(zq[t][1]));
        };
        // bias block of a chunk: group 32 of its slot, [tile][q][4 x fp32]; tile t of lane (n, q) at byte 32768 + 64 t + 16 q
        auto read_bias = [&](auto tc, f32x4& dst, unsigned slot_lane) {
            constexpr int t = decltype(tc)::value;
            const unsigned addr = slot_lane - (unsigned)(lane * 16) + (unsigned)(q * 16);
            asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(dst) : "v"(addr), "i"(32768 + 64 * t) : "memory");
        };
        // word w (0..3) of the slice formed by the finished tile pair zq: tile w >> 1, registers 2 (w & 1), + 1
        auto pair_word = [&](f32x4 (&zq)[2][2], auto w_c, float floor, u32x4& hi, u32x4& lo) {
            constexpr int w = decltype(w_c)::value, t = w >> 1, r = 2 * (w & 1);
            const HiLo v = split2_acc(zq[t][0][r], zq[t][1][r], zq[t][0][r + 1], zq[t][1][r + 1], floor);
            hi[w] = v.hi;
            lo[w] = v.lo;
        };
        // H = act(Z) for all 16 tiles (layers 0 and 5: one exposed pass)
        auto activate_all = [&](f32x4 (&Z)[16][2]) {
            asm volatile("s_nop 7" ::: "memory");   // MFMA result -> VALU read wait states (the asm below hides the reads)
#pragma unroll
            for (int s = 0; s < 8; ++s)
#pragma unroll
                for (int w = 0; w < 4; ++w) {
                    const int t = 2 * s + (w >> 1), r = 2 * (w & 1);
                    const HiLo v = split2_acc(Z[t][0][r], Z[t][1][r], Z[t][0][r + 1], Z[t][1][r + 1], 0.0f);
                    Hh[s][w] = v.hi;
                    Hl[s][w] = v.lo;
                }
        };
        auto ex_h = [&](auto sc) { return exh[decltype(sc)::value]; };
        auto ex_l = [&](auto sc) { return exl[decltype(sc)::value]; };

        stamp();  // 1: inputs + xyz encoding
        f32x4 Zq[2][2][2];
        auto dead = [&]() {   // (the pair buffers are redefined where they are dead: see mlp_lp8.hip)
#pragma unroll
            for (int b = 0; b < 2; ++b)
#pragma unroll
                for (int t = 0; t < 2; ++t) asm volatile("" : "=v"(Zq[b][t][0]), "=v"(Zq[b][t][1]));
        };
        {   // ---- layer 0: x63 (2 slices, bias in the pad column) -> 16 tiles: two slice chunks, one exposed activation pass
            f32x4 Z[16][2];
            auto ex0h = [&](auto) { return exh[0]; };
            auto ex0l = [&](auto) { return exl[0]; };
            auto ex1h = [&](auto) { return exh[1]; };
            auto ex1l = [&](auto) { return exl[1]; };
            slice_chunk(IC(16), IC(1), IC(1), Z, ex0h, ex0l, 33);     // (fetches the chunk two ahead: layer 1's first pair)
            slice_chunk(IC(16), IC(1), IC(0), Z, ex1h, ex1l, 33);
            stamp();  // 2: L0 MFMAs
            activate_all(Z);
            stamp();  // 3: L0 activation
        }
        // pair 7 of a layer (in Zq[1] when its last chunk ends) -> H[7]: rides in whatever chunk comes next -- the next pair layer's
        // first chunk, the heads' first chunk, the view branch's first chunk -- since none of them reads slice 7 before its last groups
        // and none touches Zq[1] before group 24; only layer 4 (layer 5's slice chunks need the registers) converts it exposed
        auto ride_tail = [&](auto gc_, auto first_c, auto step_c, float floor) {
            constexpr int g = decltype(gc_)::value, FIRST = decltype(first_c)::value, STEP = decltype(step_c)::value;
            if constexpr (g >= FIRST && g < FIRST + 4 * STEP && (g - FIRST) % STEP == 0) pair_word(Zq[1], IC((g - FIRST) / STEP), floor, Hh[7], Hl[7]);
        };
        auto pair_layer = [&](const int l, const bool bias_now, const bool tail_pending, const bool leave_tail) {
            // chunk c accumulates output tiles 2c, 2c+1 over all 8 input slices into Zq[c & 1]; the activation of the PREVIOUS pair rides
            // behind this chunk's MFMAs into Ho[c - 1].  The layer's input H stays live until its last chunk; there the finished slices
            // move into H, each right after the last use of the slice it replaces.  The last pair stays in Zq[1] for the next chunk
            // to convert (`leave_tail`; ride_tail), and this layer's first chunk converts its predecessor's (`tail_pending`).
            // Biases: chunk c reads the NEXT chunk's two bias vectors into Zq[(c + 1) & 1][t][0] at groups 24, 25 -- the riding activation
            // has consumed those registers by group 16 -- from c1 (tail() rotates the names at group 29).  The last chunk's reads fetch
            // the next layer's first block; where no pair layer follows they read another chunk's operand bytes, and nobody uses them.
            const float floor = l < 8 ? 0.0f : -__builtin_inff();     // feature_linear (l == 8) has no activation
            u32x4 Hoh[7], Hol[7];
            if (bias_now) {
                static_for<0, 2>([&](auto tc) { read_bias(tc, Zq[0][decltype(tc)::value][0], c0); });
                lgkm_wait<0>();
                asm volatile("" : "+v"(Zq[0][0][0]), "+v"(Zq[0][1][0]));
            }
            constexpr unsigned long long kBiasAt = 0x3ull << 24;          // groups 24, 25: one bias vector each
            static_for<0, 8>([&](auto cc) {
                constexpr int c = decltype(cc)::value, cur = c & 1, prv = cur ^ 1;
                pair_chunk(std::integral_constant<unsigned long long, kBiasAt>{}, Zq[cur], [&](auto gc_) {
                    constexpr int g = decltype(gc_)::value;
                    if constexpr (c >= 1 && g >= 4 && g <= 16 && (g & 3) == 0)       // the previous pair -> slice c - 1 of the next layer's input
                        pair_word(Zq[prv], IC((g - 4) >> 2), floor, Hoh[c - 1], Hol[c - 1]);
                    if constexpr (c == 0)
                        if (tail_pending) ride_tail(gc_, IC(4), IC(4), 0.0f);          // (a pair layer's predecessor is always a ReLU layer)
                    if constexpr (c == 7 && g >= 4) {
                        // input slice s was last used by group 4 s + 3: Ho[s] -> H[s] (hi in the first two groups after it, lo in the next two)
                        constexpr int k = g - 4, s = k >> 2, j = k & 3;
                        if constexpr (s < 7) {
                            if constexpr (j == 0) mov_slice16(Hh[s], Hoh[s]);
                            if constexpr (j == 1) mov_slice16(Hl[s], Hol[s]);
                        }
                    }
                }, [&](auto gc_) { read_bias(IC(decltype(gc_)::value - 24), Zq[prv][decltype(gc_)::value - 24][0], c1); }, c >= 6 ? (l == 8 ? 36 : 33) : 33);
            });
            stamp();  // 2 + 2l: MFMAs of layer l (with the riding activation of pairs 0..6)
            // slice 6 was last used by group 27 of the last chunk: its replacement moved at groups 28, 29
            if (!leave_tail) {
                asm volatile("s_nop 7" ::: "memory");
                static_for<0, 4>([&](auto wc) { pair_word(Zq[1], wc, floor, Hh[7], Hl[7]); });
            }
            stamp();  // 3 + 2l: the exposed rest of the activation (pair 7), unless it rides in the next chunk
        };
        dead();
#pragma unroll 1
        for (int l = 1; l <= 4; ++l) pair_layer(l, l == 1, l != 1, l != 4);
        dead();
        {   // ---- layer 5 (skip): h part slice-major over all 16 tiles (8 chunks of one slice), then the x63 part (bias in its pad column)
            f32x4 Z[16][2];
            static_for<0, 8>([&](auto sc) {
                constexpr int s = decltype(sc)::value;
                auto bh = [&](auto) { return Hh[s]; };
                auto bl = [&](auto) { return Hl[s]; };
                if constexpr (s == 0) slice_chunk(IC(16), IC(1), IC(1), Z, bh, bl, 32);
                else slice_chunk(IC(16), IC(1), IC(0), Z, bh, bl, s >= 7 ? 33 : 32);   // (the last two chunks of the part fetch the x63 part's second chunk / layer 6's first pair)
            });
            auto ex0h = [&](auto) { return exh[0]; };
            auto ex0l = [&](auto) { return exl[0]; };
            auto ex1h = [&](auto) { return exh[1]; };
            auto ex1l = [&](auto) { return exl[1]; };
            slice_chunk(IC(16), IC(1), IC(0), Z, ex0h, ex0l, 33);
            slice_chunk(IC(16), IC(1), IC(0), Z, ex1h, ex1l, 33);
            stamp();  // 12: MFMAs of layer 5
            activate_all(Z);
            stamp();  // 13: activation pass
        }
        dead();
#pragma unroll 1
        for (int l = 6; l <= 8; ++l) {
            pair_layer(l, l != 7, l == 7, true);      // (layer 7's last pair rides in the heads' first chunk, layer 8's in the view branch's)
            if (l == 7) {
                // ---- H = relu(h7): sigma head (models/nerf_mlp.py:77) and the semantic head (:79-80), all split MFMAs
                if constexpr (SEM == 0) {
                    // [bias of the raw tile][8 slices x (hi, lo): row 3 = alpha_linear]
                    pipeline16<20, 17, 0x1ull>(ring, ctx(), [&](auto ic, const f32x4& a32, const f32x4& prev) {
                        constexpr int g = decltype(ic)::value;
                        const u32x4 aop = __builtin_bit_cast(u32x4, a32);
                        if constexpr (g >= 1 && g < 17) {
                            constexpr int s = (g - 1) >> 1, p = (g - 1) & 1;
                            if constexpr (p == 0) {
                                R[0] = mfma_x(aop, Hh[s], s == 0 ? prev : R[0]);
                                R[0] = mfma_x(aop, Hl[s], R[0]);
                            } else {
                                R[1] = mfma_x(aop, Hh[s], s == 0 ? zero4 : R[1]);
                            }
                        }
                        ride_tail(ic, IC(2), IC(2), 0.0f);          // layer 7's last pair -> H[7], read at groups 15, 16
                    }, mid, tail, side, 33);
                    asm volatile("" : "+v"(R[0]), "+v"(R[1]));
                } else {
                    f32x4 S8[8][2];
                    u32x4 Sph[4], Spl[4];
                    // semantic_linear.0 on relu(h7): 4 chunks of 2 slices x 8 tiles; its bias is the first chunk's bias block (8 tiles)
                    static_for<0, 8>([&](auto tc) { read_bias(tc, S8[decltype(tc)::value][0], c0); });
                    lgkm_wait<0>();
#pragma unroll
                    for (int t = 0; t < 8; ++t) asm volatile("" : "+v"(S8[t][0]));
                    static_for<0, 4>([&](auto jc) {
                        constexpr int j = decltype(jc)::value;
                        auto bh = [&](auto sc) { return Hh[2 * j + decltype(sc)::value]; };
                        auto bl = [&](auto sc) { return Hl[2 * j + decltype(sc)::value]; };
                        if constexpr (j == 0) slice_chunk_r(IC(8), IC(2), IC(2), S8, bh, bl, [&](auto gc_) { ride_tail(gc_, IC(4), IC(4), 0.0f); }, 32);   // (+ layer 7's last pair -> H[7], read by the fourth chunk)
                        else slice_chunk(IC(8), IC(2), IC(0), S8, bh, bl, j == 2 ? (SEM == 2 ? 32 : 25) : (j == 3 ? (SEM == 2 ? 25 : 33) : 32));
                    });
                    if constexpr (SEM == 2) slice_chunk(IC(8), IC(2), IC(0), S8, ex_h, ex_l, 33);     // the x63 part (sem_with_coord)
                    auto activate_sem = [&]() {
                        asm volatile("s_nop 7" ::: "memory");
#pragma unroll
                        for (int s = 0; s < 4; ++s)
#pragma unroll
                            for (int w = 0; w < 4; ++w) {
                                const int t = 2 * s + (w >> 1), r = 2 * (w & 1);
                                const HiLo v = split2_acc(S8[t][0][r], S8[t][1][r], S8[t][0][r + 1], S8[t][1][r + 1], 0.0f);
                                Sph[s][w] = v.hi;
                                Spl[s][w] = v.lo;
                            }
                    };
                    // [bias of the raw tile][sigma: 8 slices x (hi, lo)][logits: 4 slices x (hi, lo)]
                    pipeline16<28, 25, 0x1ull>(ring, ctx(), [&](auto ic, const f32x4& a32, const f32x4& prev) {
                        constexpr int g = decltype(ic)::value;
                        const u32x4 aop = __builtin_bit_cast(u32x4, a32);
                        if constexpr (g >= 1 && g < 17) {
                            constexpr int s = (g - 1) >> 1, p = (g - 1) & 1;
                            if constexpr (p == 0) {
                                R[0] = mfma_x(aop, Hh[s], s == 0 ? prev : R[0]);
                                R[0] = mfma_x(aop, Hl[s], R[0]);
                            } else {
                                R[1] = mfma_x(aop, Hh[s], s == 0 ? zero4 : R[1]);
                            }
                        } else if constexpr (g >= 17 && g < 25) {
                            constexpr int s = (g - 17) >> 1, p = (g - 17) & 1;
                            if constexpr (g == 17) activate_sem();
                            if constexpr (p == 0) {
                                R[0] = mfma_x(aop, Sph[s], R[0]);
                                R[0] = mfma_x(aop, Spl[s], R[0]);
                            } else {
                                R[1] = mfma_x(aop, Sph[s], R[1]);
                            }
                        }
                    }, mid, tail, side, 33);
                    asm volatile("" : "+v"(R[0]), "+v"(R[1]));
                }
                stamp();  // 18 (l == 7 only; the later slots shift by one): sigma + semantic heads
            }
        }
        // ---- view branch (models/nerf_mlp.py:87-92): cat([feature, dir27]) -> 128 -> rgb.  H = feature (no activation).
        // Pair-major over the 8 hidden tiles: chunk j = tiles 2j, 2j+1 x (8 feature slices + the direction slice, whose pad column
        // carries the bias) x (hi, lo): 36 A operands; the previous pair's activation rides.
        u32x4 edh[1], edl[1];
        {
            float dv[3];
#pragma unroll
            for (int k = 0; k < 3; ++k) dv[k] = P.viewdirs[3ll * ray + k];
            poison += ((dv[0] - dv[0]) + (dv[1] - dv[1])) + (dv[2] - dv[2]);
            encode_x<1, NSOS_DIR_FREQS, 27>(edh, edl, dv, tabs + kTabDir, q);   // feature 27 (pad) = 1.0: the bias input of views_linears.0
        }
        stamp();  // direction encoding
        u32x4 Vph[4], Vpl[4];
        static_for<0, 4>([&](auto jc) {
            constexpr int j = decltype(jc)::value, cur = j & 1, prv = cur ^ 1;
            pipeline16<36, 36, 0ull>(ring, ctx(), [&](auto ic, const f32x4& a32, const f32x4&) {
                constexpr int g = decltype(ic)::value, s = g >> 2, t = (g >> 1) & 1, p = g & 1;
                const u32x4 aop = __builtin_bit_cast(u32x4, a32);
                if constexpr (s < 8) {
                    if constexpr (p == 0) {
                        Zq[cur][t][0] = mfma_x(aop, Hh[s], s == 0 ? zero4 : Zq[cur][t][0]);
                        Zq[cur][t][0] = mfma_x(aop, Hl[s], Zq[cur][t][0]);
                    } else {
                        Zq[cur][t][1] = mfma_x(aop, Hh[s], s == 0 ? zero4 : Zq[cur][t][1]);
                    }
                } else {
                    if constexpr (p == 0) {
                        Zq[cur][t][0] = mfma_x(aop, edh[0], Zq[cur][t][0]);
                        Zq[cur][t][0] = mfma_x(aop, edl[0], Zq[cur][t][0]);
                    } else {
                        Zq[cur][t][1] = mfma_x(aop, edh[0], Zq[cur][t][1]);
                    }
                }
                if constexpr (j >= 1 && g >= 8 && g <= 20 && (g & 3) == 0)
                    pair_word(Zq[prv], IC((g - 8) >> 2), 0.0f, Vph[j - 1], Vpl[j - 1]);
                if constexpr (j == 0) ride_tail(ic, IC(4), IC(4), -__builtin_inff());   // feature_linear's last pair -> H[7] (no activation), read at groups 28..31
            }, mid, tail, side, j < 2 ? 36 : (j == 2 ? 32 : 32));      // (the last two fetch the next tile's layer 0)
#pragma unroll
            for (int t = 0; t < 2; ++t) asm volatile("" : "+v"(Zq[cur][t][0]), "+v"(Zq[cur][t][1]));
        });
        stamp();  // view-branch MFMAs
        {
            asm volatile("s_nop 7" ::: "memory");
            static_for<0, 4>([&](auto wc) { pair_word(Zq[1], wc, 0.0f, Vph[3], Vpl[3]); });
        }
        // rgb_linear: rows 0..2 of the raw tile += W_rgb x the view branch's hidden activations (4 slices x 3 MFMAs), operands from LDS
        {
            f32x4 ra[8];
            const unsigned ra_addr = lds_base + (unsigned)kRgbLdsX + (unsigned)(lane * 16);
            static_for<0, 8>([&](auto kc) { lds_read_a<decltype(kc)::value * 1024>(ra[decltype(kc)::value], ra_addr); });
            // (drains the ring's look-ahead reads of the next chunk too: its counted waits then pass at once)
            asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(ra[0]), "+v"(ra[1]), "+v"(ra[2]), "+v"(ra[3]), "+v"(ra[4]), "+v"(ra[5]), "+v"(ra[6]), "+v"(ra[7]) : : "memory");
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const u32x4 ah = __builtin_bit_cast(u32x4, ra[2 * k]), al = __builtin_bit_cast(u32x4, ra[2 * k + 1]);
                R[0] = mfma_x(ah, Vph[k], R[0]);
                R[1] = mfma_x(al, Vph[k], R[1]);
                R[0] = mfma_x(ah, Vpl[k], R[0]);
            }
        }
        asm volatile("" : "+v"(R[0]), "+v"(R[1]));
        stamp();  // rgb MFMAs
        {
            asm volatile("s_nop 7" ::: "memory");
            f32x4 out;
#pragma unroll
            for (int r = 0; r < 4; ++r) out[r] = __fmaf_rn(R[1][r], 4.8828125e-4f, R[0][r]);      // main + 2^-11 lo
            // NaN / Inf in a point's inputs must come out as NaN (the reference propagates them)
            if (poison != poison) {
                const float qnan = __builtin_nanf("");
                out = f32x4{qnan, qnan, qnan, qnan};
            }
            if constexpr (C == 4) {
                // lane (n, 0) holds [r, g, b, sigma] of point n: 16 lanes x 16 B contiguous
                if (q == 0 && exists) *reinterpret_cast<f32x4*>(P.raw + gp * 4) = out;
            } else {
                const long long wave_first = (long long)tile * kTileX + wave_s * 16;
                const long long left = P.n_pts - wave_first;
                if (left >= 16) {
                    // 24 B per point: staged through 384 B of LDS per wave, then 24 lanes store 16 B each
                    float* const stage = reinterpret_cast<float*>(lds + kSlots * kSlotBytes + kAuxWords * 4) + wave_s * (kStageX / 4);
                    if (q == 0) {                                  // (rows are 24 bytes apart: 8-byte aligned pieces)
                        *reinterpret_cast<f32x2*>(stage + 6 * n) = f32x2{out[0], out[1]};
                        *reinterpret_cast<f32x2*>(stage + 6 * n + 2) = f32x2{out[2], out[3]};
                    }
                    if (q == 1) *reinterpret_cast<f32x2*>(stage + 6 * n + 4) = f32x2{out[0], out[1]};
                    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                    __builtin_amdgcn_wave_barrier();
                    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
                    if (lane < 24) *reinterpret_cast<f32x4*>(P.raw + wave_first * C + 4 * lane) = *reinterpret_cast<const f32x4*>(stage + 4 * lane);
                    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");   // the next tile's writes to the stage follow these reads
                    __builtin_amdgcn_wave_barrier();
                } else if (exists) {                               // the ragged last wave of the launch: per point
                    float* o = P.raw + gp * C;
                    if (q == 0) {
                        *reinterpret_cast<f32x2*>(o) = f32x2{out[0], out[1]};
                        *reinterpret_cast<f32x2*>(o + 2) = f32x2{out[2], out[3]};
                    } else if (q == 1) {
                        *reinterpret_cast<f32x2*>(o + 4) = f32x2{out[0], out[1]};
                    }
                }
            }
        }
        stamp();  // output stores
    }
#undef IC
    if constexpr (PROF)
        if (P.prof && blockIdx.x < 2 && lane0 == 0 && wave_s < 8) P.prof[(blockIdx.x * 8 + wave_s) * kProfSlots + kProfSlots - 1] = __builtin_readcyclecounter();
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    __syncthreads();
}

template <int SEM, bool PROF>
int32_t launch_x316p(const X316Params& p, hipStream_t stream) {
    static NsosPerDeviceFlag configured_on;
    bool& configured = configured_on.here();
    if (!configured) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&mlp_x316_kernel<SEM, PROF>),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, kLdsBytesX);
        if (e != hipSuccess) return (int32_t)e;
        configured = true;
    }
    const int cus = nsos_device_cus();
    X316Params q = p;
    q.n_tiles = (int)((p.n_pts + kTileX - 1) / kTileX);
    const int grid = q.n_tiles < cus ? q.n_tiles : cus;
    hipLaunchKernelGGL((mlp_x316_kernel<SEM, PROF>), dim3(grid), dim3(64 * kW16), kLdsBytesX, stream, q);
    return nsos_launch_status();
}
template <int SEM>
int32_t launch_x316s(const X316Params& p, hipStream_t stream) {
    if (p.prof) return launch_x316p<SEM, true>(p, stream);
    return launch_x316p<SEM, false>(p, stream);
}

// ------------------------------------------------------------------------------------------ packing
// One descriptor per chunk (36 groups of 1 KiB = 512 fp16 elements; lane (i = lane & 15, q = lane >> 4), element e).  An A operand
// of (tile t, slice s, part p) holds part p of W[16 t + i][col(s, q, e)] -- hi = fp16(w), lo = fp16((w - hi) x 2^11) -- col by the
// input's kind as in lp16_pack_kernel (hidden: col_base + 32 s + 16 (e >> 2) + 4 q + (e & 3); encoding: col_base + 32 s + 8 q + e,
// the bias in the pad column).  A bias group / block holds fp32 values.
enum KindX { kXPair = 0,     // tile pair a0 of a hidden layer: g = 4 s + 2 t + p (32 groups), then the bias block (group 32: [t][q][4 x fp32])
             kXSlice = 1,    // slice-major over slices s0 .. s0 + nsl - 1 and nt tiles, hidden input: g = (s - s0) 2 nt + 2 t + p; nbias tiles of bias block
             kXEnc = 2,      // the same over encoding slices; bias in the pad column (f == 63) when `bias`
             kXView = 3,     // pair a0 of the view branch: g = 4 s + 2 t + p, s < 8 hidden (feature), s == 8 the direction slice (bias at f == 27)
             kXTail = 4,     // [raw bias][sigma: 16][logits: 8]
             kXSigma = 5,    // [raw bias][sigma: 16]
             kXRgb = 6 };
struct ChunkX {
    const float* w;
    const float* bias;
    int in_dim, col_base, kind, a0, nt, nsl, nbias;
};
struct PackXParams {
    ChunkX ch[80];
    int n_chunks;
    const float* alpha_w; const float* alpha_b;
    const float* rgb_w; const float* rgb_b;
    const float* sem2_w; const float* sem2_b;
    unsigned short* chunks;
};

__device__ __forceinline__ int hid_col(int s, int q, int e) { return 32 * s + 16 * (e >> 2) + 4 * q + (e & 3); }

__global__ __launch_bounds__(256) void x316_pack_kernel(const PackXParams P) {
    const long long gid = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    const long long per_chunk = kSlotBytes / 2;  // 16-bit elements per slot
    const long long body = (long long)P.n_chunks * per_chunk;
    if (gid >= body + kX316TailBytes / 2) return;
    const bool in_tail = gid >= body;
    const ChunkX ck = in_tail ? ChunkX{nullptr, nullptr, 0, 0, kXRgb, 0, 0, 0, 0} : P.ch[gid / per_chunk];
    const int within = in_tail ? (int)(gid - body) : (int)(gid % per_chunk);
    const int g = within >> 9, lane = (within >> 3) & 63, e = within & 7;  // 512 elements per group
    const int i = lane & 15, q = lane >> 4;
    const int W = NSOS_NET_WIDTH, X = NSOS_XYZ_DIM;
    bool is_bias = false, is_lo = false;
    float v = 0.0f;
    const int word = (lane * 8 + e) >> 1;                // fp32 index inside a group that holds fp32 values
    auto raw_bias = [&]() {                          // the raw tile: rows 0..2 rgb, 3 sigma, 4..5 semantics
        is_bias = true;
        const int row = 4 * q + (e >> 1);
        v = row < 3 ? P.rgb_b[row] : row == 3 ? P.alpha_b[0] : (row < 6 && P.sem2_b) ? P.sem2_b[row - 4] : 0.0f;
    };
    auto bias_block = [&](int first_tile, int n_tiles) {     // [tile][q][4 x fp32]: word = 16 t + 4 q + r = feature within the block
        if (word < 16 * n_tiles) {
            is_bias = true;
            v = ck.bias ? ck.bias[16 * first_tile + word] : 0.0f;
        }
    };
    switch (ck.kind) {
        case kXPair: {
            if (g < 32) {
                const int s = g >> 2, t = 2 * ck.a0 + ((g >> 1) & 1);
                is_lo = g & 1;
                v = ck.w[(long long)(16 * t + i) * ck.in_dim + ck.col_base + hid_col(s, q, e)];
            } else if (g == 32) bias_block(2 * ck.a0, 2);
        } break;
        case kXSlice: {
            if (g < 2 * ck.nt * ck.nsl) {
                const int s = ck.a0 + g / (2 * ck.nt), t = (g % (2 * ck.nt)) >> 1;
                is_lo = g & 1;
                v = ck.w[(long long)(16 * t + i) * ck.in_dim + ck.col_base + hid_col(s, q, e)];
            } else if (g == 32 && ck.nbias) bias_block(0, ck.nbias);
        } break;
        case kXEnc: {
            if (g < 2 * ck.nt * ck.nsl) {
                const int s = ck.a0 + g / (2 * ck.nt), t = (g % (2 * ck.nt)) >> 1, f = 32 * s + 8 * q + e;
                is_lo = g & 1;
                if (f < X) v = ck.w[(long long)(16 * t + i) * ck.in_dim + ck.col_base + f];
                else if (f == 63 && ck.bias) v = ck.bias[16 * t + i];
            }
        } break;
        case kXView: {
            const int s = g >> 2, t = 2 * ck.a0 + ((g >> 1) & 1);
            is_lo = g & 1;
            if (s < 8) v = ck.w[(long long)(16 * t + i) * ck.in_dim + hid_col(s, q, e)];
            else {
                const int f = 8 * q + e;
                if (f < NSOS_DIR_DIM) v = ck.w[(long long)(16 * t + i) * ck.in_dim + W + f];
                else if (f == 27) v = ck.bias[16 * t + i];
            }
        } break;
        case kXTail: case kXSigma: {
            if (g == 0) raw_bias();
            else if (g < 17) { is_lo = (g - 1) & 1; v = i == 3 ? P.alpha_w[hid_col((g - 1) >> 1, q, e)] : 0.0f; }
            else if (g < 25 && ck.kind == kXTail) { is_lo = (g - 17) & 1; v = (i == 4 || i == 5) ? P.sem2_w[(i - 4) * (W / 2) + hid_col((g - 17) >> 1, q, e)] : 0.0f; }
        } break;
        case kXRgb: {
            if (g < 8) { is_lo = g & 1; v = i < 3 ? P.rgb_w[i * (W / 2) + hid_col(g >> 1, q, e)] : 0.0f; }
        } break;
    }
    const long long at = in_tail ? body + within : gid;
    if (is_bias) {
        const unsigned u = __builtin_bit_cast(unsigned, v);
        P.chunks[at] = (e & 1) ? (unsigned short)(u >> 16) : (unsigned short)(u & 0xffffu);
    } else {
        unsigned short h = F16::bits(v);
        if (is_lo) h = F16::bits((v - (float)__builtin_bit_cast(_Float16, h)) * 2048.0f);
        P.chunks[at] = h;
    }
}

}  // namespace

namespace nsos {
namespace x316 {

int32_t pack(const void* tensors, int32_t sem_mode, unsigned char* chunks, hipStream_t stream) {
    const nsos_mlp_tensors* T_ = static_cast<const nsos_mlp_tensors*>(tensors);
    const int X = NSOS_XYZ_DIM, W = NSOS_NET_WIDTH;
    PackXParams P = {};
    int n = 0;
    auto add = [&](const float* w, const float* bias, int in_dim, int col, int kind, int a0, int nt, int nsl, int nbias) {
        P.ch[n++] = ChunkX{w, bias, in_dim, col, kind, a0, nt, nsl, nbias};
    };
    auto pairs = [&](const float* w, const float* b) {
        for (int c = 0; c < 8; ++c) add(w, b, W, 0, kXPair, c, 2, 8, 2);
    };
    for (int s = 0; s < 2; ++s) add(T_->pts_w[0], T_->pts_b[0], X, 0, kXEnc, s, 16, 1, 0);
    for (int l = 1; l <= 4; ++l) pairs(T_->pts_w[l], T_->pts_b[l]);
    for (int s = 0; s < 8; ++s) add(T_->pts_w[5], nullptr, X + W, X, kXSlice, s, 16, 1, 0);     // skip layer: input = cat([x63, h]): h part first ...
    for (int s = 0; s < 2; ++s) add(T_->pts_w[5], T_->pts_b[5], X + W, 0, kXEnc, s, 16, 1, 0);   // ... then x63, whose pad column carries the bias
    pairs(T_->pts_w[6], T_->pts_b[6]);
    pairs(T_->pts_w[7], T_->pts_b[7]);
    if (sem_mode) {
        const int in_dim = sem_mode == NSOS_SEM_COORD ? W + X : W;
        for (int j = 0; j < 4; ++j) add(T_->sem0_w, T_->sem0_b, in_dim, 0, kXSlice, 2 * j, 8, 2, j == 0 ? 8 : 0);
        if (sem_mode == NSOS_SEM_COORD) add(T_->sem0_w, nullptr, in_dim, W, kXEnc, 0, 8, 2, 0);
        add(nullptr, nullptr, 0, 0, kXTail, 0, 0, 0, 0);
    } else {
        add(nullptr, nullptr, 0, 0, kXSigma, 0, 0, 0, 0);
    }
    pairs(T_->feature_w, T_->feature_b);
    for (int j = 0; j < 4; ++j) add(T_->views_w, T_->views_b, W + NSOS_DIR_DIM, 0, kXView, j, 2, 9, 0);
    if (n != x316_chunks(sem_mode)) return NSOS_ERR_UNSUPPORTED;
    P.n_chunks = n;
    P.alpha_w = T_->alpha_w; P.alpha_b = T_->alpha_b;
    P.rgb_w = T_->rgb_w; P.rgb_b = T_->rgb_b;
    P.sem2_w = sem_mode ? T_->sem2_w : nullptr; P.sem2_b = sem_mode ? T_->sem2_b : nullptr;
    P.chunks = reinterpret_cast<unsigned short*>(chunks);
    const long long total = (long long)n * (kSlotBytes / 2) + kX316TailBytes / 2;
    hipLaunchKernelGGL(x316_pack_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, stream, P);
    return nsos_launch_status();
}

int32_t launch(const unsigned char* chunks, int32_t sem_mode, const float* rays_o, const float* rays_d, const float* viewdirs,
               const float* z_vals, long long n_pts, int32_t n_samples, float* raw, unsigned long long* prof, hipStream_t stream) {
    X316Params p = {};
    p.chunks = chunks;
    p.rays_o = rays_o; p.rays_d = rays_d; p.viewdirs = viewdirs; p.z_vals = z_vals;
    p.raw = raw; p.n_pts = n_pts; p.n_samples = n_samples; p.prof = prof;
    switch (sem_mode) {
        case 0: return launch_x316s<0>(p, stream);
        case 1: return launch_x316s<1>(p, stream);
        default: return launch_x316s<2>(p, stream);
    }
}

}  // namespace x316
}  // namespace nsos
