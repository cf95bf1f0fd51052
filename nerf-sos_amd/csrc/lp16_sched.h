// The chunk machinery of the 16x16x32 MLP kernels, shared by mlp_lp16.hip (16-bit: one MFMA per product, two 16-point column blocks
// per wave) and mlp_x316.hip (split fp16: three MFMAs per product -- hi.hi, hi.lo, lo.hi -- one 16-point block per wave, hi / lo
// operand files where lp16 has its two blocks).  Round 6: lifted out of mlp_lp16.hip UNCHANGED -- the LDS read
// schedule of a chunk (Sched<>: which ring slot is read when, and the lgkmcnt each group waits with), the A-operand ring pipeline
// with its DMA issue slots and the one barrier per chunk (pipeline16x), the packed-word helpers and the encoding tables.  What a
// group's A operand is multiplied with -- one MFMA per block, or two on the main accumulator / one on the scaled-lo accumulator --
// is the `work` callback's business, so both kernels instantiate the same schedule.
#pragma once
#include "lp_common.h"

using namespace nsos;
using namespace nsos::lp;

namespace {

#ifndef NSOS_LP16_WAVES
#define NSOS_LP16_WAVES 8
#endif
// waves per workgroup (= per CU): 8 = two per SIMD; 16 = four per SIMD (the kernel needs <= 128 registers then): one weight slot then
// serves 512 points, so the DMA's LDS writes, the barrier and the DMA issue slots per chunk are paid once per 2 x the MFMA work
constexpr int kW16 = NSOS_LP16_WAVES;
static_assert(kW16 == 8 || kW16 == 16, "8 or 16 waves per workgroup");
constexpr int kTile16 = 32 * kW16;                                  // points per tile (= per pass over the weight stream)
constexpr int kFull16 = kSlotGroups / kW16;                         // pieces every wave fetches of a full 36-piece chunk (4 | 2) ...
constexpr int kRing16 = 4, kMid16 = 2, kDma16 = kFull16 + 1;        // ... + one more for the first waves (pieces 32..35)

// NFILL (pipeline16x): how many 1 KiB pieces the chunk TWO AHEAD holds -- the DMA issued during a chunk fetches that one, and
// copies only those pieces (the slot stride stays 36 KiB).  Copying every slot whole was 40 pieces per chunk (8 waves x 5, four
// of them redundant) = 1.6 MB per 256-point tile from L2 against the 1.28 MB the kernel reads; the weight stream is the
// largest single cost around the MFMAs (profiles/r04: a build without it runs 12 % fewer cycles per tile).  Static per call
// site (the chunk order of a tile is fixed); where a call site serves several positions, the largest count among them.

// ---- the chunk's LDS read schedule --------------------------------------------------------------------------------
// Groups 0..NG-1 (NG a multiple of 4: the ring slot of group g is g % 4 at every call site); bit g of BIAS marks a bias
// group: no MFMA of its own, its ring slot is the C operand of the NEXT group's MFMAs.  Reads are issued in target order:
// after the work of a non-bias group g everything up to target g + 4 that has not been issued yet (after a bias group:
// nothing -- its slot is still needed).  Targets NWORK..NG-1 (padding) are never read; targets >= NG are the next chunk's
// first groups (always read: resident, proven by this chunk's barrier).  LDS returns in order, so "lgkmcnt(n)" with n =
// the number of reads issued after R(g) means group g (and the bias group in front of it) has landed.
template <int NG, int NWORK, unsigned long long BIAS, unsigned long long EXTRA = 0ull>
struct Sched {
    static_assert(NG % 4 == 0 && NG >= kRing16 + kMid16 + 2 && NG <= 36, "chunk length");
    static_assert(NWORK >= kRing16 && NWORK <= NG, "the first four groups were read by the previous chunk");
    static_assert((BIAS >> (NWORK - 1)) == 0 && (BIAS & (BIAS >> 1)) == 0, "a bias group is followed by a work group");
    static constexpr bool is_bias(int g) { return g >= 0 && g < NG && ((BIAS >> g) & 1ull); }
    static constexpr bool has_extra(int g) { return g >= 0 && g < NG && ((EXTRA >> g) & 1ull); }   // one more LDS read issued after group g's ring reads
    static constexpr bool issued(int x) { return x < NWORK || x >= NG; }
    static constexpr int top_before(int g) {            // highest target whose read was issued before the work of group g
        int top = kRing16 - 1;
        for (int h = 0; h < g; ++h)
            if (!is_bias(h)) top = h + kRing16;
        return top;
    }
    static constexpr int issuer_of(int x) {             // the group after whose work target x is read (-1: the previous chunk)
        for (int h = 0; h < NG; ++h)
            if (!is_bias(h) && top_before(h) < x && x <= h + kRing16) return h;
        return -1;
    }
    static constexpr int younger(int g) {
        int n = 0;
        for (int x = g + 1; x <= top_before(g); ++x) n += issued(x) ? 1 : 0;
        // extra reads issued after R(g) and before the work of group g: after the work of groups issuer_of(g) .. g - 1 (an extra
        // follows its group's ring reads, so the issuer's own extra is younger than R(g) too)
        const int from = issuer_of(g);
        for (int h = from < 0 ? 0 : from; h < g; ++h) n += has_extra(h) ? 1 : 0;
        return n;
    }
};

// work(g, cur, prev): ring slots of group g and of group g-1 (the bias operand when g-1 is a bias group); extra(g): the EXTRA reads
template <int NG, int NWORK, unsigned long long BIAS, unsigned long long EXTRA, class M, class B, class TL, class S, class X>
__device__ __forceinline__ void pipeline16x(f32x4 (&ring)[kRing16], const ChunkCtx ctx, M&& work, B&& mid, TL&& tail, S&& side, X&& extra,
                                            const int nfill) {
    using SC = Sched<NG, NWORK, BIAS, EXTRA>;
    static_for<0, NG>([&](auto ic) {
        constexpr int g = decltype(ic)::value;
        if constexpr (g == kMid16) {
            NSOS_PIN();
            mid();
            NSOS_PIN();
        }
        if constexpr (g < NWORK && !SC::is_bias(g)) lgkm_wait<SC::younger(g)>();
        NSOS_PIN();
        work(ic, ring[g % kRing16], ring[(g + kRing16 - 1) % kRing16]);
        NSOS_PIN();
        if constexpr (!SC::is_bias(g)) {
            static_for<SC::top_before(g) + 1, g + kRing16 + 1>([&](auto xc) {
                constexpr int x = decltype(xc)::value;
                if constexpr (x < NG) {
                    if constexpr (x < NWORK) lds_read_a<x * 1024>(ring[x % kRing16], ctx.wl_cur);
                } else {
                    lds_read_a<(x - NG) * 1024>(ring[x % kRing16], ctx.wl_nxt);
                }
            });
        }
        if constexpr (SC::has_extra(g)) {
            NSOS_PIN();
            extra(ic);
            NSOS_PIN();
        }
#if defined(NSOS_LP16_NODMA)        // (A/B builds only, scripts/diag/build_variant.sh: timing without the weight stream; wrong results)
#elif defined(NSOS_LP16_DMA_STAGGER)   // (A/B: the two waves of a SIMD issue their pieces ten groups apart)
        side(std::integral_constant<int, g - kMid16>{}, std::integral_constant<int, g - kMid16 - 10>{});
#else
#ifdef NSOS_LP16_DMA_BURST            // (A/B: all five pieces back to back behind the barrier)
        if constexpr (g == kMid16) {
            NSOS_PIN();
            for (int i = 0; i < kDma16; ++i) side(i, nfill);
            NSOS_PIN();
        }
#elif defined(NSOS_LP16_DMA_LATE)     // (A/B: the pieces in the MFMA-dense middle of the chunk, two groups apart)
        if constexpr (g >= 8 && g < 8 + 2 * kDma16 && (g & 1) == 0) {
            NSOS_PIN();
            side((g - 8) >> 1, nfill);
            NSOS_PIN();
        }
#else
        if constexpr (g >= kMid16 && g < kMid16 + kDma16) {
            NSOS_PIN();
#ifdef NSOS_LP16_FULL_DMA   // (A/B builds only: every slot copied whole)
            side(g - kMid16, 36);
#else
            side(g - kMid16, nfill);
#endif
            NSOS_PIN();
        }
#endif
#endif
        if constexpr (g == NG - 3) {
            NSOS_PIN();
            tail();
            NSOS_PIN();
        }
    });
}
template <int NG, int NWORK, unsigned long long BIAS, class M, class B, class TL, class S>
__device__ __forceinline__ void pipeline16(f32x4 (&ring)[kRing16], const ChunkCtx ctx, M&& work, B&& mid, TL&& tail, S&& side, const int nfill) {
    pipeline16x<NG, NWORK, BIAS, 0ull>(ring, ctx, work, mid, tail, side, [](auto) {}, nfill);
}

#define NSOS_RELU_WORD16(W, FLOOR) do { unsigned w_ = (W); asm volatile("v_pk_max_i16 %0, %0, %1" : "+v"(w_) : "s"(FLOOR)); (W) = w_; } while (0)

__device__ __forceinline__ void mov_slice16(u32x4& dst, const u32x4& src) {   // explicit copies at a chosen point of the stream
    asm volatile("v_mov_b32 %0, %4\n\tv_mov_b32 %1, %5\n\tv_mov_b32 %2, %6\n\tv_mov_b32 %3, %7"
                 : "=&v"(dst[0]), "=&v"(dst[1]), "=&v"(dst[2]), "=&v"(dst[3]) : "v"(src[0]), "v"(src[1]), "v"(src[2]), "v"(src[3]));
}

// ---- encoding tables ----------------------------------------------------------------------------------------------
// Slot j = 8 s + e of lane group q carries encoded feature f = 32 s + 8 q + e.  Table entry (q, j) = {sx, sy, sz, phase}:
// the feature's octave scale 2^k on its coordinate (0 on the other two) and phase 0 (sin) / 0.25 revolutions (cos); all
// zero for the raw coordinates (f < 3) and the pad slots, whose values are selected separately.
constexpr int kTabXyz = 0;            // [4 q][16 slots] f32x4 = 1 KiB, in the (otherwise unused) aux region of LDS
constexpr int kTabDir = 1024;         // [4 q][8 slots] f32x4 = 512 B
__device__ __forceinline__ f32x4 enc_table_entry(int f, int n_freqs) {
    f32x4 t = {0.0f, 0.0f, 0.0f, 0.0f};
    if (f >= 3 && f < 3 + 6 * n_freqs) {
        const int m = f - 3, k = m / 6, r = m % 6, coord = r % 3;
        t[coord] = (float)(1 << k);
        t[3] = r >= 3 ? 0.25f : 0.0f;
    }
    return t;
}

}  // namespace
