// Shared device machinery of the fused MLP kernels (fp32 exact-MFMA kernel: mlp_fused.hip; fp16/bf16-input
// kernel: mlp_lp.hip): compile-time loops, the hand-counted LDS A-operand pipeline, DMA issue slots, and the
// positional encoding in MFMA B-operand form.  gfx950 only.
#pragma once
#include "common.h"

#include <type_traits>

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x2 __attribute__((ext_vector_type(2)));

namespace nsos {

constexpr int kGroupFloats = 256;  // one LDS "group" = 64 lanes x 16 B = 1 KiB of MFMA A operands

// feature index held by (tile t, reg r, half hi) in the 32x32 accumulator layout
__host__ __device__ constexpr int acc_feature(int t, int r, int hi) { return 32 * t + (r & 3) + 8 * (r >> 2) + 4 * hi; }

#define NSOS_PIN() __builtin_amdgcn_sched_barrier(0)

template <int I, int N, class F>
__device__ __forceinline__ void static_for(F&& f) {
    if constexpr (I < N) {
        f(std::integral_constant<int, I>{});
        static_for<I + 1, N>(f);
    }
}

// ---- A-operand pipeline -------------------------------------------------------------------------
// One ds_read_b128 feeds a group of MFMAs (fp32 kernel: 4 MFMAs = 256 cycles; fp16 kernel: 2 MFMAs = 64
// cycles).  hipcc, left alone, issues each read right before its use and waits lgkmcnt(0), and with
// source-level prefetching it still waits for the YOUNGEST read.  So the reads are inline asm, invisible to the
// compiler's wait-count pass, with hand-counted waits (cdna_hip_programming.md section 5.7, form iii):
//   * a ring of RING slots: slot g%RING is re-loaded for group g+RING right after group g's MFMAs were
//     issued, so while group g computes, the reads of the next RING-1 groups are in flight (LDS returns in
//     order, so "lgkmcnt(n)" with n = number of younger reads == "group g has landed");
//   * the ring never drains at a chunk boundary: during groups NG-PRE..NG-PRE+RING-1 the first RING groups of
//     the NEXT chunk (already resident in the next LDS slot) are read into `nxt`, and become the ring at the
//     chunk's end (lgkmcnt(0) there is free: those reads are >= PRE-RING groups old);
//   * extra outstanding LGKM/VM operations the compiler may issue can only make a counted wait stricter
//     (in-order return within a class), never looser, so the counts are safe.
// ONE workgroup barrier per chunk, before group MID (mid()): every wave has drained the DMA pieces that must
// have landed and arrives; passing it proves (a) the next chunk has landed for every wave and (b) every wave has
// finished the previous chunk, whose slot the DMA pieces issued right after the barrier (one per MFMA shadow)
// overwrite.  The ring reads simply continue across the barrier.  tail() advances the stream bookkeeping (slot
// rotation, DMA source pointer) in the MFMA shadow after group NG-3, after the last use of the current values.
template <int OFF_BYTES>
__device__ __forceinline__ void lds_read_a(f32x4& dst, unsigned lds_addr) {
    asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(dst) : "v"(lds_addr), "i"(OFF_BYTES) : "memory");
}
template <int N>
__device__ __forceinline__ void lgkm_wait() {
    asm volatile("s_waitcnt lgkmcnt(%0)" ::"i"(N) : "memory");
}

// Issue schedule of one chunk's LDS reads, in program order: after the MFMAs of group g:
//   R(g+RING) if g+RING < NG (ring reload),  then  N(g-(NG-PRE)) if NG-PRE <= g < NG-PRE+RING (next chunk).
// younger_reads = number of reads issued after R(g) and before group g's MFMAs (0 for g < RING, whose data
// was waited for at the previous chunk's end).
template <int RING, int PRE>
__host__ __device__ constexpr int reads_after_group(int NG, int g) {
    return (g + RING < NG ? 1 : 0) + ((g >= NG - PRE && g < NG - PRE + RING) ? 1 : 0);
}
template <int RING, int PRE>
__host__ __device__ constexpr int younger_reads(int NG, int g) {
    if (g < RING) return 0;
    int n = reads_after_group<RING, PRE>(NG, g - RING) - 1;  // R(g) is the first read issued after group g-RING
    for (int h = g - RING + 1; h < g; ++h) n += reads_after_group<RING, PRE>(NG, h);
    return n;
}

struct ChunkCtx {     // what a chunk needs from the weight stream
    unsigned wl_cur;  // LDS byte address of this lane's A operands in the current chunk's slot
    unsigned wl_nxt;  // same for the next chunk's slot
};

// NG groups; group g's A operands are the f32x4 (16 B per lane) at byte offset g*1024 from ctx.wl_cur (which
// already includes lane*16).  work(g, a) issues the MFMAs of group g plus whatever rides in their shadows.
template <int NG, int RING, int PRE, int MID, class M, class B, class T>
__device__ __forceinline__ void a_pipeline(f32x4 (&ring)[RING], const ChunkCtx ctx, M&& work, B&& mid, T&& tail) {
    static_assert(NG >= MID + 1 && NG >= PRE && PRE >= RING + 3, "chunk length outside the barrier/preload schedule");
    // the barrier before group MID is what proves that the NEXT chunk has landed in LDS: its operands may only be
    // preloaded (from group NG - PRE on) after it
    static_assert(NG - PRE > MID, "chunk too short: the next chunk's operands would be preloaded before this chunk's barrier");
    f32x4 nxt[RING];
    static_for<0, NG>([&](auto ic) {
        constexpr int g = decltype(ic)::value;
        if constexpr (g == MID) {
            NSOS_PIN();
            mid();
            NSOS_PIN();
        }
        if constexpr (g >= RING) lgkm_wait<younger_reads<RING, PRE>(NG, g)>();
        NSOS_PIN();
        work(ic, ring[g % RING]);
        NSOS_PIN();
        if constexpr (g + RING < NG) lds_read_a<(g + RING) * 1024>(ring[g % RING], ctx.wl_cur);
        if constexpr (g >= NG - PRE && g < NG - PRE + RING) lds_read_a<(g - (NG - PRE)) * 1024>(nxt[g - (NG - PRE)], ctx.wl_nxt);
        if constexpr (g == NG - 3) {
            NSOS_PIN();
            tail();
            NSOS_PIN();
        }
    });
    lgkm_wait<0>();
    NSOS_PIN();
    // The preloaded operands have landed only NOW, but hipcc considers an asm output defined where the asm statement
    // ends: with the value live across the (register-hungry) code between two chunks it would spill or copy the
    // register right after the ds_read, i.e. before the data arrives.  Re-defining the values here, after the wait,
    // gives the long-lived value a definition point at which the data is valid (scripts/check_lds_ring.py verifies).
#pragma unroll
    for (int i = 0; i < RING; ++i) asm volatile("" : "+v"(nxt[i]));
    NSOS_PIN();
    // The next chunk starts at group 0, which uses slot 0: re-assignment in order is right for every NG.
#pragma unroll
    for (int i = 0; i < RING; ++i) ring[i] = nxt[i];
}

// DMA piece I of the chunk being prefetched is issued right after the I-th MFMA that follows the barrier.
template <int SLOT, int N_PIECES, class S>
__device__ __forceinline__ void dma_slot(S&& side) {
    if constexpr (SLOT >= 0 && SLOT < N_PIECES) {
        NSOS_PIN();
        side(SLOT);
        NSOS_PIN();
    }
}

// one 1 KiB piece global -> LDS (wave-uniform SGPR addressing; lane offset voff).  Inline asm so that the whole
// issue is a few SALU + 1 VMEM instruction (through the builtin hipcc spends ~10 VALU/readfirstlane
// instructions per piece, ~110 cycles).  M0 (LDS destination) is saved/restored inside the statement.
__device__ __forceinline__ void dma_1k(const void* src_uniform, unsigned dst_lds_uniform, unsigned voff) {
    unsigned keep;
    // s_nop 2: with the two s_mov in front of it, five wait states between a v_readfirstlane that produced one of the scalar
    // operands and the vector-memory instruction that reads it -- hipcc's hazard recognizer does not look at the SGPR operands
    // of inline asm (found in sem_wgrad16.hip as a memory fault when the compiler put the readfirstlane right in front)
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %1\n\ts_nop 2\n\tglobal_load_lds_dwordx4 %2, %3\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "s"(dst_lds_uniform), "v"(voff), "s"(src_uniform) : "memory");
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

// sin and cos of a positional-encoding argument (|a| = |coordinate| * 2^k, a few thousand radians at most).
// Branch-free three-term Cody-Waite reduction with fmaf (each step rounds once; the partial remainders are
// O(1), so the reduced argument is good to ~1 ulp for |a| < 2^15) + the Cephes minimax polynomials on
// [-pi/4, pi/4] (~1 ulp).  ocml's sincosf takes its Payne-Hanek branch for arguments this large, which made the
// encoding 10k cycles per tile AND desynchronised the four waves (the next barrier waits for the slowest).
// Arguments >= 2^15 (never produced by a scene-normalised NeRF) fall back to ocml.
__device__ __forceinline__ bool sincos_in_range(float a) { return fabsf(a) < 32768.0f; }
__device__ __forceinline__ void sincos_fast(float a, float& sn, float& cs) {      // branch-free; valid for |a| < 2^15
    const float q = __builtin_rintf(a * 0.636619772367581343f);          // 2/pi
    float r = __fmaf_rn(q, -1.57079637050628662109375f, a);              // pi/2 split into three fp32 terms
    r = __fmaf_rn(q, 4.37113900018624283e-8f, r);
    r = __fmaf_rn(q, 1.71512451613343730e-15f, r);
    const int n = (int)q;
    const float r2 = r * r;
    float ps = __fmaf_rn(r2, -1.9515295891e-4f, 8.3321608736e-3f);
    ps = __fmaf_rn(r2, ps, -1.6666654611e-1f);
    ps = __fmaf_rn(r * r2, ps, r);                                       // sin(r)
    float pc = __fmaf_rn(r2, 2.443315711809948e-5f, -1.388731625493765e-3f);
    pc = __fmaf_rn(r2, pc, 4.166664568298827e-2f);
    pc = __fmaf_rn(r2 * r2, pc, __fmaf_rn(r2, -0.5f, 1.0f));             // cos(r)
    const float s0 = (n & 1) ? pc : ps, c0 = (n & 1) ? ps : pc;
    sn = (n & 2) ? -s0 : s0;
    cs = ((n + 1) & 2) ? -c0 : c0;
}
__device__ __forceinline__ void sincos_pe(float a, float& sn, float& cs) {
    if (__builtin_expect(!sincos_in_range(a), 0)) {
        sincosf(a, &sn, &cs);
        return;
    }
    sincos_fast(a, sn, cs);
}

// Encoding of a 3-vector with L octaves for an MFMA B-operand layout in which encoded feature idx lives in
// half-wave HALF::of(idx).  The sin and cos of one (octave, coordinate) pair p = 3k+c are evaluated ONCE, by
// half p&1 (job t = p>>1 evaluates pair 2t in the lo half and 2t+1 in the hi half with one instruction stream);
// values the other half needs are handed over with a lane^32 exchange (ds_bpermute: LDS pipe, not VALU): 3L/2
// sincos evaluations per lane instead of 3L.  All index bookkeeping is compile-time.
template <int L, class HALF>
struct Enc {
    static_assert((3 * L) % 2 == 0, "even number of (octave, coordinate) pairs expected");
    static constexpr int kPairs = 3 * L, kJobs = kPairs / 2;
    __host__ __device__ static constexpr int sin_idx(int p) { return 3 + 6 * (p / 3) + p % 3; }
    __host__ __device__ static constexpr int cos_idx(int p) { return 6 + 6 * (p / 3) + p % 3; }

    float own_sn[kJobs], own_cs[kJobs], recv_sn[kJobs], recv_cs[kJobs];

    __device__ __forceinline__ void evaluate(const float (&x)[3], int hi) {
        // 1. every job on the branch-free fast path (independent chains: the compiler interleaves them)
        bool big = false;
        static_for<0, kJobs>([&](auto tc) {
            constexpr int t = decltype(tc)::value, pl = 2 * t, ph = 2 * t + 1;  // pair of the lo / hi half
            const float al = x[pl % 3] * (float)(1 << (pl / 3)), ah = x[ph % 3] * (float)(1 << (ph / 3));
            const float a = hi ? ah : al;
            big |= !sincos_in_range(a);
            sincos_fast(a, own_sn[t], own_cs[t]);
        });
        // 2. ONE cold block for arguments >= 2^15 (ocml's Payne-Hanek path; never taken by a scene-normalised NeRF).  With the
        //    range check inside every job the fifteen inlined slow paths sat in the middle of the hot code: 3 000 instructions
        //    and 36 branches to skip per tile.
        if (__builtin_expect(big, 0)) {
            static_for<0, kJobs>([&](auto tc) {
                constexpr int t = decltype(tc)::value, pl = 2 * t, ph = 2 * t + 1;
                const float al = x[pl % 3] * (float)(1 << (pl / 3)), ah = x[ph % 3] * (float)(1 << (ph / 3));
                const float a = hi ? ah : al;
                if (!sincos_in_range(a)) sincosf(a, &own_sn[t], &own_cs[t]);
            });
        }
        // 3. exchange what the OTHER half needs (hi needs lo's pair pl, lo needs hi's pair ph): all shuffles back to back
        static_for<0, kJobs>([&](auto tc) {
            constexpr int t = decltype(tc)::value, pl = 2 * t, ph = 2 * t + 1;
            if constexpr (HALF::of(sin_idx(pl)) == 1 || HALF::of(sin_idx(ph)) == 0)
                recv_sn[t] = __shfl_xor(own_sn[t], 32, NSOS_WAVE);
            if constexpr (HALF::of(cos_idx(pl)) == 1 || HALF::of(cos_idx(ph)) == 0)
                recv_cs[t] = __shfl_xor(own_cs[t], 32, NSOS_WAVE);
        });
    }
    // The same for the 16-bit kernels, whose encoded features are rounded to fp16 / bf16 anyway: the hardware's v_sin_f32 /
    // v_cos_f32 (argument in revolutions) instead of the ~27-instruction Cody-Waite + minimax evaluation.  The reduction keeps
    // full accuracy for every octave: u = x / 2pi is formed ONCE per coordinate as a two-term sum uh + ul; 2^k uh is exact,
    // so its fractional part is too, and 2^k ul is a small correction -- the revolution count is good to ~1e-7 whatever the
    // octave (a plain fract(x 2^k / 2pi) in fp32 is off by 5e-4 rad at 2^9 x 15).  Job t of the hi half evaluates pair
    // 2t + 1, whose coordinate is the lo half's NEXT one: the coordinates are rotated per lane once instead of selected per job.
    __device__ __forceinline__ void evaluate_hw(const float (&x)[3], int hi) {
        constexpr float kInvHi = 0.15915493667125702f, kInvLo = 6.4206382679e-9f;   // 1 / 2pi = kInvHi + kInvLo
        float uh[3], ul[3];
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            uh[c] = x[c] * kInvHi;
            ul[c] = __fmaf_rn(x[c], kInvHi, -uh[c]) + x[c] * kInvLo;
        }
        float yh[3], yl[3], ya[3];
#pragma unroll
        for (int j = 0; j < 3; ++j) {
            yh[j] = hi ? uh[(j + 1) % 3] : uh[j];
            yl[j] = hi ? ul[(j + 1) % 3] : ul[j];
            ya[j] = hi ? x[(j + 1) % 3] : x[j];
        }
        bool big = false;
        static_for<0, kJobs>([&](auto tc) {
            constexpr int t = decltype(tc)::value, pl = 2 * t, ph = 2 * t + 1, kl = pl / 3, kh = ph / 3, j = pl % 3;
            static_assert(ph % 3 == (j + 1) % 3, "the hi half's coordinate is the lo half's next one");
            const float sc = kl == kh ? (float)(1 << kl) : (hi ? (float)(1 << kh) : (float)(1 << kl));
            const float f = __builtin_amdgcn_fractf(yh[j] * sc);            // exact: a power-of-two multiple, then its fraction
            const float g = __fmaf_rn(yl[j], sc, f);
            own_sn[t] = __builtin_amdgcn_sinf(g);
            own_cs[t] = __builtin_amdgcn_cosf(g);
            big |= !sincos_in_range(ya[j] * sc);
        });
        if (__builtin_expect(big, 0)) {            // arguments >= 2^15: as in evaluate()
            static_for<0, kJobs>([&](auto tc) {
                constexpr int t = decltype(tc)::value, pl = 2 * t, ph = 2 * t + 1;
                const float al = x[pl % 3] * (float)(1 << (pl / 3)), ah = x[ph % 3] * (float)(1 << (ph / 3));
                const float a = hi ? ah : al;
                if (!sincos_in_range(a)) sincosf(a, &own_sn[t], &own_cs[t]);
            });
        }
        static_for<0, kJobs>([&](auto tc) {
            constexpr int t = decltype(tc)::value, pl = 2 * t, ph = 2 * t + 1;
            if constexpr (HALF::of(sin_idx(pl)) == 1 || HALF::of(sin_idx(ph)) == 0)
                recv_sn[t] = __shfl_xor(own_sn[t], 32, NSOS_WAVE);
            if constexpr (HALF::of(cos_idx(pl)) == 1 || HALF::of(cos_idx(ph)) == 0)
                recv_cs[t] = __shfl_xor(own_cs[t], 32, NSOS_WAVE);
        });
    }
    // value of encoded feature IDX as seen by a lane of half H (both compile-time); IDX must satisfy HALF::of(IDX) == H
    template <int IDX, int H>
    __device__ __forceinline__ float feature(const float (&x)[3]) const {
        constexpr EncSlot e = enc_slot(IDX, L);
        if constexpr (e.octave == -1) return x[e.coord];
        else if constexpr (e.octave == -2) return 0.0f;
        else {
            constexpr int p = 3 * e.octave + e.coord, t = p >> 1;
            if constexpr ((p & 1) == H) return e.is_cos ? own_cs[t] : own_sn[t];  // my half evaluated it
            else return e.is_cos ? recv_cs[t] : recv_sn[t];
        }
    }
};

}  // namespace nsos
