// K2-LP8: the 16-bit fused positional-encoding + MLP kernel (mlp_lp.hip) re-shaped for TWO waves per SIMD.
// Same network, same packed weight stream, same arithmetic and roundings as mlp_lp_kernel -- results are bit-identical
// to it (tests/test_gpu_parity.py::test_lp8_equals_lp4) -- and the same replaced reference code
// (models/embedder.py:34-48, models/nerf_mlp.py:67-100,179-215).
//
// Why.  mlp_lp_kernel runs one 512-register wave per SIMD with 64 points each.  Its phase table
// (profiles/r01/phase_lp_*.txt) shows the matrix pipe busy only 58 % of the time: the nine activation passes
// (v_accvgpr_read + v_cvt_pk + v_pk_max, ~5.8 cycles per instruction when ONE wave issues them), the positional
// encodings and the vector-ALU heads all run with the matrix pipe idle, and nothing else is resident on the SIMD to
// use it.  A single wave issues at most one instruction every ~4 cycles while the SIMD's vector ALU takes a wave64
// instruction every 2: a second wave on the SIMD doubles the rate of every VALU phase and fills the first wave's
// barrier / ring-restart bubbles in the MFMA phases.
//
// Shape.  Workgroup = 8 waves = 512 threads (two per SIMD, <= 256 registers each); wave = ONE 32-point column; tile =
// 256 points, so the weight stream is still amortised over 256 points per CU (L2 traffic unchanged).  Per wave:
// Z = 8 x 16 accumulators (128 registers), H = 16 packed K-slices (64), xyz encoding 16, A-operand ring 16 -- all VGPRs:
// no AGPRs, so the activation pass needs no v_accvgpr_read (see activate1).
//   * One ds_read_b128 per MFMA per wave = 128 B/clk/CU, half of the LDS's ds_read_b128 rate.
//   * The A-operand ring is CONTINUOUS across chunks (no second "next chunk" register set, which does not fit in 256
//     registers): slot of group g = (g + PHASE) % 4, and the reloads behind a chunk's last four groups already fetch
//     the next chunk's first four operands; every group waits with lgkmcnt(3).  PHASE is static at every call site
//     (chunks are 32, 34 or 16 groups; every part of the network is a multiple of 4 groups long).
//   * DMA: 36 pieces per chunk over 8 waves = 5 per wave (the surplus four re-copy piece 35: same bytes, same place).
//   * The two waves of a SIMD run in lock step (same chunk, one barrier per chunk for all eight waves).  A one-chunk lag
//     between them was built and measured (NSOS_LP8_LAG below): no gain.
//
// What it buys and what it cannot.  Cycles per 256-point tile (sem+coord, bf16, profiles/r02/b_phase_*): 144.4 k -> 117.8 k,
// matrix pipe busy 58 % -> 71 %: activation pass 2.9 k -> 1.24 k per layer, encodings and heads about halved, MFMA phases
// 92 % -> 94 % of issue rate.  Wall clock improves by only 6-8 % (bf16 fine pass 1.30 -> 1.40 PFLOP/s), because the chip
// is POWER-limited on this workload: a pure stream of the same MFMAs on random operands sustains 1.54-1.90 PFLOP/s on
// the whole chip, not 2.5 (scripts/ubench/mfma_power.hip, profiles/r02/c_mfma_power.txt: the clock falls to 1.5-1.7 GHz),
// and a busier pipe is answered with a lower clock (lp4 ~2.2 GHz, lp8 ~2.0 GHz).  This kernel runs at 74-81 % of what
// the bare matrix pipe sustains on ReLU-like data.
#include "lp_common.h"

using namespace nsos;
using namespace nsos::lp;

namespace {

constexpr int kRing8 = 4, kMid8 = 2, kDma8 = 5;

// continuous-ring A-operand pipeline (see the header comment).  ring[(g + PHASE) % RING] holds group g's operand.
// Groups NWORK..NG-1 of a padded chunk carry no MFMA: their operands are NOT read (a read whose result nobody consumes
// leaves its destination "dead" for hipcc, which then reuses the register while the read is still in flight -- the
// checker found exactly that); the counted waits follow the reads that were really issued.
template <int NG, int NWORK, int RING, int g>
constexpr bool issues_read_after() {   // does the slot freed by group g get reloaded?
    return g + RING < NG ? (g + RING < NWORK) : true;          // own chunk: only consumed groups; next chunk: always
}
template <int NG, int NWORK, int RING, int g, int h = (g - RING + 1 > 0 ? g - RING + 1 : 0)>
constexpr int younger_reads_c() {       // reads issued after group g's own and before its work (within this chunk)
    if constexpr (h >= g) return (g - RING + 1 < 0) ? -(g - RING + 1) : 0;   // groups of the previous chunk always reload: count them
    else return (issues_read_after<NG, NWORK, RING, h>() ? 1 : 0) + younger_reads_c<NG, NWORK, RING, g, h + 1>();
}
template <int NG, int RING, int PHASE, int MID, int NWORK = NG, class M, class B, class TL>
__device__ __forceinline__ void a_pipeline_c(f32x4 (&ring)[RING], const ChunkCtx ctx, M&& work, B&& mid, TL&& tail) {
    static_assert(NG >= RING + MID + 2, "chunk too short: the next chunk's operands would be read before this chunk's barrier");
    static_assert(NWORK >= RING && NWORK <= NG, "the first RING groups were read by the previous chunk");
    static_for<0, NG>([&](auto ic) {
        constexpr int g = decltype(ic)::value, slot = (g + PHASE) % RING;
        if constexpr (g == MID) {
            NSOS_PIN();
            mid();
            NSOS_PIN();
        }
        if constexpr (g < NWORK) lgkm_wait<younger_reads_c<NG, NWORK, RING, g>()>();
        NSOS_PIN();
        work(ic, ring[slot]);
        NSOS_PIN();
        if constexpr (g + RING < NG) {
            if constexpr (g + RING < NWORK) lds_read_a<(g + RING) * 1024>(ring[slot], ctx.wl_cur);
        } else {
            lds_read_a<(g + RING - NG) * 1024>(ring[slot], ctx.wl_nxt);   // resident: proven by this chunk's barrier
        }
        if constexpr (g == NG - 3) {
            NSOS_PIN();
            tail();
            NSOS_PIN();
        }
    });
}

template <int NT, class A>
__device__ __forceinline__ void pin_accumulators(A& acc) {
#pragma unroll
    for (int t = 0; t < NT; ++t) asm volatile("" : "+v"(acc[t]));
}

// H[2t+u] = pack16(relu?(Z[t][8u .. 8u+7]))   -- one batched VALU pass per layer (one column).
// The kernel contains NO inline asm with AGPR ("a") operands: with <= 256 registers per wave and no AGPR use hipcc selects
// the VGPR form of the MFMAs and treats the whole budget as one file (with any "a" operand it splits 128 + 128, and
// H + encoding + ring + temporaries do not fit in 128).  So the accumulators are ordinary VGPRs here, and the pass is
// v_cvt_pk + v_pk_max_i16 per packed word: no v_accvgpr_read (the most expensive half of mlp_lp_kernel's pass).
template <class T, int NT, bool RELU>
__device__ __forceinline__ void activate1(u32x4 (&H)[2 * NT], const f32x16 (&Z)[NT]) {
    asm volatile("s_nop 15\n\ts_nop 3" ::: "memory");   // MFMA result -> VALU read wait states (the asm below hides the reads)
#pragma unroll
    for (int t = 0; t < NT; ++t) {                       // 8 conversions, then their 8 clamps: no instruction waits for the previous one
#pragma unroll
        for (int k = 0; k < 8; ++k) H[2 * t + (k >> 2)][k & 3] = T::pack2(Z[t][2 * k], Z[t][2 * k + 1]);
        if constexpr (RELU)
#pragma unroll
            for (int k = 0; k < 8; ++k) { unsigned w = H[2 * t + (k >> 2)][k & 3]; asm volatile("v_pk_max_i16 %0, %0, 0" : "+v"(w)); H[2 * t + (k >> 2)][k & 3] = w; }
    }
}

// one packed activation word = max_i16(pack16(z0, z1), floor): floor = 0 is the ReLU of both halves, 0x80008000 (two int16
// minima) passes everything (feature_linear has no activation).  2 VALU instructions; they ride behind the MFMAs of the next
// tile pair (see the layer body).
// The two instructions of a word are issued one MFMA gap apart (or in batches when exposed): v_pk_max right behind its
// v_cvt_pk stalls the in-order wave -- and the MFMAs queued behind it -- for the conversion's latency (measured: 480 cycles
// for 32 such instructions back to back).
#define NSOS_RELU_WORD(W, FLOOR) do { unsigned w_ = (W); asm volatile("v_pk_max_i16 %0, %0, %1" : "+v"(w_) : "s"(FLOOR)); (W) = w_; } while (0)   // (vector elements bind neither to references nor to asm operands)
__device__ __forceinline__ void mov_slice(u32x4& dst, const u32x4& src) {   // explicit copies at a chosen point of the stream
    asm volatile("v_mov_b32 %0, %4\n\tv_mov_b32 %1, %5\n\tv_mov_b32 %2, %6\n\tv_mov_b32 %3, %7"
                 : "=&v"(dst[0]), "=&v"(dst[1]), "=&v"(dst[2]), "=&v"(dst[3]) : "v"(src[0]), "v"(src[1]), "v"(src[2]), "v"(src[3]));
}

// fp32 vector-ALU heads on fp32 accumulators held in VGPRs (rgb: NO = 3, semantics: NO = 2): this half-wave's partial
// chains part[o] = fma(w[o][f], relu(h[f]), part[o]) over the lane's NT*16 features, weights from the LDS copy of aux;
// same order as lp_common.h's heads_partial_f32 (bit-identical results).
template <int NT, int NO>
__device__ __forceinline__ void heads_partial_v(const f32x16 (&h)[NT], const float* w_lane, float (&part)[NO]) {
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            f32x4 w[NO];
#pragma unroll
            for (int o = 0; o < NO; ++o) w[o] = *reinterpret_cast<const f32x4*>(w_lane + o * 128 + t * 16 + q * 4);
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const float x = fmaxf(h[t][q * 4 + j], 0.0f);
#pragma unroll
                for (int o = 0; o < NO; ++o) part[o] = __fmaf_rn(w[o][j], x, part[o]);
            }
        }
}

// PROF: the phase stamps (nsos_mlp_profile_rays_lp / nsos_mlp_lp_set_stamp_buffer) are compiled into their own instantiations:
// in the production kernels they cost a scalar branch per phase and kept the stamp pointer and slot counter in scratch.
// Where the SAVE variant's twenty 16-byte stores of a point's sem_in row ride (compact path).  Sites: 0 = layer 0's chunk
// (the four encoding slices, k = 16..19), 1..3 = the semantic head's chunks, 4..6 = the first three chunks of layer 8 (relu(h7),
// half-tiles k = 0..15: H is overwritten only in that layer's last chunk).  Returns k for group g of the site, or -1.
#ifndef NSOS_SAVE_SCHED
#define NSOS_SAVE_SCHED 0
#endif
template <int SEM>
constexpr int save_slot(int site, int g) {
#if NSOS_SAVE_SCHED == 0
    // at most three per chunk, six groups apart
    if (site == 0) return (g >= 3 && g < 15 && g % 3 == 0) ? 16 + g / 3 - 1 : -1;
    if (site >= 1 && site <= 3) {
        const int ch = site - 1, cnt = SEM == 2 ? (ch == 2 ? 2 : 3) : 4, first = SEM == 2 ? 3 * ch : 4 * ch, step = ch == 2 ? 4 : 6;
        if (SEM != 2 && ch == 2) return -1;
        return (g >= 3 && (g - 3) % step == 0 && (g - 3) / step < cnt) ? first + (g - 3) / step : -1;
    }
    const int ch = site - 4;
    return (g >= 3 && (g - 3) % 6 == 0 && (g - 3) / 6 < (ch == 2 ? 2 : 3)) ? 8 + 3 * ch + (g - 3) / 6 : -1;
#elif NSOS_SAVE_SCHED == 1
    // whole 128-byte lines (four consecutive k) in consecutive groups of ONE chunk: sem1, sem2, L8c1, L8c2; x63 in layer 0
    if (site == 0) return (g >= 3 && g < 7) ? 16 + g - 3 : -1;
    const int line = site == 1 ? 0 : site == 2 ? 1 : site == 4 ? 2 : site == 5 ? 3 : -1;
    return (line >= 0 && g >= 3 && g < 7) ? 4 * line + g - 3 : -1;
#elif NSOS_SAVE_SCHED == 3
    // like 0, but layer 8's first chunk stays free (the eight stores of the hidden activations land right in front of it)
    if (site == 0) return (g >= 3 && g < 15 && g % 3 == 0) ? 16 + g / 3 - 1 : -1;
    if (site >= 1 && site <= 3) {
        const int ch = site - 1, cnt = SEM == 2 ? (ch == 2 ? 2 : 3) : 4, first = SEM == 2 ? 3 * ch : 4 * ch, step = ch == 2 ? 4 : 6;
        if (SEM != 2 && ch == 2) return -1;
        return (g >= 3 && (g - 3) % step == 0 && (g - 3) / step < cnt) ? first + (g - 3) / step : -1;
    }
    const int ch = site - 4;
    if (ch == 0) return -1;
    return (g >= 3 && (g - 3) % 6 == 0 && (g - 3) / 6 < 4) ? 8 + 4 * (ch - 1) + (g - 3) / 6 : -1;
#elif NSOS_SAVE_SCHED == 2
    // whole lines, but two groups apart
    if (site == 0) return (g >= 3 && g < 11 && (g - 3) % 2 == 0) ? 16 + (g - 3) / 2 : -1;
    const int line = site == 1 ? 0 : site == 2 ? 1 : site == 4 ? 2 : site == 5 ? 3 : -1;
    return (line >= 0 && g >= 3 && g < 11 && (g - 3) % 2 == 0) ? 4 * line + (g - 3) / 2 : -1;
#endif
}

template <class T, int SEM, bool SAVE = false, bool PROF = false>
__global__ __launch_bounds__(512, 1) void mlp_lp8_kernel(const LpParams P) {
    extern __shared__ __attribute__((aligned(16))) unsigned char lds[];  // 4 x 36 KiB weight slots + 4 KiB head weights + 6 KiB output stage
    const int lane0 = threadIdx.x & 63, wave = threadIdx.x >> 6;
    constexpr int NCH = lp_chunks(SEM);
    constexpr int C = SEM ? 6 : 4;

    // ---- weight stream: slots rotate (c0 = chunk cur, c1 = cur+1, c2 = cur+2, c3 = being filled with cur+3)
    const unsigned lds_base = (unsigned)(size_t)(__attribute__((address_space(3))) unsigned char*)lds;
    const int wave_s = __builtin_amdgcn_readfirstlane(wave);
    const unsigned voff = (unsigned)(lane0 * 16);
    auto lane_addr = [&](int b) { return lds_base + (unsigned)(b * kSlotBytes + lane0 * 16); };
    auto slot_addr = [&](int b) { return lds_base + (unsigned)(b * kSlotBytes); };
    unsigned c0 = lane_addr(0), c1 = lane_addr(1), c2 = lane_addr(2), c3 = lane_addr(3);
    unsigned d0 = slot_addr(0), d1 = slot_addr(1), d2 = slot_addr(2), d3 = slot_addr(3);
    // byte offset of this wave's i-th piece inside a chunk / slot: piece wave + 8 i (the surplus ones re-copy piece 35)
    const unsigned woff = (unsigned)wave_s * 1024u;
    const unsigned wlast = wave_s + 32 < kSlotGroups ? woff + 32768u : (unsigned)(kSlotGroups - 1) * 1024u;
    auto poff = [&](int i) { return i < 4 ? woff + 8192u * (unsigned)i : wlast; };
    // ---- Experimental (-DNSOS_LP8_LAG; NOT the shipped build): the two waves of a SIMD one chunk apart.  Waves 4..7
    // ("lagging") start one barrier interval late, so that a wave's activation pass falls into an interval in which its
    // SIMD partner runs MFMAs.  Slot protocol with a lag of one chunk: leaders at chunk c, laggers at c-1; resident: c-1,
    // c, c+1; the fourth slot, which held c-2 (both groups done with it: proven by the barrier), is refilled with c+2 -- in
    // each group's own numbering "cur+2 -> slot of cur+2" for leaders and "cur+3 -> slot of cur+3" for laggers: the same
    // chunk, the same slot, issued in the same interval, 20 + 20 pieces; a chunk then has ONE interval to land, so the
    // barrier waits for all of the wave's pieces (the lock-step build uses that scheme too: it is what the fp32 kernel does).
    // Measured (profiles/r02/c_lp8_lag_vs_lockstep.txt): bit-identical results, no win -- without priorities the two waves'
    // MFMAs interleave evenly and each activation pass is then exposed on its own (layer 11.5 k cycles vs 10.5 k in lock
    // step); with s_setprio 3 around the pass's interval the layer comes back to 10.6 k, and the long VALU phases
    // (encodings) serialise between the groups; wall clock 0.5-1.5 % behind lock step.
#ifdef NSOS_LP8_LAG
    const bool lagging = wave_s >= 4;
#else
    const bool lagging = false;   // shipped: both waves of a SIMD in the same chunk (every wave fills 'cur+2')
#endif
    const unsigned char* const src_end = P.chunks + (size_t)NCH * kSlotBytes;
    const unsigned char* srcf = P.chunks + (size_t)((lagging ? 3 : 2) % NCH) * kSlotBytes;
    auto dma_piece = [&](const unsigned char* src_chunk, unsigned dst_slot, int i) {
        // the operands are wave-uniform by construction, but under SGPR pressure hipcc keeps some of the stream bookkeeping in
        // VGPRs and then hands a VGPR to the asm's "s" operands: make the uniformity explicit where it is consumed
        const unsigned long long sp = (unsigned long long)(src_chunk + poff(i));
        const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)sp), hi = __builtin_amdgcn_readfirstlane((unsigned)(sp >> 32));
        const unsigned dst = __builtin_amdgcn_readfirstlane(dst_slot + poff(i));
        dma_1k(reinterpret_cast<const void*>(((unsigned long long)hi << 32) | lo), dst, voff);
    };
    // per chunk: the fill source / destination and this wave's piece offset are made scalar ONCE (at the barrier, right
    // before the five pieces that use them), the pieces add constants on the scalar ALU
    unsigned fill_lo = 0, fill_hi = 0, fill_dst = 0, fill_w = 0, fill_wlast = 0;
    auto side = [&](int i) {
        const unsigned off = i < 4 ? fill_w + 8192u * (unsigned)i : fill_wlast;
        const unsigned long long sp = (((unsigned long long)fill_hi << 32) | fill_lo) + off;
        dma_1k(reinterpret_cast<const void*>(sp), fill_dst + off, voff);
    };
    auto mid = [&]() {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        const unsigned long long sp = (unsigned long long)srcf;
        fill_lo = __builtin_amdgcn_readfirstlane((unsigned)sp);
        fill_hi = __builtin_amdgcn_readfirstlane((unsigned)(sp >> 32));
        fill_dst = __builtin_amdgcn_readfirstlane(lagging ? d3 : d2);
        fill_w = __builtin_amdgcn_readfirstlane(woff);
        fill_wlast = __builtin_amdgcn_readfirstlane(wlast);
    };
    auto tail = [&]() {
        const unsigned tc = c0, td = d0;
        c0 = c1; c1 = c2; c2 = c3; c3 = tc;
        d0 = d1; d1 = d2; d2 = d3; d3 = td;
        srcf += kSlotBytes;
        if (srcf == src_end) srcf = P.chunks;
    };
    auto ctx = [&]() { return ChunkCtx{c0, c1}; };

    f32x4 ring[kRing8];
#pragma unroll
    for (int k = 0; k < 3; ++k)
#pragma unroll
        for (int i = 0; i < kDma8; ++i)
            dma_piece(P.chunks + (size_t)(k % NCH) * kSlotBytes, k == 0 ? d0 : (k == 1 ? d1 : d2), i);
    const unsigned* const aux_l = reinterpret_cast<const unsigned*>(lds + kSlots * kSlotBytes);
    if (threadIdx.x < kAuxWords / 4)
        *reinterpret_cast<u32x4*>(lds + kSlots * kSlotBytes + threadIdx.x * 16) = reinterpret_cast<const u32x4*>(P.aux)[threadIdx.x];
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    static_for<0, kRing8>([&](auto ic) { lds_read_a<decltype(ic)::value * 1024>(ring[decltype(ic)::value], c0); });
    NSOS_PIN();
    if (lagging) __builtin_amdgcn_s_barrier();   // the laggers' extra interval (pairs with the leaders' first chunk barrier)
    NSOS_PIN();

    // one chunk: NG groups = NG A operands `a0 + g` of a part with NT output tiles and NB leading bias operands; the
    // K-slice of operand a >= NB is (a - NB) / NT, its tile (a - NB) % NT.  zf_c != 0: the part starts the accumulation.
    auto run_chunk = [&](auto ng_c, auto nt_c, auto nb_c, auto a0_c, auto nwork_c, auto zf_c, auto ph_c, auto& acc, auto&& bsel, auto&& ride) {
        constexpr int NG = decltype(ng_c)::value, NT = decltype(nt_c)::value, NB = decltype(nb_c)::value;
        constexpr int A0 = decltype(a0_c)::value, NWORK = decltype(nwork_c)::value, PH = decltype(ph_c)::value;
        constexpr bool ZF = decltype(zf_c)::value != 0;
        a_pipeline_c<NG, kRing8, PH, kMid8, NWORK>(ring, ctx(), [&](auto ic, const f32x4& a32) {
            constexpr int g = decltype(ic)::value, a = A0 + g;
            const u32x4 aop = __builtin_bit_cast(u32x4, a32);
            if constexpr (g < NWORK) {
                if constexpr (a < NB) {
                    const f32x16 zero = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
                    const u32x4 ones = {T::kOnes, T::kOnes, T::kOnes, T::kOnes};
                    acc[a] = T::mfma(aop, ones, zero);
                } else {
                    constexpr int s = (a - NB) / NT, t = (a - NB) % NT;
                    if constexpr (ZF && s == 0) {
                        const f32x16 zero = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
                        acc[t] = T::mfma(aop, bsel(std::integral_constant<int, s>{}), zero);
                    } else {
                        acc[t] = T::mfma(aop, bsel(std::integral_constant<int, s>{}), acc[t]);
                    }
                }
            }
            dma_slot<g - kMid8, kDma8>(side);
            ride(ic);
        }, mid, tail);
        // pin the accumulators here: without a use at the chunk's end LLVM sinks whole chunks of MFMAs below later
        // branches and keeps their A operands (pending ring registers!) alive in scratch
        pin_accumulators<NT>(acc);
    };
    auto no_ride = [](auto) {};
    // one tile-pair chunk of a hidden layer (stream layout kPair8): operands [bias t0, bias t1, (s0,t0), (s0,t1), (s1,t0), ...];
    // accumulators zp[0..1]; the B operand of slice s is hsel(s)
    auto run_pair = [&](auto ph_c, auto& zp, auto&& hsel, auto&& ride) {
        constexpr int PH = decltype(ph_c)::value;
        a_pipeline_c<34, kRing8, PH, kMid8>(ring, ctx(), [&](auto ic, const f32x4& a32) {
            constexpr int g = decltype(ic)::value;
            const u32x4 aop = __builtin_bit_cast(u32x4, a32);
            if constexpr (g < 2) {
                const f32x16 zero = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
                const u32x4 ones = {T::kOnes, T::kOnes, T::kOnes, T::kOnes};
                zp[g] = T::mfma(aop, ones, zero);
            } else {
                constexpr int s = (g - 2) >> 1, t = (g - 2) & 1;
                zp[t] = T::mfma(aop, hsel(std::integral_constant<int, s>{}), zp[t]);
            }
            dma_slot<g - kMid8, kDma8>(side);
            ride(ic);
        }, mid, tail);
        pin_accumulators<2>(zp);
    };
#ifdef NSOS_LP8_LAG
    auto hi_prio = [] { NSOS_PIN(); __builtin_amdgcn_s_setprio(3); NSOS_PIN(); };
    auto lo_prio = [] { NSOS_PIN(); __builtin_amdgcn_s_setprio(0); NSOS_PIN(); };
#else
    auto hi_prio = [] {};
    auto lo_prio = [] {};
#endif
#define IC(n) std::integral_constant<int, (n)> {}

    if constexpr (PROF)
        if (P.prof && blockIdx.x < 2 && lane0 == 0) P.prof[(blockIdx.x * 8 + wave_s) * kProfSlots + kProfSlots - 2] = __builtin_readcyclecounter();
    for (int tile = blockIdx.x; tile < P.n_tiles; tile += gridDim.x) {
        // Lane-dependent constants of the tile body (half-wave selects of the encoder's octave scales, point offsets, stage
        // addresses) are loop invariants: LICM hoists them, they then live across every MFMA chunk of the tile and end up in
        // scratch.  The lane id is laundered once per tile, so they are re-derived (a handful of VALU) instead of kept.
        int lane = lane0;
        asm volatile("" : "+v"(lane));
        const int pj = lane & 31, kg = lane >> 5;
        int stamp_k = 0;
        auto stamp = [&]() {  // diagnostics only: one s_memtime per phase of the second tile of blocks 0..1 (16 rows of
            // kProfSlots stamps, the buffer contract of nsos_mlp_profile_rays_lp; the last two slots of a row hold the
            // wave's first and last cycle in the kernel)
            if constexpr (PROF) {
                if (P.prof && tile == (int)(blockIdx.x + gridDim.x) && blockIdx.x < 2) {
                    const unsigned long long t = __builtin_readcyclecounter();
                    if (lane == 0 && stamp_k < kProfSlots) P.prof[(blockIdx.x * 8 + wave_s) * kProfSlots + stamp_k] = t;
                }
                ++stamp_k;
            }
        };
        stamp();  // 0: tile start
        // ---- this lane's point: tile*256 + wave*32 + pj
        // 32-bit point indices (the entry point sends launches of >= 2^31 points to the round-1 kernel); the wave's first point
        // and how many of its 32 points exist are wave-uniform (SGPRs), only `+ pj` is per lane.
        const int n_pts = (int)P.n_pts;
        const int wave_first = tile * kTilePts + wave_s * 32;
        const int n_here = n_pts - wave_first >= 32 ? 32 : (n_pts - wave_first < 0 ? 0 : n_pts - wave_first);
        const bool exists = pj < n_here;
        const int gp = wave_first + pj;
        const int gc = exists ? gp : n_pts - 1;
        const int ray = (int)((unsigned)gc / (unsigned)P.n_samples);
        // `ray` is needed again at the far end of the tile (direction encoding, the NaN check): parked in the wave's output stage
        // (LDS, unused until the tile's last store) instead of a register the compiler would move to scratch
        int* const park = reinterpret_cast<int*>(lds + kSlots * kSlotBytes + kAuxWords * 4) + wave_s * 192 + lane;
        *park = ray;
        bool save_ok = false;
        unsigned long long save_grp = 0;
        unsigned save_off = 0;
        // store K of the tile-major sem_in: scalar group base (+ a multiple of 4 KiB) + the lane's constant 32-bit offset + an
        // immediate -- one VGPR of address for all twenty stores (generic pointers: twenty 64-bit lane addresses, i.e. spills)
        auto save_store = [&save_grp, &save_off](auto kc, const u32x4& v) {   // (named captures: an operand of an asm statement alone does not capture)
            constexpr int K = decltype(kc)::value;
            const unsigned off = save_off;
            unsigned long long b = save_grp + (unsigned long long)((K * 1024) & ~4095);
            asm volatile("global_store_dwordx4 %0, %1, %2 offset:%3" : : "v"(off), "v"(v), "s"(b), "i"((K * 1024) & 4095) : "memory");
        };
        if constexpr (SAVE) {
            save_ok = exists;     // (this kernel's SAVE variant is the compact one: sem_in16 / sem_hid16 are never NULL, see forward_rays_lp)
            // TILE-MAJOR sem_in (include/nerf_sos_hip.h, nsos_mlp_forward_rays_save16_lp): the 1 KiB a wave's store instruction holds
            // -- octet 2K + kg of its 32 points -- is contiguous: [group of 32 points][store K 0..19][kg][point][8 channels].
            // Row-major, the same instruction touched 32 lines 640 B apart (~32 cycles of the CU's address path each: what was
            // left of the training variant's overhead); the weight-gradient kernel stages sem_in through LDS anyway and places
            // the pieces itself.
            save_grp = reinterpret_cast<unsigned long long>(P.sem_in16 + (long long)(wave_first >> 5) * 5120);   // wave-uniform: scalar unit
            save_off = (unsigned)(pj * 4 + kg * 128) * 4u;                                                              // bytes
        }
        u32x4 ex[4];
        {
            const float z = P.z_vals[gc];
            float x[3];
#pragma unroll
            for (int k = 0; k < 3; ++k) {
                const float m = P.rays_d[3ll * ray + k] * z;  // models/sampler.py:70,166 (mul, then add)
                x[k] = P.rays_o[3ll * ray + k] + m;
            }
            Enc<NSOS_XYZ_FREQS, SliceHalf> e;
            e.evaluate_hw(x, kg);
            ex[0] = enc_slice<T, NSOS_XYZ_FREQS, 0, true>(e, x, kg);
            ex[1] = enc_slice<T, NSOS_XYZ_FREQS, 1, true>(e, x, kg);
            ex[2] = enc_slice<T, NSOS_XYZ_FREQS, 2, true>(e, x, kg);
            ex[3] = enc_slice<T, NSOS_XYZ_FREQS, 3, true>(e, x, kg);  // feature 63 (pad) = 1.0: layer-0 bias
        }

        f32x16 Z[8];
        u32x4 H[16];
        float sigma = 0.0f, sem_out[2] = {0.0f, 0.0f};
        auto from_ex = [&](auto sc) { return ex[decltype(sc)::value]; };
        auto from_H = [&](auto sc) { return H[decltype(sc)::value]; };
        // SAVE, compact sem_in: half-tile k of relu(h7) (features 32t + 16j + {0..15}, t = k / 2, j = k & 1) as one 16-byte
        // store per lane.  Full-sector stores: a lane holds, per 32-feature tile, the 4-feature quads q0,q2,q4,q6 (lane half 0) or
        // q1,q3,q5,q7 (half 1) of its point -- written as they are, every store put 8 B per lane = two HALF 32-byte sectors
        // per point (and the training kernel was bound by write transactions: +33 k cycles per tile).  Two v_permlane32_swap
        // per word pair hand q2 <-> q1 and q6 <-> q5 across the halves, after which half 0 owns features 0..7 / 16..23 and
        // half 1 owns 8..15 / 24..31 of the tile: one dwordx4 per lane, 32 contiguous bytes per point and instruction.
        auto store_h7 = [&](auto kc) {
            constexpr int k = decltype(kc)::value, t = k >> 1, j = k & 1;
            static_assert(k >= 0 && k < 16, "sixteen half-tile stores of relu(h7)");
            if constexpr (SAVE) {
                const auto s0 = __builtin_amdgcn_permlane32_swap(H[2 * t + j][0], H[2 * t + j][2], false, false);
                const auto s1 = __builtin_amdgcn_permlane32_swap(H[2 * t + j][1], H[2 * t + j][3], false, false);
#ifndef NSOS_LP8_SKIP_IN    // (A/B builds only: scripts/diag/build_variant.sh)
                if (save_ok)   // halves: features 32t + 16j + 8kg + {0..7} = words 16t + 8j + 4kg + {0..3}
                    save_store(IC(k), u32x4{s0[0], s1[0], s0[1], s1[1]});
#endif
            }
        };

        stamp();  // 1: inputs + xyz encoding
        hi_prio();   // (lag build only) this wave issues its MFMAs first in the interval that ends with its activation pass
        run_chunk(IC(32), IC(8), IC(0), IC(0), IC(32), IC(1), IC(0), Z, from_ex, [&](auto gc_) {
            constexpr int g = decltype(gc_)::value;      // SAVE, compact sem_in: the x63 slices (features 16s + 8kg + {0..7}; 63 is the 1.0 pad)
            if constexpr (SAVE && SEM != 0 && save_slot<SEM>(0, g) >= 16) {
                constexpr int sl = save_slot<SEM>(0, g) - 16;
                if (save_ok) save_store(IC(16 + sl), ex[sl]);
            }
        });
        stamp();  // 2: L0 MFMAs
        activate1<T, 8, true>(H, Z);
        lo_prio();
        stamp();  // 3: L0 activation
        // Layers 1..4 and 6..8 are tile-pair-major with a riding activation; the skip layer 5 (its x63 part follows the h part in
        // every accumulator) stays slice-major with one exposed pass.  Two runtime loops with ONE uniform body each (a single
        // loop with both bodies made hipcc reconcile their register assignments with 160 v_mov per layer at the back edge).
        // The exposed rest of a pair layer's activation (tiles 6,7 = the accumulators Zp[1], 16 packed words) can wait: the NEXT
        // pair layer's first chunk accumulates into Zp[0], reads the slices in order (slice 12 = the first of these words
        // only from group 26 on) and has its MFMA gaps free -- so inside a run of pair layers (1..4, 6..7) the conversion rides
        // there (groups 4..21) instead of standing between the layers with the matrix pipe idle.  The last layer of a run
        // (4, 7, 8: what follows needs all of H at once) keeps the exposed pass.  Same values, same order: bit-identical.
        f32x16 Zp[2][2];
        // (a value that is only CONDITIONALLY read at the top of a run is live, for the compiler, from its last definition --
        // through layers 0 and 5, whose 128 accumulators leave no room for 32 more registers: an empty asm "defines" the
        // accumulators afresh where they are dead, at no cost)
        auto dead = [&]() {
            asm volatile("" : "=v"(Zp[0][0]), "=v"(Zp[0][1]), "=v"(Zp[1][0]), "=v"(Zp[1][1]));
        };
        auto pair_layer = [&](const int l, const bool tail_pending, const bool leave_tail) {
            // Tile-pair-major: chunk c accumulates output tiles 2c, 2c+1 over all 16 input slices into Zp[c & 1]; the
            // activation of the PREVIOUS pair rides behind this chunk's MFMAs (2 VALU per packed word, one word per MFMA
            // gap: the accumulators are VGPRs, nothing to read back).  The layer's input H stays live until its last
            // chunk, so finished slices wait in Ho[0..7] (tiles 0..3) and move into H behind the last chunk's MFMAs, each
            // right after its final use there; tiles 4,5 are activated straight into H[8..11] once those are dead
            // (group 26 on); only tiles 6,7 (16 words) remain for after the chunk.  Per accumulator the MFMA order is
            // the round-1 kernel's (bias, slices 0..15): results stay bit-identical.
            const unsigned floor = l < 8 ? 0u : 0x80008000u;     // feature_linear (l == 8) has no activation
            u32x4 Ho[8];
            auto ride_tail = [&](auto gc_) {     // the previous layer's tiles 6,7 -> H[12..15] (always a ReLU layer: floor 0)
                constexpr int g = decltype(gc_)::value;
                if constexpr (g >= 4 && g <= 21) {
                    if (tail_pending) {
                        if constexpr (g >= 5 && g < 21) {
                            constexpr int k = g - 5;
                            NSOS_RELU_WORD(H[12 + (k >> 2)][k & 3], 0u);
                        }
                        if constexpr (g >= 4 && g < 20) {
                            constexpr int k = g - 4, tt = k >> 3, u = (k >> 2) & 1, q = k & 3;
                            H[12 + 2 * tt + u][q] = T::pack2(Zp[1][tt][8 * u + 2 * q], Zp[1][tt][8 * u + 2 * q + 1]);
                        }
                        if constexpr (g == 21) {
                            NSOS_RELU_WORD(H[11][2], 0u);
                            NSOS_RELU_WORD(H[11][3], 0u);
                        }
                    }
                }
            };
            auto ride_act = [&](auto gc_, auto src_c, auto base_c) {   // 16 words of pair buffer src -> Ho[base .. base+3]: convert in groups 4..19, clamp one group later
                constexpr int g = decltype(gc_)::value, SRC = decltype(src_c)::value, BASE = decltype(base_c)::value;
                if constexpr (g >= 5 && g < 21) {
                    constexpr int k = g - 5, tt = k >> 3, u = (k >> 2) & 1, q = k & 3;
                    NSOS_RELU_WORD(Ho[BASE + 2 * tt + u][q], floor);
                }
                if constexpr (g >= 4 && g < 20) {
                    constexpr int k = g - 4, tt = k >> 3, u = (k >> 2) & 1, q = k & 3;
                    Ho[BASE + 2 * tt + u][q] = T::pack2(Zp[SRC][tt][8 * u + 2 * q], Zp[SRC][tt][8 * u + 2 * q + 1]);
                }
            };
            // (SAVE: the second half of relu(h7)'s stores rides in layer 8's first three chunks -- a scalar branch per site)
            auto ride_save = [&](auto gc_, auto ch_c) {
                constexpr int g = decltype(gc_)::value, CH = decltype(ch_c)::value;
                if constexpr (SAVE && SEM != 0 && save_slot<SEM>(4 + CH, g) >= 0)
                    if (l == 8) store_h7(IC(save_slot<SEM>(4 + CH, g)));
            };
            run_pair(IC(0), Zp[0], from_H, [&](auto gc_) { ride_tail(gc_); ride_save(gc_, IC(0)); });
            run_pair(IC(2), Zp[1], from_H, [&](auto gc_) { ride_act(gc_, IC(0), IC(0)); ride_save(gc_, IC(1)); });
            run_pair(IC(0), Zp[0], from_H, [&](auto gc_) { ride_act(gc_, IC(1), IC(4)); ride_save(gc_, IC(2)); });
            run_pair(IC(2), Zp[1], from_H, [&](auto gc_) {
                constexpr int g = decltype(gc_)::value;
                if constexpr (g >= 4 && g <= 18 && (g & 1) == 0) {       // slice s = (g - 4) / 2 was last used by groups 2+2s, 3+2s
                    constexpr int sl = (g - 4) >> 1;
                    mov_slice(H[sl], Ho[sl]);
                }
                if constexpr (g >= 27) {                                 // clamp the two words converted one group earlier
                    static_for<0, 2>([&](auto jc) {
                        constexpr int k = (g - 27) * 2 + decltype(jc)::value, tt = k >> 3, u = (k >> 2) & 1, q = k & 3;
                        NSOS_RELU_WORD(H[8 + 2 * tt + u][q], floor);
                    });
                }
                if constexpr (g >= 26) {                                 // H[8..11] are dead: tiles 4,5 (Zp[0]) go there directly
                    static_for<0, 2>([&](auto jc) {
                        constexpr int k = (g - 26) * 2 + decltype(jc)::value, tt = k >> 3, u = (k >> 2) & 1, q = k & 3;
                        H[8 + 2 * tt + u][q] = T::pack2(Zp[0][tt][8 * u + 2 * q], Zp[0][tt][8 * u + 2 * q + 1]);
                    });
                }
            });
            stamp();  // 2 + 2l: MFMAs of layer l (with the riding activation of tiles 0..5)
            if (!leave_tail) {
                asm volatile("s_nop 15\n\ts_nop 3" ::: "memory");   // MFMA result -> VALU read wait states (asm hides the reads)
#pragma unroll
                for (int k = 0; k < 16; ++k) {                       // tiles 6,7: 16 conversions, then 16 clamps (+ the last two of tiles 4,5)
                    const int tt = k >> 3, u = (k >> 2) & 1, q = k & 3;
                    H[12 + 2 * tt + u][q] = T::pack2(Zp[1][tt][8 * u + 2 * q], Zp[1][tt][8 * u + 2 * q + 1]);
                }
                NSOS_RELU_WORD(H[11][2], floor);
                NSOS_RELU_WORD(H[11][3], floor);
#pragma unroll
                for (int k = 0; k < 16; ++k) NSOS_RELU_WORD(H[12 + (k >> 2)][k & 3], floor);
            }
            stamp();  // 3 + 2l: the exposed rest of the activation (tiles 6,7), unless it rides in the next layer
        };
        dead();
#pragma unroll 1
        for (int l = 1; l <= 4; ++l) pair_layer(l, l != 1, l != 4);
        dead();
        {   // l = 5
            // skip layer: slice-major over all 8 tiles (its x63 part follows the h part in every accumulator), one
            // exposed activation pass
            run_chunk(IC(34), IC(8), IC(8), IC(0), IC(34), IC(0), IC(0), Z, from_H, no_ride);
            run_chunk(IC(34), IC(8), IC(8), IC(34), IC(34), IC(0), IC(2), Z, from_H, no_ride);
            run_chunk(IC(34), IC(8), IC(8), IC(68), IC(34), IC(0), IC(0), Z, from_H, no_ride);
            run_chunk(IC(34), IC(8), IC(8), IC(102), IC(34), IC(0), IC(2), Z, from_H, no_ride);
            run_chunk(IC(32), IC(8), IC(0), IC(0), IC(32), IC(0), IC(0), Z, from_ex, no_ride);  // skip connection
            stamp();  // 2 + 2l: MFMAs of layer l
            activate1<T, 8, true>(H, Z);
            stamp();  // 3 + 2l: activation pass
        }
        dead();
#pragma unroll 1
        for (int l = 6; l <= 8; ++l) {
            pair_layer(l, l == 7, l == 6);
            if (l == 7) {
                dead();      // layer 7 finished its own tail: nothing of Zp is read again before layer 8 redefines it
                // sigma head: dot of the packed activations with packed weights (models/nerf_mlp.py:77)
                // four independent chains (one per packed word q of a slice), then (p0 + p1) + (p2 + p3): one chain of 64 dependent
                // v_dot2c was 1.8 k cycles per tile with nothing else for the SIMD to issue (round 3; the round-1 kernel sums in
                // the same order: bit-identical)
                const unsigned* aw = aux_l + kAuxAlphaW + kg * 64;
                float pq[4] = {kg ? 0.0f : __builtin_bit_cast(float, aux_l[kAuxScalars]), 0.0f, 0.0f, 0.0f};
#pragma unroll
                for (int s = 0; s < 16; ++s) {
                    const u32x4 w = *reinterpret_cast<const u32x4*>(aw + 4 * s);
#pragma unroll
                    for (int q = 0; q < 4; ++q) pq[q] = T::dot2(H[s][q], w[q], pq[q]);
                }
                asm volatile("s_nop 3" ::: "memory");  // v_dot2c result -> non-dot VALU read: 3 wait states hipcc cannot see (asm)
                const float pa = (pq[0] + pq[1]) + (pq[2] + pq[3]);
                sigma = both_halves(pa);
                if constexpr (SEM != 0) {  // semantic head (models/nerf_mlp.py:79-80)
                    f32x16 sacc[4];
                    // SAVE (compact operands): the twenty 16-byte stores of a point's sem_in row [relu(h7) | x63 | 1] ride in MFMA
                    // shadows EARLY in their chunks -- the next chunk's barrier waits for every outstanding vector-memory operation
                    // (its DMA pieces share the counter with the stores), and a store issued just ahead of it exposes its whole HBM
                    // latency -- and SPREAD: the store path of a CU takes 64 B per clock and HBM drains the whole chip's burst at its
                    // own rate; with 18 stores per wave behind ONE barrier (round 2) the next barrier's vmcnt(0) waited for all of it
                    // and the training variant cost what its bytes cost at the full HBM write rate, nothing overlapped (+16.4 %).
                    // Now at most three per chunk (save_slot): relu(h7) in the head's chunks and in the first three chunks of
                    // layer 8, the encoding slices in layer 0's chunk; with the hidden activations stored as 16 bits (+7 %).
                    auto ride_sem = [&](auto gc_, auto ch_c) {
                        constexpr int g = decltype(gc_)::value, CH = decltype(ch_c)::value;
                        if constexpr (SAVE && save_slot<SEM>(1 + CH, g) >= 0) store_h7(IC(save_slot<SEM>(1 + CH, g)));
                    };
                    run_chunk(IC(34), IC(4), IC(4), IC(0), IC(34), IC(0), IC(0), sacc, from_H, [&](auto gc_) { ride_sem(gc_, IC(0)); });
                    run_chunk(IC(34), IC(4), IC(4), IC(34), IC(34), IC(0), IC(2), sacc, from_H, [&](auto gc_) { ride_sem(gc_, IC(1)); });
                    if constexpr (SEM == 2) run_chunk(IC(16), IC(4), IC(0), IC(0), IC(16), IC(0), IC(0), sacc, from_ex, [&](auto gc_) { ride_sem(gc_, IC(2)); });
                    if constexpr (SAVE) {
                        auto relu_acc = [](float a) { return fmaxf(a, 0.0f); };
                        if (exists) {
#ifndef NSOS_LP8_SKIP_HID   // (A/B builds only)
                            {
                                // compact: the hidden activations in the 16-bit format too (256 B per point instead of 512).  Per
                                // tile a lane holds quads Qk = features 32t + 8k + 4kg + {0..3} (two packed words each); four
                                // half-wave swaps give half 0 features 32t + {0..15} and half 1 {16..31}: two 16-byte stores per lane
                                // and tile, 64 contiguous bytes per point.
                                unsigned* hrow16 = P.sem_hid16 + (long long)gp * 64;
#pragma unroll
                                for (int t = 0; t < 4; ++t) {
                                    unsigned w[4][2];
#pragma unroll
                                    for (int q = 0; q < 4; ++q) {
                                        w[q][0] = T::pack2(relu_acc(sacc[t][4 * q]), relu_acc(sacc[t][4 * q + 1]));
                                        w[q][1] = T::pack2(relu_acc(sacc[t][4 * q + 2]), relu_acc(sacc[t][4 * q + 3]));
                                    }
                                    const auto a = __builtin_amdgcn_permlane32_swap(w[0][0], w[2][0], false, false);
                                    const auto b = __builtin_amdgcn_permlane32_swap(w[0][1], w[2][1], false, false);
                                    const auto c = __builtin_amdgcn_permlane32_swap(w[1][0], w[3][0], false, false);
                                    const auto d = __builtin_amdgcn_permlane32_swap(w[1][1], w[3][1], false, false);
                                    *reinterpret_cast<u32x4*>(hrow16 + 16 * t + 8 * kg) = u32x4{a[0], b[0], a[1], b[1]};
                                    *reinterpret_cast<u32x4*>(hrow16 + 16 * t + 8 * kg + 4) = u32x4{c[0], d[0], c[1], d[1]};
                                }
                            }
#endif
                        }
                    }
                    float ps[2];
#pragma unroll
                    for (int o = 0; o < 2; ++o) ps[o] = kg ? 0.0f : __builtin_bit_cast(float, aux_l[kAuxScalars + 4 + o]);
                    heads_partial_v<4, 2>(sacc, reinterpret_cast<const float*>(aux_l) + kAuxSem2W + kg * 64, ps);
#pragma unroll
                    for (int o = 0; o < 2; ++o) sem_out[o] = both_halves(ps[o]);
                }
                stamp();  // 18 (l == 7 only; the later slots shift by one): sigma + semantic heads
            }
        }
        dead();
        // view branch: cat([feature, dir27]) -> 128 -> rgb   (H = feature, no activation)
        f32x16 vacc[4];
        // the ray's view direction is needed right after the two view chunks: requested here, its latency (L2) runs under their
        // MFMAs instead of in front of the direction encoding
        float dv[3];
        const int ray_d = *park;
#pragma unroll
        for (int k = 0; k < 3; ++k) dv[k] = P.viewdirs[3ll * ray_d + k];
        run_chunk(IC(34), IC(4), IC(4), IC(0), IC(34), IC(0), IC(0), vacc, from_H, no_ride);
        run_chunk(IC(34), IC(4), IC(4), IC(34), IC(34), IC(0), IC(2), vacc, from_H, no_ride);
        stamp();  // 21: view-branch MFMAs on the feature
        u32x4 ed[2];
        {
            Enc<NSOS_DIR_FREQS, SliceHalf> e;
            e.evaluate_hw(dv, kg);
            ed[0] = enc_slice<T, NSOS_DIR_FREQS, 0, false>(e, dv, kg);
            ed[1] = enc_slice<T, NSOS_DIR_FREQS, 1, false>(e, dv, kg);
        }
        auto from_ed = [&](auto sc) { return ed[decltype(sc)::value]; };
        // the inputs of the NaN check at the tile's end, re-read (L2 hits) rather than kept alive across the tile: `ray` comes
        // back from its LDS parking slot and the point index is re-derived (otherwise four 64-bit address pairs stay alive --
        // in scratch -- from the top of the tile); requested HERE, before the direction chunk, not in front of their use
        float nz, no[3], nd[3];
        {
            int pj_b = pj;
            asm volatile("" : "+v"(pj_b));
            const int gc_b = (pj_b < n_here) ? wave_first + pj_b : n_pts - 1;
            nz = P.z_vals[gc_b];
#pragma unroll
            for (int k = 0; k < 3; ++k) {
                no[k] = P.rays_o[3ll * ray_d + k];
                nd[k] = P.rays_d[3ll * ray_d + k];
            }
        }
        stamp();  // 22: direction encoding
        run_chunk(IC(16), IC(4), IC(0), IC(0), IC(8), IC(0), IC(0), vacc, from_ed, no_ride);  // 2 slices x 4 tiles; groups 8..15 are padding
        stamp();  // 23: direction MFMAs
        {
            float rgb[3];
#pragma unroll
            for (int o = 0; o < 3; ++o) rgb[o] = kg ? 0.0f : __builtin_bit_cast(float, aux_l[kAuxScalars + 1 + o]);
            heads_partial_v<4, 3>(vacc, reinterpret_cast<const float*>(aux_l) + kAuxRgbW + kg * 64, rgb);
#pragma unroll
            for (int o = 0; o < 3; ++o) rgb[o] = both_halves(rgb[o]);
            {   // NaN / Inf in the point's inputs must come out as NaN (the reference propagates them; the packed integer
                // ReLU would launder them).  The inputs are re-read here (L2 hits) rather than kept alive across the tile.
                // (requested before the direction chunk -- see there -- so that the L2 round trips run under its MFMAs)
                float chk = nz - nz;
#pragma unroll
                for (int k = 0; k < 3; ++k) chk += ((no[k] - no[k]) + (nd[k] - nd[k])) + (dv[k] - dv[k]);
                if (chk != chk) {
                    const float qnan = __builtin_nanf("");
                    rgb[0] = rgb[1] = rgb[2] = sigma = sem_out[0] = sem_out[1] = qnan;
                }
            }
            if constexpr (C == 4) {
                if (exists && kg == 0) *reinterpret_cast<f32x4*>(P.raw + (long long)gp * C) = f32x4{rgb[0], rgb[1], rgb[2], sigma};
            } else {
                // 24 B per point.  Written per lane (8 B pieces from the two halves) every store instruction covered a third of
                // each 32-byte sector it touched and the memory-side write counter showed 2.9x the bytes (partial-line writes).
                // The wave's 32 points are 768 contiguous bytes: staged through 768 B of LDS per wave, then 48 lanes store 16 B
                // each -- full sectors, one store instruction.  (Both halves hold every value after both_halves.)
                float* const stage = reinterpret_cast<float*>(lds + kSlots * kSlotBytes + kAuxWords * 4) + wave_s * 192;
                if (n_here == 32) {
                    if (kg == 0) {
                        *reinterpret_cast<f32x2*>(stage + 6 * pj) = f32x2{rgb[0], rgb[1]};
                        *reinterpret_cast<f32x2*>(stage + 6 * pj + 2) = f32x2{rgb[2], sigma};
                        *reinterpret_cast<f32x2*>(stage + 6 * pj + 4) = f32x2{sem_out[0], sem_out[1]};
                    }
                    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                    __builtin_amdgcn_wave_barrier();
                    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
                    if (lane < 48) *reinterpret_cast<f32x4*>(P.raw + (long long)wave_first * C + 4 * lane) = *reinterpret_cast<const f32x4*>(stage + 4 * lane);
                    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");   // the next tile's writes to the stage follow these reads
                    __builtin_amdgcn_wave_barrier();
                } else if (pj < n_here) {          // the ragged last wave of the launch: per point
                    float* out = P.raw + (long long)(wave_first + pj) * C;
                    if (kg == 0) {
                        *reinterpret_cast<f32x2*>(out) = f32x2{rgb[0], rgb[1]};
                        *reinterpret_cast<f32x2*>(out + 2) = f32x2{rgb[2], sigma};
                    } else {
                        *reinterpret_cast<f32x2*>(out + 4) = f32x2{sem_out[0], sem_out[1]};
                    }
                }
            }
        }
        stamp();  // 24: rgb head + stores
    }
#undef IC
    if constexpr (PROF)
        if (P.prof && blockIdx.x < 2 && lane0 == 0) P.prof[(blockIdx.x * 8 + wave_s) * kProfSlots + kProfSlots - 1] = __builtin_readcyclecounter();
    if (!lagging) __builtin_amdgcn_s_barrier();  // pairs with the laggers' last chunk barrier
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    __syncthreads();
}


constexpr int kLdsBytes8 = kSlots * kSlotBytes + kAuxWords * 4 + 8 * 768;   // 4 weight slots + head weights + the raw-output stage (768 B per wave)

template <class T, int SEM, bool SAVE, bool PROF>
int32_t launch8p(const LpParams& p, hipStream_t stream) {
    static NsosPerDeviceFlag configured_on;
    bool& configured = configured_on.here();
    if (!configured) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&mlp_lp8_kernel<T, SEM, SAVE, PROF>),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, kLdsBytes8);
        if (e != hipSuccess) return (int32_t)e;
        configured = true;
    }
    const int cus = nsos_device_cus();
    const int grid = p.n_tiles < cus ? p.n_tiles : cus;
    hipLaunchKernelGGL((mlp_lp8_kernel<T, SEM, SAVE, PROF>), dim3(grid), dim3(512), kLdsBytes8, stream, p);
    return nsos_launch_status();
}
template <class T, int SEM, bool SAVE>
int32_t launch8(const LpParams& p, hipStream_t stream) {
    if (p.prof) {   // diagnostics: the stamped instantiations exist for the shapes the phase-profile scripts use
        if constexpr (SEM != 1) return launch8p<T, SEM, SAVE, true>(p, stream);
        else return NSOS_ERR_UNSUPPORTED;
    }
    return launch8p<T, SEM, SAVE, false>(p, stream);
}

}  // namespace

namespace nsos {
namespace lp {

// dispatch used by forward_rays_lp (mlp_lp.hip); sem_mode and dtype were validated there
int32_t launch_lp8(const LpParams& p, int32_t sem_mode, bool is_f16, bool save, hipStream_t st) {
    if (save) {
        if (is_f16) return sem_mode == 1 ? launch8<F16, 1, true>(p, st) : launch8<F16, 2, true>(p, st);
        return sem_mode == 1 ? launch8<BF16, 1, true>(p, st) : launch8<BF16, 2, true>(p, st);
    }
    if (is_f16) {
        switch (sem_mode) {
            case 0: return launch8<F16, 0, false>(p, st);
            case 1: return launch8<F16, 1, false>(p, st);
            default: return launch8<F16, 2, false>(p, st);
        }
    }
    switch (sem_mode) {
        case 0: return launch8<BF16, 0, false>(p, st);
        case 1: return launch8<BF16, 1, false>(p, st);
        default: return launch8<BF16, 2, false>(p, st);
    }
}

}  // namespace lp
}  // namespace nsos
