// K2-LP16: the 16-bit fused positional-encoding + MLP kernel on v_mfma_f32_16x16x32 (round 4; the default 16-bit kernel).
// Same network and the same replaced reference code as mlp_lp.hip / mlp_lp8.hip (models/embedder.py:34-48,
// models/nerf_mlp.py:67-100,179-215), same workgroup shape as mlp_lp8_kernel (8 waves = two per SIMD, <= 256 registers,
// 32 points per wave, 256-point tiles, four 36 KiB weight slots fed by global->LDS DMA, one barrier per chunk, the
// continuous 4-deep A-operand ring) -- re-tiled for the OTHER 16-bit MFMA shape.
//
// Why.  The chip runs this kernel against its power limit, and what the matrix pipe sustains there depends on the MFMA
// shape: scripts/ubench/mfma_mix.hip (profiles/r04/a_mfma_mix.txt), network-like operands, two waves per SIMD, pipe 98 %
// busy in both cases --
//     v_mfma_f32_32x32x16_bf16  operands in registers 1805 TF (1.75 GHz)   A operand through LDS 1626 TF (1.58 GHz)
//     v_mfma_f32_16x16x32_bf16                         2150 TF (2.09 GHz)                         1878 TF (1.84 GHz)
// i.e. +15-19 % at equal occupancy (per 32 kFLOP the 32x32 shape moves A 4 + B 4 + C 16 + D 16 register vectors, the 16x16
// shape 2 x (B 4 + C 4 + D 4) + A 4).  Three rounds of removing cycles from mlp_lp8_kernel had returned about half of
// every saving as a lower clock; this changes what a cycle costs.
//
// Shape.  lane = 16 q + n.  A wave's 32 points are two COLUMN BLOCKS c of 16 points (point 16 c + n); every lane works
// for both.  v_mfma_f32_16x16x32: A[row = lane & 15][k = 8 (lane >> 4) + e], B[k = 8 (lane >> 4) + e][col = lane & 15],
// D[row = 4 (lane >> 4) + r][col = lane & 15].  An output TILE is 16 features: accumulator Z[t][c] (4 registers) of lane
// (n, q) holds features 16 t + 4 q + r of point 16 c + n.  A K-SLICE is 32 input features = two tiles: the B operand of
// slice s for block c is H[s][c] = pack16(Z[2s][c][0..3], Z[2s+1][c][0..3]) -- the accumulator layout IS the operand
// layout (k position (q, e) of slice s carries feature 32 s + 16 (e >> 2) + 4 q + (e & 3); the weight stream is packed in
// that order), so activations never leave the lane: v_cvt_pk + v_pk_max_i16 per packed word, as in mlp_lp8_kernel.
// One 1 KiB A operand (16 rows x 32 k) feeds TWO MFMAs (the two column blocks): one ds_read_b128 per 32 kFLOP and wave =
// 128 B/clk/CU, as before.  Register budget as before: 16 tiles x 2 blocks x 4 = 128 accumulator registers per 256-wide
// layer, 8 slices x 2 blocks x 4 = 64 for H.
//
// What else changed against mlp_lp8_kernel:
//   * NO bias MFMAs.  There a bias was an A operand multiplied by a B of ones: 2 of a chunk's 34 MFMAs (5.9 % of the matrix
//     pipe's time; with 16-row tiles it would be 11 %).  Here a hidden layer's chunk carries its four tiles' biases as a 256-byte
//     block behind its 32 A operands (lane (n, q) of tile t: the four fp32 values bias[16 t + 4 q + r]); the PREVIOUS chunk
//     reads them straight into the accumulators' block-0 registers (read_bias / quad_layer) once the riding activation is done
//     with those registers, and the tile's first two MFMAs take them as their C operand.  (First version: the bias as a ring
//     group of its own in front of each tile's first A operand -- the four MFMA-less groups at the head of every chunk halved
//     the ring's look-ahead there: 10.26 k cycles per layer against 9.79 k now, profiles/r04.)  The raw tile's bias (heads)
//     still travels through the ring as a group whose slot is the next group's C operand (Sched's BIAS mask: its slot is
//     re-loaded one group late).  Layers with an encoding input (0, 5, sem+coord head, view branch) carry the bias in the
//     encoding's pad column (input 1.0) instead.
//   * Hidden layers are tile-QUAD-major (chunk c = output tiles 4c..4c+3 over all 8 slices: 32 A operands + the bias block,
//     64 MFMAs), with the previous quad's activation riding behind the MFMAs exactly as the tile
//     pairs did (same register counts: a quad buffer is 32 registers, a quad's activation 16 packed words).
//   * The sigma, rgb and semantic-logit heads are MFMAs into ONE 16-row "raw" tile R[c] (rows 0..2 rgb, 3 sigma, 4..5
//     semantics): 8 + 4 + 4 A operands, 32 MFMAs per wave and tile instead of ~5 k cycles of v_dot2c / fp32 FMA chains on
//     the vector ALU.  rgb and semantics therefore see the hidden activations and their weights rounded to 16 bits (the
//     sigma head always did); the tests hold this to an emulation with exactly these roundings (tests/test_gpu_parity.py).  Lane
//     (n, 0) ends up with [r, g, b, sigma] of its point: the C = 4 output is one 16-byte store per point.
//     rgb_linear's four A operands (4 KiB behind the stream's chunks) are fetched once per workgroup and stay in LDS: its 8 MFMAs
//     run right behind the view branch's activation (a chunk of their own -- barrier, ring restart, DMA pieces -- cost 2.6 k
//     cycles per tile for 8 MFMAs; folding them into the next tile's layer-0 chunk instead made hipcc keep the 32 registers of
//     the pending activations in scratch across the tile boundary: +15 % cycles, profiles/r04).
//   * The encodings are evaluated for the lane's 2 x 16 feature slots from a per-(q, slot) table in LDS (octave scale per
//     coordinate + phase: sin(2 pi (frac + 1/4)) = cos), the same two-term revolution arithmetic as Enc::evaluate_hw.
// Results are NOT bit-identical to mlp_lp_kernel / mlp_lp8_kernel (other contraction order inside the MFMAs, 16-bit heads):
// tests/test_gpu_parity.py holds all three to the same emulation.
#include "lp_common.h"
#include "lp16_sched.h"

using namespace nsos;
using namespace nsos::lp;

namespace {

constexpr int kRgbLds = kSlots * kSlotBytes + kAuxWords * 4 + kW16 * 768;   // LDS offset of rgb_linear's four resident A operands
// the NS slices of one point's encoding in B-operand form: word w of slice s = slots (8 s + 2 w, 8 s + 2 w + 1)
template <class T, int NS, int L, int ONE_AT>   // ONE_AT: the feature index that is the constant 1.0 (bias input), in the last lane group
__device__ __forceinline__ void encode16(u32x4 (&out)[NS], const float (&x)[3], const unsigned char* lds_tab, int q) {
    constexpr float kInvHi = 0.15915493667125702f, kInvLo = 6.4206382679e-9f;   // 1 / 2pi = kInvHi + kInvLo (Enc::evaluate_hw)
    float uh[3], ul[3];
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        uh[c] = x[c] * kInvHi;
        ul[c] = __fmaf_rn(x[c], kInvHi, -uh[c]) + x[c] * kInvLo;
    }
    const float amax = fmaxf(fmaxf(fabsf(x[0]), fabsf(x[1])), fabsf(x[2])) * (float)(1 << (L - 1));
    const bool big = !(amax < 32768.0f);                // also true for NaN / Inf inputs (the cold path propagates them)
    float val[8 * NS];
    const f32x4* tab = reinterpret_cast<const f32x4*>(lds_tab) + q * (8 * NS);
    static_for<0, 8 * NS>([&](auto jc) {
        constexpr int j = decltype(jc)::value;
        const f32x4 t = tab[j];
        // exact: one scale is a power of two, the others are zero (a NaN / Inf coordinate poisons the point, as it must)
        const float ah = __fmaf_rn(uh[2], t[2], __fmaf_rn(uh[1], t[1], uh[0] * t[0]));
        const float fr = __builtin_amdgcn_fractf(ah);
        const float g = __fmaf_rn(ul[0], t[0], __fmaf_rn(ul[1], t[1], __fmaf_rn(ul[2], t[2], fr))) + t[3];
        val[j] = __builtin_amdgcn_sinf(g);
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
    // raw coordinates: features 0..2 = slots 0..2 of lane group 0; the constant: slot ONE_AT - 24 of lane group 3 (last slice)
#pragma unroll
    for (int e = 0; e < 3; ++e) val[e] = q == 0 ? x[e] : val[e];
    constexpr int one_slot = 8 * (NS - 1) + (ONE_AT - 32 * (NS - 1) - 24);
    static_assert(one_slot >= 8 * (NS - 1) && one_slot < 8 * NS, "the constant input lives in lane group 3 of the last slice");
    val[one_slot] = q == 3 ? 1.0f : val[one_slot];
    // (pad slots between the last feature and the constant have all-zero table entries: sin(0) = 0)
#pragma unroll
    for (int s = 0; s < NS; ++s)
#pragma unroll
        for (int w = 0; w < 4; ++w) out[s][w] = T::pack2(val[8 * s + 2 * w], val[8 * s + 2 * w + 1]);
}

template <class T, int SEM, bool SAVE = false, bool PROF = false>
__global__ __launch_bounds__(64 * kW16, 1) void mlp_lp16_kernel(const LpParams P) {
    extern __shared__ __attribute__((aligned(16))) unsigned char lds[];  // 4 x 36 KiB weight slots + 4 KiB tables + 6 KiB output stage
    const int lane0 = threadIdx.x & 63, wave = threadIdx.x >> 6;
    constexpr int NCH = lp16_chunks(SEM);
    constexpr int C = SEM ? 6 : 4;

    // ---- weight stream: slots rotate (c0 = chunk cur, c1 = cur+1, c2 = cur+2, c3 = the slot that becomes free at the next barrier)
    const unsigned lds_base = (unsigned)(size_t)(__attribute__((address_space(3))) unsigned char*)lds;
    const int wave_s = __builtin_amdgcn_readfirstlane(wave);
    const unsigned voff = (unsigned)(lane0 * 16);
    auto lane_addr = [&](int b) { return lds_base + (unsigned)(b * kSlotBytes + lane0 * 16); };
    auto slot_addr = [&](int b) { return lds_base + (unsigned)(b * kSlotBytes); };
    unsigned c0 = lane_addr(0), c1 = lane_addr(1), c2 = lane_addr(2), c3 = lane_addr(3);
    unsigned d0 = slot_addr(0), d1 = slot_addr(1), d2 = slot_addr(2), d3 = slot_addr(3);
    // byte offset of this wave's i-th piece inside a chunk / slot: piece wave + kW16 i (the surplus ones re-copy piece 35)
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
#ifdef NSOS_LP16_DMA_STAGGER
    unsigned fill_wlast = 0;
#endif
    unsigned fill_lo = 0, fill_hi = 0, fill_dst = 0, fill_w = 0;
#ifdef NSOS_LP16_DMA_STAGGER
    auto side = [&](auto early_c, auto late_c) {
        constexpr int EARLY = decltype(early_c)::value, LATE = decltype(late_c)::value;
        auto piece = [&](int i) {
            const unsigned off = i < kFull16 ? fill_w + (unsigned)(kW16 * 1024) * (unsigned)i : fill_wlast;
            const unsigned long long sp = (((unsigned long long)fill_hi << 32) | fill_lo) + off;
            dma_1k(reinterpret_cast<const void*>(sp), fill_dst + off, voff);
        };
        if constexpr (EARLY >= 0 && EARLY < kDma16) { if (wave_s < 4) { NSOS_PIN(); piece(EARLY); NSOS_PIN(); } }
        if constexpr (LATE >= 0 && LATE < kDma16) { if (wave_s >= 4) { NSOS_PIN(); piece(LATE); NSOS_PIN(); } }
    };
#else
    auto side = [&](int i, int nfill) {
        // this wave's i-th piece of the chunk being fetched: piece wave + kW16 i -- if the chunk holds it.  The test is wave-uniform, but
        // a BRANCH around the copy cuts the chunk's straight-line code into basic blocks and the register allocator then spills
        // across them (8-38 scratch instructions per tile): where the test does not fold at compile time the piece is issued with
        // EXEC cleared instead -- a no-op that costs an issue slot -- inside one asm statement.
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
#endif
    auto mid = [&]() {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        const unsigned long long sp = (unsigned long long)srcf;
        fill_lo = __builtin_amdgcn_readfirstlane((unsigned)sp);
        fill_hi = __builtin_amdgcn_readfirstlane((unsigned)(sp >> 32));
        fill_dst = __builtin_amdgcn_readfirstlane(d2);
        fill_w = __builtin_amdgcn_readfirstlane(woff);
#ifdef NSOS_LP16_DMA_STAGGER
        fill_wlast = __builtin_amdgcn_readfirstlane(wlast);
#endif
    };
    auto tail = [&]() {
        const unsigned tc = c0, td = d0;
        c0 = c1; c1 = c2; c2 = c3; c3 = tc;
        d0 = d1; d1 = d2; d2 = d3; d3 = td;
        srcf += kSlotBytes;
        if (srcf == src_end) srcf = P.chunks;
    };
    auto ctx = [&]() { return ChunkCtx{c0, c1}; };

#ifdef NSOS_LP16_PRIO   // (A/B: static priority for the second-dispatched half of the workgroup, MI355X_MICROARCH.md "two waves per SIMD" item 4)
    if (wave_s >= 4) __builtin_amdgcn_s_setprio(NSOS_LP16_PRIO);
#endif
    f32x4 ring[kRing16];
#pragma unroll
    for (int k = 0; k < 3; ++k)
#pragma unroll
        for (int i = 0; i < kDma16; ++i)
            dma_piece(P.chunks + (size_t)(k % NCH) * kSlotBytes, k == 0 ? d0 : (k == 1 ? d1 : d2), i);
    // rgb_linear's four A operands stay resident in LDS (kRgbLds): its 8 MFMAs per wave and tile run right behind the view branch's
    // activation instead of in a chunk of their own (a barrier, a ring restart and five DMA pieces for 8 MFMAs: 2.6 k cycles per tile)
    if (wave_s < 4) {
        const unsigned long long sp = (unsigned long long)(P.chunks + (size_t)NCH * kSlotBytes + (size_t)wave_s * 1024);
        const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)sp), hi = __builtin_amdgcn_readfirstlane((unsigned)(sp >> 32));
        const unsigned dst = __builtin_amdgcn_readfirstlane(lds_base + (unsigned)(kRgbLds + wave_s * 1024));
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
        // (lane-dependent loop invariants are re-derived per tile instead of living -- in scratch -- across every chunk: see mlp_lp8.hip)
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
        // ---- this lane's two points: tile*256 + wave*32 + 16 c + n
        const int n_pts = (int)P.n_pts;
        const int wave_first = tile * kTile16 + wave_s * 32;
        const int n_here = n_pts - wave_first >= 32 ? 32 : (n_pts - wave_first < 0 ? 0 : n_pts - wave_first);
        int* const park = reinterpret_cast<int*>(lds + kSlots * kSlotBytes + kAuxWords * 4) + wave_s * 192 + lane;   // [2][64] ints of the wave's output stage
        bool save_ok[2] = {false, false};
        unsigned long long save_grp = 0;
        unsigned save_off = 0, save_off_x = 0;
        // store of 16 bytes per lane into the tile-major sem_in: scalar group base (+ a multiple of 4 KiB) + the lane's constant
        // offset + an immediate
        auto save_store = [&save_grp](auto boff_c, unsigned off, const u32x4& v) {
            constexpr int BOFF = decltype(boff_c)::value;
            unsigned long long b = save_grp + (unsigned long long)(BOFF & ~4095);
            asm volatile("global_store_dwordx4 %0, %1, %2 offset:%3 nt\n\ts_nop 1" : : "v"(off), "v"(v), "s"(b), "i"(BOFF & 4095) : "memory");   // (s_nop: see the hidden-activation stores)
        };
        if constexpr (SAVE) {
            // TILE-MAJOR sem_in (include/nerf_sos_hip.h): [group of 32 points][octet 0..39][point][8 channels] -- store K, half kg of
            // mlp_lp8_kernel is octet 2K + kg.  Lane (n, q) ends up (after one v_permlane16_swap per word pair) with octet
            // 4 s + 2 (q & 1) + (q >> 1) of slice s of relu(h7), and holds octet 32 + 4 s + q of the encoding slice s as it is.
            save_grp = reinterpret_cast<unsigned long long>(P.sem_in16 + (long long)(wave_first >> 5) * 5120);
            save_off = (unsigned)((2 * (q & 1) + (q >> 1)) * 512 + n * 16);
            save_off_x = (unsigned)(q * 512 + n * 16);
        }
        u32x4 ex[2][2];
        {
#pragma unroll
            for (int c = 0; c < 2; ++c) {
                const int pw = 16 * c + n;
                const bool exists = pw < n_here;
                const int gc = exists ? wave_first + pw : n_pts - 1;
                const int ray = (int)((unsigned)gc / (unsigned)P.n_samples);
                park[64 * c] = ray;
                if constexpr (SAVE) save_ok[c] = exists;
                const float z = P.z_vals[gc];
                float x[3];
#pragma unroll
                for (int k = 0; k < 3; ++k) {
                    const float m = P.rays_d[3ll * ray + k] * z;  // models/sampler.py:70,166 (mul, then add)
                    x[k] = P.rays_o[3ll * ray + k] + m;
                }
                u32x4 e2[2];
#ifdef NSOS_LP16_NOENC      // (A/B builds only: what the xyz encoding costs; wrong results)
                e2[0] = e2[1] = u32x4{__float_as_uint(x[0]), __float_as_uint(x[1]), __float_as_uint(x[2]), 0u};
#else
                encode16<T, 2, NSOS_XYZ_FREQS, 63>(e2, x, tabs + kTabXyz, q);   // feature 63 (pad) = 1.0: the bias input of layers 0, 5 and the sem+coord head
#endif
                ex[0][c] = e2[0];
                ex[1][c] = e2[1];
            }
        }

        u32x4 H[8][2];
        f32x4 R[2] = {zero4, zero4};      // the "raw" tile: rows 0..2 rgb, 3 sigma, 4..5 semantics
        auto from_ex = [&](auto sc, auto cc) { return ex[decltype(sc)::value][decltype(cc)::value]; };
        // SAVE: slice s, block c of relu(h7) as one 16-byte store per lane.  The lane holds channels 32 s + 4 q + {0..3} (words 0, 1)
        // and 32 s + 16 + 4 q + {0..3} (words 2, 3) of its point; v_permlane16_swap (odd rows of the first operand <-> even rows of
        // the second) hands the even groups their right neighbour's first half and the odd groups their left neighbour's second:
        // 8 consecutive channels per lane, 256 contiguous bytes per 16-lane group.
        auto store_h7 = [&](auto sc, auto cc) {
            constexpr int s = decltype(sc)::value, c = decltype(cc)::value;
            if constexpr (SAVE) {
                const auto r0 = __builtin_amdgcn_permlane16_swap(H[s][c][0], H[s][c][2], false, false);
                const auto r1 = __builtin_amdgcn_permlane16_swap(H[s][c][1], H[s][c][3], false, false);
#ifndef NSOS_LP16_SKIP_IN    // (A/B builds only: scripts/diag/build_variant.sh)
                if (save_ok[c]) save_store(IC(s * 2048 + c * 256), save_off, u32x4{r0[0], r1[0], r0[1], r1[1]});
#endif
            }
        };
        // the k-th of the 16 stores of relu(h7) (k = 2 s + c) / of the 4 stores of the encoding slices
        auto store_h7_k = [&](auto kc) { constexpr int k = decltype(kc)::value; store_h7(IC(k >> 1), IC(k & 1)); };
        auto store_ex_k = [&](auto kc) {
            constexpr int k = decltype(kc)::value, s = k >> 1, c = k & 1;
            if constexpr (SAVE)
                if (save_ok[c]) save_store(IC(16 * 1024 + s * 2048 + c * 256), save_off_x, ex[s][c]);
        };

        // ---- generic chunk runners ----------------------------------------------------------------------------
        // slice-major chunk over NT tiles: group g = (slice S0 + g / NT, tile g % NT); acc[t][c] (+)= A x B(slice)[c]; ZF: slice S0 starts the sum
        auto slice_chunk = [&](auto nt_c, auto nsl_c, auto zf_c, auto& acc, auto&& bsel, auto&& ride, const int nfill) {
            constexpr int NT = decltype(nt_c)::value, NSL = decltype(nsl_c)::value;
            constexpr bool ZF = decltype(zf_c)::value != 0;
            pipeline16<NT * NSL, NT * NSL, 0ull>(ring, ctx(), [&](auto ic, const f32x4& a32, const f32x4&) {
                constexpr int g = decltype(ic)::value, sl = g / NT, t = g % NT;
                const u32x4 aop = __builtin_bit_cast(u32x4, a32);
                if constexpr (ZF && sl == 0) {
                    acc[t][0] = T::mfma_k32(aop, bsel(IC(sl), IC(0)), zero4);
                    acc[t][1] = T::mfma_k32(aop, bsel(IC(sl), IC(1)), zero4);
                } else {
                    acc[t][0] = T::mfma_k32(aop, bsel(IC(sl), IC(0)), acc[t][0]);
                    acc[t][1] = T::mfma_k32(aop, bsel(IC(sl), IC(1)), acc[t][1]);
                }
                ride(ic);
            }, mid, tail, side, nfill);
#pragma unroll
            for (int t = 0; t < NT; ++t) asm volatile("" : "+v"(acc[t][0]), "+v"(acc[t][1]));   // (keeps LLVM from sinking the chunk: see mlp_lp8.hip)
        };
        // tile-quad chunk of a hidden layer: 32 A operands, group g = (slice g >> 2, tile g & 3); zq[t][0] holds the tile's bias when the
        // chunk starts (read from LDS by the PREVIOUS chunk -- bias_next below -- or by bias_now): the C operand of both first MFMAs.
        // EXTRA / extra: this chunk's own reads of the NEXT quad chunk's bias (its counted part; see quad_layer)
        auto quad_chunk = [&](auto extra_c, auto& zq, auto&& ride, auto&& extra, const int nfill) {
            constexpr unsigned long long EXTRA = (unsigned long long)decltype(extra_c)::value;
            pipeline16x<32, 32, 0ull, EXTRA>(ring, ctx(), [&](auto ic, const f32x4& a32, const f32x4&) {
                constexpr int g = decltype(ic)::value, s = g >> 2, t = g & 3;
                const u32x4 aop = __builtin_bit_cast(u32x4, a32);
                if constexpr (s == 0) {             // block 1 first: zq[t][0] is still the bias
                    zq[t][1] = T::mfma_k32(aop, H[0][1], zq[t][0]);
                    zq[t][0] = T::mfma_k32(aop, H[0][0], zq[t][0]);
                } else {
                    zq[t][0] = T::mfma_k32(aop, H[s][0], zq[t][0]);
                    zq[t][1] = T::mfma_k32(aop, H[s][1], zq[t][1]);
                }
#ifndef NSOS_LP16_NORIDE              // (A/B builds only: timing without the riding activation; wrong results)
                ride(ic);
#endif
            }, mid, tail, side, extra, nfill);
#pragma unroll
            for (int t = 0; t < 4; ++t) asm volatile("" : "+v"(zq[t][0]), "+v"(zq[t][1]));
        };
        // the bias block of a quad chunk: group 32 of its slot, [tile][q][4 x fp32]; tile t of lane (n, q) at byte 32768 + 64 t + 16 q.
        // slot_lane = the lane's A-operand address in that slot (slot + 16 lane).
        auto read_bias = [&](auto tc, f32x4& dst, unsigned slot_lane) {
            constexpr int t = decltype(tc)::value;
            const unsigned addr = slot_lane - (unsigned)(lane * 16) + (unsigned)(q * 16);
            asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(dst) : "v"(addr), "i"(32768 + 64 * t) : "memory");
        };
        auto no_ride = [](auto) {};
        // packed word w (0..3) of local slice ls (0..1), block c of a finished quad zq: tiles 2 ls, 2 ls + 1
        auto quad_word = [&](auto& zq, int ls, int c, int w) {
            return T::pack2(zq[2 * ls + (w >> 1)][c][2 * (w & 1)], zq[2 * ls + (w >> 1)][c][2 * (w & 1) + 1]);
        };

        stamp();  // 1: inputs + xyz encoding
        f32x4 Zq[2][4][2];
        auto dead = [&]() {   // (see mlp_lp8.hip: the quad buffers are redefined where they are dead)
#pragma unroll
            for (int b = 0; b < 2; ++b)
#pragma unroll
                for (int t = 0; t < 4; ++t) asm volatile("" : "=v"(Zq[b][t][0]), "=v"(Zq[b][t][1]));
        };
        // H = relu(Z) for all 16 tiles (layers 0 and 5: exposed pass)
        auto activate_all = [&](const f32x4 (&Z)[16][2]) {
            asm volatile("s_nop 7" ::: "memory");   // MFMA result -> VALU read wait states (the asm below hides the reads)
#pragma unroll
            for (int s = 0; s < 8; ++s) {
#pragma unroll
                for (int c = 0; c < 2; ++c)
#pragma unroll
                    for (int w = 0; w < 4; ++w) H[s][c][w] = T::pack2(Z[2 * s + (w >> 1)][c][2 * (w & 1)], Z[2 * s + (w >> 1)][c][2 * (w & 1) + 1]);
#pragma unroll
                for (int c = 0; c < 2; ++c)
#pragma unroll
                    for (int w = 0; w < 4; ++w) NSOS_RELU_WORD16(H[s][c][w], 0u);
            }
        };
        {   // ---- layer 0: x63 (2 slices, bias in the pad column) -> 16 tiles, slice-major, one exposed pass
            f32x4 Z[16][2];
            slice_chunk(IC(16), IC(2), IC(1), Z, from_ex, [&](auto gc_) {
                constexpr int g = decltype(gc_)::value;     // SAVE: the four stores of the encoding slices
                if constexpr (SAVE && SEM != 0 && g >= 3 && g < 15 && g % 3 == 0) store_ex_k(IC(g / 3 - 1));
            }, 33);     // (fetches the chunk two ahead: layer 1's second quad)
            stamp();  // 2: L0 MFMAs
            activate_all(Z);
            stamp();  // 3: L0 activation
        }
        auto quad_layer = [&](const int l, const bool tail_pending, const bool leave_tail, const bool bias_now, const bool next_quad) {
            // chunk c accumulates output tiles 4c..4c+3 over all 8 input slices into Zq[c & 1]; the activation of the PREVIOUS quad
            // rides behind this chunk's MFMAs.  The layer's input H stays live until its last chunk: finished slices 0..3 wait in
            // Ho and move into H behind the last chunk's MFMAs, each right after its final use there; quad 2 is activated straight
            // into H[4..5] once those are dead; quad 3 is the tail (rides in the next layer's first chunk, or one exposed pass).
            // Biases: chunk c reads the NEXT chunk's four bias vectors into Zq[(c + 1) & 1][t][0] once the riding activation is done
            // with those registers (groups 24..27; the last chunk: 22, 24, 26, 28, and only when a quad layer follows directly --
            // `next_quad`, a runtime flag: those four reads are not in the counted waits, which makes the waits behind them stricter,
            // never looser).  `bias_now`: the layer's first chunk follows some other kind of chunk: read and wait here.
            const unsigned floor = l < 8 ? 0u : 0x80008000u;     // feature_linear (l == 8) has no activation
            u32x4 Ho[4][2];
            if (bias_now) {
                static_for<0, 4>([&](auto tc) { read_bias(tc, Zq[0][decltype(tc)::value][0], c0); });
                lgkm_wait<0>();
                asm volatile("" : "+v"(Zq[0][0][0]), "+v"(Zq[0][1][0]), "+v"(Zq[0][2][0]), "+v"(Zq[0][3][0]));
            }
            auto ride_tail = [&](auto gc_) {     // the previous layer's quad 3 (Zq[1]) -> H[6..7] (always a ReLU layer): input slices 6, 7 are read from group 24 on
                constexpr int g = decltype(gc_)::value;
                if constexpr (g >= 4 && g <= 20) {
                    if (tail_pending) {
                        if constexpr (g >= 5) {
                            constexpr int k = g - 5;
                            NSOS_RELU_WORD16(H[6 + (k >> 3)][(k >> 2) & 1][k & 3], 0u);
                        }
                        if constexpr (g < 20) {
                            constexpr int k = g - 4;
                            H[6 + (k >> 3)][(k >> 2) & 1][k & 3] = quad_word(Zq[1], k >> 3, (k >> 2) & 1, k & 3);
                        }
                    }
                }
            };
            auto ride_act = [&](auto gc_, auto src_c, auto base_c) {   // 16 words of quad buffer src -> Ho[base .. base+1]: convert in groups 4..19, clamp one group later
                constexpr int g = decltype(gc_)::value, SRC = decltype(src_c)::value, BASE = decltype(base_c)::value;
                if constexpr (g >= 5 && g <= 20) {
                    constexpr int k = g - 5;
                    NSOS_RELU_WORD16(Ho[BASE + (k >> 3)][(k >> 2) & 1][k & 3], floor);
                }
                if constexpr (g >= 4 && g < 20) {
                    constexpr int k = g - 4;
                    Ho[BASE + (k >> 3)][(k >> 2) & 1][k & 3] = quad_word(Zq[SRC], k >> 3, (k >> 2) & 1, k & 3);
                }
            };
            // (SAVE: the second half of relu(h7)'s stores rides in layer 8's first three chunks -- H is overwritten only in the last)
            auto ride_save = [&](auto gc_, auto ch_c) {
                constexpr int g = decltype(gc_)::value, CH = decltype(ch_c)::value;
                if constexpr (SAVE && SEM != 0 && g >= 3 && (g - 3) % 6 == 0 && (g - 3) / 6 < (CH == 2 ? 2 : 3))
                    if (l == 8) store_h7_k(IC(8 + 3 * CH + (g - 3) / 6));
            };
            constexpr unsigned long long kBiasAt = 0xFull << 24;          // groups 24..27: one bias vector each (c1 = the next chunk's slot: tail() rotates the names at group 29)
            quad_chunk(std::integral_constant<unsigned long long, kBiasAt>{}, Zq[0], [&](auto gc_) { ride_tail(gc_); ride_save(gc_, IC(0)); },
                       [&](auto gc_) { read_bias(IC(decltype(gc_)::value - 24), Zq[1][decltype(gc_)::value - 24][0], c1); }, 33);
            quad_chunk(std::integral_constant<unsigned long long, kBiasAt>{}, Zq[1], [&](auto gc_) { ride_act(gc_, IC(0), IC(0)); ride_save(gc_, IC(1)); },
                       [&](auto gc_) { read_bias(IC(decltype(gc_)::value - 24), Zq[0][decltype(gc_)::value - 24][0], c1); }, 33);
            // (the last two chunks fetch the NEXT part's first two chunks: quads / slice chunks / the heads' chunks (<= 33 pieces) --
            //  or, behind feature_linear, the view branch's full slots)
            const int nfill_next = l == 8 ? 36 : 33;
            quad_chunk(std::integral_constant<unsigned long long, kBiasAt>{}, Zq[0], [&](auto gc_) { ride_act(gc_, IC(1), IC(2)); ride_save(gc_, IC(2)); },
                       [&](auto gc_) { read_bias(IC(decltype(gc_)::value - 24), Zq[1][decltype(gc_)::value - 24][0], c1); }, nfill_next);
            quad_chunk(std::integral_constant<unsigned long long, 0ull>{}, Zq[1], [&](auto gc_) {
                constexpr int g = decltype(gc_)::value;
                // input slice s was last used by group 4 s + 3: Ho[s] -> H[s], one block per group
                if constexpr (g >= 4 && g <= 17 && ((g - 4) & 3) < 2) mov_slice16(H[(g - 4) >> 2][(g - 4) & 1], Ho[(g - 4) >> 2][(g - 4) & 1]);
                // quad 2 (Zq[0]) -> H[4] (dead from group 20: 8 words in groups 20..27) and H[5] (dead from group 24: groups 24..31)
                if constexpr (g >= 21 && g <= 28) { constexpr int k = g - 21; NSOS_RELU_WORD16(H[4][k >> 2][k & 3], floor); }
                if constexpr (g >= 25) { constexpr int k = g - 25; NSOS_RELU_WORD16(H[5][k >> 2][k & 3], floor); }
                if constexpr (g >= 20 && g <= 27) { constexpr int k = g - 20; H[4][k >> 2][k & 3] = quad_word(Zq[0], 0, k >> 2, k & 3); }
                if constexpr (g >= 24) { constexpr int k = g - 24; H[5][k >> 2][k & 3] = quad_word(Zq[0], 1, k >> 2, k & 3); }
                // the next layer's first bias vectors into Zq[0][t][0], each once its register's last word has been converted
                // (tile 0: group 21, tile 1: 23, tile 2: 25, tile 3: 27); uncounted (see above)
                if constexpr (g == 22 || g == 24 || g == 26 || g == 28) {
                    if (next_quad) { NSOS_PIN(); read_bias(IC((g - 22) >> 1), Zq[0][(g - 22) >> 1][0], c1); NSOS_PIN(); }
                }
            }, [](auto) {}, nfill_next);
            NSOS_RELU_WORD16(H[5][1][3], floor);     // the word converted in the chunk's last group
            stamp();  // 2 + 2l: MFMAs of layer l (with the riding activation of quads 0..2)
            if (!leave_tail) {
                asm volatile("s_nop 7" ::: "memory");   // MFMA result -> VALU read wait states
#pragma unroll
                for (int k = 0; k < 16; ++k) H[6 + (k >> 3)][(k >> 2) & 1][k & 3] = quad_word(Zq[1], k >> 3, (k >> 2) & 1, k & 3);
#pragma unroll
                for (int k = 0; k < 16; ++k) NSOS_RELU_WORD16(H[6 + (k >> 3)][(k >> 2) & 1][k & 3], floor);
            }
            stamp();  // 3 + 2l: the exposed rest of the activation (quad 3), unless it rides in the next layer
        };
        dead();
#pragma unroll 1
        for (int l = 1; l <= 4; ++l) quad_layer(l, l != 1, l != 4, l == 1, l != 4);
        dead();
        {   // ---- layer 5 (skip): h part slice-major over all 16 tiles (4 chunks of 2 slices), then the x63 part (bias in its pad column)
            f32x4 Z[16][2];
            auto from_H01 = [&](auto sc, auto cc) { return H[0 + decltype(sc)::value][decltype(cc)::value]; };
            auto from_H23 = [&](auto sc, auto cc) { return H[2 + decltype(sc)::value][decltype(cc)::value]; };
            auto from_H45 = [&](auto sc, auto cc) { return H[4 + decltype(sc)::value][decltype(cc)::value]; };
            auto from_H67 = [&](auto sc, auto cc) { return H[6 + decltype(sc)::value][decltype(cc)::value]; };
            slice_chunk(IC(16), IC(2), IC(1), Z, from_H01, no_ride, 32);
            slice_chunk(IC(16), IC(2), IC(0), Z, from_H23, no_ride, 32);
            slice_chunk(IC(16), IC(2), IC(0), Z, from_H45, no_ride, 32);
            slice_chunk(IC(16), IC(2), IC(0), Z, from_H67, no_ride, 33);     // (layer 6's first quad)
            slice_chunk(IC(16), IC(2), IC(0), Z, from_ex, no_ride, 33);
            stamp();  // 12: MFMAs of layer 5
            activate_all(Z);
            stamp();  // 13: activation pass
        }
        dead();
#pragma unroll 1
        for (int l = 6; l <= 8; ++l) {
            quad_layer(l, l == 7, l == 6, l != 7, l == 6);
            if (l == 7) {
                dead();
                // ---- H = relu(h7): sigma head (models/nerf_mlp.py:77) and the semantic head (:79-80), all on the matrix pipe
                if constexpr (SEM == 0) {
                    // [bias of the raw tile][8 A operands: row 3 = alpha_linear]
                    pipeline16<12, 9, 0x1ull>(ring, ctx(), [&](auto ic, const f32x4& a32, const f32x4& prev) {
                        constexpr int g = decltype(ic)::value;
                        const u32x4 aop = __builtin_bit_cast(u32x4, a32);
                        if constexpr (g == 1) {
                            R[0] = T::mfma_k32(aop, H[0][0], prev);
                            R[1] = T::mfma_k32(aop, H[0][1], prev);
                        } else if constexpr (g >= 2 && g < 9) {
                            R[0] = T::mfma_k32(aop, H[g - 1][0], R[0]);
                            R[1] = T::mfma_k32(aop, H[g - 1][1], R[1]);
                        }
                    }, mid, tail, side, 33);
                    asm volatile("" : "+v"(R[0]), "+v"(R[1]));
                } else {
                    f32x4 S8[8][2];
                    u32x4 Sp[4][2];
                    auto from_Hlo = [&](auto sc, auto cc) { return H[decltype(sc)::value][decltype(cc)::value]; };
                    auto from_Hhi = [&](auto sc, auto cc) { return H[4 + decltype(sc)::value][decltype(cc)::value]; };
                    // SAVE: the first eight stores of relu(h7) ride here (at most three per chunk, six groups apart: see mlp_lp8.hip)
                    auto ride_sem = [&](auto gc_, auto ch_c) {
                        constexpr int g = decltype(gc_)::value, CH = decltype(ch_c)::value;
                        if constexpr (SAVE && g >= 3 && (g - 3) % 6 == 0 && (g - 3) / 6 < (CH == 2 ? 2 : 3)) store_h7_k(IC(3 * CH + (g - 3) / 6));
                    };
                    slice_chunk(IC(8), IC(4), IC(1), S8, from_Hlo, [&](auto gc_) { ride_sem(gc_, IC(0)); }, SEM == 2 ? 29 : 21);   // (the head's tail chunk)
#ifdef NSOS_LP16_HEAD_STAMPS      // (A/B builds only, scripts/diag/build_variant.sh: the heads' three chunks stamped one by one)
                    stamp();
#endif
                    slice_chunk(IC(8), IC(4), IC(0), S8, from_Hhi, [&](auto gc_) { ride_sem(gc_, IC(1)); }, 33);
#ifdef NSOS_LP16_HEAD_STAMPS
                    stamp();
#endif
                    // the head's hidden activations -> Sp (one exposed pass, inside the tail chunk below, before the logit MFMAs)
                    auto activate_sem = [&]() {
                        asm volatile("s_nop 7" ::: "memory");
#pragma unroll
                        for (int s = 0; s < 4; ++s)
#pragma unroll
                            for (int c = 0; c < 2; ++c)
#pragma unroll
                                for (int w = 0; w < 4; ++w) Sp[s][c][w] = T::pack2(S8[2 * s + (w >> 1)][c][2 * (w & 1)], S8[2 * s + (w >> 1)][c][2 * (w & 1) + 1]);
#pragma unroll
                        for (int s = 0; s < 4; ++s)
#pragma unroll
                            for (int c = 0; c < 2; ++c)
#pragma unroll
                                for (int w = 0; w < 4; ++w) NSOS_RELU_WORD16(Sp[s][c][w], 0u);
                        if constexpr (SAVE) {
                            // the hidden activations as the logit MFMAs consume them, 16-bit, TILE-MAJOR like sem_in ([group of 32 points][octet
                            // 0..15][point][8 channels], include/nerf_sos_hip.h): a lane holds channels 16 t + 4 q + {0..3} of tile t; one
                            // v_permlane16_swap per word pair -> 8 consecutive channels per lane = octet 4 s + 2 (q & 1) + (q >> 1) of its point:
                            // the lane offset of the relu(h7) stores serves these too, and a store writes four runs of 256 contiguous bytes
                            const unsigned long long hid_grp = reinterpret_cast<unsigned long long>(P.sem_hid16 + (long long)(wave_first >> 5) * 2048);
#pragma unroll
                            for (int s = 0; s < 4; ++s)
#pragma unroll
                                for (int c = 0; c < 2; ++c) {
                                    const auto r0 = __builtin_amdgcn_permlane16_swap(Sp[s][c][0], Sp[s][c][2], false, false);
                                    const auto r1 = __builtin_amdgcn_permlane16_swap(Sp[s][c][1], Sp[s][c][3], false, false);
                                    const u32x4 v = {r0[0], r1[0], r0[1], r1[1]};
#ifndef NSOS_LP16_SKIP_HID   // (A/B builds only)
                                    if (save_ok[c])
                                        // (s_nop: a store of more than 8 bytes followed by a write of its data registers needs wait states, and the
                                        //  hazard recognizer does not look inside asm -- eight stores back to back all got v[184:187] and stored each
                                        //  other's words)
                                        asm volatile("global_store_dwordx4 %0, %1, %2 offset:%3 nt\n\ts_nop 1" : : "v"(save_off), "v"(v), "s"(hid_grp + (unsigned long long)((s * 2048 + c * 256) & ~4095)),
                                                     "i"((s * 2048 + c * 256) & 4095) : "memory");
#endif
                                }
                        }
                    };
                    if constexpr (SEM == 2) {
                        // [x63 part of semantic_linear.0: 2 slices x 8 tiles, bias in the pad column][bias of the raw tile][8 x sigma][4 x logits]
                        pipeline16<32, 29, 1ull << 16>(ring, ctx(), [&](auto ic, const f32x4& a32, const f32x4& prev) {
                            constexpr int g = decltype(ic)::value;
                            const u32x4 aop = __builtin_bit_cast(u32x4, a32);
                            if constexpr (g < 16) {
                                constexpr int sl = g >> 3, t = g & 7;
                                S8[t][0] = T::mfma_k32(aop, ex[sl][0], S8[t][0]);
                                S8[t][1] = T::mfma_k32(aop, ex[sl][1], S8[t][1]);
                            } else if constexpr (g == 17) {
                                R[0] = T::mfma_k32(aop, H[0][0], prev);
                                R[1] = T::mfma_k32(aop, H[0][1], prev);
                            } else if constexpr (g >= 18 && g < 25) {
                                R[0] = T::mfma_k32(aop, H[g - 17][0], R[0]);
                                R[1] = T::mfma_k32(aop, H[g - 17][1], R[1]);
                            } else if constexpr (g >= 25 && g < 29) {
                                if constexpr (g == 25) activate_sem();
                                R[0] = T::mfma_k32(aop, Sp[g - 25][0], R[0]);
                                R[1] = T::mfma_k32(aop, Sp[g - 25][1], R[1]);
                            }
                            ride_sem(ic, IC(2));
                        }, mid, tail, side, 33);
                    } else {
                        // [bias of semantic_linear.0 as a constant K-slice: 8 tiles][bias of the raw tile][8 x sigma][4 x logits]
                        const unsigned one_w = q == 0 ? (unsigned)(T::kOnes & 0xffffu) : 0u;     // B = e_0: k position (q 0, e 0) is 1.0
                        const u32x4 ones = {one_w, 0u, 0u, 0u};
                        pipeline16<24, 21, 1ull << 8>(ring, ctx(), [&](auto ic, const f32x4& a32, const f32x4& prev) {
                            constexpr int g = decltype(ic)::value;
                            const u32x4 aop = __builtin_bit_cast(u32x4, a32);
                            if constexpr (g < 8) {
                                S8[g][0] = T::mfma_k32(aop, ones, S8[g][0]);
                                S8[g][1] = T::mfma_k32(aop, ones, S8[g][1]);
                            } else if constexpr (g == 9) {
                                R[0] = T::mfma_k32(aop, H[0][0], prev);
                                R[1] = T::mfma_k32(aop, H[0][1], prev);
                            } else if constexpr (g >= 10 && g < 17) {
                                R[0] = T::mfma_k32(aop, H[g - 9][0], R[0]);
                                R[1] = T::mfma_k32(aop, H[g - 9][1], R[1]);
                            } else if constexpr (g >= 17 && g < 21) {
                                if constexpr (g == 17) activate_sem();
                                R[0] = T::mfma_k32(aop, Sp[g - 17][0], R[0]);
                                R[1] = T::mfma_k32(aop, Sp[g - 17][1], R[1]);
                            }
                            ride_sem(ic, IC(2));
                        }, mid, tail, side, 33);
                    }
                    asm volatile("" : "+v"(R[0]), "+v"(R[1]));
                }
                stamp();  // 18 (l == 7 only; the later slots shift by one): sigma + semantic heads
            }
        }
        dead();
        // ---- view branch (models/nerf_mlp.py:87-92): cat([feature, dir27]) -> 128 -> rgb.  H = feature (no activation).
        // Quad-major over the 8 hidden tiles: chunk j = tiles 4j..4j+3 x (8 feature slices + the direction slice, whose pad column
        // carries the bias): 36 A operands; the first quad's activation rides in the second chunk.
        u32x4 ed[2];
        {
#pragma unroll
            for (int c = 0; c < 2; ++c) {
                const int ray_c = park[64 * c];
                float dv[3];
#pragma unroll
                for (int k = 0; k < 3; ++k) dv[k] = P.viewdirs[3ll * ray_c + k];
                u32x4 e1[1];
                encode16<T, 1, NSOS_DIR_FREQS, 27>(e1, dv, tabs + kTabDir, q);   // feature 27 (pad) = 1.0: the bias input of views_linears.0
                ed[c] = e1[0];
            }
        }
        stamp();  // direction encoding
        u32x4 Vp[4][2];
        auto view_chunk = [&](auto& zq, auto&& ride, const int nfill) {
            pipeline16<36, 36, 0ull>(ring, ctx(), [&](auto ic, const f32x4& a32, const f32x4&) {
                constexpr int g = decltype(ic)::value, s = g >> 2, t = g & 3;
                const u32x4 aop = __builtin_bit_cast(u32x4, a32);
                if constexpr (s == 0) {
                    zq[t][0] = T::mfma_k32(aop, H[0][0], zero4);
                    zq[t][1] = T::mfma_k32(aop, H[0][1], zero4);
                } else if constexpr (s < 8) {
                    zq[t][0] = T::mfma_k32(aop, H[s][0], zq[t][0]);
                    zq[t][1] = T::mfma_k32(aop, H[s][1], zq[t][1]);
                } else {
                    zq[t][0] = T::mfma_k32(aop, ed[0], zq[t][0]);
                    zq[t][1] = T::mfma_k32(aop, ed[1], zq[t][1]);
                }
                ride(ic);
            }, mid, tail, side, nfill);
#pragma unroll
            for (int t = 0; t < 4; ++t) asm volatile("" : "+v"(zq[t][0]), "+v"(zq[t][1]));
        };
        view_chunk(Zq[0], no_ride, 32);      // (fetches the next tile's layer 0)
        view_chunk(Zq[1], [&](auto gc_) {
            constexpr int g = decltype(gc_)::value;
            if constexpr (g >= 9 && g <= 24) { constexpr int k = g - 9; NSOS_RELU_WORD16(Vp[k >> 3][(k >> 2) & 1][k & 3], 0u); }
            if constexpr (g >= 8 && g < 24) { constexpr int k = g - 8; Vp[k >> 3][(k >> 2) & 1][k & 3] = quad_word(Zq[0], k >> 3, (k >> 2) & 1, k & 3); }
        }, 33);                              // (the next tile's layer 1, first quad)
        stamp();  // view-branch MFMAs
        {
            asm volatile("s_nop 7" ::: "memory");
#pragma unroll
            for (int k = 0; k < 16; ++k) Vp[2 + (k >> 3)][(k >> 2) & 1][k & 3] = quad_word(Zq[1], k >> 3, (k >> 2) & 1, k & 3);
#pragma unroll
            for (int k = 0; k < 16; ++k) NSOS_RELU_WORD16(Vp[2 + (k >> 3)][(k >> 2) & 1][k & 3], 0u);
        }
        // the inputs of the NaN check, re-read (L2 hits) rather than kept alive across the tile; requested before the rgb chunk
        float chk[2];
#pragma unroll
        for (int c = 0; c < 2; ++c) {
            const int ray_c = park[64 * c];
            int pw = 16 * c + n;
            asm volatile("" : "+v"(pw));
            const int gc_b = (pw < n_here) ? wave_first + pw : n_pts - 1;
            const float nz = P.z_vals[gc_b];
            float acc = nz - nz;
#pragma unroll
            for (int k = 0; k < 3; ++k) {
                const float o = P.rays_o[3ll * ray_c + k], d = P.rays_d[3ll * ray_c + k], v = P.viewdirs[3ll * ray_c + k];
                acc += ((o - o) + (d - d)) + (v - v);
            }
            chk[c] = acc;
        }
        // rgb_linear: rows 0..2 of the raw tile += W_rgb x the view branch's hidden activations (4 slices), operands from LDS
        {
            f32x4 ra[4];
            const unsigned ra_addr = lds_base + (unsigned)kRgbLds + (unsigned)(lane * 16);
            static_for<0, 4>([&](auto kc) { lds_read_a<decltype(kc)::value * 1024>(ra[decltype(kc)::value], ra_addr); });
            // (drains the ring's look-ahead reads of the next chunk too: its counted waits then pass at once)
            asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(ra[0]), "+v"(ra[1]), "+v"(ra[2]), "+v"(ra[3]) : : "memory");
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const u32x4 aop = __builtin_bit_cast(u32x4, ra[k]);
                R[0] = T::mfma_k32(aop, Vp[k][0], R[0]);
                R[1] = T::mfma_k32(aop, Vp[k][1], R[1]);
            }
        }
        asm volatile("" : "+v"(R[0]), "+v"(R[1]));
        stamp();  // rgb MFMAs
        {
            asm volatile("s_nop 7" ::: "memory");
            // NaN / Inf in a point's inputs must come out as NaN (the reference propagates them; the packed integer ReLU launders them)
#pragma unroll
            for (int c = 0; c < 2; ++c)
                if (chk[c] != chk[c]) {
                    const float qnan = __builtin_nanf("");
                    R[c] = f32x4{qnan, qnan, qnan, qnan};
                }
            if constexpr (C == 4) {
                // lane (n, 0) holds [r, g, b, sigma] of point 16 c + n: 16 lanes x 16 B contiguous per block
#pragma unroll
                for (int c = 0; c < 2; ++c)
                    if (q == 0 && 16 * c + n < n_here) *reinterpret_cast<f32x4*>(P.raw + (long long)(wave_first + 16 * c + n) * 4) = R[c];
            } else {
                // 24 B per point: staged through 768 B of LDS per wave, then 48 lanes store 16 B each (full sectors: see mlp_lp8.hip)
                float* const stage = reinterpret_cast<float*>(lds + kSlots * kSlotBytes + kAuxWords * 4) + wave_s * 192;
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");    // the parked ray indices were read above
                __builtin_amdgcn_wave_barrier();
                if (n_here == 32) {
#pragma unroll
                    for (int c = 0; c < 2; ++c) {
                        if (q == 0) {                                  // (rows are 24 bytes apart: 8-byte aligned pieces)
                            *reinterpret_cast<f32x2*>(stage + 6 * (16 * c + n)) = f32x2{R[c][0], R[c][1]};
                            *reinterpret_cast<f32x2*>(stage + 6 * (16 * c + n) + 2) = f32x2{R[c][2], R[c][3]};
                        }
                        if (q == 1) *reinterpret_cast<f32x2*>(stage + 6 * (16 * c + n) + 4) = f32x2{R[c][0], R[c][1]};
                    }
                    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                    __builtin_amdgcn_wave_barrier();
                    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
                    if (lane < 48) *reinterpret_cast<f32x4*>(P.raw + (long long)wave_first * C + 4 * lane) = *reinterpret_cast<const f32x4*>(stage + 4 * lane);
                    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");   // the next tile's writes to the stage follow these reads
                    __builtin_amdgcn_wave_barrier();
                } else {
#pragma unroll
                    for (int c = 0; c < 2; ++c)
                        if (16 * c + n < n_here) {          // the ragged last wave of the launch: per point
                            float* out = P.raw + (long long)(wave_first + 16 * c + n) * C;
                            if (q == 0) {
                                *reinterpret_cast<f32x2*>(out) = f32x2{R[c][0], R[c][1]};
                                *reinterpret_cast<f32x2*>(out + 2) = f32x2{R[c][2], R[c][3]};
                            } else if (q == 1) {
                                *reinterpret_cast<f32x2*>(out + 4) = f32x2{R[c][0], R[c][1]};
                            }
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

constexpr int kLdsBytes16 = kRgbLds + kLp16TailBytes;   // 4 weight slots + encoding tables + the output stage (768 B per wave) + rgb_linear's operands
static_assert(kLdsBytes16 <= 160 * 1024, "LDS");

template <class T, int SEM, bool SAVE, bool PROF>
int32_t launch16p(const LpParams& p, hipStream_t stream) {
    static NsosPerDeviceFlag configured_on;
    bool& configured = configured_on.here();
    if (!configured) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&mlp_lp16_kernel<T, SEM, SAVE, PROF>),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, kLdsBytes16);
        if (e != hipSuccess) return (int32_t)e;
        configured = true;
    }
    const int cus = nsos_device_cus();
    LpParams q = p;
    q.n_tiles = (int)((p.n_pts + kTile16 - 1) / kTile16);
    const int grid = q.n_tiles < cus ? q.n_tiles : cus;
    hipLaunchKernelGGL((mlp_lp16_kernel<T, SEM, SAVE, PROF>), dim3(grid), dim3(64 * kW16), kLdsBytes16, stream, q);
    return nsos_launch_status();
}
template <class T, int SEM, bool SAVE>
int32_t launch16(const LpParams& p, hipStream_t stream) {
    if (p.prof) {   // diagnostics: the stamped instantiations exist for the shapes the phase-profile scripts use
        if constexpr (SEM != 1) return launch16p<T, SEM, SAVE, true>(p, stream);
        else return NSOS_ERR_UNSUPPORTED;
    }
    return launch16p<T, SEM, SAVE, false>(p, stream);
}

// ------------------------------------------------------------------------------------------ packing
// One descriptor per chunk of the stream (36 groups of 1 KiB = 512 16-bit elements; lane (i = lane & 15, q = lane >> 4), element e):
//   an A operand of (tile t, slice s) holds W[16 t + i][col(s, q, e)], col by the input's kind:
//     hidden input (an accumulator-layout H):  col_base + 32 s + 16 (e >> 2) + 4 q + (e & 3)
//     encoding input (natural order):          col_base + 32 s + 8 q + e;  past n_enc features: the bias if this is the pad column, else 0
//   a bias group holds, as fp32, bias[16 t + 4 q + r] in the lane's four words.
enum Kind16 { kQ16 = 0,      // tile quad a0 of a hidden layer: 32 A operands, g = 4 s + t, then the bias block (group 32: [t][q][4 x fp32])
              kSlice16 = 1,  // slice-major: g = (s - s0) * nt + t; hidden input
              kEnc16 = 2,    // slice-major over encoding slices s0..: g = (s - s0) * nt + t; bias in the pad column (f == pad_at)
              kView16 = 3,   // quad a0 of the view branch: g = s * 4 + t, s < 8 hidden (feature), s == 8 the direction slice (bias at f == 27)
              kSemTail2 = 4, kSemTail1 = 5, kSigma16 = 6, kRgb16 = 7 };
struct Chunk16 {
    const float* w;
    const float* bias;
    int in_dim, col_base, kind, a0, nt;
};
struct Pack16Params {
    Chunk16 ch[48];
    int n_chunks;
    int first, count;     // the chunks [first, first + count) are written
    int tail;             // != 0: also the four rgb_linear operands behind the last chunk
    const float* alpha_w; const float* alpha_b;
    const float* rgb_w; const float* rgb_b;
    const float* sem0_w; const float* sem0_b;
    const float* sem2_w; const float* sem2_b;
    unsigned short* chunks;
};

__device__ __forceinline__ int hid_col(int s, int q, int e) { return 32 * s + 16 * (e >> 2) + 4 * q + (e & 3); }

template <class T>
__global__ __launch_bounds__(256) void lp16_pack_kernel(const Pack16Params P) {
    const long long gid = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    const long long per_chunk = kSlotBytes / 2;  // 16-bit elements per slot
    const long long body = (long long)P.count * per_chunk;
    if (gid >= body + (P.tail ? kLp16TailBytes / 2 : 0)) return;
    const bool in_tail = gid >= body;
    const Chunk16 ck = in_tail ? Chunk16{nullptr, nullptr, 0, 0, kRgb16, 0, 0} : P.ch[P.first + gid / per_chunk];
    const int within = in_tail ? (int)(gid - body) : (int)(gid % per_chunk);
    const int g = within >> 9, lane = (within >> 3) & 63, e = within & 7;  // 512 elements per group
    const int i = lane & 15, q = lane >> 4;
    const int W = NSOS_NET_WIDTH, X = NSOS_XYZ_DIM;
    // a group is either 512 16-bit operand elements or 256 fp32 bias values (two 16-bit halves each)
    bool is_bias = false;
    float v = 0.0f;
    auto raw_bias = [&]() {                          // the raw tile: rows 0..2 rgb, 3 sigma, 4..5 semantics
        is_bias = true;
        const int row = 4 * q + (e >> 1);
        v = row < 3 ? P.rgb_b[row] : row == 3 ? P.alpha_b[0] : (row < 6 && P.sem2_b) ? P.sem2_b[row - 4] : 0.0f;
    };
    auto sigma_a = [&](int s) { v = i == 3 ? P.alpha_w[hid_col(s, q, e)] : 0.0f; };
    auto logits_a = [&](int s) { v = (i == 4 || i == 5) ? P.sem2_w[(i - 4) * (W / 2) + hid_col(s, q, e)] : 0.0f; };
    switch (ck.kind) {
        case kQ16: {
            if (g < 32) {
                const int sl = g >> 2, t = 4 * ck.a0 + (g & 3);
                v = ck.w[(long long)(16 * t + i) * ck.in_dim + ck.col_base + hid_col(sl, q, e)];
            } else if (g == 32) {     // the bias block: [tile 0..3][q][4 x fp32] = the first 128 16-bit elements of the group
                const int word = (lane * 8 + e) >> 1;            // fp32 index inside the group
                if (word < 64) {
                    is_bias = true;
                    v = ck.bias ? ck.bias[16 * (4 * ck.a0 + (word >> 4)) + (word & 15)] : 0.0f;    // word = 16 t + 4 q + r = feature within the quad
                }
            }
        } break;
        case kSlice16: {
            if (g < 32) {
                const int s = ck.a0 + g / ck.nt, t = g % ck.nt;
                v = ck.w[(long long)(16 * t + i) * ck.in_dim + ck.col_base + hid_col(s, q, e)];
            }
        } break;
        case kEnc16: {
            if (g < 2 * ck.nt) {
                const int s = g / ck.nt, t = g % ck.nt, f = 32 * s + 8 * q + e;
                if (f < X) v = ck.w[(long long)(16 * t + i) * ck.in_dim + ck.col_base + f];
                else if (f == 63 && ck.bias) v = ck.bias[16 * t + i];
            }
        } break;
        case kView16: {
            const int s = g >> 2, t = 4 * ck.a0 + (g & 3);
            if (s < 8) v = ck.w[(long long)(16 * t + i) * ck.in_dim + hid_col(s, q, e)];
            else {
                const int f = 8 * q + e;
                if (f < NSOS_DIR_DIM) v = ck.w[(long long)(16 * t + i) * ck.in_dim + W + f];
                else if (f == 27) v = ck.bias[16 * t + i];
            }
        } break;
        case kSemTail2: {   // [x63 part of semantic_linear.0: 16][raw bias][sigma: 8][logits: 4]
            if (g < 16) {
                const int s = g >> 3, t = g & 7, f = 32 * s + 8 * q + e;
                if (f < X) v = P.sem0_w[(long long)(16 * t + i) * (W + X) + W + f];
                else if (f == 63) v = P.sem0_b[16 * t + i];
            } else if (g == 16) raw_bias();
            else if (g < 25) sigma_a(g - 17);
            else if (g < 29) logits_a(g - 25);
        } break;
        case kSemTail1: {   // [bias of semantic_linear.0 as a constant slice: 8][raw bias][sigma: 8][logits: 4]
            if (g < 8) v = (q == 0 && e == 0) ? P.sem0_b[16 * g + i] : 0.0f;
            else if (g == 8) raw_bias();
            else if (g < 17) sigma_a(g - 9);
            else if (g < 21) logits_a(g - 17);
        } break;
        case kSigma16: {
            if (g == 0) raw_bias();
            else if (g < 9) sigma_a(g - 1);
        } break;
        case kRgb16: {
            if (g < 4) v = i < 3 ? P.rgb_w[i * (W / 2) + hid_col(g, q, e)] : 0.0f;
        } break;
    }
    const long long at = in_tail ? (long long)P.n_chunks * per_chunk + within : (long long)P.first * per_chunk + gid;
    if (is_bias) {
        const unsigned u = __builtin_bit_cast(unsigned, v);
        P.chunks[at] = (e & 1) ? (unsigned short)(u >> 16) : (unsigned short)(u & 0xffffu);
    } else {
        P.chunks[at] = T::bits(v);
    }
}

}  // namespace

namespace nsos {
namespace lp {

int32_t pack_lp16(const void* tensors, int32_t sem_mode, bool is_f16, unsigned char* chunks, hipStream_t stream, bool heads_only) {
    const nsos_mlp_tensors* T_ = static_cast<const nsos_mlp_tensors*>(tensors);
    const int X = NSOS_XYZ_DIM, W = NSOS_NET_WIDTH;
    Pack16Params P = {};
    int n = 0;
    auto add = [&](const float* w, const float* bias, int in_dim, int col, int kind, int a0, int nt) {
        P.ch[n++] = Chunk16{w, bias, in_dim, col, kind, a0, nt};
    };
    auto quads = [&](const float* w, const float* b) {
        for (int c = 0; c < 4; ++c) add(w, b, W, 0, kQ16, c, 4);
    };
    add(T_->pts_w[0], T_->pts_b[0], X, 0, kEnc16, 0, 16);
    for (int l = 1; l <= 4; ++l) quads(T_->pts_w[l], T_->pts_b[l]);
    for (int j = 0; j < 4; ++j) add(T_->pts_w[5], nullptr, X + W, X, kSlice16, 2 * j, 16);   // skip layer: input = cat([x63, h]): h part first ...
    add(T_->pts_w[5], T_->pts_b[5], X + W, 0, kEnc16, 0, 16);                                 // ... then x63, whose pad column carries the bias
    quads(T_->pts_w[6], T_->pts_b[6]);
    quads(T_->pts_w[7], T_->pts_b[7]);
    if (sem_mode) {
        const int in_dim = sem_mode == NSOS_SEM_COORD ? W + X : W;
        for (int j = 0; j < 2; ++j) add(T_->sem0_w, nullptr, in_dim, 0, kSlice16, 4 * j, 8);
        add(nullptr, nullptr, 0, 0, sem_mode == NSOS_SEM_COORD ? kSemTail2 : kSemTail1, 0, 0);
    } else {
        add(nullptr, nullptr, 0, 0, kSigma16, 0, 0);
    }
    quads(T_->feature_w, T_->feature_b);
    for (int j = 0; j < 2; ++j) add(T_->views_w, T_->views_b, W + NSOS_DIR_DIM, 0, kView16, j, 4);
    if (n != lp16_chunks(sem_mode)) return NSOS_ERR_UNSUPPORTED;
    P.n_chunks = n;
    P.first = heads_only ? 30 : 0;                // the head's chunks: two slice chunks + the tail (sigma, raw bias, logits)
    P.count = heads_only ? 3 : n;
    P.alpha_w = T_->alpha_w; P.alpha_b = T_->alpha_b;
    P.rgb_w = T_->rgb_w; P.rgb_b = T_->rgb_b;
    P.sem0_w = sem_mode ? T_->sem0_w : nullptr; P.sem0_b = sem_mode ? T_->sem0_b : nullptr;
    P.sem2_w = sem_mode ? T_->sem2_w : nullptr; P.sem2_b = sem_mode ? T_->sem2_b : nullptr;
    P.chunks = reinterpret_cast<unsigned short*>(chunks);
    P.tail = heads_only ? 0 : 1;                  // rgb_linear's operands behind the chunks (kLp16TailBytes)
    const long long total = (long long)P.count * (kSlotBytes / 2) + (P.tail ? kLp16TailBytes / 2 : 0);
    const dim3 grid((unsigned)((total + 255) / 256)), block(256);
    if (is_f16) hipLaunchKernelGGL(lp16_pack_kernel<F16>, grid, block, 0, stream, P);
    else hipLaunchKernelGGL(lp16_pack_kernel<BF16>, grid, block, 0, stream, P);
    return nsos_launch_status();
}

// dispatch used by forward_rays_lp (mlp_lp.hip); sem_mode and dtype were validated there
int32_t launch_lp16(const LpParams& p, int32_t sem_mode, bool is_f16, bool save, hipStream_t st) {
    if (save) {
        if (is_f16) return sem_mode == 1 ? launch16<F16, 1, true>(p, st) : launch16<F16, 2, true>(p, st);
        return sem_mode == 1 ? launch16<BF16, 1, true>(p, st) : launch16<BF16, 2, true>(p, st);
    }
    if (is_f16) {
        switch (sem_mode) {
            case 0: return launch16<F16, 0, false>(p, st);
            case 1: return launch16<F16, 1, false>(p, st);
            default: return launch16<F16, 2, false>(p, st);
        }
    }
    switch (sem_mode) {
        case 0: return launch16<BF16, 0, false>(p, st);
        case 1: return launch16<BF16, 1, false>(p, st);
        default: return launch16<BF16, 2, false>(p, st);
    }
}

}  // namespace lp
}  // namespace nsos
