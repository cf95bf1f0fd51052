// K2-LP: the fused positional-encoding + MLP kernel with 16-bit (fp16 or bf16) MFMA inputs and fp32
// accumulation, for the reduced-precision configurations (BASELINE C3 bf16, C5 fp16 eval-only).  Same network,
// same replaced reference code as mlp_fused.hip (models/embedder.py:34-48, models/nerf_mlp.py:67-100,179-215);
// NOT used by the fp32 parity path or by bench.py's headline number.
//
// What changes against the exact-fp32 kernel (DESIGN.md "K2-LP"):
//   * v_mfma_f32_32x32x16_{f16,bf16}: 32 cycles, K = 16 -> 16x the MAC rate, so everything that was free next to a
//     64-cycle fp32 MFMA now matters: A operands are 16 B per lane per MFMA, so each wave owns TWO column tiles
//     (64 points) and every ds_read_b128 feeds two MFMAs; a 256 x 256 layer is only 8192 cycles of matrix pipe.
//   * Accumulator -> B-operand feedback still works: a K-slice of 16 features is (tile t, reg half u): lane half kg
//     supplies k-slots 8kg..8kg+7 = accumulator regs 8u..8u+7 = features 32t + 16u + {0,1,2,3,8,9,10,11} + 4kg, so
//     after each layer ONE batched pass converts Z (fp32, AGPRs) to packed 16-bit pairs (v_cvt_pk) with the ReLU as
//     v_pk_max_i16(x, 0) (sign bit set <=> negative int16), 4 VGPRs per slice, named directly as MFMA srcB.
//   * Weights are packed to 16 bit in MFMA order; chunks are <= 34 A operands (1 KiB each) in 36 KiB slots, FOUR
//     LDS slots (144 KiB): chunk c consumed, c+1 resident, c+2 landing, c+3 being issued; the per-chunk barrier
//     waits with a COUNTED vmcnt (the 9 newest DMA pieces stay in flight), giving the DMA two chunks of lead.
//   * Bias: leading K-slice with B = 1.0 (hidden layers, heads); layer 0 carries it in the encoding's pad slot.
//   * sigma head: v_dot2c on the packed activations; rgb / semantics heads: fp32 VALU on the fp32 accumulators.
// Compiled with -ffp-contract=off (x = o + d*z stays a separately rounded multiply and add).
#include <cstdlib>

#include "lp_common.h"

using namespace nsos;

using namespace nsos::lp;

namespace {


// ------------------------------------------------------------------------------------------ the kernel
// SAVE: training-mode variant that also stores what the semantic head's backward needs (K5, frozen backbone)
template <class T, int SEM, bool SAVE = false>
__global__ __launch_bounds__(256, 1) void mlp_lp_kernel(const LpParams P) {
    extern __shared__ __attribute__((aligned(16))) unsigned char lds[];  // 4 x 36 KiB weight slots + 4 KiB head weights
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int pj = lane & 31, kg = lane >> 5;
    constexpr int NCH = lp_chunks(SEM);
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
        // all DMA pieces except the newest kDmaPieces (chunk cur+2, issued one chunk ago) must have landed:
        // that is chunk cur+1, which the end of this chunk starts to read.  (Extra outstanding VM operations
        // of the compiler only make this counted wait stricter.)
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
    // head weights and biases (aux, 4 KiB) live in LDS behind the slots: a global load per use would put an L2
    // round trip in front of every few VALU instructions of the heads
    const unsigned* const aux_l = reinterpret_cast<const unsigned*>(lds + kSlots * kSlotBytes);
    *reinterpret_cast<u32x4*>(lds + kSlots * kSlotBytes + threadIdx.x * 16) = reinterpret_cast<const u32x4*>(P.aux)[threadIdx.x];
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    static_for<0, kRing>([&](auto ic) { lds_read_a<decltype(ic)::value * 1024>(ring[decltype(ic)::value], c0); });
    lgkm_wait<0>();
    NSOS_PIN();

    // one chunk: NG groups, group g = A operand `a0 + g` of a part with NT output tiles and NB leading bias
    // A operands; the K-slice of A operand a >= NB is (a - NB) / NT, its tile (a - NB) % NT.
    // zf_c != 0: the part starts the accumulation (slice 0 uses C = 0 instead of the old accumulator contents).
    // ride(g): extra work placed in the MFMA shadows after group g (the compact SAVE stores of the semantic head's input)
    auto run_chunk = [&](auto ng_c, auto nt_c, auto nb_c, auto a0_c, auto nwork_c, auto zf_c, auto& acc, auto&& bsel, auto&& ride) {
        constexpr int NG = decltype(ng_c)::value, NT = decltype(nt_c)::value, NB = decltype(nb_c)::value;
        constexpr int A0 = decltype(a0_c)::value, NWORK = decltype(nwork_c)::value;
        constexpr bool ZF = decltype(zf_c)::value != 0;
        a_pipeline<NG, kRing, kPre, kMid>(ring, ctx(), [&](auto ic, const f32x4& a32) {
            constexpr int g = decltype(ic)::value, a = A0 + g;
            const u32x4 aop = __builtin_bit_cast(u32x4, a32);
            static_for<0, 2>([&](auto cc) {
                constexpr int c = decltype(cc)::value;
                if constexpr (g < NWORK) {
                    if constexpr (a < NB) {
                        const f32x16 zero = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
                        const u32x4 ones = {T::kOnes, T::kOnes, T::kOnes, T::kOnes};
                        acc[c][a] = T::mfma(aop, ones, zero);
                    } else {
                        constexpr int s = (a - NB) / NT, t = (a - NB) % NT;
                        if constexpr (ZF && s == 0) {
                            const f32x16 zero = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
                            acc[c][t] = T::mfma(aop, bsel(cc, std::integral_constant<int, s>{}), zero);
                        } else {
                            acc[c][t] = T::mfma(aop, bsel(cc, std::integral_constant<int, s>{}), acc[c][t]);
                        }
                    }
                }
                dma_slot<(g - kMid) * 2 + c, kDmaPieces>(side);
            });
            ride(ic);
        }, mid, tail);
    };
    auto no_ride = [](auto) {};
#define IC(n) std::integral_constant<int, (n)> {}

    for (int tile = blockIdx.x; tile < P.n_tiles; tile += gridDim.x) {
        int stamp_k = 0;
        auto stamp = [&]() {  // diagnostics only: one s_memtime per phase of the second tile of blocks 0..3
            if (P.prof && tile == (int)(blockIdx.x + gridDim.x) && blockIdx.x < 4) {
                const unsigned long long t = __builtin_readcyclecounter();
                if (lane == 0 && stamp_k < kProfSlots) P.prof[(blockIdx.x * 4 + wave) * kProfSlots + stamp_k] = t;
            }
            ++stamp_k;
        };
        stamp();  // 0: tile start
        // ---- this lane's two points (column tile c: point tile*256 + wave*64 + c*32 + pj)
        int ray_of[2];  // n_rays <= n_pts < 2^31 * 256, rays themselves < 2^31 (checked at the entry point)
        bool save_ok[2] = {false, false};       // SAVE, compact: this lane's point is in range and the 16-bit matrix is wanted
        unsigned* save_row[2] = {nullptr, nullptr};
        u32x4 ex[2][4];
#pragma unroll
        for (int c = 0; c < 2; ++c) {
            const long long gp = (long long)tile * kTilePts + wave * 64 + c * 32 + pj;
            const long long gc = gp < P.n_pts ? gp : P.n_pts - 1;
            const long long ray = gc / P.n_samples;
            const float z = P.z_vals[gc];
            float x[3];
            ray_of[c] = (int)ray;
            if constexpr (SAVE) {
                save_ok[c] = gp < P.n_pts && P.sem_in16 != nullptr;
                save_row[c] = P.sem_in16 + gc * 160;
            }
#pragma unroll
            for (int k = 0; k < 3; ++k) {
                const float m = P.rays_d[3 * ray + k] * z;  // models/sampler.py:70,166 (mul, then add)
                x[k] = P.rays_o[3 * ray + k] + m;
            }
            {
                Enc<NSOS_XYZ_FREQS, SliceHalf> e;
                e.evaluate_hw(x, kg);
                ex[c][0] = enc_slice<T, NSOS_XYZ_FREQS, 0, true>(e, x, kg);
                ex[c][1] = enc_slice<T, NSOS_XYZ_FREQS, 1, true>(e, x, kg);
                ex[c][2] = enc_slice<T, NSOS_XYZ_FREQS, 2, true>(e, x, kg);
                ex[c][3] = enc_slice<T, NSOS_XYZ_FREQS, 3, true>(e, x, kg);  // feature 63 (pad) = 1.0: layer-0 bias
            }
        }

        f32x16 Z[2][8];
        u32x4 H[2][16];
        float sigma[2] = {0.0f, 0.0f}, sem_out[2][2] = {{0.0f, 0.0f}, {0.0f, 0.0f}};
        auto from_ex = [&](auto cc, auto sc) { return ex[decltype(cc)::value][decltype(sc)::value]; };
        auto from_H = [&](auto cc, auto sc) { return H[decltype(cc)::value][decltype(sc)::value]; };

        // pts_linears.0: 4 encoded slices x 8 tiles; the first slice of every tile starts from C = 0
        stamp();  // 1: inputs + xyz encoding
        run_chunk(IC(32), IC(8), IC(0), IC(0), IC(32), IC(1), Z, from_ex, no_ride);
        stamp();  // 2: L0 MFMAs
        activate<T, 8, true>(H, Z);
        stamp();  // 3: L0 activation
        // pts_linears.1..7 (l = 1..7) and feature_linear (l = 8)
#pragma unroll 1
        for (int l = 1; l <= 8; ++l) {
            run_chunk(IC(34), IC(8), IC(8), IC(0), IC(34), IC(0), Z, from_H, no_ride);
            run_chunk(IC(34), IC(8), IC(8), IC(34), IC(34), IC(0), Z, from_H, no_ride);
            run_chunk(IC(34), IC(8), IC(8), IC(68), IC(34), IC(0), Z, from_H, no_ride);
            run_chunk(IC(34), IC(8), IC(8), IC(102), IC(34), IC(0), Z, from_H, no_ride);
            if (l == 5) run_chunk(IC(32), IC(8), IC(0), IC(0), IC(32), IC(0), Z, from_ex, no_ride);  // skip connection
            stamp();  // 2 + 2l: MFMAs of layer l
            if (l < 8) activate<T, 8, true>(H, Z); else activate<T, 8, false>(H, Z);
            stamp();  // 3 + 2l: activation pass
            if (l == 7) {
                // sigma head: dot of the packed activations with packed weights (models/nerf_mlp.py:77)
                const unsigned* aw = aux_l + kAuxAlphaW + kg * 64;
                // four chains per column (one per packed word q), then (p0 + p1) + (p2 + p3): the order mlp_lp8_kernel sums in
                float pq[2][4];
#pragma unroll
                for (int c = 0; c < 2; ++c) {
                    pq[c][0] = kg ? 0.0f : __builtin_bit_cast(float, aux_l[kAuxScalars]);
                    pq[c][1] = pq[c][2] = pq[c][3] = 0.0f;
                }
#pragma unroll
                for (int s = 0; s < 16; ++s) {
                    const u32x4 w = *reinterpret_cast<const u32x4*>(aw + 4 * s);
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        pq[0][q] = T::dot2(H[0][s][q], w[q], pq[0][q]);
                        pq[1][q] = T::dot2(H[1][s][q], w[q], pq[1][q]);
                    }
                }
                asm volatile("s_nop 3" ::: "memory");  // v_dot2c result -> non-dot VALU read: 3 wait states hipcc cannot see (asm)
                sigma[0] = both_halves((pq[0][0] + pq[0][1]) + (pq[0][2] + pq[0][3]));
                sigma[1] = both_halves((pq[1][0] + pq[1][1]) + (pq[1][2] + pq[1][3]));
                if constexpr (SEM != 0) {  // semantic head (models/nerf_mlp.py:79-80)
                    f32x16 sacc[2][4];
                    // SAVE, compact sem_in: the 72 stores of [relu(h7) | x63 | 1] (packed words as they are) ride in these two chunks'
                    // MFMA shadows, two per group from group 11 on -- as a burst they made the next barrier wait for HBM
                    auto ride_sem = [&](auto gc_, auto ch_c) {
                        constexpr int g = decltype(gc_)::value, CH = decltype(ch_c)::value;
                        if constexpr (SAVE && g >= 11) {
                            static_for<0, 2>([&](auto jc) {
                                constexpr int k = CH * 46 + (g - 11) * 2 + decltype(jc)::value;
                                if constexpr (k < 64) {          // word pair (q, q+1) of slice 2t+u of column c = features 32t + 8(2u + q/2) + 4kg + {0..3}
                                    constexpr int c = k >> 5, t = (k >> 2) & 7, u = (k >> 1) & 1, q = 2 * (k & 1);
                                    if (save_ok[c])
                                        *reinterpret_cast<u32x2*>(save_row[c] + (32 * t + 8 * (2 * u + (q >> 1)) + 4 * kg) / 2) =
                                            u32x2{H[c][2 * t + u][q], H[c][2 * t + u][q + 1]};
                                } else if constexpr (k < 72) {   // slice words = features 16s + 8kg + {0..7}; 63 is the 1.0 pad
                                    constexpr int c = (k - 64) >> 2, sl = (k - 64) & 3;
                                    if (save_ok[c]) *reinterpret_cast<u32x4*>(save_row[c] + 128 + 8 * sl + 4 * kg) = ex[c][sl];
                                }
                            });
                        }
                    };
                    run_chunk(IC(34), IC(4), IC(4), IC(0), IC(34), IC(0), sacc, from_H, [&](auto gc_) { ride_sem(gc_, IC(0)); });
                    run_chunk(IC(34), IC(4), IC(4), IC(34), IC(34), IC(0), sacc, from_H, [&](auto gc_) { ride_sem(gc_, IC(1)); });
                    if constexpr (SEM == 2) run_chunk(IC(16), IC(4), IC(0), IC(0), IC(16), IC(0), sacc, from_ex, no_ride);
                    if constexpr (SAVE) {
                        asm volatile("s_nop 15\n\ts_nop 3" ::: "memory");  // MFMA result -> VALU read wait states
                        auto relu_acc = [](float a) {   // AGPR read inside asm: see pack8_acc
                            float x;
                            asm volatile("v_accvgpr_read_b32 %0, %1\n\tv_max_f32 %0, 0, %0" : "=v"(x) : "a"(a));
                            return x;
                        };
#pragma unroll
                        for (int c = 0; c < 2; ++c) {
                            const long long gp = (long long)tile * kTilePts + wave * 64 + c * 32 + pj;
                            if (gp < P.n_pts) {
                                float* row = P.sem_in + gp * 320;
                                float* hrow = P.sem_hid + gp * 128;
                                if (!P.sem_in16) {
#pragma unroll
                                for (int t = 0; t < 8; ++t)
#pragma unroll
                                    for (int u = 0; u < 2; ++u)
#pragma unroll
                                        for (int q = 0; q < 4; q += 2) {   // words q, q+1 = accumulator elements 8u+2q .. +3 of tile t: 4 consecutive features
                                            const unsigned w0 = H[c][2 * t + u][q], w1 = H[c][2 * t + u][q + 1];
                                            *reinterpret_cast<f32x4*>(row + 32 * t + 8 * (2 * u + (q >> 1)) + 4 * kg) =
                                                f32x4{T::lo(w0), T::hi(w0), T::lo(w1), T::hi(w1)};
                                        }
#pragma unroll
                                for (int sl = 0; sl < 4; ++sl)
#pragma unroll
                                    for (int q = 0; q < 4; q += 2) {    // slice words q, q+1 = features 16s + 8kg + 2q .. +3; 63 is the 1.0 pad
                                        const unsigned w0 = ex[c][sl][q], w1 = ex[c][sl][q + 1];
                                        *reinterpret_cast<f32x4*>(row + 256 + 16 * sl + 8 * kg + 2 * q) = f32x4{T::lo(w0), T::hi(w0), T::lo(w1), T::hi(w1)};
                                    }
                                }
                                if (P.sem_in16) {   // compact: the hidden activations in the 16-bit format too (features 32t + 8q + 4kg + {0..3} = 2 words)
                                    unsigned* hrow16 = P.sem_hid16 + gp * 64;
#pragma unroll
                                    for (int t = 0; t < 4; ++t)
#pragma unroll
                                        for (int q = 0; q < 4; ++q)
                                            *reinterpret_cast<u32x2*>(hrow16 + 16 * t + 4 * q + 2 * kg) =
                                                u32x2{T::pack2(relu_acc(sacc[c][t][4 * q]), relu_acc(sacc[c][t][4 * q + 1])),
                                                      T::pack2(relu_acc(sacc[c][t][4 * q + 2]), relu_acc(sacc[c][t][4 * q + 3]))};
                                } else {
#pragma unroll
                                for (int t = 0; t < 4; ++t)
#pragma unroll
                                    for (int q = 0; q < 4; ++q)
                                        *reinterpret_cast<f32x4*>(hrow + 32 * t + 8 * q + 4 * kg) =
                                            f32x4{relu_acc(sacc[c][t][4 * q]), relu_acc(sacc[c][t][4 * q + 1]),
                                                  relu_acc(sacc[c][t][4 * q + 2]), relu_acc(sacc[c][t][4 * q + 3])};
                                }
                            }
                        }
                    }
#pragma unroll
                    for (int c = 0; c < 2; ++c) {
                        float ps[2];
#pragma unroll
                        for (int o = 0; o < 2; ++o) ps[o] = kg ? 0.0f : __builtin_bit_cast(float, aux_l[kAuxScalars + 4 + o]);
                        heads_partial_f32<4, 2>(sacc[c], reinterpret_cast<const float*>(aux_l) + kAuxSem2W + kg * 64, ps);
#pragma unroll
                        for (int o = 0; o < 2; ++o) sem_out[c][o] = both_halves(ps[o]);
                    }
                }
                stamp();  // 18 (l == 7 only; the later slots shift by one): sigma + semantic heads
            }
        }
        // view branch: cat([feature, dir27]) -> 128 -> rgb   (H = feature, no activation)
        f32x16 vacc[2][4];
        run_chunk(IC(34), IC(4), IC(4), IC(0), IC(34), IC(0), vacc, from_H, no_ride);
        run_chunk(IC(34), IC(4), IC(4), IC(34), IC(34), IC(0), vacc, from_H, no_ride);
        stamp();  // 21: view-branch MFMAs on the feature
        // the direction encoding is evaluated only now (12 sincos per column): keeping its 16 VGPRs alive through
        // the trunk pushes the kernel into spilling, and a spilled "pending" ring register is a race
        u32x4 ed[2][2];
#pragma unroll
        for (int c = 0; c < 2; ++c) {
            float dv[3];
#pragma unroll
            for (int k = 0; k < 3; ++k) dv[k] = P.viewdirs[3ll * ray_of[c] + k];
            Enc<NSOS_DIR_FREQS, SliceHalf> e;
            e.evaluate_hw(dv, kg);
            ed[c][0] = enc_slice<T, NSOS_DIR_FREQS, 0, false>(e, dv, kg);
            ed[c][1] = enc_slice<T, NSOS_DIR_FREQS, 1, false>(e, dv, kg);
        }
        auto from_ed = [&](auto cc, auto sc) { return ed[decltype(cc)::value][decltype(sc)::value]; };
        stamp();  // 22: direction encoding
        run_chunk(IC(16), IC(4), IC(0), IC(0), IC(8), IC(0), vacc, from_ed, no_ride);  // 2 slices x 4 tiles; groups 8..15 are padding
        stamp();  // 23: direction MFMAs
#pragma unroll
        for (int c = 0; c < 2; ++c) {
            float rgb[3];
#pragma unroll
            for (int o = 0; o < 3; ++o) rgb[o] = kg ? 0.0f : __builtin_bit_cast(float, aux_l[kAuxScalars + 1 + o]);
            heads_partial_f32<4, 3>(vacc[c], reinterpret_cast<const float*>(aux_l) + kAuxRgbW + kg * 64, rgb);
#pragma unroll
            for (int o = 0; o < 3; ++o) rgb[o] = both_halves(rgb[o]);
            const long long gp = (long long)tile * kTilePts + wave * 64 + c * 32 + pj;
            {   // NaN / Inf in the point's inputs must come out as NaN (the reference propagates them; the packed integer
                // ReLU would launder them).  The inputs are re-read here (L2 hits) rather than kept alive across the tile.
                const long long gc = gp < P.n_pts ? gp : P.n_pts - 1, ray = ray_of[c];
                const float z = P.z_vals[gc];
                float chk = z - z;
#pragma unroll
                for (int k = 0; k < 3; ++k) {
                    const float o = P.rays_o[3 * ray + k], d = P.rays_d[3 * ray + k], v = P.viewdirs[3 * ray + k];
                    chk += ((o - o) + (d - d)) + (v - v);
                }
                if (chk != chk) {
                    const float qnan = __builtin_nanf("");
                    rgb[0] = rgb[1] = rgb[2] = sigma[c] = sem_out[c][0] = sem_out[c][1] = qnan;
                }
            }
            if (gp < P.n_pts) {
                float* out = P.raw + gp * C;
                if constexpr (C == 4) {
                    if (kg == 0) *reinterpret_cast<f32x4*>(out) = f32x4{rgb[0], rgb[1], rgb[2], sigma[c]};
                } else {
                    if (kg == 0) {
                        *reinterpret_cast<f32x2*>(out) = f32x2{rgb[0], rgb[1]};
                        *reinterpret_cast<f32x2*>(out + 2) = f32x2{rgb[2], sigma[c]};
                    } else {
                        *reinterpret_cast<f32x2*>(out + 4) = f32x2{sem_out[c][0], sem_out[c][1]};
                    }
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
struct LpChunk {
    const float* w;
    const float* bias;  // leading bias A operands of the part (a < NB), or the pad-slot bias (kEnc8 of layer 0), or NULL
    int in_dim, col_base, kind, a0, n_groups;
};
struct LpPackParams {
    LpChunk ch[40];
    int n_chunks;
    int first, count;     // the chunks [first, first + count) are written (all of them, or the semantic head's: nsos_mlp_pack_lp_heads)
    const float* alpha_w; const float* alpha_b;
    const float* rgb_w; const float* rgb_b;
    const float* sem2_w; const float* sem2_b;
    unsigned* aux;
    unsigned short* chunks;
};

template <class T>
__global__ __launch_bounds__(256) void lp_pack_kernel(const LpPackParams P) {
    const long long gid = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (gid < kAuxWords) {
        const int a = (int)gid;
        unsigned v = 0;
        auto feat128 = [](int rem) { return acc_feature((rem & 63) >> 4, rem & 15, rem >> 6); };
        if (a < kAuxRgbW) {  // sigma head weights, packed pairs in H order: word = [kg][s][q] -> features of slice s
            const int kgl = a >> 6, s = (a & 63) >> 2, q = a & 3;
            const int f0 = acc_feature(s >> 1, 8 * (s & 1) + 2 * q, kgl), f1 = acc_feature(s >> 1, 8 * (s & 1) + 2 * q + 1, kgl);
            v = (unsigned)T::bits(P.alpha_w[f0]) | ((unsigned)T::bits(P.alpha_w[f1]) << 16);
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
    if (gid >= (long long)P.count * per_chunk) return;
    const LpChunk ck = P.ch[P.first + gid / per_chunk];
    const int within = (int)(gid % per_chunk);
    const int g = within >> 9, lane = (within >> 3) & 63, e = within & 7;  // 512 elements per A operand
    const int i = lane & 31, kgl = lane >> 5, m = 8 * kgl + e;
    float v = 0.0f;
    if (g < ck.n_groups && ck.kind == kPair8) {   // a0 = tile pair c: operands [bias 2c, bias 2c+1, then slice-major over the two tiles]
        const int t = 2 * ck.a0 + (g < 2 ? g : ((g - 2) & 1));
        if (g < 2) v = (m == 0) ? ck.bias[32 * t + i] : 0.0f;
        else {
            const int s = (g - 2) >> 1;
            v = ck.w[(long long)(32 * t + i) * ck.in_dim + ck.col_base + acc_feature(s >> 1, 8 * (s & 1) + e, kgl)];
        }
    } else if (g < ck.n_groups) {
        const int a = ck.a0 + g;
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
    P.chunks[(long long)P.first * per_chunk + gid] = T::bits(v);
}


constexpr int kLdsBytes = kSlots * kSlotBytes + kAuxWords * 4;

template <class T, int SEM, bool SAVE = false>
int32_t launch_lp(const LpParams& p, hipStream_t stream) {
    static NsosPerDeviceFlag configured_on;
    bool& configured = configured_on.here();
    if (!configured) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&mlp_lp_kernel<T, SEM, SAVE>),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, kLdsBytes);
        if (e != hipSuccess) return (int32_t)e;
        configured = true;
    }
    const int cus = nsos_device_cus();
    const int grid = p.n_tiles < cus ? p.n_tiles : cus;
    hipLaunchKernelGGL((mlp_lp_kernel<T, SEM, SAVE>), dim3(grid), dim3(256), kLdsBytes, stream, p);
    return nsos_launch_status();
}

}  // namespace

// ------------------------------------------------------------------------------------------ C ABI
extern "C" size_t nsos_mlp_packed_bytes_lp(int32_t sem_mode) {
    if (sem_mode < 0 || sem_mode > 2) return 0;
    // aux + the slice-major stream of mlp_lp_kernel + the stream of mlp_lp8_kernel (tile-pair-major hidden layers) + the stream
    // of mlp_lp16_kernel (16x16x32 tiles, tile-quad-major hidden layers)
    return (size_t)kAuxWords * 4 + 2 * (size_t)lp_chunks(sem_mode) * kSlotBytes + lp16_stream_bytes(sem_mode);
}

static int32_t pack_lp_impl(const nsos_mlp_tensors* T_, int32_t sem_mode, int32_t dtype, void* packed, size_t packed_bytes, void* stream,
                            bool heads_only);
static int lp_waves_per_simd();
extern "C" int32_t nsos_mlp_pack_lp(const nsos_mlp_tensors* T_, int32_t sem_mode, int32_t dtype, void* packed,
                                    size_t packed_bytes, void* stream) {
    return pack_lp_impl(T_, sem_mode, dtype, packed, packed_bytes, stream, false);
}
// Only what depends on semantic_linear.*: the head's chunks of all three streams (chunks 30.. of each: the trunk's 30 chunks come
// first in every layout) and the aux block.  For the shipped training recipe (--fix_backbone: only the semantic heads train,
// run_nerf.py:307-318) a step re-packs 3 chunks instead of 37-40 -- and only in the stream of the kernel that is selected NOW
// (nsos_mlp_lp_selected_kernel: one launch per net and step instead of three); `packed` must hold a full pack of the same trunk.
// CONTRACT: after a heads-only re-pack only the selected kernel's stream is current.  A launch that takes another stream -- after
// nsos_mlp_lp_select_kernel, or forward_rays_lp's fall-back to the round-1 kernel on stream 0 (fp32 sem_in saves:
// nsos_mlp_forward_rays_save_lp; launches of >= 2^31 points) -- needs a FULL pack first.  The Python layer enforces it: ops.PackPlan
// tags the buffer and ops.mlp_forward_rays_lp / mlp_forward_rays_save raise instead of rendering stale heads (re-packing
// stream 0 on every step as well was tried first: +10 us per C3 step for a path no training step takes).
extern "C" int32_t nsos_mlp_pack_lp_heads(const nsos_mlp_tensors* T_, int32_t sem_mode, int32_t dtype, void* packed,
                                          size_t packed_bytes, void* stream) {
    NSOS_REQUIRE(sem_mode == NSOS_SEM_PLAIN || sem_mode == NSOS_SEM_COORD, NSOS_ERR_UNSUPPORTED);
    return pack_lp_impl(T_, sem_mode, dtype, packed, packed_bytes, stream, true);
}
static int32_t pack_lp_impl(const nsos_mlp_tensors* T_, int32_t sem_mode, int32_t dtype, void* packed, size_t packed_bytes, void* stream,
                            bool heads_only) {
    NSOS_REQUIRE(T_ && packed, NSOS_ERR_NULL_POINTER);
    NSOS_REQUIRE(sem_mode >= 0 && sem_mode <= 2 && (dtype == NSOS_DTYPE_F16 || dtype == NSOS_DTYPE_BF16), NSOS_ERR_UNSUPPORTED);
    NSOS_REQUIRE(packed_bytes >= nsos_mlp_packed_bytes_lp(sem_mode), NSOS_ERR_BUFFER_TOO_SMALL);
    NSOS_REQUIRE(((uintptr_t)packed & 15) == 0, NSOS_ERR_MISALIGNED);
    for (int l = 0; l < NSOS_NET_DEPTH; ++l) NSOS_REQUIRE(T_->pts_w[l] && T_->pts_b[l], NSOS_ERR_NULL_POINTER);
    NSOS_REQUIRE(T_->alpha_w && T_->alpha_b && T_->feature_w && T_->feature_b && T_->views_w && T_->views_b &&
                     T_->rgb_w && T_->rgb_b, NSOS_ERR_NULL_POINTER);
    if (sem_mode) NSOS_REQUIRE(T_->sem0_w && T_->sem0_b && T_->sem2_w && T_->sem2_b, NSOS_ERR_NULL_POINTER);

    const int X = NSOS_XYZ_DIM, W = NSOS_NET_WIDTH;
    const int selected = lp_waves_per_simd();
    for (int layout = 0; layout < 2; ++layout) {   // 0: slice-major hidden layers (mlp_lp_kernel), 1: tile-pair-major (mlp_lp8_kernel)
        if (heads_only && selected != layout + 1) continue;
        LpPackParams P = {};
        int n = 0;
        auto add = [&](const float* w, const float* bias, int in_dim, int col, int kind, int a0, int ng) {
            P.ch[n++] = LpChunk{w, bias, in_dim, col, kind, a0, ng};
        };
        auto hidden8 = [&](const float* w, const float* b, int in_dim, int col, bool pairs) {
            for (int c = 0; c < 4; ++c) pairs ? add(w, b, in_dim, col, kPair8, c, 34) : add(w, b, in_dim, col, kHid8, 34 * c, 34);
        };
        auto hidden4 = [&](const float* w, const float* b, int in_dim, int col) {
            for (int c = 0; c < 2; ++c) add(w, b, in_dim, col, kHid4, 34 * c, 34);
        };
        const bool pm = layout == 1;
        add(T_->pts_w[0], T_->pts_b[0], X, 0, kEnc8, 0, 32);  // bias in the pad slot
        for (int l = 1; l <= 4; ++l) hidden8(T_->pts_w[l], T_->pts_b[l], W, 0, pm);
        hidden8(T_->pts_w[5], T_->pts_b[5], X + W, X, false);  // skip layer: h part (with its bias slice), slice-major in both ...
        add(T_->pts_w[5], nullptr, X + W, 0, kEnc8, 0, 32);    // ... then the x63 part
        hidden8(T_->pts_w[6], T_->pts_b[6], W, 0, pm);
        hidden8(T_->pts_w[7], T_->pts_b[7], W, 0, pm);
        if (sem_mode) {
            const int in_dim = sem_mode == NSOS_SEM_COORD ? W + X : W;
            hidden4(T_->sem0_w, T_->sem0_b, in_dim, 0);
            if (sem_mode == NSOS_SEM_COORD) add(T_->sem0_w, nullptr, in_dim, W, kEnc4, 0, 16);
        }
        hidden8(T_->feature_w, T_->feature_b, W, 0, pm);
        hidden4(T_->views_w, T_->views_b, W + NSOS_DIR_DIM, 0);
        add(T_->views_w, nullptr, W + NSOS_DIR_DIM, W, kDir4, 0, 8);
        NSOS_REQUIRE(n == lp_chunks(sem_mode), NSOS_ERR_UNSUPPORTED);
        P.n_chunks = n;
        P.first = heads_only ? 30 : 0;                       // L0 (1) + L1-4 (16) + L5 (5) + L6-7 (8) chunks precede the head's
        P.count = heads_only ? (sem_mode == NSOS_SEM_COORD ? 3 : 2) : n;
        P.alpha_w = T_->alpha_w; P.alpha_b = T_->alpha_b;
        P.rgb_w = T_->rgb_w; P.rgb_b = T_->rgb_b;
        P.sem2_w = sem_mode ? T_->sem2_w : nullptr;
        P.sem2_b = sem_mode ? T_->sem2_b : nullptr;
        P.aux = static_cast<unsigned*>(packed);
        P.chunks = reinterpret_cast<unsigned short*>(P.aux + kAuxWords) + (size_t)layout * n * (kSlotBytes / 2);
        const long long total = (long long)P.count * (kSlotBytes / 2);
        const dim3 grid((unsigned)((total + 255) / 256)), block(256);
        if (dtype == NSOS_DTYPE_F16) hipLaunchKernelGGL(lp_pack_kernel<F16>, grid, block, 0, (hipStream_t)stream, P);
        else hipLaunchKernelGGL(lp_pack_kernel<BF16>, grid, block, 0, (hipStream_t)stream, P);
    }
    const int32_t rc = nsos_launch_status();
    if (rc != NSOS_OK) return rc;
    unsigned char* stream16 = reinterpret_cast<unsigned char*>(static_cast<unsigned*>(packed) + kAuxWords) + 2 * (size_t)lp_chunks(sem_mode) * kSlotBytes;
    if (heads_only && selected != 3) return NSOS_OK;
    return pack_lp16(T_, sem_mode, dtype == NSOS_DTYPE_F16, stream16, (hipStream_t)stream, heads_only);
}

// which kernel serves the 16-bit path: 3 = mlp_lp16_kernel (round 4: two waves per SIMD on v_mfma_f32_16x16x32; default),
// 2 = mlp_lp8_kernel (rounds 2-3: two 256-register waves per SIMD on 32x32x16), 1 = mlp_lp_kernel (round 1: one 512-register
// wave per SIMD, 64 points).  NSOS_LP_KERNEL=lp16|lp8|lp4 (or the older NSOS_LP_WAVES=4) in the environment or
// nsos_mlp_lp_select_kernel(3|2|1) select one for A/B measurements; lp8 and lp4 are bit-identical to each other, lp16 agrees
// with them to the 16-bit formats' rounding (other contraction order, 16-bit heads: mlp_lp16.hip).
static int g_lp_waves_per_simd = 0;
static int lp_waves_per_simd() {
    if (g_lp_waves_per_simd == 0) {
        const char* e = getenv("NSOS_LP_WAVES");
        const char* k = getenv("NSOS_LP_KERNEL");
        g_lp_waves_per_simd = 3;
        if (e && e[0] == '4') g_lp_waves_per_simd = 1;
        if (k && k[0] == 'l' && k[1] == 'p') g_lp_waves_per_simd = k[2] == '4' ? 1 : (k[2] == '8' ? 2 : 3);
    }
    return g_lp_waves_per_simd;
}

// diagnostics: a stamp buffer for EVERY following 16-bit launch (inference and training variants alike), or NULL to stop;
// layout as nsos_mlp_profile_rays_lp
static unsigned long long* g_lp_stamps = nullptr;
extern "C" int32_t nsos_mlp_lp_set_stamp_buffer(uint64_t* stamps) {
    g_lp_stamps = reinterpret_cast<unsigned long long*>(stamps);
    return NSOS_OK;
}

extern "C" int32_t nsos_mlp_lp_selected_kernel(void) { return lp_waves_per_simd(); }

extern "C" int32_t nsos_mlp_lp_select_kernel(int32_t waves_per_simd) {
    NSOS_REQUIRE(waves_per_simd >= 1 && waves_per_simd <= 3, NSOS_ERR_UNSUPPORTED);
    g_lp_waves_per_simd = waves_per_simd;
    return NSOS_OK;
}

static int32_t forward_rays_lp(const void* packed, int32_t sem_mode, int32_t dtype, const float* rays_o,
                               const float* rays_d, const float* viewdirs, const float* z_vals, int64_t n_rays,
                               int32_t n_samples, float* raw, unsigned long long* prof, float* sem_in, float* sem_hid,
                               void* stream, unsigned* sem_in16 = nullptr, unsigned* sem_hid16 = nullptr) {
    if (n_rays == 0) return NSOS_OK;
    NSOS_REQUIRE(packed && rays_o && rays_d && viewdirs && z_vals && raw, NSOS_ERR_NULL_POINTER);
    NSOS_REQUIRE(n_rays > 0 && n_samples >= 1, NSOS_ERR_BAD_SHAPE);
    NSOS_REQUIRE(n_rays < (1ll << 31), NSOS_ERR_UNSUPPORTED);
    NSOS_REQUIRE(sem_mode >= 0 && sem_mode <= 2 && (dtype == NSOS_DTYPE_F16 || dtype == NSOS_DTYPE_BF16), NSOS_ERR_UNSUPPORTED);
    NSOS_REQUIRE(((uintptr_t)packed & 15) == 0 && ((uintptr_t)raw & 15) == 0, NSOS_ERR_MISALIGNED);
    const long long n_pts = (long long)n_rays * n_samples;
    NSOS_REQUIRE((n_pts + kTilePts - 1) / kTilePts < (1ll << 31), NSOS_ERR_UNSUPPORTED);
    LpParams p = {};
    p.aux = static_cast<const unsigned*>(packed);
    p.chunks = reinterpret_cast<const unsigned char*>(p.aux + kAuxWords);
    p.rays_o = rays_o; p.rays_d = rays_d; p.viewdirs = viewdirs; p.z_vals = z_vals;
    p.raw = raw; p.n_pts = n_pts; p.n_samples = n_samples;
    p.n_tiles = (int)((n_pts + kTilePts - 1) / kTilePts);
    p.prof = prof ? prof : g_lp_stamps;
    p.sem_in = sem_in;
    p.sem_in16 = sem_in16;
    p.sem_hid = sem_hid;
    p.sem_hid16 = sem_hid16;
    const hipStream_t st = (hipStream_t)stream;
    // NSOS_LP_WAVES=4 selects the one-wave-per-SIMD kernel of round 1 (A/B measurements); default: two waves per SIMD
    // (mlp_lp8_kernel indexes its points with 32 bits: launches of 2^31 points or more -- 11 M rays x 192 samples -- take the
    //  round-1 kernel, whose results are bit-identical)
    // (its training variant stores the compact 16-bit operands only: the fp32 sem_in / sem_hid of nsos_mlp_forward_rays_save_lp
    //  -- tests and the exact-kernel backward -- come from the round-1 kernel as well)
    if (lp_waves_per_simd() >= 2 && n_pts < (1ll << 31) && !(sem_in && !sem_in16)) {
        if (sem_in || sem_in16) NSOS_REQUIRE((sem_in16 ? (void*)sem_hid16 : (void*)sem_hid) && sem_mode != NSOS_SEM_NONE, NSOS_ERR_UNSUPPORTED);
        if (lp_waves_per_simd() == 3) {
            p.chunks += 2 * (size_t)lp_chunks(sem_mode) * kSlotBytes;   // the third stream: 16x16x32 tiles
            return launch_lp16(p, sem_mode, dtype == NSOS_DTYPE_F16, sem_in || sem_in16, st);
        }
        p.chunks += (size_t)lp_chunks(sem_mode) * kSlotBytes;   // the second stream: tile-pair-major hidden layers
        return launch_lp8(p, sem_mode, dtype == NSOS_DTYPE_F16, sem_in || sem_in16, st);
    }
    if (sem_in || sem_in16) {
        NSOS_REQUIRE((sem_in16 ? (void*)sem_hid16 : (void*)sem_hid) && sem_mode != NSOS_SEM_NONE, NSOS_ERR_UNSUPPORTED);
        if (dtype == NSOS_DTYPE_F16) return sem_mode == 1 ? launch_lp<F16, 1, true>(p, st) : launch_lp<F16, 2, true>(p, st);
        return sem_mode == 1 ? launch_lp<BF16, 1, true>(p, st) : launch_lp<BF16, 2, true>(p, st);
    }
    if (dtype == NSOS_DTYPE_F16) {
        switch (sem_mode) {
            case 0: return launch_lp<F16, 0>(p, st);
            case 1: return launch_lp<F16, 1>(p, st);
            default: return launch_lp<F16, 2>(p, st);
        }
    }
    switch (sem_mode) {
        case 0: return launch_lp<BF16, 0>(p, st);
        case 1: return launch_lp<BF16, 1>(p, st);
        default: return launch_lp<BF16, 2>(p, st);
    }
}

extern "C" int32_t nsos_mlp_forward_rays_lp(const void* packed, int32_t sem_mode, int32_t dtype, const float* rays_o,
                                            const float* rays_d, const float* viewdirs, const float* z_vals,
                                            int64_t n_rays, int32_t n_samples, float* raw, void* stream) {
    return forward_rays_lp(packed, sem_mode, dtype, rays_o, rays_d, viewdirs, z_vals, n_rays, n_samples, raw, nullptr, nullptr,
                           nullptr, stream);
}

extern "C" int32_t nsos_mlp_forward_rays_save_lp(const void* packed, int32_t sem_mode, int32_t dtype, const float* rays_o,
                                                 const float* rays_d, const float* viewdirs, const float* z_vals,
                                                 int64_t n_rays, int32_t n_samples, float* raw, float* sem_in,
                                                 float* sem_hid, void* stream) {
    if (n_rays == 0) return NSOS_OK;
    NSOS_REQUIRE(sem_in && sem_hid, NSOS_ERR_NULL_POINTER);
    NSOS_REQUIRE(((uintptr_t)sem_in & 15) == 0 && ((uintptr_t)sem_hid & 15) == 0, NSOS_ERR_MISALIGNED);
    return forward_rays_lp(packed, sem_mode, dtype, rays_o, rays_d, viewdirs, z_vals, n_rays, n_samples, raw, nullptr, sem_in,
                           sem_hid, stream);
}

extern "C" int32_t nsos_mlp_save16_layout(int64_t n_points) {   // the same condition forward_rays_lp selects the kernel by
    if (!(lp_waves_per_simd() >= 2 && n_points < (1ll << 31))) return NSOS_SEM_IN_ROWS;
    return lp_waves_per_simd() == 3 ? (NSOS_SEM_IN_TILED | NSOS_SEM_HID_TILED) : NSOS_SEM_IN_TILED;
}

extern "C" int32_t nsos_mlp_forward_rays_save16_lp(const void* packed, int32_t sem_mode, int32_t dtype, const float* rays_o,
                                                   const float* rays_d, const float* viewdirs, const float* z_vals,
                                                   int64_t n_rays, int32_t n_samples, float* raw, void* sem_in16,
                                                   void* sem_hid16, void* stream) {
    if (n_rays == 0) return NSOS_OK;
    NSOS_REQUIRE(sem_in16 && sem_hid16, NSOS_ERR_NULL_POINTER);
    NSOS_REQUIRE(((uintptr_t)sem_in16 & 15) == 0 && ((uintptr_t)sem_hid16 & 15) == 0, NSOS_ERR_MISALIGNED);
    return forward_rays_lp(packed, sem_mode, dtype, rays_o, rays_d, viewdirs, z_vals, n_rays, n_samples, raw, nullptr, nullptr,
                           nullptr, stream, static_cast<unsigned*>(sem_in16), static_cast<unsigned*>(sem_hid16));
}

extern "C" int32_t nsos_mlp_profile_rays_lp(const void* packed, int32_t sem_mode, int32_t dtype, const float* rays_o,
                                            const float* rays_d, const float* viewdirs, const float* z_vals,
                                            int64_t n_rays, int32_t n_samples, float* raw, uint64_t* stamps,
                                            void* stream) {
    if (n_rays == 0) return NSOS_OK;
    NSOS_REQUIRE(stamps, NSOS_ERR_NULL_POINTER);
    return forward_rays_lp(packed, sem_mode, dtype, rays_o, rays_d, viewdirs, z_vals, n_rays, n_samples, raw,
                           reinterpret_cast<unsigned long long*>(stamps), nullptr, nullptr, stream);
}
