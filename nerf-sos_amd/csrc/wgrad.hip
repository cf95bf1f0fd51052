// K7 (full backward): weight-gradient reductions and the ReLU mask for gfx950.
//
//   dW[m][n] = sum_p G[p][m] * X[p][n]      db[m] = sum_p G[p][m]           (autograd of y = x W^T + b over P points)
// G [P, ldg] = gradient w.r.t. a layer's pre-activation output, X [P, ldx] = that layer's input; both row-major
// slices of larger buffers (the saved activations of nsos_mlp_forward_rays_save_all, gradient ping-pong buffers).
// The shape is a reduction over millions of points into a tiny output -- M, N <= 256 and K = P -- for which the BLAS
// library's heuristics pick 16x16 macro-tiles (measured ~3 % of the fp32 MFMA rate).  Here: persistent
// grid; a workgroup owns a contiguous block of points; v_mfma_f32_32x32x2_f32 (exact fp32) with K = points, two per
// MFMA; the 4 waves split the 32-row tiles of the output (RW = min(M/32, 4) row groups) and, when M/32 < 4, also the
// points (KW = 4/RW interleaved subsets).  A operand = G[pt][32 mt + i], B operand = X[pt][32 nt + j]: both are
// coalesced 128 B rows.  Per-(workgroup, k-subset) partials are summed in a fixed order in fp64: deterministic.
// MFMA-bound for M = N = 256 (65 536 MAC per point per layer).
#include "common.h"
#include "x3_common.h"   // split2 / mfma16 of the split-fp16 kernels (wgrad_x3_kernel)

typedef float f32x16 __attribute__((ext_vector_type(16)));

#ifndef NSOS_WG_U
#define NSOS_WG_U 4   // k-steps per operand group of the 16-tile variant (prefetch distance)
#endif

namespace {
constexpr int kWgradMaxBlocks = 256;

// RT = row tiles per wave (1 or 2), NT = column tiles (1..8)
// XT: element type of X -- float, or _Float16 for activations saved in 16 bits (nsos_mlp_forward_rays_save_all16_x3: exact in fp32)
template <int RT, int NT, class XT>
__global__ __launch_bounds__(256, 1) void wgrad_kernel(const float* __restrict__ G, int ldg, const XT* __restrict__ X, int ldx,
                                                       long long n_pts, int M, int N, float* __restrict__ partial) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, i = lane & 31, k = lane >> 5;
    const int MT = M >> 5;
    const int RW = MT < 4 ? MT : 4, KW = 4 / RW;            // MT in {1,2,4,8}
    const int rw = wave % RW, kw = wave / RW;
    long long chunk = (n_pts + gridDim.x - 1) / gridDim.x;
    chunk = (chunk + 2 * KW - 1) / (2 * KW) * (2 * KW);
    const long long start = (long long)blockIdx.x * chunk;
    const long long end = start + chunk < n_pts ? start + chunk : n_pts;
    f32x16 acc[RT][NT];
#pragma unroll
    for (int r = 0; r < RT; ++r)
#pragma unroll
        for (int t = 0; t < NT; ++t)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[r][t][e] = 0.0f;
    float bsum[RT];
#pragma unroll
    for (int r = 0; r < RT; ++r) bsum[r] = 0.0f;

    // Operand fetch runs a group of U k-steps (2U points) ahead of the MFMAs: one k-step is only RT*NT*64 cycles of
    // MFMA work, far less than an HBM round trip.  Two operand sets are used alternately (no register copies: hipcc
    // hoists a `cur = nxt` copy -- and the wait for the loads behind it -- up into the MFMA stream), and the hot loop
    // only touches groups whose points are all in range, so its loads are unconditional (a `valid ? load : 0` becomes a
    // branch with a vmcnt wait right behind every load and serialises the whole pipeline on the memory latency).
    constexpr int U = RT * NT >= 16 ? NSOS_WG_U : (RT * NT >= 8 ? 6 : 8);
    struct In { float a[RT]; float b[NT]; };
    const long long step = 2 * KW;
    const int kw_s = __builtin_amdgcn_readfirstlane(kw), rw_s = __builtin_amdgcn_readfirstlane(rw);
    const long long base = start + 2 * kw_s;                       // wave-uniform
    // addressing: wave-uniform base pointers (SGPRs, advanced per group on the scalar unit) + per-lane 32-bit offsets
    // that never change, so the hot loop carries no address arithmetic on the vector unit (a VALU instruction between
    // two fp32 MFMAs costs ~12 cycles)
    int goff[U], xoff[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
        goff[u] = (int)((u * step + k) * ldg) + 32 * rw_s + i;
        xoff[u] = (int)((u * step + k) * ldx) + i;
    }
    auto fetch_grp = [&](const float* gb, const XT* xb, In (&buf)[U]) {   // all points of the group are in range
#pragma unroll
        for (int u = 0; u < U; ++u) {
#pragma unroll
            for (int r = 0; r < RT; ++r) buf[u].a[r] = gb[goff[u] + 32 * RW * r];   // row tile rw + RW*r
#pragma unroll
            for (int t = 0; t < NT; ++t) buf[u].b[t] = (float)xb[xoff[u] + 32 * t];
        }
    };
    auto compute1 = [&](const In& in) {
#pragma unroll
        for (int r = 0; r < RT; ++r) {
            bsum[r] += in.a[r];
#pragma unroll
            for (int t = 0; t < NT; ++t) acc[r][t] = __builtin_amdgcn_mfma_f32_32x32x2f32(in.a[r], in.b[t], acc[r][t], 0, 0, 0);
        }
    };
    auto compute = [&](const In (&buf)[U]) {
#pragma unroll
        for (int u = 0; u < U; ++u) compute1(buf[u]);
    };
    // k-steps of this wave: j = 0, 1, ... at points base + j*step + {0,1}; n_full of them lie completely below `end`
    const long long n_ks = base < end ? (end - base + step - 1) / step : 0;              // k-steps with at least one point
    const long long n_full = base + 1 < end ? (end - 2 - base) / step + 1 : 0;           // k-steps with both points
    const long long n_grp = n_full / U;                                                   // complete groups of U k-steps
    if (n_grp > 0) {
        In bufA[U], bufB[U];
        const float* const g0 = G + base * ldg;
        const XT* const x0 = X + base * ldx;
        const long long gstride = (long long)U * step * ldg, xstride = (long long)U * step * ldx;
        auto grp_ptr = [&](auto* p0, long long stride, long long g) { return p0 + (g < n_grp ? g : n_grp - 1) * stride; };
        fetch_grp(g0, x0, bufA);
        long long g = 0;
        for (; g + 1 < n_grp; g += 2) {
            // loads of the next group, THEN the MFMAs of the current one: the barriers keep hipcc from sinking loads
            // into the MFMA stream (its in-order vmcnt waits would then stall on the freshest load)
            fetch_grp(grp_ptr(g0, gstride, g + 1), grp_ptr(x0, xstride, g + 1), bufB);
            __builtin_amdgcn_sched_barrier(0);
            compute(bufA);
            __builtin_amdgcn_sched_barrier(0);
            fetch_grp(grp_ptr(g0, gstride, g + 2), grp_ptr(x0, xstride, g + 2), bufA);   // past the end: re-fetch the last group
            __builtin_amdgcn_sched_barrier(0);
            compute(bufB);
            __builtin_amdgcn_sched_barrier(0);
        }
        if (g < n_grp) compute(bufA);
    }
    for (long long j = n_grp * U; j < n_ks; ++j) {   // ragged tail: fewer than U k-steps, the last possibly half valid
        const long long pt = base + j * step + k;
        const bool valid = pt < end;
        const long long pc = valid ? pt : (n_pts - 1);
        In in;
        const float* grow = G + pc * ldg + 32 * rw + i;
        const XT* xrow = X + pc * ldx + i;
#pragma unroll
        for (int r = 0; r < RT; ++r) in.a[r] = valid ? grow[32 * RW * r] : 0.0f;
#pragma unroll
        for (int t = 0; t < NT; ++t) in.b[t] = (float)xrow[32 * t];
        compute1(in);
    }
    // partial [(block * KW + kw)][M*N + M]
    float* out = partial + ((size_t)blockIdx.x * KW + kw) * ((size_t)M * N + M);
#pragma unroll
    for (int r = 0; r < RT; ++r) {
        const int mt = rw + RW * r;
#pragma unroll
        for (int t = 0; t < NT; ++t)
#pragma unroll
            for (int e = 0; e < 16; ++e)   // accumulator element e of lane (i, k): row (e&3) + 8(e>>2) + 4k, column i
                out[(size_t)(32 * mt + (e & 3) + 8 * (e >> 2) + 4 * k) * N + 32 * t + i] = acc[r][t][e];
        const float s = bsum[r] + __shfl_xor(bsum[r], 32, NSOS_WAVE);
        if (k == 0) out[(size_t)M * N + 32 * mt + i] = s;
    }
}

__global__ __launch_bounds__(256) void wgrad_reduce_kernel(const float* __restrict__ partial, int n_part, int M, int N,
                                                           float* __restrict__ dW, int ldw, float* __restrict__ db) {
    // 32 output elements per workgroup x 8 slices of the partials; slice sums are combined in slice order (fixed order)
    __shared__ double sm[8][32];
    const int x = threadIdx.x & 31, y = threadIdx.x >> 5, tot = M * N + M;
    const int e = blockIdx.x * 32 + x;
    double s = 0.0;
    if (e < tot) {
        const int per = (n_part + 7) / 8, b0 = y * per, b1 = b0 + per < n_part ? b0 + per : n_part;
        for (int b = b0; b < b1; ++b) s += (double)partial[(size_t)b * tot + e];
    }
    sm[y][x] = s;
    __syncthreads();
    if (y == 0 && e < tot) {
        double t = sm[0][x];
#pragma unroll
        for (int k = 1; k < 8; ++k) t += sm[k][x];
        if (e < M * N) dW[(size_t)(e / N) * ldw + e % N] = (float)t;
        else if (db) db[e - M * N] = (float)t;
    }
}

// g[p][c] = h[p][c] > 0 ? g[p][c] : 0   (ReLU backward), 4 columns per thread
__global__ __launch_bounds__(256) void relu_mask_kernel(float* __restrict__ g, int ldg, const float* __restrict__ h, int ldh,
                                                        long long n_pts, int n_cols) {
    const int per_row = n_cols >> 2;
    const long long gid = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (gid >= n_pts * per_row) return;
    const long long p = gid / per_row;
    const int c = (int)(gid % per_row) * 4;
    typedef float f32x4 __attribute__((ext_vector_type(4)));
    f32x4 gv = *reinterpret_cast<f32x4*>(g + p * ldg + c);
    const f32x4 hv = *reinterpret_cast<const f32x4*>(h + p * ldh + c);
#pragma unroll
    for (int j = 0; j < 4; ++j) gv[j] = hv[j] > 0.0f ? gv[j] : 0.0f;
    *reinterpret_cast<f32x4*>(g + p * ldg + c) = gv;
}

// ---- split-fp16 variant for the 256 x 256 layers (K7-X3) -------------------------------------------------------------
// Same reduction on the 16-bit matrix pipe: both operands are split on the fly into hi = fp16(v), lo = fp16(v - hi) and
// every 16-point k-step runs hi.hi + lo.hi + hi.lo (v_mfma_f32_32x32x16_f16, fp32 accumulate): 3 x 32 cycles per 16 points
// per 32x32 tile instead of 8 x 64.  With K = points an MFMA operand is 8 CONSECUTIVE POINTS of one column per lane, i.e. a
// strided gather; so the workgroup gathers each k-step ONCE (1024 operand slots = 16 column tiles x 64 lanes, 4 per
// thread: 8 coalesced dword loads, 4 split2, two ds_write_b128), stages it in LDS in MFMA operand order (double buffered,
// one barrier per k-step) and all 4 waves read their A (2 row tiles) and B (8 column tiles) operands from there.
// Operand loads run two k-steps ahead in registers.  G is expected in fp16 range (the fused input-gradient kernel's
// power-of-two scale takes care of that); db is accumulated from the fp32 values.  Output: the same partial layout as
// wgrad_kernel (KW = 1), reduced by wgrad_reduce_kernel.
template <class XT>   // X: float, or _Float16 (activations saved in 16 bits: their lo parts are zero, the three-MFMA product stays)
__global__ __launch_bounds__(256, 1) void wgrad_x3_kernel(const float* __restrict__ G, int ldg, const XT* __restrict__ X, int ldx,
                                                          long long n_pts, float* __restrict__ partial) {
    constexpr int M = 256, N = 256;
    __shared__ __attribute__((aligned(16))) unsigned char lds[2 * 16 * 2 * 1024];   // [buffer][tile 0..15][hi, lo][64 lanes x 16 B]
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, i = lane & 31, kg = lane >> 5;
    // this workgroup's k-steps (16 points each): a contiguous range; the last step of the whole problem may be ragged
    const long long n_steps = (n_pts + 15) / 16;
    const long long per = (n_steps + gridDim.x - 1) / gridDim.x;
    const long long s0 = (long long)blockIdx.x * per, s1 = s0 + per < n_steps ? s0 + per : n_steps;
    const long long n_full_all = n_pts / 16;
    const long long f1 = s1 < n_full_all ? s1 : n_full_all;            // full steps are [s0, f1)
    const bool ragged = s1 > f1 && s1 > s0;                              // ... and possibly one ragged step, index f1
    f32x16 acc[2][8];
#pragma unroll
    for (int r = 0; r < 2; ++r)
#pragma unroll
        for (int t = 0; t < 8; ++t)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[r][t][e] = 0.0f;
    float bsum[2] = {0.0f, 0.0f};

    // slot j of this thread: column tile T = 4j + wave of [G | X] (T < 8: G), lane position (i, kg): points 8kg .. 8kg+7 of the step.
    // Addressing as in wgrad_kernel: wave-uniform row pointers (scalar unit) + one constant 32-bit lane offset per slot.
    const int wave_s = __builtin_amdgcn_readfirstlane(wave);
    // (slots 0, 1 are G tiles, slots 2, 3 X tiles for every wave: T < 8 <=> j < 2; addresses in bytes: the two element types differ)
    const char* src[4];
    long long ldb[4];     // row stride in bytes
    unsigned voff[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int T = 4 * j + wave_s;
        const int esz = j < 2 ? 4 : (int)sizeof(XT);
        src[j] = j < 2 ? reinterpret_cast<const char*>(G + 32 * (T & 7)) : reinterpret_cast<const char*>(X + 32 * (T & 7));
        ldb[j] = (long long)(j < 2 ? ldg : ldx) * esz;
        voff[j] = (unsigned)(8 * kg * ldb[j] + i * esz);   // bytes
    }
    auto fetch = [&](long long step, float (&raw)[4][8]) {               // every point of the step is in range
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const unsigned long long rb = (unsigned long long)(src[j] + step * 16 * ldb[j]);   // wave-uniform; tell hipcc so
            typedef const __attribute__((address_space(1))) char* gptr;
            const gptr row = (gptr)(((unsigned long long)(unsigned)__builtin_amdgcn_readfirstlane((unsigned)(rb >> 32)) << 32) |
                                    (unsigned long long)(unsigned)__builtin_amdgcn_readfirstlane((unsigned)rb));   // (the builtin returns int)
            unsigned vo = voff[j];
            asm volatile("" : "+v"(vo));   // ... and keep LICM from hoisting zext(lane offset) out of the loop, which loses the saddr form
#pragma unroll
            for (int e = 0; e < 8; ++e) {   // scalar base + zero-extended 32-bit lane offset: global_load_dword v, v_off, s[base]
                unsigned long long re = (unsigned long long)(row + (long long)e * ldb[j]);
                asm("" : "+s"(re));         // keep the row pointer a scalar of its own (hipcc otherwise folds e * ld into the lane offset)
                const auto at = reinterpret_cast<const __attribute__((address_space(1))) char*>(re) + vo;
                if (j < 2) raw[j][e] = *reinterpret_cast<const __attribute__((address_space(1))) float*>(at);
                else raw[j][e] = (float)*reinterpret_cast<const __attribute__((address_space(1))) XT*>(at);
            }
        }
    };
    auto fetch_masked = [&](long long step, float (&raw)[4][8]) {
        const long long p0 = step * 16 + 8 * kg;
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                const long long p = p0 + e;
                const char* at = src[j] + (p < n_pts ? p : n_pts - 1) * ldb[j] + i * (j < 2 ? 4 : (int)sizeof(XT));
                const float v = j < 2 ? *reinterpret_cast<const float*>(at) : (float)*reinterpret_cast<const XT*>(at);
                raw[j][e] = p < n_pts ? v : 0.0f;
            }
    };
    // split one slot and put it into LDS in operand order; w = 1 if the step is a real one (0: a re-fetched step past the end)
    auto stage_slot = [&](int j, const float (&raw)[4][8], int buf, float w) {
        u32x4 h, l;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            unsigned a, b;
            split2(raw[j][2 * q], raw[j][2 * q + 1], a, b);
            h[q] = a; l[q] = b;
        }
        if (j < 2) bsum[j] += w * (((raw[j][0] + raw[j][1]) + (raw[j][2] + raw[j][3])) + ((raw[j][4] + raw[j][5]) + (raw[j][6] + raw[j][7])));
        unsigned char* dst = lds + ((buf * 16 + 4 * j + wave_s) * 2) * 1024 + lane * 16;
        *reinterpret_cast<u32x4*>(dst) = h;
        *reinterpret_cast<u32x4*>(dst + 1024) = l;
    };
    auto operand = [&](int buf, int T, int part) {
        return *reinterpret_cast<const u32x4*>(lds + ((buf * 16 + T) * 2 + part) * 1024 + lane * 16);
    };
    // the 48 MFMAs of one k-step in 4 quarters (column-tile pairs); `between(q)` runs after quarter q
    auto compute = [&](int buf, auto&& between) {
        u32x4 ah[2], al[2];
#pragma unroll
        for (int r = 0; r < 2; ++r) { ah[r] = operand(buf, wave_s + 4 * r, 0); al[r] = operand(buf, wave_s + 4 * r, 1); }
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            u32x4 bh[2], bl[2];
#pragma unroll
            for (int c = 0; c < 2; ++c) { bh[c] = operand(buf, 8 + 2 * q + c, 0); bl[c] = operand(buf, 8 + 2 * q + c, 1); }
            // product-major: the same accumulator comes back every 4th MFMA (free from distance 4, mfma_chain.hip)
#pragma unroll
            for (int c = 0; c < 2; ++c)
#pragma unroll
                for (int r = 0; r < 2; ++r) acc[r][2 * q + c] = mfma16(ah[r], bh[c], acc[r][2 * q + c]);
#pragma unroll
            for (int c = 0; c < 2; ++c)
#pragma unroll
                for (int r = 0; r < 2; ++r) acc[r][2 * q + c] = mfma16(al[r], bh[c], acc[r][2 * q + c]);
#pragma unroll
            for (int c = 0; c < 2; ++c)
#pragma unroll
                for (int r = 0; r < 2; ++r) acc[r][2 * q + c] = mfma16(ah[r], bl[c], acc[r][2 * q + c]);
            between(q);
        }
    };

    const long long nf = f1 > s0 ? f1 - s0 : 0;                           // full steps of this workgroup
    if (nf > 0) {
        float R0[4][8], R1[4][8];
        const long long s0u = __builtin_amdgcn_readfirstlane((int)s0);   // < 2^31 steps
        auto clampf = [&](long long j) { return s0u + (j < nf ? j : nf - 1); };   // past the end: re-fetch the last full step
        fetch(clampf(0), R0);
#pragma unroll
        for (int j = 0; j < 4; ++j) stage_slot(j, R0, 0, 1.0f);
        fetch(clampf(1), R1);
        fetch(clampf(2), R0);
        __syncthreads();
        // pairs of steps (2p, 2p+1), no branch inside: at the top buffer 0 holds step 2p, R1 step 2p+1, R0 step 2p+2
        for (long long pp = 0; pp < nf / 2; ++pp) {
            compute(0, [&](int q) { stage_slot(q, R1, 1, 1.0f); });
            fetch(clampf(2 * pp + 3), R1);
            __syncthreads();
            const float w0 = 2 * pp + 2 < nf ? 1.0f : 0.0f;       // past the end this stages a re-fetched step that is never computed
            compute(1, [&](int q) { stage_slot(q, R0, 0, w0); });
            fetch(clampf(2 * pp + 4), R0);
            __syncthreads();
        }
        if (nf & 1) compute(0, [&](int) {});                      // the odd last full step is already staged in buffer 0
    }
    if (ragged) {
        float R[4][8];
        fetch_masked(f1, R);
#pragma unroll
        for (int j = 0; j < 4; ++j) stage_slot(j, R, 1, 1.0f);    // buffer 1 is free: its last readers passed a barrier
        __syncthreads();
        compute(1, [&](int) {});
    }
    float* out = partial + (size_t)blockIdx.x * ((size_t)M * N + M);
#pragma unroll
    for (int r = 0; r < 2; ++r) {
        const int mt = wave + 4 * r;
#pragma unroll
        for (int t = 0; t < 8; ++t)
#pragma unroll
            for (int e = 0; e < 16; ++e)   // accumulator element e of lane (i, kg): row (e&3) + 8(e>>2) + 4kg, column i
                out[(size_t)(32 * mt + (e & 3) + 8 * (e >> 2) + 4 * kg) * N + 32 * t + i] = acc[r][t][e];
        const float sb = bsum[r] + __shfl_xor(bsum[r], 32, NSOS_WAVE);   // slot j = r is column tile 4r + wave = mt of G
        if (kg == 0) out[(size_t)M * N + 32 * mt + i] = sb;
    }
}

template <int RT, int NT, class XT>
void launch_wgrad(int blocks, hipStream_t st, const float* G, int ldg, const XT* X, int ldx, long long n_pts, int M, int N, float* ws) {
    hipLaunchKernelGGL((wgrad_kernel<RT, NT, XT>), dim3(blocks), dim3(256), 0, st, G, ldg, X, ldx, n_pts, M, N, ws);
}
}  // namespace

// largest case: M = 32 (KW = 4 k-subsets), N = 256, or M = N = 256 (KW = 1)
extern "C" size_t nsos_wgrad_workspace_bytes(void) { return (size_t)kWgradMaxBlocks * (256 * 256 + 256) * sizeof(float); }

template <class XT>
static int32_t wgrad_entry(const float* G, int32_t ldg, const XT* X, int32_t ldx, int64_t n_pts, int32_t M, int32_t N,
                           float* dW, int32_t ldw, float* db, void* workspace, size_t workspace_bytes, void* stream) {
    NSOS_REQUIRE(dW && workspace, NSOS_ERR_NULL_POINTER);
    NSOS_REQUIRE(n_pts >= 0, NSOS_ERR_BAD_SHAPE);
    NSOS_REQUIRE(n_pts == 0 || (G && X), NSOS_ERR_NULL_POINTER);
    NSOS_REQUIRE((M == 32 || M == 64 || M == 128 || M == 256) && (N == 32 || N == 64 || N == 128 || N == 256), NSOS_ERR_UNSUPPORTED);
    NSOS_REQUIRE(ldg >= M && ldx >= N && ldw >= N, NSOS_ERR_BAD_SHAPE);
    const int MT = M / 32, RW = MT < 4 ? MT : 4, KW = 4 / RW, RT = MT / RW, NT = N / 32;
    int blocks = kWgradMaxBlocks;
    const long long units = (n_pts + 2 * KW - 1) / (2 * KW);
    if (units < blocks) blocks = (int)(units > 0 ? units : 1);
    const size_t need = (size_t)blocks * KW * ((size_t)M * N + M) * sizeof(float);
    NSOS_REQUIRE(workspace_bytes >= need, NSOS_ERR_BUFFER_TOO_SMALL);
    const hipStream_t st = (hipStream_t)stream;
    float* ws = static_cast<float*>(workspace);
#define NSOS_WG(R, T) launch_wgrad<R, T, XT>(blocks, st, G, ldg, X, ldx, (long long)n_pts, M, N, ws)
    if (RT == 1) { switch (NT) { case 1: NSOS_WG(1, 1); break; case 2: NSOS_WG(1, 2); break; case 4: NSOS_WG(1, 4); break; default: NSOS_WG(1, 8); break; } }
    else         { switch (NT) { case 1: NSOS_WG(2, 1); break; case 2: NSOS_WG(2, 2); break; case 4: NSOS_WG(2, 4); break; default: NSOS_WG(2, 8); break; } }
#undef NSOS_WG
    const int tot = M * N + M;
    hipLaunchKernelGGL(wgrad_reduce_kernel, dim3((tot + 31) / 32), dim3(256), 0, st, ws, blocks * KW, M, N, dW, ldw, db);
    return nsos_launch_status();
}

extern "C" int32_t nsos_wgrad(const float* G, int32_t ldg, const float* X, int32_t ldx, int64_t n_pts, int32_t M, int32_t N,
                              float* dW, int32_t ldw, float* db, void* workspace, size_t workspace_bytes, void* stream) {
    return wgrad_entry<float>(G, ldg, X, ldx, n_pts, M, N, dW, ldw, db, workspace, workspace_bytes, stream);
}
extern "C" int32_t nsos_wgrad_xh(const float* G, int32_t ldg, const void* X_f16, int32_t ldx, int64_t n_pts, int32_t M, int32_t N,
                                 float* dW, int32_t ldw, float* db, void* workspace, size_t workspace_bytes, void* stream) {
    return wgrad_entry<_Float16>(G, ldg, static_cast<const _Float16*>(X_f16), ldx, n_pts, M, N, dW, ldw, db, workspace, workspace_bytes, stream);
}

// A whole list of reductions over column blocks of ONE (G, X) pair of row-major buffers -- the weight gradients of a generic-architecture
// net (backward.py: generic_mlp_backward): one call, and no host work per item, instead of one Python-level nsos_wgrad call with three
// tensor slicings per (Linear, segment, 256 / 128 / 64 / 32 tile).  Items run in list order on the stream; they share the workspace, so
// each item is its kernel + its reduction exactly as nsos_wgrad issues them (same numbers).
extern "C" int32_t nsos_wgrad_batch(const nsos_wgrad_item* items, int32_t n_items, const float* G, int32_t ldg, const float* X, int32_t ldx,
                                    int64_t n_pts, float* out, void* workspace, size_t workspace_bytes, void* stream) {
    NSOS_REQUIRE(n_items >= 0 && (n_items == 0 || items), NSOS_ERR_NULL_POINTER);
    NSOS_REQUIRE(n_items == 0 || out, NSOS_ERR_NULL_POINTER);
    for (int i = 0; i < n_items; ++i) {
        const nsos_wgrad_item& it = items[i];
        NSOS_REQUIRE(it.g_col >= 0 && it.x_col >= 0 && it.w_off >= 0 && it.g_col + it.M <= ldg && it.x_col + it.N <= ldx, NSOS_ERR_BAD_SHAPE);
        const int32_t rc = wgrad_entry<float>(G + it.g_col, ldg, X + it.x_col, ldx, n_pts, it.M, it.N, out + it.w_off, it.ldw,
                                              it.b_off >= 0 ? out + it.b_off : nullptr, workspace, workspace_bytes, stream);
        if (rc != NSOS_OK) return rc;
    }
    return NSOS_OK;
}

extern "C" int32_t nsos_relu_mask(float* g, int32_t ldg, const float* h, int32_t ldh, int64_t n_pts, int32_t n_cols, void* stream) {
    if (n_pts == 0) return NSOS_OK;
    NSOS_REQUIRE(g && h, NSOS_ERR_NULL_POINTER);
    NSOS_REQUIRE(n_pts > 0 && n_cols > 0 && (n_cols & 3) == 0 && (ldg & 3) == 0 && (ldh & 3) == 0, NSOS_ERR_BAD_SHAPE);
    NSOS_REQUIRE((((uintptr_t)g | (uintptr_t)h) & 15) == 0, NSOS_ERR_MISALIGNED);
    const long long tot = (long long)n_pts * (n_cols >> 2);
    NSOS_REQUIRE((tot + 255) / 256 < (1ll << 31), NSOS_ERR_UNSUPPORTED);
    hipLaunchKernelGGL(relu_mask_kernel, dim3((unsigned)((tot + 255) / 256)), dim3(256), 0, (hipStream_t)stream, g, ldg, h, ldh,
                       (long long)n_pts, n_cols);
    return nsos_launch_status();
}

template <class XT>
static int32_t wgrad_x3_entry(const float* G, int32_t ldg, const XT* X, int32_t ldx, int64_t n_pts, float* dW, int32_t ldw,
                              float* db, void* workspace, size_t workspace_bytes, void* stream) {
    constexpr int M = 256, N = 256;
    NSOS_REQUIRE(dW && workspace, NSOS_ERR_NULL_POINTER);
    NSOS_REQUIRE(n_pts >= 0, NSOS_ERR_BAD_SHAPE);
    NSOS_REQUIRE(n_pts == 0 || (G && X), NSOS_ERR_NULL_POINTER);
    NSOS_REQUIRE(ldg >= M && ldx >= N && ldw >= N, NSOS_ERR_BAD_SHAPE);
    int blocks = kWgradMaxBlocks;
    const long long steps = (n_pts + 15) / 16;
    if (steps < blocks) blocks = (int)(steps > 0 ? steps : 1);
    NSOS_REQUIRE(workspace_bytes >= (size_t)blocks * ((size_t)M * N + M) * sizeof(float), NSOS_ERR_BUFFER_TOO_SMALL);
    const hipStream_t st = (hipStream_t)stream;
    float* ws = static_cast<float*>(workspace);
    hipLaunchKernelGGL(wgrad_x3_kernel<XT>, dim3(blocks), dim3(256), 0, st, G, ldg, X, ldx, (long long)n_pts, ws);
    const int tot = M * N + M;
    hipLaunchKernelGGL(wgrad_reduce_kernel, dim3((tot + 31) / 32), dim3(256), 0, st, ws, blocks, M, N, dW, ldw, db);
    return nsos_launch_status();
}

extern "C" int32_t nsos_wgrad_x3(const float* G, int32_t ldg, const float* X, int32_t ldx, int64_t n_pts, float* dW, int32_t ldw,
                                 float* db, void* workspace, size_t workspace_bytes, void* stream) {
    return wgrad_x3_entry<float>(G, ldg, X, ldx, n_pts, dW, ldw, db, workspace, workspace_bytes, stream);
}
extern "C" int32_t nsos_wgrad_x3_xh(const float* G, int32_t ldg, const void* X_f16, int32_t ldx, int64_t n_pts, float* dW, int32_t ldw,
                                    float* db, void* workspace, size_t workspace_bytes, void* stream) {
    return wgrad_x3_entry<_Float16>(G, ldg, static_cast<const _Float16*>(X_f16), ldx, n_pts, dW, ldw, db, workspace, workspace_bytes, stream);
}
