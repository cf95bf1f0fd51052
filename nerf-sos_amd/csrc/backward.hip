// K5 (first part): backward of the semantic head for the reference's shipped training recipe, in which the
// backbone is frozen (run_nerf.py:307-318, scripts/train_*_node0.sh --fix_backbone) and only
// semantic_linear.{0,2} receive gradients (82 436 of 1 274 124 parameters).
//
//   semantics[r,:] = sum_s w[r,s] * logits[r,s,:]                      (models/renderer.py:64-66; w is backbone-only)
//   logits[p,:]    = W2 relu(W1 [h7(p), x63(p)] + b1) + b2              (models/nerf_mlp.py:61,79-80)
// so with G = dL/dsemantics:
//   g_logits[p,k] = w[p] * G[ray(p),k]
//   g_hid[p,f]    = (hid[p,f] > 0) * sum_k g_logits[p,k] * W2[k,f]
// and the weight gradients are plain GEMMs over the P points, done by the host with the BLAS library on the
// tensors this kernel writes and the ones the SAVE variant of the fused MLP kernel stored:
//   dW2 = g_logits^T hid,  db2 = g_logits^T 1,  [dW1 | db1] = g_hid^T [h7, x63, 1].
// This kernel is the element-wise part: HBM-bound, reads 512 B + writes 520 B per point.
#include "common.h"
#include "x3_common.h"   // split2 / mfma16 of the split-fp16 kernels (sem_head_wgrad_x3_kernel)

typedef float f32x4 __attribute__((ext_vector_type(4)));

__global__ __launch_bounds__(256) void sem_head_backward_kernel(const float* __restrict__ weights,
                                                                const float* __restrict__ g_sem,
                                                                const float* __restrict__ w2,
                                                                const float* __restrict__ hid, int64_t n_pts, int S,
                                                                float* __restrict__ g_hid,
                                                                float* __restrict__ g_logits) {
    const int64_t gid = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;  // one thread per (point, 4 features)
    const int64_t p = gid >> 5;
    if (p >= n_pts) return;
    const int f = (int)(gid & 31) * 4;
    const int64_t r = p / S;
    const float w = weights[p];
    const float gl0 = w * g_sem[2 * r], gl1 = w * g_sem[2 * r + 1];
    const f32x4 h = *reinterpret_cast<const f32x4*>(hid + p * 128 + f);
    const f32x4 a = *reinterpret_cast<const f32x4*>(w2 + f);
    const f32x4 b = *reinterpret_cast<const f32x4*>(w2 + 128 + f);
    f32x4 o;
#pragma unroll
    for (int j = 0; j < 4; ++j) o[j] = h[j] > 0.0f ? gl0 * a[j] + gl1 * b[j] : 0.0f;
    *reinterpret_cast<f32x4*>(g_hid + p * 128 + f) = o;
    if (f == 0) {
        g_logits[2 * p] = gl0;
        g_logits[2 * p + 1] = gl1;
    }
}

extern "C" int32_t nsos_sem_head_backward(const float* weights, const float* g_semantics, const float* sem2_w,
                                          const float* sem_hid, int64_t n_rays, int32_t n_samples, float* g_hid,
                                          float* g_logits, void* stream) {
    if (n_rays == 0) return NSOS_OK;
    NSOS_REQUIRE(weights && g_semantics && sem2_w && sem_hid && g_hid && g_logits, NSOS_ERR_NULL_POINTER);
    NSOS_REQUIRE(n_rays > 0 && n_samples >= 1, NSOS_ERR_BAD_SHAPE);
    NSOS_REQUIRE((((uintptr_t)sem2_w | (uintptr_t)sem_hid | (uintptr_t)g_hid) & 15) == 0, NSOS_ERR_MISALIGNED);
    const int64_t n_pts = n_rays * n_samples, total = n_pts * 32;
    NSOS_REQUIRE((total + 255) / 256 < (int64_t)1 << 31, NSOS_ERR_UNSUPPORTED);
    hipLaunchKernelGGL(sem_head_backward_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0,
                       (hipStream_t)stream, weights, g_semantics, sem2_w, sem_hid, n_pts, n_samples, g_hid, g_logits);
    return nsos_launch_status();
}

// ------------------------------------------------------------------------------------------------------------------
// K5 (second part): the weight gradients themselves, fused with the element-wise part above.
//   [dW1 | db1] [128,320] = sum_p g_hid[p,:]^T sem_in[p,:]      dW2 [2,128] = sum_p g_logits[p,:]^T hid[p,:]
//   db2 [2] = sum_p g_logits[p,:]
// These are reductions over millions of points into tiny outputs (M = 128 or 2, N = 320 or 128, K = R*S): the BLAS
// library picks 16x16 macro-tiles for that shape and runs at ~3 % of the fp32 MFMA rate (19 ms per pass for the 6.3 M
// points of a 32 768-ray step).  Here: persistent grid, workgroup b owns a contiguous block of points; wave w owns
// the 32 hidden features [32w, 32w+32) x all 320 columns = 10 accumulator tiles of v_mfma_f32_32x32x2_f32 (exact
// fp32), K = points, two per MFMA.  The A operand (g_hid) is formed in registers from sem_hid, the compositing
// weight and dL/dsemantics -- g_hid and g_logits are never written.  MFMA-bound: 81 920 MAC per point.
// Partials [n_blocks][128*320 + 2*128 + 2] are then summed in block order (fp64): deterministic, no atomics.
typedef float f32x16 __attribute__((ext_vector_type(16)));

namespace {
constexpr int kWgradOut = 128 * 320 + 2 * 128 + 2;

struct WgradIn {
    float a;        // g_hid of (feature 32w+i, point pt0+k)
    float b[10];    // sem_in columns 32nt+i of that point
};

__global__ __launch_bounds__(256, 1) void sem_head_wgrad_kernel(const float* __restrict__ weights, const float* __restrict__ g_sem,
                                                                const float* __restrict__ w2, const float* __restrict__ hid,
                                                                const float* __restrict__ sem_in, long long n_pts, int S,
                                                                float* __restrict__ partial) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, i = lane & 31, k = lane >> 5;
    long long chunk = (n_pts + gridDim.x - 1) / gridDim.x;
    chunk += chunk & 1;
    const long long start = (long long)blockIdx.x * chunk;
    const long long end = start + chunk < n_pts ? start + chunk : n_pts;
    const float w2a = w2[32 * wave + i], w2b = w2[128 + 32 * wave + i];
    f32x16 acc[10];
#pragma unroll
    for (int t = 0; t < 10; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[t][r] = 0.0f;
    float gw2[2] = {0.0f, 0.0f}, gb2[2] = {0.0f, 0.0f};

    auto fetch = [&](long long pt0, WgradIn& in) {
        const long long pt = pt0 + k;
        const bool valid = pt < end;
        const long long pc = valid ? pt : (n_pts - 1);
        float w = weights[pc];   // unconditional load of a clamped row, then a select: `valid ? load : 0` compiles to a
        w = valid ? w : 0.0f;    // branch with a vmcnt wait behind every load, which serialises the prefetch
        const long long r = pc / S;
        const float gl0 = w * g_sem[2 * r], gl1 = w * g_sem[2 * r + 1];   // g_logits (models/renderer.py:64-66)
        const float h = hid[pc * 128 + 32 * wave + i];
        in.a = h > 0.0f ? gl0 * w2a + gl1 * w2b : 0.0f;                    // g_hid (models/nerf_mlp.py:61)
        gw2[0] += gl0 * h;                                                 // hid is stored after its ReLU
        gw2[1] += gl1 * h;
        gb2[0] += gl0;
        gb2[1] += gl1;
        const float* row = sem_in + pc * 320 + i;
#pragma unroll
        for (int t = 0; t < 10; ++t) in.b[t] = row[32 * t];
    };
    constexpr int U = 4;   // fetch runs 4 k-steps (8 points, 2 560 MFMA cycles) ahead: one k-step is shorter than an HBM trip
    WgradIn bufA[U], bufB[U];   // used alternately: no register copies for hipcc to hoist into the MFMA stream
    auto compute = [&](const WgradIn (&in)[U]) {
#pragma unroll
        for (int u = 0; u < U; ++u)
#pragma unroll
            for (int t = 0; t < 10; ++t) acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(in[u].a, in[u].b[t], acc[t], 0, 0, 0);
    };
#pragma unroll
    for (int u = 0; u < U; ++u) fetch(start + 2 * u, bufA[u]);       // points >= end contribute zeros (valid == false)
    for (long long pt0 = start; pt0 < end; pt0 += 4 * U) {
#pragma unroll
        for (int u = 0; u < U; ++u) fetch(pt0 + 2 * (U + u), bufB[u]);
        compute(bufA);
#pragma unroll
        for (int u = 0; u < U; ++u) fetch(pt0 + 2 * (2 * U + u), bufA[u]);
        compute(bufB);
    }
    float* out = partial + (size_t)blockIdx.x * kWgradOut;
#pragma unroll
    for (int t = 0; t < 10; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r)   // accumulator element r of lane (i, k): row (r&3) + 8(r>>2) + 4k, column i
            out[(size_t)(32 * wave + (r & 3) + 8 * (r >> 2) + 4 * k) * 320 + 32 * t + i] = acc[t][r];
#pragma unroll
    for (int o = 0; o < 2; ++o) {
        const float s = gw2[o] + __shfl_xor(gw2[o], 32, NSOS_WAVE);
        if (k == 0) out[128 * 320 + o * 128 + 32 * wave + i] = s;
        const float sb = gb2[o] + __shfl_xor(gb2[o], 32, NSOS_WAVE);
        if (wave == 0 && lane == 0) out[128 * 320 + 256 + o] = sb;
    }
}

// The same reduction on the 16-bit matrix pipe with split operands (hi = fp16(v), lo = fp16(v - hi); hi.hi + lo.hi + hi.lo,
// fp32 accumulate): 3 x 32 cycles per 16 points per 32x32 tile instead of 8 x 64 -- the kernel becomes HBM-bound (1.8 KB per
// point).  Structure of wgrad_x3_kernel (wgrad.hip): per 16-point k-step the workgroup forms / gathers every operand ONCE
// (14 column tiles x 64 lanes = 896 slots: wave w forms g_hid tile w from sem_hid, the compositing weight and dL/dsemantics,
// and gathers sem_in tiles w, 4+w and -- waves 0, 1 -- 8+w), splits it and stages it in LDS in MFMA operand order (double
// buffered, one barrier per k-step); wave w then runs its 30 MFMAs (A = g_hid tile w, B = the 10 sem_in tiles).  g_hid has
// no natural scale: the caller passes a power of two (device scalar) that brings it into fp16 range and divides gw1 by it;
// gw2 / gb2 stay plain fp32 sums exactly as in sem_head_wgrad_kernel.  n_samples >= 8, n_pts < 2^31.
// sem_in here is the fp32 matrix (mlp_precision = "fp16x3"); the compact 16-bit matrix of the bf16 / fp16 paths has a kernel of
// its own (sem_wgrad16.hip).
__global__ __launch_bounds__(256, 1) void sem_head_wgrad_x3_kernel(const float* __restrict__ weights, const float* __restrict__ g_sem,
                                                                   const float* __restrict__ w2, const float* __restrict__ hid,
                                                                   const float* __restrict__ sem_in, const float* __restrict__ scale_p,
                                                                   long long n_pts, int S, float* __restrict__ partial) {
    __shared__ __attribute__((aligned(16))) unsigned char lds[2 * 14 * 2 * 1024];   // [buffer][tile 0..13][hi, lo][64 lanes x 16 B]
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, i = lane & 31, kg = lane >> 5;
    const int wave_s = __builtin_amdgcn_readfirstlane(wave);
    const long long n_steps = (n_pts + 15) / 16;
    const long long per = (n_steps + gridDim.x - 1) / gridDim.x;
    const long long s0 = (long long)blockIdx.x * per, s1 = s0 + per < n_steps ? s0 + per : n_steps;
    const long long n_full_all = n_pts / 16;
    const long long f1 = s1 < n_full_all ? s1 : n_full_all;            // full steps are [s0, f1), then possibly one ragged step f1
    const bool ragged = s1 > f1 && s1 > s0;
    const float scale = *scale_p;
    const float w2a = w2[32 * wave + i], w2b = w2[128 + 32 * wave + i];
    f32x16 acc[10];
#pragma unroll
    for (int t = 0; t < 10; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[t][r] = 0.0f;
    float gw2[2] = {0.0f, 0.0f}, gb2[2] = {0.0f, 0.0f};
    // sem_in tiles staged by this wave: w, 4+w and 8 + (w & 1): waves 2, 3 duplicate tiles 8, 9 (identical values into the same LDS
    // slot) rather than branch around a third of their loads
    const int xt2 = 8 + (wave_s & 1);

    struct Raw { float h[8], wt[8], g0[8], g1[8], x[3][8]; };
    // Addressing of the full steps as in wgrad_x3_kernel: wave-uniform row pointers (scalar unit) + constant 32-bit lane offsets
    // (the half-wave's 8-point shift is part of the lane offset), so the loads are `global_load_dword v, v_off, s[row]`.
    typedef const __attribute__((address_space(1))) float* gptr;
    typedef const __attribute__((address_space(1))) char* gbytes;
    auto uniform = [](const float* p) {                                   // tell hipcc the pointer is wave-uniform
        const unsigned long long b = (unsigned long long)p;
        return (gptr)(((unsigned long long)(unsigned)__builtin_amdgcn_readfirstlane((unsigned)(b >> 32)) << 32) |
                      (unsigned long long)(unsigned)__builtin_amdgcn_readfirstlane((unsigned)b));
    };
    auto at = [](gptr row, unsigned byte_off) {
        unsigned long long r = (unsigned long long)row;
        asm("" : "+s"(r));                                                // keep the row pointer a scalar of its own
        return *reinterpret_cast<gptr>(reinterpret_cast<gbytes>(r) + byte_off);
    };
    const unsigned off_w = 8u * kg * 4u, off_h = (8u * kg * 128u + i) * 4u, off_x = (8u * kg * 320u + i) * 4u;
    auto fetch_full = [&](long long step, Raw& R) {
        const unsigned p0 = (unsigned)(step * 16) + 8u * (unsigned)kg;
        const unsigned q0 = p0 / (unsigned)S, rem0 = p0 - q0 * (unsigned)S;     // ray of the first point; S >= 8: at most one crossing below
        unsigned ow = off_w, oh = off_h, ox = off_x;
        asm volatile("" : "+v"(ow), "+v"(oh), "+v"(ox));                  // ... and keep LICM from hoisting their zero-extension
        const gptr wrow = uniform(weights + step * 16), hrow = uniform(hid + step * 16 * 128 + 32 * wave_s);
        gptr xrow[3];
#pragma unroll
        for (int j = 0; j < 3; ++j) {
            const int T = j < 2 ? 4 * j + wave_s : xt2;
            xrow[j] = uniform(sem_in + step * 16 * 320 + 32 * T);
        }
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const unsigned r = q0 + ((rem0 + e) >= (unsigned)S ? 1u : 0u);
            R.wt[e] = at(wrow + e, ow);
            R.g0[e] = g_sem[2ull * r];
            R.g1[e] = g_sem[2ull * r + 1];
            R.h[e] = at(hrow + e * 128, oh);
#pragma unroll
            for (int j = 0; j < 3; ++j) R.x[j][e] = at(xrow[j] + e * 320, ox);
        }
    };
    // the (single) ragged step -- rows clamped, out-of-range points get weight 0 (their g_hid is then 0)
    auto fetch_masked = [&](long long step, Raw& R) {
        const unsigned p0 = (unsigned)(step * 16) + 8u * (unsigned)kg;
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            unsigned p = p0 + e;
            const bool ok = p < (unsigned)n_pts;
            p = ok ? p : (unsigned)n_pts - 1u;
            const unsigned r = p / (unsigned)S;
            const float w = weights[p];
            R.wt[e] = ok ? w : 0.0f;
            R.g0[e] = g_sem[2ull * r];
            R.g1[e] = g_sem[2ull * r + 1];
            R.h[e] = hid[(unsigned long long)p * 128 + 32 * wave + i];
#pragma unroll
            for (int j = 0; j < 3; ++j) R.x[j][e] = sem_in[(unsigned long long)p * 320 + 32 * (j < 2 ? 4 * j + wave : xt2) + i];
        }
    };
    auto put = [&](int buf, int T, const float (&v)[8]) {               // split 8 points of one column and store the operand pair
        u32x4 h, l;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            unsigned a, b;
            split2(v[2 * q], v[2 * q + 1], a, b);
            h[q] = a; l[q] = b;
        }
        unsigned char* dst = lds + ((buf * 14 + T) * 2) * 1024 + lane * 16;
        *reinterpret_cast<u32x4*>(dst) = h;
        *reinterpret_cast<u32x4*>(dst + 1024) = l;
    };
    // part 0: this wave's g_hid tile (and the plain fp32 sums); parts 1..3: its sem_in tiles.  w = 0 for a re-fetched step
    auto stage_part = [&](int part, const Raw& R, int buf, float w) {
        if (part == 0) {
            float a[8];
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                const float gl0 = R.wt[e] * R.g0[e], gl1 = R.wt[e] * R.g1[e];       // g_logits (models/renderer.py:64-66)
                const float gh = R.h[e] > 0.0f ? gl0 * w2a + gl1 * w2b : 0.0f;       // g_hid (models/nerf_mlp.py:61)
                a[e] = gh * scale;
                gw2[0] += w * (gl0 * R.h[e]);                                        // hid is stored after its ReLU
                gw2[1] += w * (gl1 * R.h[e]);
                gb2[0] += w * gl0;
                gb2[1] += w * gl1;
            }
            put(buf, wave_s, a);
        } else {
            put(buf, 4 + (part < 3 ? 4 * (part - 1) + wave_s : xt2), R.x[part - 1]);
        }
    };
    auto operand = [&](int buf, int T, int part) {
        return *reinterpret_cast<const u32x4*>(lds + ((buf * 14 + T) * 2 + part) * 1024 + lane * 16);
    };
    // 30 MFMAs of one k-step: two groups of 5 column tiles, product-major inside a group (an accumulator returns every 5th MFMA)
    auto compute = [&](int buf, auto&& between) {
        const u32x4 ah = operand(buf, wave_s, 0), al = operand(buf, wave_s, 1);
#pragma unroll
        for (int g = 0; g < 2; ++g) {
            u32x4 bh[5], bl[5];
#pragma unroll
            for (int c = 0; c < 5; ++c) { bh[c] = operand(buf, 4 + 5 * g + c, 0); bl[c] = operand(buf, 4 + 5 * g + c, 1); }
#pragma unroll
            for (int c = 0; c < 5; ++c) acc[5 * g + c] = mfma16(ah, bh[c], acc[5 * g + c]);
            between(2 * g);
#pragma unroll
            for (int c = 0; c < 5; ++c) acc[5 * g + c] = mfma16(al, bh[c], acc[5 * g + c]);
#pragma unroll
            for (int c = 0; c < 5; ++c) acc[5 * g + c] = mfma16(ah, bl[c], acc[5 * g + c]);
            between(2 * g + 1);
        }
    };
    const long long nf = f1 > s0 ? f1 - s0 : 0;
    if (nf > 0) {
        Raw R0, R1;
        const long long s0u = __builtin_amdgcn_readfirstlane((int)s0);
        auto clampf = [&](long long j) { return s0u + (j < nf ? j : nf - 1); };   // past the end: re-fetch the last full step
        fetch_full(clampf(0), R0);
#pragma unroll
        for (int part = 0; part < 4; ++part) stage_part(part, R0, 0, 1.0f);
        fetch_full(clampf(1), R1);
        fetch_full(clampf(2), R0);
        __syncthreads();
        // pairs of steps (2p, 2p+1), no branch inside: at the top buffer 0 holds step 2p, R1 step 2p+1, R0 step 2p+2
        for (long long pp = 0; pp < nf / 2; ++pp) {
            compute(0, [&](int q) { stage_part(q, R1, 1, 1.0f); });
            fetch_full(clampf(2 * pp + 3), R1);
            __syncthreads();
            const float w0 = 2 * pp + 2 < nf ? 1.0f : 0.0f;              // past the end: staged, never computed, not summed
            compute(1, [&](int q) { stage_part(q, R0, 0, w0); });
            fetch_full(clampf(2 * pp + 4), R0);
            __syncthreads();
        }
        if (nf & 1) compute(0, [&](int) {});
    }
    if (ragged) {
        Raw R;
        fetch_masked(f1, R);
#pragma unroll
        for (int part = 0; part < 4; ++part) stage_part(part, R, 1, 1.0f);
        __syncthreads();
        compute(1, [&](int) {});
    }
    float* out = partial + (size_t)blockIdx.x * kWgradOut;
#pragma unroll
    for (int t = 0; t < 10; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r)   // accumulator element r of lane (i, kg): row (r&3) + 8(r>>2) + 4kg, column i
            out[(size_t)(32 * wave + (r & 3) + 8 * (r >> 2) + 4 * kg) * 320 + 32 * t + i] = acc[t][r];
#pragma unroll
    for (int o = 0; o < 2; ++o) {
        const float s = gw2[o] + __shfl_xor(gw2[o], 32, NSOS_WAVE);
        if (kg == 0) out[128 * 320 + o * 128 + 32 * wave + i] = s;
        const float sb = gb2[o] + __shfl_xor(gb2[o], 32, NSOS_WAVE);
        if (wave == 0 && lane == 0) out[128 * 320 + 256 + o] = sb;
    }
}

// 64 outputs x 4 slabs of partials per workgroup, two independent fp64 chains per thread, slabs folded in order through LDS
// (a single thread walking all the partials of its output was 256 dependent loads: 63 us per call)
// gw1 is divided by *scale_p (a power of two; nullptr: 1) on the way out
// gb1 != nullptr: gw1 is written as the contiguous [128, in_dim] weight gradient and the bias gradient (column 319 of the
// augmented matrix) to gb1 [128] -- what autograd wants, without the two slicing copies per net and step
__global__ __launch_bounds__(256) void sem_head_wgrad_reduce_kernel(const float* __restrict__ partial, int n_blocks,
                                                                    const float* __restrict__ scale_p, float* __restrict__ gw1,
                                                                    float* __restrict__ gw2, float* __restrict__ gb2,
                                                                    float* __restrict__ gb1, int in_dim) {
    __shared__ double fold[3][64];
    const int el = threadIdx.x & 63, slab = threadIdx.x >> 6;
    const int e = blockIdx.x * 64 + el;
    const int per = (n_blocks + 3) / 4, b0 = slab * per, b1 = b0 + per < n_blocks ? b0 + per : n_blocks;
    double s0 = 0.0, s1 = 0.0;
    if (e < kWgradOut) {
        int b = b0;
        for (; b + 1 < b1; b += 2) {
            s0 += (double)partial[(size_t)b * kWgradOut + e];
            s1 += (double)partial[(size_t)(b + 1) * kWgradOut + e];
        }
        if (b < b1) s0 += (double)partial[(size_t)b * kWgradOut + e];
    }
    double s = s0 + s1;
    if (slab > 0) fold[slab - 1][el] = s;
    __syncthreads();
    if (slab != 0 || e >= kWgradOut) return;
    s += fold[0][el];
    s += fold[1][el];
    s += fold[2][el];
    if (e < 128 * 320) {
        const float v = (float)s * (scale_p ? 1.0f / *scale_p : 1.0f);
        if (!gb1) gw1[e] = v;
        else {
            const int row = e / 320, col = e - 320 * row;
            if (col < in_dim) gw1[row * in_dim + col] = v;
            else if (col == 319) gb1[row] = v;
        }
    }
    else if (e < 128 * 320 + 256) gw2[e - 128 * 320] = (float)s;
    else gb2[e - 128 * 320 - 256] = (float)s;
}

int wgrad_blocks() { return nsos_device_cus(); }
}  // namespace

// 1024 blocks of partial sums, then one float: the scale nsos_sem_head_wgrad_x3 derives when the caller passes none
extern "C" size_t nsos_sem_head_wgrad_workspace_bytes(void) { return (size_t)1024 * kWgradOut * sizeof(float) + 256; }

extern "C" int32_t nsos_sem_head_wgrad(const float* weights, const float* g_semantics, const float* sem2_w,
                                       const float* sem_hid, const float* sem_in, int64_t n_rays, int32_t n_samples,
                                       float* gw1_aug, float* gw2, float* gb2, void* workspace, size_t workspace_bytes,
                                       float* gb1, int32_t in_dim, void* stream) {
    NSOS_REQUIRE(gw1_aug && gw2 && gb2, NSOS_ERR_NULL_POINTER);
    NSOS_REQUIRE(!gb1 || (in_dim >= 1 && in_dim <= 319), NSOS_ERR_BAD_SHAPE);
    NSOS_REQUIRE(n_rays >= 0 && n_samples >= 1, NSOS_ERR_BAD_SHAPE);
    NSOS_REQUIRE(n_rays == 0 || (weights && g_semantics && sem2_w && sem_hid && sem_in && workspace), NSOS_ERR_NULL_POINTER);
    NSOS_REQUIRE(workspace_bytes >= nsos_sem_head_wgrad_workspace_bytes(), NSOS_ERR_BUFFER_TOO_SMALL);
    const long long n_pts = (long long)n_rays * n_samples;
    int blocks = wgrad_blocks();
    if (blocks > 1024) blocks = 1024;
    const long long pairs = (n_pts + 1) / 2;
    if (pairs < blocks) blocks = (int)(pairs > 0 ? pairs : 1);
    const hipStream_t st = (hipStream_t)stream;
    hipLaunchKernelGGL(sem_head_wgrad_kernel, dim3(blocks), dim3(256), 0, st, weights, g_semantics, sem2_w, sem_hid, sem_in,
                       n_pts, (int)n_samples, static_cast<float*>(workspace));
    hipLaunchKernelGGL(sem_head_wgrad_reduce_kernel, dim3((kWgradOut + 63) / 64), dim3(256), 0, st,
                       static_cast<const float*>(workspace), blocks, (const float*)nullptr, gw1_aug, gw2, gb2, gb1, (int)in_dim);
    return nsos_launch_status();
}

namespace nsos_detail {
int32_t sem_head_wgrad16(const float* weights, const float* g_semantics, const float* sem2_w, const void* sem_hid,
                         const void* sem_in, int32_t sem_in_dtype, int64_t n_rays, int32_t n_samples, const float* scale,
                         float* partial, int blocks, hipStream_t st, float* scale_out);   // sem_wgrad16.hip
}

namespace {
// The power of two that brings g_hid into fp16 range: |g_hid| <= max|dL/dsemantics| * max_m (|W2[0,m]| + |W2[1,m]|) (compositing
// weights are <= 1); that bound goes to 2^8.  One workgroup: 2 n_rays + 128 values.
__global__ __launch_bounds__(1024) void sem_head_scale_kernel(const float* __restrict__ g_sem, long long n, const float* __restrict__ w2,
                                                              float* __restrict__ scale_out) {
    __shared__ float red[16];
    float m = 0.0f;
    for (long long j = threadIdx.x; j < n; j += 1024) m = fmaxf(m, fabsf(g_sem[j]));
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) m = fmaxf(m, __shfl_xor(m, off, NSOS_WAVE));
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = m;
    __syncthreads();
    if (threadIdx.x >= 64) return;
    m = threadIdx.x < 16 ? red[threadIdx.x] : 0.0f;
    float c = fmaxf(fabsf(w2[threadIdx.x]) + fabsf(w2[128 + threadIdx.x]), fabsf(w2[64 + threadIdx.x]) + fabsf(w2[192 + threadIdx.x]));
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) {
        m = fmaxf(m, __shfl_xor(m, off, NSOS_WAVE));
        c = fmaxf(c, __shfl_xor(c, off, NSOS_WAVE));
    }
    if (threadIdx.x == 0) {
        const float bound = fmaxf(m * c, 1e-30f);
        const float e = fminf(fmaxf(floorf(log2f(256.0f / bound)), -60.0f), 60.0f);
        *scale_out = exp2f(e);
    }
}
}  // namespace

extern "C" int32_t nsos_sem_head_wgrad_x3(const float* weights, const float* g_semantics, const float* sem2_w,
                                          const void* sem_hid, const void* sem_in, int32_t sem_in_dtype, int64_t n_rays,
                                          int32_t n_samples, const float* scale, float* gw1_aug, float* gw2, float* gb2,
                                          void* workspace, size_t workspace_bytes, float* gb1, int32_t in_dim, void* stream) {
    NSOS_REQUIRE(!gb1 || (in_dim >= 1 && in_dim <= 319), NSOS_ERR_BAD_SHAPE);
    const int32_t tiled = sem_in_dtype & (NSOS_SEM_IN_TILED | NSOS_SEM_HID_TILED);
    sem_in_dtype &= ~(NSOS_SEM_IN_TILED | NSOS_SEM_HID_TILED);
    NSOS_REQUIRE(sem_in_dtype >= 0 && sem_in_dtype <= 2 && !(tiled && sem_in_dtype == 0), NSOS_ERR_UNSUPPORTED);
    NSOS_REQUIRE(gw1_aug && gw2 && gb2, NSOS_ERR_NULL_POINTER);
    NSOS_REQUIRE(n_rays >= 0 && n_samples >= 1, NSOS_ERR_BAD_SHAPE);
    NSOS_REQUIRE(n_rays == 0 || (weights && g_semantics && sem2_w && sem_hid && sem_in && workspace), NSOS_ERR_NULL_POINTER);
    NSOS_REQUIRE(workspace_bytes >= nsos_sem_head_wgrad_workspace_bytes(), NSOS_ERR_BUFFER_TOO_SMALL);
    const long long n_pts = (long long)n_rays * n_samples;
    NSOS_REQUIRE(n_samples >= 8 && n_pts < (1ll << 31), NSOS_ERR_UNSUPPORTED);
    int blocks = wgrad_blocks();
    if (blocks > 1024) blocks = 1024;
    const long long steps = (n_pts + 15) / 16;
    if (steps < blocks) blocks = (int)(steps > 0 ? steps : 1);
    const hipStream_t st = (hipStream_t)stream;
    float* ws = static_cast<float*>(workspace);
    float* const derived = ws + (size_t)1024 * kWgradOut;
    const bool in_kernel = !scale && sem_in_dtype != 0;      // the 16-bit kernel derives the scale itself (one launch less per call)
    if (!scale && !in_kernel) {
        hipLaunchKernelGGL(sem_head_scale_kernel, dim3(1), dim3(1024), 0, st, g_semantics, 2ll * n_rays, sem2_w, derived);
        scale = derived;
    }
    switch (sem_in_dtype) {
        case 0: hipLaunchKernelGGL(sem_head_wgrad_x3_kernel, dim3(blocks), dim3(256), 0, st, weights, g_semantics, sem2_w, static_cast<const float*>(sem_hid), static_cast<const float*>(sem_in), scale, n_pts, (int)n_samples, ws); break;
        default: {
            const int32_t rc = nsos_detail::sem_head_wgrad16(weights, g_semantics, sem2_w, sem_hid, sem_in, sem_in_dtype | tiled, n_rays, n_samples, scale, ws, blocks, st, derived);
            if (rc != NSOS_OK) return rc;
            if (in_kernel) scale = derived;
        }
    }
    hipLaunchKernelGGL(sem_head_wgrad_reduce_kernel, dim3((kWgradOut + 63) / 64), dim3(256), 0, st,
                       static_cast<const float*>(workspace), blocks, scale, gw1_aug, gw2, gb2, gb1, (int)in_dim);
    return nsos_launch_status();
}
