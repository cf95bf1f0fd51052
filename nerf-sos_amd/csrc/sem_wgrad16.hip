// K5 on the 16-bit training paths (C3 / C4): the semantic head's weight gradients when the forward kernel kept sem_in in its
// own 16-bit format (nsos_mlp_forward_rays_save16_lp).  Same reduction and the same partial-sum layout as
// sem_head_wgrad_x3_kernel (backward.hip):
//   [dW1 | db1] [128,320] = sum_p g_hid[p,:]^T sem_in[p,:],   dW2 [2,128] = sum_p g_logits[p,:]^T hid[p,:],   db2 = sum_p g_logits
//   g_logits[p,k] = w[p] G[ray(p),k],   g_hid[p,f] = (hid[p,f] > 0) sum_k g_logits[p,k] W2[k,f]          (models/nerf_mlp.py:61,79-80)
// g_hid is split into fp16 hi + lo; a 16-bit sem_in IS its own fp16 image (fp16: the stored word; bf16: widened and
// re-rounded -- exact from 2^-14 up, 8 significant bits into 11; below that the error is < 2^-25 absolute), so a product is
// TWO MFMAs (hi.x + lo.x), not three.
//
// Why a kernel of its own.  The 4-wave kernel reads 1.16 KB per point and ran at 2.3 TB/s -- neither HBM- nor MFMA-bound
// (L2-resident inputs: -19 %; MFMAs removed: -17 %).  At 248 live VGPRs hipcc sinks half of each operand fetch next to its
// use and drains vmcnt to ~0 at the top of every k-step: the memory latency is exposed once per 16 points.  Here:
//   * every global load is an asm statement and every wait a hand-counted s_waitcnt that pins the registers it releases:
//     three operand sets in flight per wave (48 points ahead), the set of step s+1 is staged while step s is multiplied and
//     refilled with step s+4 right away;
//   * 8 waves = two per SIMD at <= 256 registers: wave w multiplies g_hid tile (w & 3) by the five sem_in column tiles of
//     half (w >> 2).  Waves 0..3 fetch and form the g_hid tiles (and one sem_in tile: 8 + (w & 1)), waves 4..7 fetch two
//     sem_in tiles each; a SIMD's two waves are one of each kind, so the staging VALU of one runs under the MFMAs of the other.
// Launch: grid = #CUs, 512 threads, n_samples >= 8 (at most one ray crossing inside 8 consecutive points), n_pts < 2^31.
#include "common.h"
#include "x3_common.h"

namespace nsos_detail {
int32_t sem_head_wgrad16(const float* weights, const float* g_semantics, const float* sem2_w, const float* sem_hid,
                         const void* sem_in, int32_t sem_in_dtype, int64_t n_rays, int32_t n_samples, const float* scale,
                         float* partial, int blocks, hipStream_t st);
}

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));

namespace {
constexpr int kWgradOut = 128 * 320 + 2 * 128 + 2;      // as in backward.hip
constexpr int kBufBytes = (4 * 2 + 10) * 1024;           // g_hid tiles 0..3 (hi, lo), then sem_in tiles 0..9
constexpr int kLoadsG = 20, kLoadsX = 16;                // loads per fetch of a g-kind / x-kind wave


// the operand set of one 16-point step, as one wave holds it
struct SetG {            // waves 0..3
    f32x4 wt[2];         // compositing weights of the lane half's 8 points
    f32x2 ga, gb;        // dL/dsemantics of the ray of the first point and of the next ray
    float h[8];          // sem_hid column 32 gt + i
    unsigned x[8];       // sem_in column of the extra tile, raw 16 bit
    int cross;           // points e >= cross belong to the next ray
};
struct SetX {            // waves 4..7
    unsigned x[2][8];
};

template <int OFF>
__device__ __forceinline__ unsigned ld_u16(unsigned voff, unsigned long long base) {
    unsigned v;
    asm volatile("global_load_ushort %0, %1, %2 offset:%3" : "=v"(v) : "v"(voff), "s"(base), "i"(OFF));
    return v;
}
template <int OFF>
__device__ __forceinline__ float ld_f32(unsigned voff, unsigned long long base) {
    float v;
    asm volatile("global_load_dword %0, %1, %2 offset:%3" : "=v"(v) : "v"(voff), "s"(base), "i"(OFF));
    return v;
}
template <int OFF>
__device__ __forceinline__ f32x4 ld_f32x4(unsigned voff, unsigned long long base) {
    f32x4 v;
    asm volatile("global_load_dwordx4 %0, %1, %2 offset:%3" : "=v"(v) : "v"(voff), "s"(base), "i"(OFF));
    return v;
}
__device__ __forceinline__ f32x2 ld_f32x2(unsigned voff, unsigned long long base) {
    f32x2 v;
    asm volatile("global_load_dwordx2 %0, %1, %2" : "=v"(v) : "v"(voff), "s"(base));
    return v;
}
__device__ __forceinline__ unsigned long long uniform64(const void* p) {   // the pointer is wave-uniform: say so
    const unsigned long long b = (unsigned long long)p;
    return ((unsigned long long)(unsigned)__builtin_amdgcn_readfirstlane((unsigned)(b >> 32)) << 32) |
           (unsigned long long)(unsigned)__builtin_amdgcn_readfirstlane((unsigned)b);
}

// s_waitcnt vmcnt(N) that owns the registers it releases: nothing that reads them can be scheduled above it.  ("+v" operands
// count twice towards the 30 an asm statement may have: the set is pinned in two statements, which keep their order.)
template <int N>
__device__ __forceinline__ void wait_set(SetG& s) {
    asm volatile("s_waitcnt vmcnt(%10)"
                 : "+v"(s.wt[0]), "+v"(s.wt[1]), "+v"(s.ga), "+v"(s.gb), "+v"(s.h[0]), "+v"(s.h[1]), "+v"(s.h[2]), "+v"(s.h[3]),
                   "+v"(s.h[4]), "+v"(s.h[5])
                 : "i"(N));
    asm volatile("" : "+v"(s.h[6]), "+v"(s.h[7]), "+v"(s.x[0]), "+v"(s.x[1]), "+v"(s.x[2]), "+v"(s.x[3]), "+v"(s.x[4]), "+v"(s.x[5]),
                      "+v"(s.x[6]), "+v"(s.x[7]));
}
template <int N>
__device__ __forceinline__ void wait_set(SetX& s) {
    asm volatile("s_waitcnt vmcnt(%8)"
                 : "+v"(s.x[0][0]), "+v"(s.x[0][1]), "+v"(s.x[0][2]), "+v"(s.x[0][3]), "+v"(s.x[0][4]), "+v"(s.x[0][5]),
                   "+v"(s.x[0][6]), "+v"(s.x[0][7])
                 : "i"(N));
    asm volatile("" : "+v"(s.x[1][0]), "+v"(s.x[1][1]), "+v"(s.x[1][2]), "+v"(s.x[1][3]), "+v"(s.x[1][4]), "+v"(s.x[1][5]),
                      "+v"(s.x[1][6]), "+v"(s.x[1][7]));
}

template <int XFMT>
__device__ __forceinline__ u32x4 pack16(const unsigned (&v)[8]) {
    u32x4 h;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        if constexpr (XFMT == 1) {
            h[q] = v[2 * q] | (v[2 * q + 1] << 16);
        } else {
            unsigned w;
            asm("v_cvt_pk_f16_f32 %0, %1, %2" : "=v"(w) : "v"(v[2 * q] << 16), "v"(v[2 * q + 1] << 16));
            h[q] = w;
        }
    }
    return h;
}

template <int XFMT>   // 1: fp16, 2: bf16
__global__ __launch_bounds__(512, 1) void sem_head_wgrad16_kernel(const float* __restrict__ weights, const float* __restrict__ g_sem,
                                                                  const float* __restrict__ w2, const float* __restrict__ hid,
                                                                  const unsigned short* __restrict__ sem_in,
                                                                  const float* __restrict__ scale_p, long long n_pts,
                                                                  long long n_rays, int S, float* __restrict__ partial) {
    __shared__ __attribute__((aligned(16))) unsigned char lds[2 * kBufBytes];
    const int lane = threadIdx.x & 63, i = lane & 31, kg = lane >> 5;
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const int gt = wave & 3, ch = wave >> 2;
    const bool kind_g = wave < 4;
    const int xt0 = kind_g ? 8 + (wave & 1) : wave - 4, xt1 = wave;      // sem_in tiles this wave stages (x-kind: w - 4 and w)

    const long long n_full = n_pts / 16;
    const long long per = (n_full + gridDim.x - 1) / gridDim.x;
    const long long s0 = (long long)blockIdx.x * per;
    const long long s1 = s0 + per < n_full ? s0 + per : n_full;
    const int nf = __builtin_amdgcn_readfirstlane((int)(s1 > s0 ? s1 - s0 : 0));
    const float scale = *scale_p;
    const float w2a = w2[32 * gt + i] * scale, w2b = w2[128 + 32 * gt + i] * scale;   // scale is a power of two: exact

    f32x16 acc[5];
#pragma unroll
    for (int t = 0; t < 5; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[t][r] = 0.0f;
    float gw2[2] = {0.0f, 0.0f}, gb2[2] = {0.0f, 0.0f};

    // lane offsets (bytes) inside a step's rows; the half-wave's 8-point shift is part of them
    const unsigned off_w = 32u * kg, off_h = (8u * kg * 128u + i) * 4u, off_x = (8u * kg * 320u + i) * 2u;
    const unsigned long long g_base = uniform64(g_sem);

    auto fetch_g = [&](long long step, SetG& s) {
        const unsigned p0 = (unsigned)(step * 16) + 8u * (unsigned)kg;
        const unsigned q0 = p0 / (unsigned)S;
        const unsigned q1 = q0 + 1 < (unsigned)n_rays ? q0 + 1 : q0;
        s.cross = (int)((q0 + 1) * (unsigned)S - p0);
        const unsigned long long wb = uniform64(weights + step * 16);
        const unsigned long long hb = uniform64(hid + step * 16 * 128 + 32 * gt);
        const unsigned long long xb = uniform64(sem_in + step * 16 * 320 + 4 * 320 + 32 * xt0);   // row of point 4: offsets fit 13 bits
        s.wt[0] = ld_f32x4<0>(off_w, wb);
        s.wt[1] = ld_f32x4<16>(off_w, wb);
        s.ga = ld_f32x2(q0 * 8u, g_base);
        s.gb = ld_f32x2(q1 * 8u, g_base);
        s.h[0] = ld_f32<0 * 512>(off_h, hb); s.h[1] = ld_f32<1 * 512>(off_h, hb); s.h[2] = ld_f32<2 * 512>(off_h, hb);
        s.h[3] = ld_f32<3 * 512>(off_h, hb); s.h[4] = ld_f32<4 * 512>(off_h, hb); s.h[5] = ld_f32<5 * 512>(off_h, hb);
        s.h[6] = ld_f32<6 * 512>(off_h, hb); s.h[7] = ld_f32<7 * 512>(off_h, hb);
        s.x[0] = ld_u16<-4 * 640>(off_x, xb); s.x[1] = ld_u16<-3 * 640>(off_x, xb); s.x[2] = ld_u16<-2 * 640>(off_x, xb);
        s.x[3] = ld_u16<-1 * 640>(off_x, xb); s.x[4] = ld_u16<0>(off_x, xb);        s.x[5] = ld_u16<640>(off_x, xb);
        s.x[6] = ld_u16<2 * 640>(off_x, xb);  s.x[7] = ld_u16<3 * 640>(off_x, xb);
    };
    auto fetch_x = [&](long long step, SetX& s) {
        const unsigned long long xa = uniform64(sem_in + step * 16 * 320 + 4 * 320 + 32 * xt0);
        const unsigned long long xb = uniform64(sem_in + step * 16 * 320 + 4 * 320 + 32 * xt1);
        s.x[0][0] = ld_u16<-4 * 640>(off_x, xa); s.x[0][1] = ld_u16<-3 * 640>(off_x, xa); s.x[0][2] = ld_u16<-2 * 640>(off_x, xa);
        s.x[0][3] = ld_u16<-1 * 640>(off_x, xa); s.x[0][4] = ld_u16<0>(off_x, xa);        s.x[0][5] = ld_u16<640>(off_x, xa);
        s.x[0][6] = ld_u16<2 * 640>(off_x, xa);  s.x[0][7] = ld_u16<3 * 640>(off_x, xa);
        s.x[1][0] = ld_u16<-4 * 640>(off_x, xb); s.x[1][1] = ld_u16<-3 * 640>(off_x, xb); s.x[1][2] = ld_u16<-2 * 640>(off_x, xb);
        s.x[1][3] = ld_u16<-1 * 640>(off_x, xb); s.x[1][4] = ld_u16<0>(off_x, xb);        s.x[1][5] = ld_u16<640>(off_x, xb);
        s.x[1][6] = ld_u16<2 * 640>(off_x, xb);  s.x[1][7] = ld_u16<3 * 640>(off_x, xb);
    };
    auto tile = [&](int buf, int slot) { return reinterpret_cast<u32x4*>(lds + buf * kBufBytes + slot * 1024 + lane * 16); };
    auto stage_g = [&](const SetG& s, int buf) {
        float a[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const bool next = e >= s.cross;
            const float wte = s.wt[e >> 2][e & 3];
            const float gl0 = wte * (next ? s.gb[0] : s.ga[0]), gl1 = wte * (next ? s.gb[1] : s.ga[1]);   // g_logits (models/renderer.py:64-66)
            a[e] = s.h[e] > 0.0f ? __fmaf_rn(gl1, w2b, gl0 * w2a) : 0.0f;                              // g_hid x scale
            gw2[0] = __fmaf_rn(gl0, s.h[e], gw2[0]);                                                      // hid is stored after its ReLU
            gw2[1] = __fmaf_rn(gl1, s.h[e], gw2[1]);
            gb2[0] += gl0;
            gb2[1] += gl1;
        }
        u32x4 h, l;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            unsigned x, y;
            split2(a[2 * q], a[2 * q + 1], x, y);
            h[q] = x; l[q] = y;
        }
        *tile(buf, 2 * gt) = h;
        *tile(buf, 2 * gt + 1) = l;
        *tile(buf, 8 + xt0) = pack16<XFMT>(s.x);
    };
    auto stage_x = [&](const SetX& s, int buf) {
        *tile(buf, 8 + xt0) = pack16<XFMT>(s.x[0]);
        *tile(buf, 8 + xt1) = pack16<XFMT>(s.x[1]);
    };
    auto compute = [&](int buf) {
        const u32x4 ah = *tile(buf, 2 * gt), al = *tile(buf, 2 * gt + 1);
        u32x4 b[5];
#pragma unroll
        for (int c = 0; c < 5; ++c) b[c] = *tile(buf, 8 + 5 * ch + c);
#pragma unroll
        for (int c = 0; c < 5; ++c) acc[c] = mfma16(ah, b[c], acc[c]);
#pragma unroll
        for (int c = 0; c < 5; ++c) acc[c] = mfma16(al, b[c], acc[c]);
    };

    if (nf > 0) {
        const long long base = __builtin_amdgcn_readfirstlane((int)s0);
        auto at = [&](int j) { return base + (j < nf ? j : nf - 1); };       // past the end: re-fetch the last step (never staged)
        // One k-step of a wave of either kind.  J: the operand set that holds step s + 1 (sets rotate with the step)
        if (kind_g) {
            SetG A, B, C;
            fetch_g(at(0), A); fetch_g(at(1), B); fetch_g(at(2), C);
            wait_set<2 * kLoadsG>(A);
            stage_g(A, 0);
            fetch_g(at(3), A);
            __syncthreads();
            auto step = [&](int s, SetG& J) {
                if (s + 1 < nf) {
                    wait_set<2 * kLoadsG>(J);
                    stage_g(J, (s + 1) & 1);
                }
                fetch_g(at(s + 4), J);
                compute(s & 1);
                __syncthreads();
            };
            for (int s = 0; s < nf; s += 3) {
                step(s, B);
                if (s + 1 < nf) step(s + 1, C);
                if (s + 2 < nf) step(s + 2, A);
            }
        } else {
            SetX A, B, C;
            fetch_x(at(0), A); fetch_x(at(1), B); fetch_x(at(2), C);
            wait_set<2 * kLoadsX>(A);
            stage_x(A, 0);
            fetch_x(at(3), A);
            __syncthreads();
            auto step = [&](int s, SetX& J) {
                compute(s & 1);
                if (s + 1 < nf) {
                    wait_set<2 * kLoadsX>(J);
                    stage_x(J, (s + 1) & 1);
                }
                fetch_x(at(s + 4), J);
                __syncthreads();
            };
            for (int s = 0; s < nf; s += 3) {
                step(s, B);
                if (s + 1 < nf) step(s + 1, C);
                if (s + 2 < nf) step(s + 2, A);
            }
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");     // the re-fetched sets are still landing in registers
    }
    // the ragged step (n_pts % 16 points), by workgroup 0: rows clamped, out-of-range points get weight 0 (their g_hid is 0)
    if (blockIdx.x == 0 && (n_pts & 15)) {
        const unsigned p0 = (unsigned)(n_full * 16) + 8u * (unsigned)kg;
        auto row = [&](int e) { const unsigned p = p0 + e; return p < (unsigned)n_pts ? p : (unsigned)n_pts - 1u; };
        if (kind_g) {
            float a[8];
            unsigned xv[8];
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                const unsigned p = row(e);
                const float wte = p0 + e < (unsigned)n_pts ? weights[p] : 0.0f;
                const unsigned r = p / (unsigned)S;
                const float gl0 = wte * g_sem[2ull * r], gl1 = wte * g_sem[2ull * r + 1];
                const float hv = hid[(unsigned long long)p * 128 + 32 * gt + i];
                a[e] = hv > 0.0f ? __fmaf_rn(gl1, w2b, gl0 * w2a) : 0.0f;
                gw2[0] = __fmaf_rn(gl0, hv, gw2[0]);
                gw2[1] = __fmaf_rn(gl1, hv, gw2[1]);
                gb2[0] += gl0;
                gb2[1] += gl1;
                xv[e] = sem_in[(unsigned long long)p * 320 + 32 * xt0 + i];
            }
            u32x4 h, l;
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                unsigned x, y;
                split2(a[2 * q], a[2 * q + 1], x, y);
                h[q] = x; l[q] = y;
            }
            *tile(0, 2 * gt) = h;
            *tile(0, 2 * gt + 1) = l;
            *tile(0, 8 + xt0) = pack16<XFMT>(xv);
        } else {
            unsigned xa[8], xb[8];
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                const unsigned long long r = (unsigned long long)row(e) * 320 + i;
                xa[e] = sem_in[r + 32 * xt0];
                xb[e] = sem_in[r + 32 * xt1];
            }
            *tile(0, 8 + xt0) = pack16<XFMT>(xa);
            *tile(0, 8 + xt1) = pack16<XFMT>(xb);
        }
        __syncthreads();
        compute(0);
    }
    float* out = partial + (size_t)blockIdx.x * kWgradOut;
#pragma unroll
    for (int t = 0; t < 5; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r)   // accumulator element r of lane (i, kg): row (r&3) + 8(r>>2) + 4kg, column i
            out[(size_t)(32 * gt + (r & 3) + 8 * (r >> 2) + 4 * kg) * 320 + 32 * (5 * ch + t) + i] = acc[t][r];
    if (kind_g) {
#pragma unroll
        for (int o = 0; o < 2; ++o) {
            const float s = gw2[o] + __shfl_xor(gw2[o], 32, NSOS_WAVE);
            if (kg == 0) out[128 * 320 + o * 128 + 32 * gt + i] = s;
            const float sb = gb2[o] + __shfl_xor(gb2[o], 32, NSOS_WAVE);
            if (wave == 0 && lane == 0) out[128 * 320 + 256 + o] = sb;
        }
    }
}
}  // namespace

int32_t nsos_detail::sem_head_wgrad16(const float* weights, const float* g_semantics, const float* sem2_w, const float* sem_hid,
                                      const void* sem_in, int32_t sem_in_dtype, int64_t n_rays, int32_t n_samples,
                                      const float* scale, float* partial, int blocks, hipStream_t st) {
    const long long n_pts = (long long)n_rays * n_samples;
    const unsigned short* x = static_cast<const unsigned short*>(sem_in);
    if (sem_in_dtype == 1)
        hipLaunchKernelGGL(sem_head_wgrad16_kernel<1>, dim3(blocks), dim3(512), 0, st, weights, g_semantics, sem2_w, sem_hid, x,
                           scale, n_pts, (long long)n_rays, (int)n_samples, partial);
    else
        hipLaunchKernelGGL(sem_head_wgrad16_kernel<2>, dim3(blocks), dim3(512), 0, st, weights, g_semantics, sem2_w, sem_hid, x,
                           scale, n_pts, (long long)n_rays, (int)n_samples, partial);
    return nsos_launch_status();
}
