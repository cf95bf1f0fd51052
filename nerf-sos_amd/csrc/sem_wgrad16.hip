// K5 on the 16-bit training paths (C3 / C4): the semantic head's weight gradients when the forward kernel kept sem_in in its
// own 16-bit format (nsos_mlp_forward_rays_save16_lp).  Same reduction and the same partial-sum layout as
// sem_head_wgrad_x3_kernel (backward.hip):
//   [dW1 | db1] [128,320] = sum_p g_hid[p,:]^T sem_in[p,:],   dW2 [2,128] = sum_p g_logits[p,:]^T hid[p,:],   db2 = sum_p g_logits
//   g_logits[p,k] = w[p] G[ray(p),k],   g_hid[p,f] = (hid[p,f] > 0) sum_k g_logits[p,k] W2[k,f]          (models/nerf_mlp.py:61,79-80)
// g_hid is split into a hi + lo pair of sem_in's OWN 16-bit format and multiplied on that format's MFMA (fp16: 22 significant
// bits of g_hid; bf16: 16 -- against a sem_in that carries 8), so sem_in goes from memory to LDS to the matrix pipe as it is
// stored and a product is TWO MFMAs (hi.x + lo.x), not three.  (Until round 3 a bf16 sem_in was re-rounded to fp16 on the
// way into LDS: 36 VALU instructions per wave and step on a kernel whose step was VALU-bound, see below.)
//
// Why a kernel of its own.  The 4-wave kernel reads 1.16 KB per point and ran at 2.3 TB/s -- neither HBM- nor MFMA-bound
// (L2-resident inputs: -19 %; MFMAs removed: -17 %).  At 248 live VGPRs hipcc sinks half of each operand fetch next to its
// use and drains vmcnt to ~0 at the top of every k-step: the memory latency is exposed once per 16 points.  Here:
//   * every global load is an asm statement and every wait a hand-counted s_waitcnt that pins the registers it releases:
//     three operand sets in flight per wave (48 points ahead), the set of step s+1 is staged while step s is multiplied and
//     refilled with step s+4 right away;
//   * 8 waves = two per SIMD at <= 256 registers: wave w multiplies g_hid tile (w & 3) by the five sem_in column tiles of
//     half (w >> 2).  Waves 0..3 fetch and form g_hid, waves 4..7 move sem_in;
//   * g_hid is formed in the order the hidden activations lie in memory: a lane owns ONE point and 8 consecutive features
//     (one coalesced dwordx4 of `hid`, one weight, one ray gradient: three loads per step and lane, ~70 VALU), writes the
//     split words row-major ([16 points][128 features], rows padded to 288 B) and the MFMA operand comes back through the same
//     transposing LDS read as sem_in.  (Round 2 formed it in operand order -- a lane owned one feature of 8 points: eight
//     2-byte loads, per-point ray bookkeeping with an integer division per fetch, ~190 VALU per step on the four waves every
//     barrier waited for: 2.1 k cycles per 16-point step against 0.64 k of MFMA.)
//   * sem_in goes to LDS as it lies in memory -- [16 points][320 channels], rows padded to 672 B -- with three coalesced
//     dwordx4 loads per wave and step (a 32-channel x 8-point MFMA operand gathered from global memory was 8 two-byte loads per
//     lane: 80 load instructions per CU and step), and the MFMA operand (8 consecutive POINTS of one channel per lane) comes out
//     of LDS through the transposing read: a 16-lane group of ds_read_b64_tr_b16 turns a [4 rows][16 columns] block into
//     column-per-lane order (scripts/ubench/tr_read.hip prints what it returns), two reads per operand.  The 672-byte row
//     stride puts the four rows of a group 8 banks apart.
// Launch: grid = #CUs, 512 threads, n_samples >= 8 (at most one ray crossing inside 8 consecutive points), n_pts < 2^31.
#include "common.h"
#include "x3_common.h"

namespace nsos_detail {
int32_t sem_head_wgrad16(const float* weights, const float* g_semantics, const float* sem2_w, const void* sem_hid,
                         const void* sem_in, int32_t sem_in_dtype, int64_t n_rays, int32_t n_samples, const float* scale,
                         float* partial, int blocks, hipStream_t st, float* scale_out);
}

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef unsigned u32x2 __attribute__((ext_vector_type(2)));

#ifdef NSOS_WG_PROF   // diagnostic build (scripts/diag/wg_prof.py): s_memtime stamps around the phases of a step
__device__ unsigned long long nsos_wg_prof[8][8];
#endif
namespace {
constexpr int kWgradOut = 128 * 320 + 2 * 128 + 2;      // as in backward.hip
constexpr int kRowBytes = 672;                           // LDS row of one point's 320 channels (640 B) + 32 B: rows 8 banks apart
constexpr int kGRow = 288;                               // LDS row of one point's 128 g_hid words (256 B) + 32 B
constexpr int kGBytes = 2 * 16 * kGRow;                  // g_hid of the step's 16 points, row-major: hi words, then lo words
constexpr int kBufBytes = kGBytes + 16 * kRowBytes;      // ... then the step's 16 sem_in rows as they lie in memory
constexpr int kLoadsG = 3, kLoadsX = 3;                  // loads per fetch of a g-kind / x-kind wave


// the operand set of one 16-point step, as one wave holds it
struct SetG {            // waves 0..3: lane = (point 4 w + (l >> 4), features 8 (l & 15) .. + 7)
    float wt;            // compositing weight of the point
    f32x2 g;             // dL/dsemantics of its ray
    u32x4 h;             // its 8 hidden activations, 16-bit words as stored
};
struct SetX {            // waves 4..7: 3 x 16 B of the wave's four sem_in rows (160 chunks of 16 B over 64 lanes)
    u32x4 x[3];
};

__device__ __forceinline__ u32x4 ld_u32x4(unsigned voff, unsigned long long base) {
    u32x4 v;
    asm volatile("global_load_dwordx4 %0, %1, %2" : "=v"(v) : "v"(voff), "s"(base));
    return v;
}
template <int OFF>
__device__ __forceinline__ float ld_f32(unsigned voff, unsigned long long base) {
    float v;
    asm volatile("global_load_dword %0, %1, %2 offset:%3" : "=v"(v) : "v"(voff), "s"(base), "i"(OFF));
    return v;
}
template <int OFF>
__device__ __forceinline__ unsigned ld_u16(unsigned voff, unsigned long long base) {
    unsigned v;
    asm volatile("global_load_ushort %0, %1, %2 offset:%3" : "=v"(v) : "v"(voff), "s"(base), "i"(OFF));
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
    unsigned long long u = ((unsigned long long)(unsigned)__builtin_amdgcn_readfirstlane((unsigned)(b >> 32)) << 32) |
                           (unsigned long long)(unsigned)__builtin_amdgcn_readfirstlane((unsigned)b);
    // A v_readfirstlane result read by a vector-memory instruction as its scalar base needs five wait states, and hipcc's
    // hazard recognizer does not look at the SGPR operands of inline asm: when the address arithmetic happens to run on the
    // VALU, the readfirstlane lands directly in front of the asm load, which then reads the stale pair (memory fault).
    asm volatile("s_nop 4" : "+s"(u));
    return u;
}

// s_waitcnt vmcnt(N) that owns the registers it releases: nothing that reads them can be scheduled above it.
template <int N>
__device__ __forceinline__ void wait_set(SetG& s) {
    asm volatile("s_waitcnt vmcnt(%3)" : "+v"(s.wt), "+v"(s.g), "+v"(s.h) : "i"(N));
}
template <int N>
__device__ __forceinline__ void wait_set(SetX& s) {
    asm volatile("s_waitcnt vmcnt(%3)" : "+v"(s.x[0]), "+v"(s.x[1]), "+v"(s.x[2]) : "i"(N));
}

template <int XFMT>
__device__ __forceinline__ f32x16 mfma_fmt(u32x4 a, u32x4 b, f32x16 c) {
    typedef __bf16 bf16x8_t __attribute__((ext_vector_type(8)));
    if constexpr (XFMT == 1) return mfma16(a, b, c);
    else return __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8_t, a), __builtin_bit_cast(bf16x8_t, b), c, 0, 0, 0);
}
// (hi, lo) packed words of the fp32 pair (v0, v1) in format XFMT: hi = round(v), lo = round(v - hi)
template <int XFMT>
__device__ __forceinline__ void split_pair(float v0, float v1, unsigned& hi, unsigned& lo) {
    if constexpr (XFMT == 1) split2(v0, v1, hi, lo);
    else {
        asm("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(hi) : "v"(v0), "v"(v1));
        const float r0 = v0 - __builtin_bit_cast(float, hi << 16), r1 = v1 - __builtin_bit_cast(float, hi & 0xffff0000u);
        asm("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(lo) : "v"(r0), "v"(r1));
    }
}

// 8 consecutive 16-bit values as fp16 words: fp16 as they are; bf16 widened and re-rounded (exact from 2^-14 up: 8 significant
// bits into 11; below that the error is < 2^-25 absolute)
template <int XFMT>
__device__ __forceinline__ u32x4 to_f16(u32x4 v) {
    if constexpr (XFMT == 1) return v;
    u32x4 h;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        unsigned w;
        asm("v_cvt_pk_f16_f32 %0, %1, %2" : "=v"(w) : "v"(v[q] << 16), "v"(v[q] & 0xffff0000u));
        h[q] = w;
    }
    return h;
}

// the transposing LDS read: lane (r = (l & 15) >> 2, q = l & 3) of a 16-lane group passes the address of row r, columns 4q..4q+3
// of a [4][16] block of 16-bit elements; lane c = l & 15 receives the block's column c, rows 0..3
template <int OFF>
__device__ __forceinline__ u32x2 lds_read_tr(unsigned addr) {
    u32x2 v;
    asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(v) : "v"(addr), "i"(OFF) : "memory");   // stays behind the barrier
    return v;
}

template <int XFMT, bool TILED, bool HTILED>   // XFMT 1: fp16, 2: bf16; TILED / HTILED: sem_in / hid in the tile-major layouts of nsos_mlp_forward_rays_save16_lp
__global__ __launch_bounds__(512, 1) void sem_head_wgrad16_kernel(const float* __restrict__ weights, const float* __restrict__ g_sem,
                                                                  const float* __restrict__ w2, const unsigned short* __restrict__ hid,
                                                                  const unsigned short* __restrict__ sem_in,
                                                                  const float* __restrict__ scale_p, long long n_pts,
                                                                  long long n_rays, int S, float* __restrict__ partial,
                                                                  float* __restrict__ scale_out) {
    __shared__ __attribute__((aligned(16))) unsigned char lds[3 * kBufBytes];
    const int lane = threadIdx.x & 63, i = lane & 31, kg = lane >> 5;
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const int gt = wave & 3, ch = wave >> 2;
    const bool kind_g = wave < 4;
    const int xw = wave - 4;                       // x-kind: this wave moves rows 4 xw .. 4 xw + 3 of every step

    const long long n_full = n_pts / 16;
    const long long per = (n_full + gridDim.x - 1) / gridDim.x;
    const long long s0 = (long long)blockIdx.x * per;
    const long long s1 = s0 + per < n_full ? s0 + per : n_full;
    const int nf = __builtin_amdgcn_readfirstlane((int)(s1 > s0 ? s1 - s0 : 0));
    // The power of two that brings g_hid into the 16-bit range: handed in, or (scale_p == NULL) derived here by EVERY workgroup
    // from the same 2 n_rays + 256 values -- what sem_head_scale_kernel computes (backward.hip), without its launch: a maximum
    // does not depend on the order, so all workgroups agree bit for bit; workgroup 0 publishes it for the reduction kernel.
    float scale;
    if (scale_p) scale = *scale_p;
    else {
        float* const red = reinterpret_cast<float*>(lds);
        float m = 0.0f;
        // (independent 16-byte loads, four in flight per thread: a scalar loop was sixteen dependent L2 round trips = 6 us per call)
        const long long n_g = 2 * n_rays, n_q = (((unsigned long long)g_sem & 15) == 0) ? n_g >> 2 : 0;
        const f32x4* const g4 = reinterpret_cast<const f32x4*>(g_sem);
        for (long long j = threadIdx.x; j < n_q; j += 4 * 512) {
            f32x4 v[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) v[u] = g4[j + 512 * u < n_q ? j + 512 * u : j];
#pragma unroll
            for (int u = 0; u < 4; ++u) m = fmaxf(fmaxf(m, fmaxf(fabsf(v[u][0]), fabsf(v[u][1]))), fmaxf(fabsf(v[u][2]), fabsf(v[u][3])));
        }
        for (long long j = 4 * n_q + threadIdx.x; j < n_g; j += 512) m = fmaxf(m, fabsf(g_sem[j]));
        float c = threadIdx.x < 128 ? fabsf(w2[threadIdx.x]) + fabsf(w2[128 + threadIdx.x]) : 0.0f;
#pragma unroll
        for (int off = 32; off >= 1; off >>= 1) {
            m = fmaxf(m, __shfl_xor(m, off, 64));
            c = fmaxf(c, __shfl_xor(c, off, 64));
        }
        if (lane == 0) { red[wave] = m; red[8 + wave] = c; }
        __syncthreads();
#pragma unroll
        for (int k = 0; k < 8; ++k) { m = fmaxf(m, red[k]); c = fmaxf(c, red[8 + k]); }
        const float bound = fmaxf(m * c, 1e-30f);
        scale = exp2f(fminf(fmaxf(floorf(log2f(256.0f / bound)), -60.0f), 60.0f));
        if (blockIdx.x == 0 && threadIdx.x == 0) *scale_out = scale;
        __syncthreads();                               // the staging images reuse this LDS
    }
    // g-kind lane: point pl of the step, features 8 fo .. 8 fo + 7.  Row-major hid: a wave takes 4 points x all 16 octets (four
    // 256-byte rows); tile-major hid ([group][octet][point][8]): 16 points x 4 octets (four 256-byte runs)
    const int pl = HTILED ? (lane & 15) : 4 * gt + (lane >> 4), fo = HTILED ? 4 * gt + (lane >> 4) : (lane & 15);
    f32x2 w2a[4], w2b[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) {                                                    // scale is a power of two: exact
        w2a[q] = f32x2{w2[8 * fo + 2 * q] * scale, w2[8 * fo + 2 * q + 1] * scale};
        w2b[q] = f32x2{w2[128 + 8 * fo + 2 * q] * scale, w2[128 + 8 * fo + 2 * q + 1] * scale};
    }

    f32x16 acc[5];
#pragma unroll
    for (int t = 0; t < 5; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[t][r] = 0.0f;
    f32x2 gw2[2][4];
    float gb2[2] = {0.0f, 0.0f};
#pragma unroll
    for (int q = 0; q < 4; ++q) gw2[0][q] = gw2[1][q] = f32x2{0.0f, 0.0f};

    const unsigned off_w = 4u * (unsigned)pl, off_h = HTILED ? (unsigned)fo * 512u + (unsigned)pl * 16u : ((unsigned)pl * 128u + 8u * (unsigned)fo) * 2u;   // hid: 16-bit like sem_in
    auto widen = [](unsigned w) { return XFMT == 1 ? (float)__builtin_bit_cast(_Float16, (unsigned short)w) : __builtin_bit_cast(float, w << 16); };
    const unsigned long long g_base = uniform64(g_sem);
    // the ray of the lane's point, kept by increments: point s0 * 16 + pl now, + 16 per fetched step (n_samples >= 8: two wraps at most)
    unsigned ray_q = (unsigned)((unsigned long long)(s0 * 16 + pl) / (unsigned)S), ray_r = (unsigned)((unsigned long long)(s0 * 16 + pl) % (unsigned)S);
    if (nf == 0) ray_q = 0;
    // x-kind: chunk c = lane + 64 j (j = 0..2) of the wave's 4 rows x 40 chunks of 16 B; chunks past 159 repeat chunk 159 (the
    // load is issued by every lane so that the vmcnt arithmetic holds; the duplicate is not written to LDS)
    unsigned xg_off[3], xl_off[3];
    bool x_own[3];
#pragma unroll
    for (int j = 0; j < 3; ++j) {
        const int c_raw = lane + 64 * j, c = c_raw < 160 ? c_raw : 159;
        // row-major: the wave moves rows 4 xw .. 4 xw + 3 (40 pieces each).  Tile-major: the wave moves stores K = 5 xw .. 5 xw + 4
        // of the step's half group: piece c = (K - 5 xw, kg, point): 256-byte runs of 16 points; octet 2 K + kg of its point.
        const int row = TILED ? (c & 15) : 4 * xw + c / 40, col = TILED ? 2 * (5 * xw + (c >> 5)) + ((c >> 4) & 1) : c % 40;
        x_own[j] = c_raw < 160;
        xg_off[j] = TILED ? (unsigned)((5 * xw + (c >> 5)) * 1024 + (((c >> 4) & 1) * 32 + (c & 15)) * 16) : (unsigned)(row * 640 + col * 16);
        xl_off[j] = (unsigned)(kGBytes + row * kRowBytes + col * 16);
        asm volatile("" : "+v"(xg_off[j]), "+v"(xl_off[j]));     // per-lane constants: keep them in registers (no re-derivation per fetch)
    }
    // operand fetch of column tile T: lane (r = (l & 15) >> 2, q = l & 3) of 16-lane group g = l >> 4 passes the address of point
    // 8 (g >> 1) + r, channels 32 T + 16 (g & 1) + 4 q .. + 3; a second read four rows further gives points + 4 .. + 7.
    // The g_hid operand (feature tile gt) is fetched the same way from its own rows.
    const unsigned tr_off = (unsigned)(kGBytes + (8 * (lane >> 5) + ((lane & 15) >> 2)) * kRowBytes + (16 * ((lane >> 4) & 1) + 4 * (lane & 3)) * 2);
    const unsigned tr_off_g = (unsigned)((8 * (lane >> 5) + ((lane & 15) >> 2)) * kGRow + (32 * gt + 16 * ((lane >> 4) & 1) + 4 * (lane & 3)) * 2);
    const unsigned lds_base = (unsigned)(size_t)(__attribute__((address_space(3))) unsigned char*)lds;

    // fetch of step `step` for a lane whose point lies on ray ray_q; `advance`: move the ray bookkeeping on to the next step
    // The three input streams advance by one 16-point step per fetch: their wave-uniform base addresses live in SGPRs and move
    // on the scalar unit (`advance` is false once the last step is reached: it is fetched again, never staged).  (A base
    // re-derived per fetch from the step index was two v_readfirstlane + s_nop 4 + 64-bit multiplies: 316 of the 2 000 cycles
    // a step took on the g-kind waves.)
    // (tile-major sem_in: step s is half (s & 1) of group s >> 1: + 256 B into the group's second half, then on to the next group)
    unsigned long long wb_run = uniform64(weights + s0 * 16), hb_run = uniform64(HTILED ? hid + (s0 >> 1) * (32 * 128) + (s0 & 1) * (16 * 8) : hid + s0 * 16 * 128),
                       xb_run = uniform64(TILED ? sem_in + (s0 >> 1) * (32 * 320) + (s0 & 1) * (16 * 8) : sem_in + s0 * 16 * 320);
    int x_half = __builtin_amdgcn_readfirstlane((int)(s0 & 1));
    int h_half = x_half;
    auto fetch_g = [&](bool advance, SetG& s) {
        s.wt = ld_f32<0>(off_w, wb_run);
        s.g = ld_f32x2(ray_q * 8u, g_base);
        s.h = ld_u32x4(off_h, hb_run);
        if (advance) {
            wb_run += 16 * 4;
            if constexpr (HTILED) {       // step s is half (s & 1) of group s >> 1: + 256 B into the second half, then on to the next group
                unsigned inc = h_half ? 32u * 128u * 2u - 256u : 256u;
                asm volatile("" : "+s"(inc), "+s"(h_half));
                hb_run += inc;
                h_half ^= 1;
                asm volatile("" : "+s"(hb_run));
            }
            else hb_run += 16 * 128 * 2;
            ray_r += 16u;
            if (ray_r >= (unsigned)S) { ray_r -= (unsigned)S; ++ray_q; }
            if (ray_r >= (unsigned)S) { ray_r -= (unsigned)S; ++ray_q; }
        }
    };
    auto fetch_x = [&](bool advance, SetX& s) {
        s.x[0] = ld_u32x4(xg_off[0], xb_run);
        s.x[1] = ld_u32x4(xg_off[1], xb_run);
        s.x[2] = ld_u32x4(xg_off[2], xb_run);
        if (advance) {
            if constexpr (TILED) {
                unsigned inc = x_half ? 32u * 320u * 2u - 256u : 256u;
                asm volatile("" : "+s"(inc), "+s"(x_half));        // wave-uniform: keep the stream's base on the scalar unit
                xb_run += inc;
                x_half ^= 1;
                asm volatile("" : "+s"(xb_run));
            }
            else xb_run += 16 * 320 * 2;
        }
    };
    // g_hid of the lane's point for its 8 features, from the compositing weight, the ray's dL/dsemantics and the hidden
    // activations; split words to the step's row-major image
    auto stage_g = [&](const SetG& s, int buf) {
        const float gl0 = s.wt * s.g[0], gl1 = s.wt * s.g[1];                        // g_logits (models/renderer.py:64-66)
        gb2[0] += gl0;
        gb2[1] += gl1;
        const f32x2 gl0v = {gl0, gl0}, gl1v = {gl1, gl1};
        u32x4 h, l;
#pragma unroll
        for (int q = 0; q < 4; ++q) {                                                // features 8 fo + 2 q, + 1: packed fp32 math
            const unsigned w = s.h[q];
            const f32x2 hv = {widen(w), XFMT == 1 ? widen(w >> 16) : __builtin_bit_cast(float, w & 0xffff0000u)};
            f32x2 a = __builtin_elementwise_fma(gl1v, w2b[q], gl0v * w2a[q]);        // g_hid x scale (models/nerf_mlp.py:61)
            a[0] = (w & 0xffffu) ? a[0] : 0.0f;                                      // hid is stored after its ReLU: > 0 <=> bits != 0
            a[1] = (w >> 16) ? a[1] : 0.0f;
            gw2[0][q] = __builtin_elementwise_fma(gl0v, hv, gw2[0][q]);
            gw2[1][q] = __builtin_elementwise_fma(gl1v, hv, gw2[1][q]);
            unsigned x, y;
            split_pair<XFMT>(a[0], a[1], x, y);
            h[q] = x; l[q] = y;
        }
        unsigned char* row = lds + buf * kBufBytes + pl * kGRow + fo * 16;
        *reinterpret_cast<u32x4*>(row) = h;
        *reinterpret_cast<u32x4*>(row + 16 * kGRow) = l;
    };
    auto stage_x = [&](const SetX& s, int buf) {
#pragma unroll
        for (int j = 0; j < 3; ++j)
            if (x_own[j]) *reinterpret_cast<u32x4*>(lds + buf * kBufBytes + xl_off[j]) = s.x[j];     // as stored: the MFMA is sem_in's format's
    };
    // The MFMA operands of one step as a wave holds them: 14 transposing reads, issued one step ahead of their MFMAs (the image
    // of step s + 1 is complete at the barrier that opens step s: three LDS images rotate) and released by ONE wait at the end
    // of the step, behind the MFMAs and the staging they overlap with.
    struct Ops { u32x2 lo[5], hi[5], gq[4]; };
    auto read_ops = [&](int buf, Ops& o) {
        const unsigned tg = lds_base + (unsigned)(buf * kBufBytes) + tr_off_g;
        const unsigned ta = lds_base + (unsigned)(buf * kBufBytes) + tr_off + (unsigned)(5 * ch * 64);
        o.gq[0] = lds_read_tr<0>(tg); o.gq[1] = lds_read_tr<4 * kGRow>(tg);                  // hi words: points + 0..3, + 4..7
        o.gq[2] = lds_read_tr<16 * kGRow>(tg); o.gq[3] = lds_read_tr<20 * kGRow>(tg);        // lo words
        o.lo[0] = lds_read_tr<0 * 64>(ta); o.hi[0] = lds_read_tr<0 * 64 + 4 * kRowBytes>(ta);
        o.lo[1] = lds_read_tr<1 * 64>(ta); o.hi[1] = lds_read_tr<1 * 64 + 4 * kRowBytes>(ta);
        o.lo[2] = lds_read_tr<2 * 64>(ta); o.hi[2] = lds_read_tr<2 * 64 + 4 * kRowBytes>(ta);
        o.lo[3] = lds_read_tr<3 * 64>(ta); o.hi[3] = lds_read_tr<3 * 64 + 4 * kRowBytes>(ta);
        o.lo[4] = lds_read_tr<4 * 64>(ta); o.hi[4] = lds_read_tr<4 * 64 + 4 * kRowBytes>(ta);
    };
    // the reads are asm: hipcc does not count them.  One wait that owns their registers.
    auto wait_ops = [&](Ops& o) {
        asm volatile("s_waitcnt lgkmcnt(0)"
                     : "+v"(o.lo[0]), "+v"(o.hi[0]), "+v"(o.lo[1]), "+v"(o.hi[1]), "+v"(o.lo[2]), "+v"(o.hi[2]), "+v"(o.lo[3]), "+v"(o.hi[3]),
                       "+v"(o.lo[4]), "+v"(o.hi[4]), "+v"(o.gq[0]), "+v"(o.gq[1]), "+v"(o.gq[2]), "+v"(o.gq[3]));
    };
    auto mfmas = [&](const Ops& o) {
        const u32x4 ah = u32x4{o.gq[0][0], o.gq[0][1], o.gq[1][0], o.gq[1][1]}, al = u32x4{o.gq[2][0], o.gq[2][1], o.gq[3][0], o.gq[3][1]};
        u32x4 b[5];
#pragma unroll
        for (int c = 0; c < 5; ++c) b[c] = u32x4{o.lo[c][0], o.lo[c][1], o.hi[c][0], o.hi[c][1]};
#pragma unroll
        for (int c = 0; c < 5; ++c) acc[c] = mfma_fmt<XFMT>(ah, b[c], acc[c]);
#pragma unroll
        for (int c = 0; c < 5; ++c) acc[c] = mfma_fmt<XFMT>(al, b[c], acc[c]);
    };
    auto compute = [&](int buf) { Ops o; read_ops(buf, o); wait_ops(o); mfmas(o); };

#ifdef NSOS_WG_PROF   // (the stamps do not wait for anything: s_memtime returns in order with the other scalar-memory results only)
    unsigned long long pt[8] = {0, 0, 0, 0, 0, 0, 0, 0}, pacc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
#define PROF_T(k) do { pt[k] = __builtin_amdgcn_s_memtime(); } while (0)
#define PROF_ACC() do { for (int k_ = 1; k_ < 8; ++k_) pacc[k_] += (pt[k_] >= pt[k_ - 1] ? pt[k_] - pt[k_ - 1] : 0); pacc[0] += 1; } while (0)
#else
#define PROF_T(k)
#define PROF_ACC()
#endif
    if (nf > 0) {
        // Step s: request the operands of step s + 1 from image (s + 1) % 3, stage the fetched set J that holds step s + 2 into
        // image (s + 2) % 3 (sets rotate C, A, B with the step), refill it with step s + 5, multiply step s from the operands
        // requested a step ago.  Two fetches have been issued since J's own: vmcnt(2 x loads per fetch) releases it.  The g-kind
        // wave of a SIMD stages before its MFMAs, the x-kind wave after: one feeds the matrix pipe while the other is on the VALU.
        Ops O0, O1;
        if (kind_g) {
            SetG A, B, C;
            auto fetch = [&](int j, SetG& J) { fetch_g(j + 1 < nf, J); };            // consecutive steps; past the end the last one again
            fetch(0, A); fetch(1, B); fetch(2, C);
            wait_set<2 * kLoadsG>(A);
            stage_g(A, 0);
            fetch(3, A);
            wait_set<2 * kLoadsG>(B);
            if (nf > 1) stage_g(B, 1);
            fetch(4, B);
            __syncthreads();
            read_ops(0, O0);
            wait_ops(O0);
            auto step = [&](int s, int b1, int b2, SetG& J, Ops& cur, Ops& nxt) {
                PROF_T(0);
                if (s + 1 < nf) read_ops(b1, nxt);
                PROF_T(1);
                if (s + 2 < nf) {
                    wait_set<2 * kLoadsG>(J);
                    PROF_T(2);
                    stage_g(J, b2);
                }
                PROF_T(3);
                fetch(s + 5, J);
                PROF_T(4);
                mfmas(cur);
                PROF_T(5);
                wait_ops(nxt);
                PROF_T(6);
                __syncthreads();
                PROF_T(7);
                PROF_ACC();
            };
            for (int s = 0; s < nf; s += 6) {
                step(s, 1, 2, C, O0, O1);
                if (s + 1 < nf) step(s + 1, 2, 0, A, O1, O0);
                if (s + 2 < nf) step(s + 2, 0, 1, B, O0, O1);
                if (s + 3 < nf) step(s + 3, 1, 2, C, O1, O0);
                if (s + 4 < nf) step(s + 4, 2, 0, A, O0, O1);
                if (s + 5 < nf) step(s + 5, 0, 1, B, O1, O0);
            }
        } else {
            SetX A, B, C;
            auto fetch = [&](int j, SetX& J) { fetch_x(j + 1 < nf, J); };
            fetch(0, A); fetch(1, B); fetch(2, C);
            wait_set<2 * kLoadsX>(A);
            stage_x(A, 0);
            fetch(3, A);
            wait_set<2 * kLoadsX>(B);
            if (nf > 1) stage_x(B, 1);
            fetch(4, B);
            __syncthreads();
            read_ops(0, O0);
            wait_ops(O0);
            auto step = [&](int s, int b1, int b2, SetX& J, Ops& cur, Ops& nxt) {
                PROF_T(0);
                if (s + 1 < nf) read_ops(b1, nxt);
                PROF_T(1);
                mfmas(cur);
                PROF_T(2);
                if (s + 2 < nf) {
                    wait_set<2 * kLoadsX>(J);
                    PROF_T(3);
                    stage_x(J, b2);
                }
                PROF_T(4);
                fetch(s + 5, J);
                PROF_T(5);
                wait_ops(nxt);
                PROF_T(6);
                __syncthreads();
                PROF_T(7);
                PROF_ACC();
            };
            for (int s = 0; s < nf; s += 6) {
                step(s, 1, 2, C, O0, O1);
                if (s + 1 < nf) step(s + 1, 2, 0, A, O1, O0);
                if (s + 2 < nf) step(s + 2, 0, 1, B, O0, O1);
                if (s + 3 < nf) step(s + 3, 1, 2, C, O1, O0);
                if (s + 4 < nf) step(s + 4, 2, 0, A, O0, O1);
                if (s + 5 < nf) step(s + 5, 0, 1, B, O1, O0);
            }
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");     // the re-fetched sets are still landing in registers
    }
#ifdef NSOS_WG_PROF
    if (blockIdx.x == 1 && lane == 0)
        for (int k = 0; k < 8; ++k) nsos_wg_prof[wave][k] = pacc[k];
#endif
    // the ragged step (n_pts % 16 points), by workgroup 0: rows clamped, out-of-range points get weight 0 (their g_hid is 0)
    if (blockIdx.x == 0 && (n_pts & 15)) {
        if (kind_g) {
            const unsigned p = (unsigned)(n_full * 16) + (unsigned)pl;
            const bool valid = p < (unsigned)n_pts;
            const unsigned pc = valid ? p : (unsigned)n_pts - 1u, r = pc / (unsigned)S;
            SetG t;
            t.wt = valid ? weights[pc] : 0.0f;
            t.g = f32x2{g_sem[2ull * r], g_sem[2ull * r + 1]};
            t.h = *reinterpret_cast<const u32x4*>(HTILED ? hid + (((unsigned long long)(pc >> 5) * 16 + fo) * 32 + (pc & 31)) * 8 : hid + (unsigned long long)pc * 128 + 8 * fo);
            stage_g(t, 0);
        } else {
#pragma unroll
            for (int j = 0; j < 3; ++j) {
                const int c_raw = lane + 64 * j, c = c_raw < 160 ? c_raw : 159;
                const int row = TILED ? (c & 15) : 4 * xw + c / 40, col = TILED ? 2 * (5 * xw + (c >> 5)) + ((c >> 4) & 1) : c % 40;
                const unsigned long long p = (unsigned long long)(n_full * 16 + row) < (unsigned long long)n_pts ? (unsigned long long)(n_full * 16 + row)
                                                                                                                : (unsigned long long)n_pts - 1;
                const unsigned short* src = TILED ? sem_in + (((p >> 5) * 20 + (col >> 1)) * 64 + (col & 1) * 32 + (p & 31)) * 8 : sem_in + p * 320 + col * 8;
                const u32x4 v = *reinterpret_cast<const u32x4*>(src);
                if (x_own[j]) *reinterpret_cast<u32x4*>(lds + xl_off[j]) = v;
            }
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
    // dW2 / db2: the g-kind lanes' sums, folded over the 16 point slots in slot order through LDS
    __syncthreads();
    float* red = reinterpret_cast<float*>(lds);              // [16 slots][2][128] then [16][2]
    if (kind_g) {
#pragma unroll
        for (int o = 0; o < 2; ++o) {
#pragma unroll
            for (int k = 0; k < 8; ++k) red[(pl * 2 + o) * 128 + 8 * fo + k] = gw2[o][k >> 1][k & 1];
            if (fo == 0) red[16 * 256 + pl * 2 + o] = gb2[o];
        }
    }
    __syncthreads();
    if (threadIdx.x < 256) {
        float t = 0.0f;
        for (int k = 0; k < 16; ++k) t += red[k * 256 + threadIdx.x];
        out[128 * 320 + threadIdx.x] = t;
    } else if (threadIdx.x < 258) {
        float t = 0.0f;
        for (int k = 0; k < 16; ++k) t += red[16 * 256 + k * 2 + (threadIdx.x - 256)];
        out[128 * 320 + 256 + (threadIdx.x - 256)] = t;
    }
}
}  // namespace

int32_t nsos_detail::sem_head_wgrad16(const float* weights, const float* g_semantics, const float* sem2_w, const void* sem_hid,
                                      const void* sem_in, int32_t sem_in_dtype, int64_t n_rays, int32_t n_samples,
                                      const float* scale, float* partial, int blocks, hipStream_t st, float* scale_out) {
    const long long n_pts = (long long)n_rays * n_samples;
    const unsigned short* x = static_cast<const unsigned short*>(sem_in);
    const unsigned short* h = static_cast<const unsigned short*>(sem_hid);
    const bool tiled = (sem_in_dtype & NSOS_SEM_IN_TILED) != 0, htiled = (sem_in_dtype & NSOS_SEM_HID_TILED) != 0;
    const int fmt = sem_in_dtype & ~(NSOS_SEM_IN_TILED | NSOS_SEM_HID_TILED);
    if (htiled && !tiled) return NSOS_ERR_UNSUPPORTED;       // (the kernels store either sem_in alone or both matrices tile-major)
#define NSOS_WG16_LAUNCH(F, T, H) hipLaunchKernelGGL((sem_head_wgrad16_kernel<F, T, H>), dim3(blocks), dim3(512), 0, st, weights, g_semantics, sem2_w, h, x, \
                                                     scale, n_pts, (long long)n_rays, (int)n_samples, partial, scale_out)
    if (fmt == 1) { if (htiled) NSOS_WG16_LAUNCH(1, true, true); else if (tiled) NSOS_WG16_LAUNCH(1, true, false); else NSOS_WG16_LAUNCH(1, false, false); }
    else { if (htiled) NSOS_WG16_LAUNCH(2, true, true); else if (tiled) NSOS_WG16_LAUNCH(2, true, false); else NSOS_WG16_LAUNCH(2, false, false); }
#undef NSOS_WG16_LAUNCH
    return nsos_launch_status();
}

#ifdef NSOS_WG_PROF
extern "C" int32_t nsos_wg_prof_read(unsigned long long* out) {
    return (int32_t)hipMemcpyFromSymbol(out, HIP_SYMBOL(nsos_wg_prof), sizeof(unsigned long long) * 64);
}
#endif
