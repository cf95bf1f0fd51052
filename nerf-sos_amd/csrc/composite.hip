// K3: volumetric compositing for gfx950 -- VolumetricRenderer.forward (models/renderer.py:35-85).
//
// One 64-lane wave per ray.  Lane l owns the IPL = ceil(S/64) consecutive samples
// [l*IPL, (l+1)*IPL) (1 for the coarse pass, 3 for the 192-sample fine pass), so
//   * the exclusive transmittance product is: a lane-local running product, ONE wave-level
//     exclusive scan of the 64 lane products (6 shuffle steps), and a lane-local fix-up;
//   * the five weighted sums are lane-local partials + one butterfly reduction each.
// The product and the sums are carried in fp64 and rounded to fp32 once per output: that is what
// torch-CPU cumprod does and makes the result independent of the scan's association
// (an fp64 re-association error is ~1e-16, invisible after rounding to fp32).
// HBM-bound: reads 4*(C+1) B and writes 4 B per sample; every byte of a ray's row is consumed by
// its wave, so all fetched cache lines are fully used.
// Compiled with -ffp-contract=off: element-wise fp32 expressions match the reference's op order.
#include "common.h"
#include "importance_device.h"

typedef float f32x4v __attribute__((ext_vector_type(4)));

// exp rounded correctly to fp32 (fp64 evaluation, one rounding).  alpha = 1 - exp(-sigma*delta) cancels: for a thin
// sample the result lives on the 6e-8 grid of fl(exp) near 1, so an ulp of libm difference in exp moves a small alpha by
// 1e-3 of itself -- and with it the coarse weights, the cdf and the importance samples (tests/test_gpu_pins.py::
// test_free_running_render_at_c2_size).  The reference's CPU exp (SLEEF u10 / glibc) is correctly rounded in all but a
// few per cent / per mille of the cases; ocml's expf is a different <= 1 ulp function.  This kernel is HBM-bound: the
// fp64 exp costs nothing measurable.
__device__ __forceinline__ float exp_cr(float x) { return (float)exp((double)x); }

// One ray per wave: VolumetricRenderer.forward for ray r (models/renderer.py:35-85).  Returns the weight of the lane's FIRST
// sample (for IPL == 1 -- S <= 64 -- that is lane j's sample j: what the hierarchical sampler consumes next).
template <int IPL>
__device__ __forceinline__ float composite_ray(const int64_t r, const int lane, const float* __restrict__ raw,
                                               const float* __restrict__ z_vals, const float* __restrict__ rays_d,
                                               const float* __restrict__ noise, float noise_std, int S, int C, int white_bkgd,
                                               float* __restrict__ weights, float* __restrict__ rgb, float* __restrict__ sem,
                                               float* __restrict__ depth, float* __restrict__ acc, float* __restrict__ disp,
                                               float* z_first, const float* staged = nullptr) {
    // `staged`: this ray's raw row [S, C] in LDS (composite_kernel stages it with coalesced 16-byte loads when a lane owns
    // several samples: read per lane from global memory, every dword load gathered 64 values 4 C IPL bytes apart)
    float w_first = 0.0f;
    const float dx = rays_d[3 * r], dy = rays_d[3 * r + 1], dz = rays_d[3 * r + 2];
    const float norm = (float)sqrt((double)dx * dx + (double)dy * dy + (double)dz * dz);  // :38

    const float* zr = z_vals + r * S;
    const float* rr = raw + r * (int64_t)S * C;
    const int s0 = lane * IPL;

    float z[IPL + 1], alpha[IPL], col[IPL][3], smv[IPL][2];
#pragma unroll
    for (int i = 0; i <= IPL; ++i) z[i] = (s0 + i < S) ? zr[s0 + i] : 0.0f;

    double prod = 1.0;  // product of (1 - alpha + 1e-10) over this lane's samples
    double tloc[IPL];   // lane-local exclusive prefix
#pragma unroll
    for (int i = 0; i < IPL; ++i) {
        const int s = s0 + i;
        const bool live = s < S;
        float c[6];
        if (staged) {
            const float* lp = staged + (live ? s : 0) * C;
#pragma unroll
            for (int k = 0; k < 6; ++k) c[k] = k < C ? lp[k] : 0.0f;
        } else {
            const float* gp = rr + (int64_t)(live ? s : 0) * C;
            if (C == 6 || C == 4) {              // rows of 24 / 16 bytes: 8-byte aligned pairs (three / two loads instead of six / four)
                const float2* gp2 = reinterpret_cast<const float2*>(gp);
                const float2 a = gp2[0], b = gp2[1];
                c[0] = a.x; c[1] = a.y; c[2] = b.x; c[3] = b.y;
                c[4] = c[5] = 0.0f;
                if (C == 6) { const float2 e = gp2[2]; c[4] = e.x; c[5] = e.y; }
            } else {
#pragma unroll
                for (int k = 0; k < 6; ++k) c[k] = k < C ? gp[k] : 0.0f;
            }
        }
        float dist = (s + 1 < S) ? (z[i + 1] - z[i]) : 1e10f;  // :35-37
        dist = dist * norm;
        float sigma = c[3];
        if (noise) sigma = sigma + noise[r * S + (live ? s : 0)] * noise_std;  // :46-50
        const float relu = sigma > 0.0f ? sigma : 0.0f;
        const float a = live ? (1.0f - exp_cr(-relu * dist)) : 0.0f;  // :52
        alpha[i] = a;
#pragma unroll
        for (int k = 0; k < 3; ++k) col[i][k] = 1.0f / (1.0f + expf(-c[k]));  // sigmoid :41
        smv[i][0] = C > 4 ? c[4] : 0.0f;
        smv[i][1] = C > 5 ? c[5] : 0.0f;
        tloc[i] = prod;
        if (live) prod *= (double)((1.0f - a) + 1e-10f);  // :57
    }
    // exclusive scan of the lane products across the wave (DPP: no LDS round trips, common.h)
    const double incl = nsos_wave_scan_incl<true>(prod);
    const double excl = nsos_wave_shr1(incl, 1.0);

    double s_rgb[3] = {0, 0, 0}, s_sem[2] = {0, 0}, s_depth = 0, s_acc = 0;
#pragma unroll
    for (int i = 0; i < IPL; ++i) {
        const int s = s0 + i;
        if (s < S) {
            const float T = (float)(excl * tloc[i]);  // :58 (fp64 running product, rounded per element)
            const float w = alpha[i] * T;             // :61
            weights[r * S + s] = w;
            if (i == 0) w_first = w;
#pragma unroll
            for (int k = 0; k < 3; ++k) s_rgb[k] += (double)(w * col[i][k]);  // :62
            s_sem[0] += (double)(w * smv[i][0]);                              // :64-66
            s_sem[1] += (double)(w * smv[i][1]);
            s_depth += (double)(w * z[i]);  // :69
            s_acc += (double)w;             // :71
        }
    }
#pragma unroll
    for (int k = 0; k < 3; ++k) s_rgb[k] = nsos_wave_sum(s_rgb[k]);
    if (C > 4) {
        s_sem[0] = nsos_wave_sum(s_sem[0]);
        s_sem[1] = nsos_wave_sum(s_sem[1]);
    }
    s_depth = nsos_wave_sum(s_depth);
    s_acc = nsos_wave_sum(s_acc);

    if (lane == 0) {
        const float a = (float)s_acc;
        float dep = (float)s_depth;
        if (a <= 1e-10f) dep = 1e10f;  // :72
        const float q = dep / a;
        disp[r] = 1.0f / (q > 1e-10f ? q : (q != q ? q : 1e-10f));  // :74 (torch.max propagates NaN)
        depth[r] = dep;
        acc[r] = a;
        const float bg = white_bkgd ? (1.0f - a) : 0.0f;  // :77-81
#pragma unroll
        for (int k = 0; k < 3; ++k) rgb[3 * r + k] = (float)s_rgb[k] + bg;
        if (C > 4) {
            sem[(C - 4) * r] = (float)s_sem[0] + bg;
            if (C > 5) sem[(C - 4) * r + 1] = (float)s_sem[1] + bg;
        }
    }
    if (z_first) *z_first = z[0];
    return w_first;
}

template <int IPL>
__global__ __launch_bounds__(256) void composite_kernel(const float* __restrict__ raw,
                                                        const float* __restrict__ z_vals,
                                                        const float* __restrict__ rays_d,
                                                        const float* __restrict__ noise, float noise_std,
                                                        int64_t n_rays, int S, int C, int white_bkgd,
                                                        float* __restrict__ weights, float* __restrict__ rgb,
                                                        float* __restrict__ sem, float* __restrict__ depth,
                                                        float* __restrict__ acc, float* __restrict__ disp) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int64_t r = (int64_t)blockIdx.x * (blockDim.x >> 6) + wave;
    if (r >= n_rays) return;   // whole wave exits together; only wave-level synchronisation below
    if constexpr (IPL >= 2 && IPL <= 4) {
        // the ray's raw row (S x C floats, contiguous, 16-byte aligned when S C is a multiple of 4) through LDS: 64 lanes x 16 B
        // per load instruction instead of 64 strided dwords
        __shared__ __attribute__((aligned(16))) float stage[4][IPL * 64 * 6];
        const int n = S * C;
        if ((n & 3) == 0) {
            const f32x4v* src = reinterpret_cast<const f32x4v*>(raw + r * (int64_t)n);
            f32x4v* dst = reinterpret_cast<f32x4v*>(stage[wave]);
            for (int i = lane; i < (n >> 2); i += 64) dst[i] = src[i];
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
            composite_ray<IPL>(r, lane, raw, z_vals, rays_d, noise, noise_std, S, C, white_bkgd, weights, rgb, sem, depth, acc, disp, nullptr,
                               stage[wave]);
            return;
        }
    }
    composite_ray<IPL>(r, lane, raw, z_vals, rays_d, noise, noise_std, S, C, white_bkgd, weights, rgb, sem, depth, acc, disp, nullptr);
}

// Coarse compositing + hierarchical resampling of the same ray in ONE launch (models/nerf_net.py:98-113: renderer, then
// importance_sampler on ret['weights']): the coarse weights go from the compositing registers straight into the pdf -- one
// launch and one round trip of the weights less per step.  Same device code as the two separate kernels: bit-identical outputs.
__global__ __launch_bounds__(256) void composite_importance_kernel(const float* __restrict__ raw, const float* __restrict__ z_vals,
                                                                   const float* __restrict__ rays_d, const float* __restrict__ noise,
                                                                   float noise_std, int64_t n_rays, int S, int C, int white_bkgd,
                                                                   float* __restrict__ weights, float* __restrict__ rgb,
                                                                   float* __restrict__ sem, float* __restrict__ depth,
                                                                   float* __restrict__ acc, float* __restrict__ disp,
                                                                   const float* __restrict__ u_in, int N, float* __restrict__ z_fine,
                                                                   float* __restrict__ z_samples, float* __restrict__ z_std) {
    __shared__ ImportanceLds lds_all[4];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int64_t r = (int64_t)blockIdx.x * 4 + wave;
    if (r >= n_rays) return;  // whole wave exits together; no block-level barrier below
    float z = 0.0f;
    const float w = composite_ray<1>(r, lane, raw, z_vals, rays_d, noise, noise_std, S, C, white_bkgd, weights, rgb, sem, depth, acc,
                                     disp, &z);
    importance_ray(lds_all[wave], r, lane, lane < S ? z : 0.0f, lane < S ? w : 0.0f, u_in, nullptr, S, N, z_fine, z_samples, z_std,
                   nullptr, nullptr);
}

extern "C" int32_t nsos_composite(const float* raw, const float* z_vals, const float* rays_d, const float* noise,
                                  float noise_std, int64_t n_rays, int32_t n_samples, int32_t n_ch,
                                  int32_t white_bkgd, float* weights, float* rgb, float* sem, float* depth,
                                  float* acc, float* disp, void* stream) {
    if (n_rays == 0) return NSOS_OK;  // empty batch: nothing to launch (empty tensors have NULL data pointers)
    NSOS_REQUIRE(raw && z_vals && rays_d && weights && rgb && depth && acc && disp, NSOS_ERR_NULL_POINTER);
    NSOS_REQUIRE(n_ch == 4 || n_ch == 5 || n_ch == 6, NSOS_ERR_UNSUPPORTED);
    NSOS_REQUIRE(n_ch == 4 || sem, NSOS_ERR_NULL_POINTER);
    NSOS_REQUIRE(n_rays >= 0 && n_samples >= 1, NSOS_ERR_BAD_SHAPE);
    NSOS_REQUIRE(n_samples <= 512, NSOS_ERR_UNSUPPORTED);
    if (n_rays == 0) return NSOS_OK;
    NSOS_REQUIRE((n_rays + 3) / 4 < (int64_t)1 << 31, NSOS_ERR_UNSUPPORTED);
    const dim3 grid((unsigned)((n_rays + 3) / 4)), block(256);
    const int ipl = (n_samples + 63) / 64;
#define NSOS_LAUNCH_COMPOSITE(I)                                                                              \
    hipLaunchKernelGGL(composite_kernel<I>, grid, block, 0, (hipStream_t)stream, raw, z_vals, rays_d, noise, \
                       noise_std, n_rays, n_samples, n_ch, white_bkgd, weights, rgb, sem, depth, acc, disp)
    switch (ipl) {
        case 1: NSOS_LAUNCH_COMPOSITE(1); break;
        case 2: NSOS_LAUNCH_COMPOSITE(2); break;
        case 3: NSOS_LAUNCH_COMPOSITE(3); break;
        case 4: NSOS_LAUNCH_COMPOSITE(4); break;
        default: NSOS_LAUNCH_COMPOSITE(8); break;
    }
#undef NSOS_LAUNCH_COMPOSITE
    return nsos_launch_status();
}

// ------------------------------------------------------------------------------------------------------------------
// Backward of the compositing (autograd of models/renderer.py:35-85 w.r.t. `raw`): one wave per ray, same sample
// ownership as the forward kernel.  alpha / T / weights are recomputed from `raw` (cheaper than storing them), then
//   gw_j     = dL/dw_j = g_weights_j + g_rgb.sigmoid(c_j) + g_sem.s_j + G_depth z_j + G_acc
//   dL/da_i  = gw_i T_i - (1/t_i) sum_{j>i} gw_j w_j          (w_j = a_j prod_{i<j} t_i,  t_i = 1 - a_i + 1e-10)
//   g_sigma_i = dL/da_i * dist_i * exp(-relu(sigma_i) dist_i) * [sigma_i > 0]
//   g_c_i    = g_rgb * w_i * sig (1 - sig),   g_s_i = g_sem * w_i
// with G_depth / G_acc collecting the per-ray terms (white background, the depth -> 1e10 replacement for empty rays,
// disp = 1/max(1e-10, depth/acc)).  The suffix sum is a lane-local reverse pass + one wave-level scan, in fp64.
// z_vals, rays_d and the noise carry no gradient in the reference (samples are detached, models/sampler.py:159).
// Any of the upstream gradients may be NULL (= zero).  Rays with acc <= 1e-10 take no gradient through disp (the
// reference's autograd produces NaN there: -0 * inf).
// NS = 2: the shipped head (sem_dim <= 2, the arithmetic order the parity tests pin); NS = 8: any sem_dim the generic nets have
template <int IPL, int NS>
__global__ __launch_bounds__(256) void composite_backward_kernel(
    const float* __restrict__ raw, const float* __restrict__ z_vals, const float* __restrict__ rays_d,
    const float* __restrict__ noise, float noise_std, int64_t n_rays, int S, int C, int white_bkgd,
    const float* __restrict__ g_rgb, const float* __restrict__ g_sem, const float* __restrict__ g_depth,
    const float* __restrict__ g_acc, const float* __restrict__ g_disp, const float* __restrict__ g_weights,
    float* __restrict__ g_raw) {
    const int lane = threadIdx.x & 63;
    const int64_t r = (int64_t)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    if (r >= n_rays) return;
    const float dx = rays_d[3 * r], dy = rays_d[3 * r + 1], dz = rays_d[3 * r + 2];
    const float norm = (float)sqrt((double)dx * dx + (double)dy * dy + (double)dz * dz);
    const float* zr = z_vals + r * S;
    const float* rr = raw + r * (int64_t)S * C;
    const int s0 = lane * IPL;
    const int nsem = C - 4;

    float z[IPL + 1], alpha[IPL], tt[IPL], dexp[IPL], col[IPL][3], smv[IPL][NS];
#pragma unroll
    for (int i = 0; i <= IPL; ++i) z[i] = (s0 + i < S) ? zr[s0 + i] : 0.0f;
    double prod = 1.0, tloc[IPL];
#pragma unroll
    for (int i = 0; i < IPL; ++i) {
        const int s = s0 + i;
        const bool live = s < S;
        const float* c = rr + (int64_t)(live ? s : 0) * C;
        float dist = (s + 1 < S) ? (z[i + 1] - z[i]) : 1e10f;
        dist = dist * norm;
        float sigma = c[3];
        if (noise) sigma = sigma + noise[r * S + (live ? s : 0)] * noise_std;
        const float relu = sigma > 0.0f ? sigma : 0.0f;
        const float e = exp_cr(-relu * dist);
        const float a = live ? (1.0f - e) : 0.0f;
        alpha[i] = a;
        dexp[i] = (live && sigma > 0.0f) ? dist * e : 0.0f;   // d alpha / d sigma
#pragma unroll
        for (int k = 0; k < 3; ++k) col[i][k] = 1.0f / (1.0f + expf(-c[k]));
#pragma unroll
        for (int k = 0; k < NS; ++k) smv[i][k] = nsem > k ? c[4 + k] : 0.0f;
        tloc[i] = prod;
        tt[i] = (1.0f - a) + 1e-10f;
        if (live) prod *= (double)tt[i];
    }
    double incl = prod;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
        const double o = __shfl_up(incl, off, NSOS_WAVE);
        if (lane >= off) incl *= o;
    }
    double excl = __shfl_up(incl, 1, NSOS_WAVE);
    if (lane == 0) excl = 1.0;
    float T[IPL], w[IPL];
    double s_depth = 0, s_acc = 0;
#pragma unroll
    for (int i = 0; i < IPL; ++i) {
        T[i] = (float)(excl * tloc[i]);
        w[i] = (s0 + i < S) ? alpha[i] * T[i] : 0.0f;
        s_depth += (double)(w[i] * z[i]);
        s_acc += (double)w[i];
    }
    s_depth = nsos_wave_sum(s_depth);
    s_acc = nsos_wave_sum(s_acc);
    const float a_ray = (float)s_acc, dep = (float)s_depth;
    const bool empty = a_ray <= 1e-10f;   // depth replaced by 1e10: no gradient through the sum (:72)

    float grgb[3] = {0, 0, 0}, gsem[NS];
#pragma unroll
    for (int k = 0; k < NS; ++k) gsem[k] = 0.0f;
    if (g_rgb)
#pragma unroll
        for (int k = 0; k < 3; ++k) grgb[k] = g_rgb[3 * r + k];
    if (g_sem)
#pragma unroll
        for (int k = 0; k < NS; ++k) if (nsem > k) gsem[k] = g_sem[nsem * r + k];
    float G_acc = g_acc ? g_acc[r] : 0.0f;
    float G_dep = (g_depth && !empty) ? g_depth[r] : 0.0f;
    if (g_disp && !empty) {
        const float q = dep / a_ray;
        if (q > 1e-10f) {                           // disp = 1/q (:74)
            const float gq = -g_disp[r] / (q * q);
            G_dep += gq / a_ray;
            G_acc += -gq * dep / (a_ray * a_ray);
        }
    }
    if (white_bkgd) {                                                                // rgb, sem += 1 - acc (:77-81)
        float gs = gsem[0] + gsem[1];
#pragma unroll
        for (int k = 2; k < NS; ++k) gs += gsem[k];
        G_acc -= (grgb[0] + grgb[1] + grgb[2]) + gs;
    }

    float gw[IPL];
    double suf_loc[IPL], tail = 0.0;   // suf_loc[i] = sum over this lane's samples j > i of gw_j w_j
#pragma unroll
    for (int i = IPL - 1; i >= 0; --i) {
        const int s = s0 + i;
        float g = g_weights && s < S ? g_weights[r * S + s] : 0.0f;
#pragma unroll
        for (int k = 0; k < 3; ++k) g += grgb[k] * col[i][k];
        {
            float gs = gsem[0] * smv[i][0] + gsem[1] * smv[i][1];
#pragma unroll
            for (int k = 2; k < NS; ++k) gs += gsem[k] * smv[i][k];
            g += gs;
        }
        g += G_dep * z[i] + G_acc;
        gw[i] = s < S ? g : 0.0f;
        suf_loc[i] = tail;
        tail += (double)(gw[i] * w[i]);
    }
    // exclusive suffix scan of the lane totals (lanes above this one)
    double incl_s = tail;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
        const double o = __shfl_down(incl_s, off, NSOS_WAVE);
        if (lane + off < 64) incl_s += o;
    }
    double above = __shfl_down(incl_s, 1, NSOS_WAVE);
    if (lane == 63) above = 0.0;
#pragma unroll
    for (int i = 0; i < IPL; ++i) {
        const int s = s0 + i;
        if (s >= S) continue;
        const double suffix = above + suf_loc[i];
        const float ga = (float)((double)(gw[i] * T[i]) - suffix / (double)tt[i]);
        float* o = g_raw + (r * S + s) * (int64_t)C;
#pragma unroll
        for (int k = 0; k < 3; ++k) o[k] = grgb[k] * w[i] * (col[i][k] * (1.0f - col[i][k]));
        o[3] = ga * dexp[i];
#pragma unroll
        for (int k = 0; k < NS; ++k) if (nsem > k) o[4 + k] = gsem[k] * w[i];
    }
}

extern "C" int32_t nsos_composite_backward(const float* raw, const float* z_vals, const float* rays_d, const float* noise,
                                           float noise_std, int64_t n_rays, int32_t n_samples, int32_t n_ch,
                                           int32_t white_bkgd, const float* g_rgb, const float* g_sem,
                                           const float* g_depth, const float* g_acc, const float* g_disp,
                                           const float* g_weights, float* g_raw, void* stream) {
    if (n_rays == 0) return NSOS_OK;
    NSOS_REQUIRE(raw && z_vals && rays_d && g_raw, NSOS_ERR_NULL_POINTER);
    NSOS_REQUIRE(n_ch >= 4 && n_ch <= 12, NSOS_ERR_UNSUPPORTED);     // 4 + sem_dim, sem_dim <= 8 (generic nets)
    NSOS_REQUIRE(n_rays >= 0 && n_samples >= 1, NSOS_ERR_BAD_SHAPE);
    NSOS_REQUIRE(n_samples <= 512, NSOS_ERR_UNSUPPORTED);
    NSOS_REQUIRE((n_rays + 3) / 4 < (int64_t)1 << 31, NSOS_ERR_UNSUPPORTED);
    const dim3 grid((unsigned)((n_rays + 3) / 4)), block(256);
    const int ipl = (n_samples + 63) / 64;
#define NSOS_LAUNCH_CB(I)                                                                                              \
    if (n_ch <= 6) hipLaunchKernelGGL((composite_backward_kernel<I, 2>), grid, block, 0, (hipStream_t)stream, raw, z_vals, rays_d, noise, \
                       noise_std, n_rays, n_samples, n_ch, white_bkgd, g_rgb, g_sem, g_depth, g_acc, g_disp, g_weights, g_raw);        \
    else hipLaunchKernelGGL((composite_backward_kernel<I, 8>), grid, block, 0, (hipStream_t)stream, raw, z_vals, rays_d, noise, \
                       noise_std, n_rays, n_samples, n_ch, white_bkgd, g_rgb, g_sem, g_depth, g_acc, g_disp, g_weights, g_raw)
    switch (ipl) {
        case 1: NSOS_LAUNCH_CB(1); break;
        case 2: NSOS_LAUNCH_CB(2); break;
        case 3: NSOS_LAUNCH_CB(3); break;
        case 4: NSOS_LAUNCH_CB(4); break;
        default: NSOS_LAUNCH_CB(8); break;
    }
#undef NSOS_LAUNCH_CB
    return nsos_launch_status();
}

extern "C" int32_t nsos_composite_importance(const float* raw, const float* z_vals, const float* rays_d, const float* noise,
                                             float noise_std, int64_t n_rays, int32_t n_coarse, int32_t n_ch, int32_t white_bkgd,
                                             float* weights, float* rgb, float* sem, float* depth, float* acc, float* disp,
                                             const float* u, int32_t n_importance, float* z_fine, float* z_samples, float* z_std,
                                             void* stream) {
    if (n_rays == 0) return NSOS_OK;
    NSOS_REQUIRE(raw && z_vals && rays_d && weights && rgb && depth && acc && disp && z_fine && z_samples && z_std, NSOS_ERR_NULL_POINTER);
    NSOS_REQUIRE(n_ch == 4 || n_ch == 5 || n_ch == 6, NSOS_ERR_UNSUPPORTED);
    NSOS_REQUIRE(n_ch == 4 || sem, NSOS_ERR_NULL_POINTER);
    NSOS_REQUIRE(n_rays > 0 && n_importance >= 1, NSOS_ERR_BAD_SHAPE);
    NSOS_REQUIRE(n_coarse >= 2 && n_coarse <= 64 && n_importance <= NSOS_MAX_IMPORTANCE, NSOS_ERR_UNSUPPORTED);
    NSOS_REQUIRE((n_rays + 3) / 4 < (int64_t)1 << 31, NSOS_ERR_UNSUPPORTED);
    hipLaunchKernelGGL(composite_importance_kernel, dim3((unsigned)((n_rays + 3) / 4)), dim3(256), 0, (hipStream_t)stream, raw, z_vals,
                       rays_d, noise, noise_std, n_rays, n_coarse, n_ch, white_bkgd, weights, rgb, sem, depth, acc, disp, u,
                       n_importance, z_fine, z_samples, z_std);
    return nsos_launch_status();
}
