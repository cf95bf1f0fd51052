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

template <int IPL>
__global__ __launch_bounds__(256) void composite_kernel(const float* __restrict__ raw,
                                                        const float* __restrict__ z_vals,
                                                        const float* __restrict__ rays_d,
                                                        const float* __restrict__ noise, float noise_std,
                                                        int64_t n_rays, int S, int C, int white_bkgd,
                                                        float* __restrict__ weights, float* __restrict__ rgb,
                                                        float* __restrict__ sem, float* __restrict__ depth,
                                                        float* __restrict__ acc, float* __restrict__ disp) {
    const int lane = threadIdx.x & 63;
    const int64_t r = (int64_t)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    if (r >= n_rays) return;

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
        const float* c = rr + (int64_t)(live ? s : 0) * C;
        float dist = (s + 1 < S) ? (z[i + 1] - z[i]) : 1e10f;  // :35-37
        dist = dist * norm;
        float sigma = c[3];
        if (noise) sigma = sigma + noise[r * S + (live ? s : 0)] * noise_std;  // :46-50
        const float relu = sigma > 0.0f ? sigma : 0.0f;
        const float a = live ? (1.0f - expf(-relu * dist)) : 0.0f;  // :52
        alpha[i] = a;
#pragma unroll
        for (int k = 0; k < 3; ++k) col[i][k] = 1.0f / (1.0f + expf(-c[k]));  // sigmoid :41
        smv[i][0] = C > 4 ? c[4] : 0.0f;
        smv[i][1] = C > 5 ? c[5] : 0.0f;
        tloc[i] = prod;
        if (live) prod *= (double)((1.0f - a) + 1e-10f);  // :57
    }
    // exclusive scan of the lane products across the wave
    double incl = prod;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
        const double o = __shfl_up(incl, off, NSOS_WAVE);
        if (lane >= off) incl *= o;
    }
    double excl = __shfl_up(incl, 1, NSOS_WAVE);
    if (lane == 0) excl = 1.0;

    double s_rgb[3] = {0, 0, 0}, s_sem[2] = {0, 0}, s_depth = 0, s_acc = 0;
#pragma unroll
    for (int i = 0; i < IPL; ++i) {
        const int s = s0 + i;
        if (s < S) {
            const float T = (float)(excl * tloc[i]);  // :58 (fp64 running product, rounded per element)
            const float w = alpha[i] * T;             // :61
            weights[r * S + s] = w;
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
