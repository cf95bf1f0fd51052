/* ============================================================================================
 * TEST INFRASTRUCTURE -- not product code.
 *
 * Plain-C CPU restatement of the NeRF-SOS volumetric-rendering hot path.  Only tests/,
 * __graft_entry__.smoke() and bench.py's cpu_baseline leg may load this library; the product
 * package (nerf-sos_amd/) never links, imports or calls it.
 *
 * Parity pin: tests/test_oracle_golden.py checks every function below against the fixtures in
 * tests/golden/ (.npz), which tests/golden/make_goldens.py captured from the real reference
 * (imported unmodified from /root/reference in the build container).
 *
 * The reference does all arithmetic in ATen fp32 ops whose reduction order is an
 * implementation detail of the host BLAS / SIMD width.  This restatement fixes ONE canonical
 * order so that the HIP kernels can be compared with it bit for bit:
 *   - scans / sums / norms : accumulated in fp64 in index order, rounded to fp32 per output
 *     (this IS what torch-CPU cumsum/cumprod/std do -- verified bitwise; torch.sum and
 *     torch.norm differ from it by at most 1 ulp);
 *   - Linear layers        : fp32 fmaf chain  acc = b[o]; acc = fmaf(W[o][k], x[k], acc)  with
 *     k running in CHAIN ORDER (below), which is the order the gfx950 exact-fp32 MFMA
 *     (v_mfma_f32_32x32x2_f32) consumes the activations when a layer's accumulator registers
 *     are fed straight back as the next layer's B operand;
 *   - element-wise ops     : single fp32 operations in the reference's expression order, no
 *     fused multiply-add (build with -ffp-contract=off).
 * ============================================================================================ */
#include "nerf_oracle.h"

#include <math.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

enum { W = 256, HALF = 128, XD = 63, VD = 27, LX = 10, LV = 4 };

int32_t oracle_num_threads(void) {
#ifdef _OPENMP
    return (int32_t)omp_get_max_threads();
#else
    return 1;
#endif
}

/* torch.linspace(start, end, n) on CPU, fp32: step = (end-start)/(n-1);
 * i < n/2 : fma(step, i, start)   else   fma(-step, n-1-i, end).   (verified bitwise) */
static inline float linspace01(int i, int n) {
    if (n == 1) return 0.0f; /* torch.linspace(0, 1, 1) == [0] */
    const float step = 1.0f / (float)(n - 1);
    return (i < n / 2) ? fmaf(step, (float)i, 0.0f) : fmaf(-step, (float)(n - 1 - i), 1.0f);
}

/* chain order of a hidden activation vector: within every block of 8 features the order is
 * 0,4,1,5,2,6,3,7 (the lo/hi half-wave pairing of the 32x32 MFMA accumulator layout). */
static inline int chain_feature(int j) {
    const int q = j & 7;
    return (j & ~7) + (q >> 1) + 4 * (q & 1);
}

/* ---------------------------------------------------------------------------- ray generation
 * utils/ray.py:12-22 get_persp_rays: pixel (i, j) of an H x W image, intrinsics fx, fy, cx, cy, pose c2w[3][4]. */
void oracle_generate_rays(int32_t H, int32_t W, float fx, float fy, float cx, float cy, const float* c2w,
                          float* rays_o, float* rays_d) {
    for (int j = 0; j < H; ++j)
        for (int i = 0; i < W; ++i) {
            const float d0 = ((float)i - cx) / fx, d1 = -(((float)j - cy) / fy), d2 = -1.0f;
            float* o = rays_o + 3 * ((int64_t)j * W + i);
            float* d = rays_d + 3 * ((int64_t)j * W + i);
            for (int c = 0; c < 3; ++c) {
                const float p0 = d0 * c2w[4 * c], p1 = d1 * c2w[4 * c + 1], p2 = d2 * c2w[4 * c + 2];
                d[c] = (p0 + p1) + p2; /* torch.sum over the 3 products, left to right (verified bitwise) */
                o[c] = c2w[4 * c + 3];
            }
        }
}

/* ---------------------------------------------------------------------------- ray setup
 * models/nerf_net.py:164-165 (viewdirs = d/|d|) and models/sampler.py:46-68 (stratified z). */
void oracle_ray_setup(const float* rays_o, const float* rays_d, const float* near, const float* far,
                      const float* t_rand, int64_t n_rays, int32_t n_samples, float* z_vals, float* viewdirs) {
    (void)rays_o;
    for (int64_t r = 0; r < n_rays; ++r) {
        if (viewdirs) {
            const float* d = rays_d + 3 * r;
            const float n = (float)sqrt((double)d[0] * d[0] + (double)d[1] * d[1] + (double)d[2] * d[2]);
            for (int c = 0; c < 3; ++c) viewdirs[3 * r + c] = d[c] / n;
        }
        float* z = z_vals + r * n_samples;
        for (int s = 0; s < n_samples; ++s) {
            const float t = linspace01(s, n_samples);
            z[s] = near[r] * (1.0f - t) + far[r] * t; /* models/sampler.py:48 */
        }
        if (t_rand) { /* models/sampler.py:54-68 */
            float prev_mid = 0.0f;
            const float zlast = z[n_samples - 1];
            float zs = z[0];
            for (int s = 0; s < n_samples; ++s) {
                const float znext = (s + 1 < n_samples) ? z[s + 1] : zlast;
                const float mid = 0.5f * (znext + zs);
                const float lower = (s == 0) ? z[0] : prev_mid;
                const float upper = (s + 1 < n_samples) ? mid : zlast;
                prev_mid = mid;
                zs = znext;
                z[s] = lower + (upper - lower) * t_rand[r * n_samples + s];
            }
        }
    }
}

/* models/sampler.py:70,166: pts = o + d * z (separately rounded mul and add). */
void oracle_ray_points(const float* rays_o, const float* rays_d, const float* z_vals, int64_t n_rays,
                       int32_t n_samples, float* pts) {
    for (int64_t r = 0; r < n_rays; ++r)
        for (int s = 0; s < n_samples; ++s)
            for (int c = 0; c < 3; ++c) {
                const float m = rays_d[3 * r + c] * z_vals[r * n_samples + s];
                pts[(r * n_samples + s) * 3 + c] = rays_o[3 * r + c] + m;
            }
}

/* ---------------------------------------------------------------------------- encoding
 * models/embedder.py:34-48: [x, sin(2^0 x), cos(2^0 x), sin(2^1 x), ...] (3 + 6L values). */
static void encode3(const float* x, int n_freqs, float* out) {
    out[0] = x[0];
    out[1] = x[1];
    out[2] = x[2];
    for (int k = 0; k < n_freqs; ++k) {
        const float f = (float)(1 << k);
        for (int c = 0; c < 3; ++c) {
            const float a = x[c] * f;
            out[3 + 6 * k + c] = sinf(a);
            out[3 + 6 * k + 3 + c] = cosf(a);
        }
    }
}

void oracle_posenc(const float* x, int64_t n_pts, int32_t n_freqs, float* out) {
    const int dim = 3 + 6 * n_freqs;
    for (int64_t p = 0; p < n_pts; ++p) encode3(x + 3 * p, n_freqs, out + p * dim);
}

/* ---------------------------------------------------------------------------- MLP
 * models/nerf_mlp.py:67-100.  y[o] = fmaf-chain over the inputs in chain order. */
static inline float dot_natural(const float* w, const float* x, int n, float acc) {
    for (int k = 0; k < n; ++k) acc = fmaf(w[k], x[k], acc);
    return acc;
}
static inline float dot_chain(const float* w, const float* h, int n, float acc) {
    for (int j = 0; j < n; ++j) {
        const int f = chain_feature(j);
        acc = fmaf(w[f], h[f], acc);
    }
    return acc;
}
/* small heads are evaluated on the vector ALU by the two half-waves separately:
 * lo = b + sum over features with bit2 clear, hi = sum over features with bit2 set, out = lo+hi */
static inline float dot_halves(const float* w, const float* h, int n, float bias) {
    float lo = bias, hi = 0.0f;
    for (int f = 0; f < n; ++f) {
        if (f & 4) hi = fmaf(w[f], h[f], hi);
        else       lo = fmaf(w[f], h[f], lo);
    }
    return lo + hi;
}

static void mlp_point(const oracle_mlp_weights* w, const float* pt, const float* dir, float* raw,
                      const oracle_mlp_taps* taps, int64_t p) {
    float ex[XD + 1], ed[VD + 1], h[W], t[W];
    encode3(pt, LX, ex);
    encode3(dir, LV, ed);
    for (int l = 0; l < 8; ++l) {
        const int in_dim = (l == 0) ? XD : (l == 5 ? XD + W : W);
        for (int o = 0; o < W; ++o) {
            const float* row = w->pts_w[l] + (int64_t)o * in_dim;
            float acc = w->pts_b[l][o];
            if (l == 0) acc = dot_natural(row, ex, XD, acc);
            else if (l == 5) { /* input cat([x63, h]) (models/nerf_mlp.py:73-74); canonical order: h part, then x63 */
                acc = dot_chain(row + XD, h, W, acc);
                acc = dot_natural(row, ex, XD, acc);
            } else acc = dot_chain(row, h, W, acc);
            t[o] = acc > 0.0f ? acc : 0.0f;
        }
        memcpy(h, t, sizeof h);
        if (taps && taps->h[l]) memcpy(taps->h[l] + p * W, h, sizeof h);
    }
    raw[3] = dot_halves(w->alpha_w, h, W, w->alpha_b[0]); /* :77 */
    if (w->use_semantics) {                                /* :79-80 */
        float s[HALF];
        const int in_dim = w->sem_with_coord ? W + XD : W;
        for (int o = 0; o < HALF; ++o) {
            const float* row = w->sem0_w + (int64_t)o * in_dim;
            float acc = dot_chain(row, h, W, w->sem0_b[o]); /* cat([h, x63]): h first */
            if (w->sem_with_coord) acc = dot_natural(row + W, ex, XD, acc);
            s[o] = acc > 0.0f ? acc : 0.0f;
        }
        if (taps && taps->sem_hidden) memcpy(taps->sem_hidden + p * HALF, s, sizeof s);
        for (int o = 0; o < 2; ++o) raw[4 + o] = dot_halves(w->sem2_w + o * HALF, s, HALF, w->sem2_b[o]);
    }
    for (int o = 0; o < W; ++o) t[o] = dot_chain(w->feature_w + (int64_t)o * W, h, W, w->feature_b[o]); /* :86 */
    if (taps && taps->feature) memcpy(taps->feature + p * W, t, sizeof t);
    float v[HALF];
    for (int o = 0; o < HALF; ++o) { /* :87-90, cat([feature, dir27]) */
        const float* row = w->views_w + (int64_t)o * (W + VD);
        float acc = dot_chain(row, t, W, w->views_b[o]);
        acc = dot_natural(row + W, ed, VD, acc);
        v[o] = acc > 0.0f ? acc : 0.0f;
    }
    if (taps && taps->view_hidden) memcpy(taps->view_hidden + p * HALF, v, sizeof v);
    for (int o = 0; o < 3; ++o) raw[o] = dot_halves(w->rgb_w + o * HALF, v, HALF, w->rgb_b[o]); /* :92 */
}

/* dirs_stride_pts: number of consecutive points that share one direction row (S for rays,
 * 1 for explicit per-point directions as in NeRFMLP.forward, models/nerf_mlp.py:179). */
void oracle_mlp(const oracle_mlp_weights* w, const float* pts, const float* dirs, int64_t n_pts,
                int64_t dirs_stride_pts, float* raw, const oracle_mlp_taps* taps) {
    const int n_ch = w->use_semantics ? 6 : 4;
#pragma omp parallel for schedule(static)
    for (int64_t p = 0; p < n_pts; ++p)
        mlp_point(w, pts + 3 * p, dirs + 3 * (p / dirs_stride_pts), raw + p * n_ch, taps, p);
}

/* ---------------------------------------------------------------------------- compositing
 * models/renderer.py:35-85. */
void oracle_composite(const float* raw, const float* z_vals, const float* rays_d, const float* noise,
                      float noise_std, int64_t n_rays, int32_t n_samples, int32_t n_ch, int32_t white_bkgd,
                      float* weights, float* rgb, float* sem, float* depth, float* acc, float* disp) {
    const int S = n_samples;
    for (int64_t r = 0; r < n_rays; ++r) {
        const float* d = rays_d + 3 * r;
        const float norm = (float)sqrt((double)d[0] * d[0] + (double)d[1] * d[1] + (double)d[2] * d[2]);
        const float* z = z_vals + r * S;
        double T = 1.0, s_rgb[3] = {0, 0, 0}, s_sem[2] = {0, 0}, s_depth = 0, s_acc = 0;
        for (int s = 0; s < S; ++s) {
            const float* c = raw + (r * S + s) * n_ch;
            float dist = (s + 1 < S) ? (z[s + 1] - z[s]) : 1e10f; /* :35-37 */
            dist = dist * norm;                                    /* :38 */
            float sigma = c[3];
            if (noise) sigma = sigma + noise[r * S + s] * noise_std; /* :46-50 */
            const float relu = sigma > 0.0f ? sigma : 0.0f;
            const float alpha = 1.0f - expf(-relu * dist);           /* :52 */
            const float wgt = alpha * (float)T;                      /* :57-61 */
            T *= (double)((1.0f - alpha) + 1e-10f);
            weights[r * S + s] = wgt;
            for (int k = 0; k < 3; ++k) {
                const float col = 1.0f / (1.0f + expf(-c[k])); /* sigmoid :41 */
                s_rgb[k] += (double)(wgt * col);
            }
            if (n_ch > 4)
                for (int k = 0; k < n_ch - 4; ++k) s_sem[k] += (double)(wgt * c[4 + k]); /* :64-66 */
            s_depth += (double)(wgt * z[s]);
            s_acc += (double)wgt;
        }
        float a = (float)s_acc, dep = (float)s_depth;
        if (a <= 1e-10f) dep = 1e10f;          /* :72 */
        const float q = dep / a;
        disp[r] = 1.0f / (q > 1e-10f ? q : (q != q ? q : 1e-10f)); /* torch.max propagates NaN :74 */
        depth[r] = dep;
        acc[r] = a;
        for (int k = 0; k < 3; ++k) rgb[3 * r + k] = (float)s_rgb[k] + (white_bkgd ? (1.0f - a) : 0.0f);
        if (n_ch > 4)
            for (int k = 0; k < n_ch - 4; ++k)
                sem[(n_ch - 4) * r + k] = (float)s_sem[k] + (white_bkgd ? (1.0f - a) : 0.0f);
    }
}

/* ---------------------------------------------------------------------------- importance sampling
 * torch.searchsorted(cdf, u, right=True): number of cdf entries <= u. */
void oracle_searchsorted_right(const float* cdf, int32_t n_cdf, const float* u, int32_t n_u, int64_t n_rows,
                               int64_t* inds) {
    for (int64_t r = 0; r < n_rows; ++r)
        for (int i = 0; i < n_u; ++i) {
            int lo = 0, hi = n_cdf;
            const float v = u[r * n_u + i];
            while (lo < hi) {
                const int mid = (lo + hi) >> 1;
                if (cdf[r * n_cdf + mid] <= v) lo = mid + 1; else hi = mid;
            }
            inds[r * n_u + i] = lo;
        }
}

static int cmp_float(const void* a, const void* b) {
    const float x = *(const float*)a, y = *(const float*)b;
    return (x > y) - (x < y);
}

/* models/sampler.py:91-167 + models/nerf_net.py:124.  u == NULL <=> det (linspace).
 * cdf_in != NULL replaces the computed cdf (stage-wise index pinning). */
void oracle_importance(const float* z_vals, const float* weights, const float* u, const float* cdf_in,
                       int64_t n_rays, int32_t n_coarse, int32_t n_importance, float* cdf_out, int64_t* inds_out,
                       float* z_samples, float* z_fine, float* z_std) {
    const int S = n_coarse, NB = S - 1, N = n_importance;
    float* bins = (float*)malloc(sizeof(float) * (size_t)NB * 2);
    float* cdf = bins + NB;
    for (int64_t r = 0; r < n_rays; ++r) {
        const float* z = z_vals + r * S;
        for (int j = 0; j < NB; ++j) bins[j] = 0.5f * (z[j + 1] + z[j]); /* :155 */
        if (cdf_in) memcpy(cdf, cdf_in + r * NB, sizeof(float) * NB);
        else {
            double sum = 0;                                                /* :93-97 */
            for (int j = 0; j < NB - 1; ++j) sum += (double)(weights[r * S + 1 + j] + 1e-5f);
            const float fsum = (float)sum;
            double run = 0;
            cdf[0] = 0.0f;
            for (int j = 0; j < NB - 1; ++j) {
                run += (double)((weights[r * S + 1 + j] + 1e-5f) / fsum);
                cdf[j + 1] = (float)run;
            }
        }
        if (cdf_out) memcpy(cdf_out + r * NB, cdf, sizeof(float) * NB);
        float* zs = z_samples + r * N;
        for (int i = 0; i < N; ++i) {
            const float v = u ? u[r * N + i] : linspace01(i, N); /* :100-103 */
            int lo = 0, hi = NB;
            while (lo < hi) { const int mid = (lo + hi) >> 1; if (cdf[mid] <= v) lo = mid + 1; else hi = mid; }
            if (inds_out) inds_out[r * N + i] = lo;
            const int below = lo - 1 > 0 ? lo - 1 : 0;             /* :118-119 */
            const int above = lo < NB - 1 ? lo : NB - 1;
            float denom = cdf[above] - cdf[below];                 /* :128-131 */
            if (denom < 1e-5f) denom = 1.0f;
            const float t = (v - cdf[below]) / denom;
            const float span = bins[above] - bins[below];
            zs[i] = bins[below] + t * span;
        }
        float* zf = z_fine + r * (S + N);                          /* :161 */
        memcpy(zf, z, sizeof(float) * S);
        memcpy(zf + S, zs, sizeof(float) * N);
        qsort(zf, (size_t)(S + N), sizeof(float), cmp_float);
        if (z_std) {                                               /* models/nerf_net.py:124 */
            double m = 0, v2 = 0;
            for (int i = 0; i < N; ++i) m += zs[i];
            m /= N;
            for (int i = 0; i < N; ++i) v2 += (zs[i] - m) * (zs[i] - m);
            z_std[r] = (float)sqrt(v2 / N);
        }
    }
    free(bins);
}
