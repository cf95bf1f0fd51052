/* TEST INFRASTRUCTURE -- not product code.  See nerf_oracle.c. */
#ifndef NERF_ORACLE_H
#define NERF_ORACLE_H
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* One NeRF-SOS MLP (models/nerf_mlp.py:24-100), tensors in PyTorch [out,in] row-major fp32.
 * Fixed architecture D=8, W=256, skip after layer 4, multires 10 / 4 (every shipped config). */
typedef struct {
    const float* pts_w[8];   /* [256,63] [256,256]x4 [256,319] [256,256]x2 */
    const float* pts_b[8];   /* [256] */
    const float* alpha_w;    /* [1,256] */
    const float* alpha_b;    /* [1] */
    const float* feature_w;  /* [256,256] */
    const float* feature_b;  /* [256] */
    const float* views_w;    /* [128,283] */
    const float* views_b;    /* [128] */
    const float* rgb_w;      /* [3,128] */
    const float* rgb_b;      /* [3] */
    const float* sem0_w;     /* [128,319] or [128,256]; NULL when use_semantics == 0 */
    const float* sem0_b;     /* [128] */
    const float* sem2_w;     /* [2,128] */
    const float* sem2_b;     /* [2] */
    int32_t use_semantics;   /* 0/1 */
    int32_t sem_with_coord;  /* 0/1 */
} oracle_mlp_weights;

/* per-point intermediate activations (optional, for the layer-wise golden checks) */
typedef struct {
    float* h[8];        /* each [P,256], post-ReLU */
    float* feature;     /* [P,256] */
    float* view_hidden; /* [P,128] */
    float* sem_hidden;  /* [P,128] */
} oracle_mlp_taps;

void oracle_generate_rays(int32_t H, int32_t W, float fx, float fy, float cx, float cy, const float* c2w,
                          float* rays_o, float* rays_d);
void oracle_ray_setup(const float* rays_o, const float* rays_d, const float* near, const float* far,
                      const float* t_rand, int64_t n_rays, int32_t n_samples, float* z_vals, float* viewdirs);
void oracle_ray_points(const float* rays_o, const float* rays_d, const float* z_vals, int64_t n_rays,
                       int32_t n_samples, float* pts);
void oracle_posenc(const float* x, int64_t n_pts, int32_t n_freqs, float* out);
void oracle_mlp(const oracle_mlp_weights* w, const float* pts, const float* dirs, int64_t n_pts,
                int64_t dirs_stride_pts, float* raw, const oracle_mlp_taps* taps);
void oracle_composite(const float* raw, const float* z_vals, const float* rays_d, const float* noise,
                      float noise_std, int64_t n_rays, int32_t n_samples, int32_t n_ch, int32_t white_bkgd,
                      float* weights, float* rgb, float* sem, float* depth, float* acc, float* disp);
void oracle_searchsorted_right(const float* cdf, int32_t n_cdf, const float* u, int32_t n_u, int64_t n_rows,
                               int64_t* inds);
void oracle_importance(const float* z_vals, const float* weights, const float* u, const float* cdf_in,
                       int64_t n_rays, int32_t n_coarse, int32_t n_importance, float* cdf_out, int64_t* inds_out,
                       float* z_samples, float* z_fine, float* z_std);
int32_t oracle_num_threads(void);

#ifdef __cplusplus
}
#endif
#endif
