// Training-batch assembly on the device: what the reference's dataset classes and collaters
// hand to `train_one_step`, produced from the scene's images resident in HBM and its poses -- no [N,H,W,2,3] ray file on
// disk, no per-step host gather, no per-step H2D copy of rays / colours / masks.
//
//   nsos_patch_batch : PatchNeRFDataset.__getitem__ (data/datasets.py:240-254: strided crop at a random origin, the item's
//                      pose and crop origin) of B items +
//                      PatchBatchCollater (data/collater.py:31-61), rays from the item's pose through K0's arithmetic
//                      (utils/ray.py:12-22 get_persp_rays; bit-identical to the stored rays, see generate_rays_kernel);
//   nsos_pixel_batch : an explicit pixel list -- RayNeRFDataset's flattened [N*H*W] items (data/datasets.py:149-152,159-171 +
//                      RayBatchCollater) and ViewNeRFDataset's random pixels of one view (:279-300 + ViewBatchCollater).
//
// HBM-bound gather kernels: per selected pixel 12 B (pose: broadcast) + 4*rgb_ch + 4*mask_words read, 24 B + the same written.
#include "common.h"

struct PixelSource {
    int H, W;
    float fx, fy, cx, cy;
    const float* poses;   // [n_images, pose_rows, pose_cols]: rows 3 (LLFF [R | t | hwf], cols 5) or 4 (blender / toydesk /
    int pose_rows, pose_cols, n_images;   // tankstemple 4x4, data/gen_dataset.py:228-233 saves poses[i_split] unsliced); rays use [:3,:4]
    const float* rgbs;    // [n_images, H, W, rgb_ch] or NULL
    int rgb_ch;
    const unsigned* masks;   // [n_images, H, W, mask_words] 4-byte words (int64 label = 2 words, float = 1) or NULL
    int mask_words;
};

// one selected pixel (image n, row y, column x) -> slot `o` of the outputs
__device__ __forceinline__ void emit_pixel(const PixelSource& S, int n, int y, int x, int64_t o, float* __restrict__ rays_o,
                                           float* __restrict__ rays_d, float* __restrict__ target, unsigned* __restrict__ masks_out) {
    n = n < 0 ? 0 : (n >= S.n_images ? S.n_images - 1 : n);   // descriptors that live on the device cannot be validated by the
    y = y < 0 ? 0 : (y >= S.H ? S.H - 1 : y);                 // host: clamp instead of reading out of bounds
    x = x < 0 ? 0 : (x >= S.W ? S.W - 1 : x);
    if (rays_o) {
        const float* c2w = S.poses + (size_t)n * S.pose_rows * S.pose_cols;
        const float d0 = ((float)x - S.cx) / S.fx, d1 = -(((float)y - S.cy) / S.fy), d2 = -1.0f;   // utils/ray.py:16
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            const float* row = c2w + c * S.pose_cols;
            const float p0 = d0 * row[0], p1 = d1 * row[1], p2 = d2 * row[2];
            rays_d[3 * o + c] = (p0 + p1) + p2;                                                     // utils/ray.py:18 (left to right)
            rays_o[3 * o + c] = row[3];                                                             // utils/ray.py:20
        }
    }
    const size_t pix = ((size_t)n * S.H + y) * S.W + x;
    if (target)
        for (int c = 0; c < S.rgb_ch; ++c) target[o * S.rgb_ch + c] = S.rgbs[pix * S.rgb_ch + c];
    if (masks_out)
        for (int c = 0; c < S.mask_words; ++c) masks_out[o * S.mask_words + c] = S.masks[pix * S.mask_words + c];
}

#define NSOS_PATCH_SEL_BY_VALUE 64
struct PatchSel { int v[NSOS_PATCH_SEL_BY_VALUE * 3]; };   // (image, h_idx, w_idx) per patch, in the kernel arguments

// grid: one thread per (patch, output pixel); `sel_dev` (device, [B,3] int32) wins over the by-value copy when given
__global__ __launch_bounds__(256) void patch_batch_kernel(const PixelSource S, const PatchSel sel, const int* __restrict__ sel_dev,
                                                          int n_patches, int P, int stride, float* __restrict__ rays_o,
                                                          float* __restrict__ rays_d, float* __restrict__ target,
                                                          unsigned* __restrict__ masks_out, float* __restrict__ poses_out,
                                                          float* __restrict__ start_out, int64_t out_base, int patch_base) {
    const int64_t gid = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int64_t per = (int64_t)P * P;
    if (gid >= per * n_patches) return;
    const int b = (int)(gid / per);
    const int q = (int)(gid - b * per);
    const int a = q / P, c = q - a * P;                     // rays[h_idx + a*stride, w_idx + c*stride] (data/datasets.py:245)
    const int* d = sel_dev ? sel_dev + 3 * b : sel.v + 3 * b;
    // the item's pose and crop origin (data/datasets.py:251-252: `poses = self.poses[i]`, `start_idx = Tensor([h_idx, w_idx])`);
    // a patch has >= 1 pixel, so its first thread writes them (serial loop: P*P may be smaller than pose_rows*pose_cols)
    if (q == 0) {
        const int n = d[0] < 0 ? 0 : (d[0] >= S.n_images ? S.n_images - 1 : d[0]);
        if (poses_out)
            for (int k = 0; k < S.pose_rows * S.pose_cols; ++k)
                poses_out[(size_t)(patch_base + b) * S.pose_rows * S.pose_cols + k] = S.poses[(size_t)n * S.pose_rows * S.pose_cols + k];
        if (start_out) {
            start_out[2 * (patch_base + b)] = (float)d[1];
            start_out[2 * (patch_base + b) + 1] = (float)d[2];
        }
    }
    emit_pixel(S, d[0], d[1] + a * stride, d[2] + c * stride, out_base + gid, rays_o, rays_d, target, masks_out);
}

// grid: one thread per listed pixel; pix = (n*H + y)*W + x
__global__ __launch_bounds__(256) void pixel_batch_kernel(const PixelSource S, const int64_t* __restrict__ pix, int64_t n,
                                                          float* __restrict__ rays_o, float* __restrict__ rays_d,
                                                          float* __restrict__ target, unsigned* __restrict__ masks_out) {
    const int64_t gid = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (gid >= n) return;
    const int64_t p = pix[gid];
    const int64_t hw = (int64_t)S.H * S.W;
    const int img = (int)(p / hw);
    const int rem = (int)(p - (int64_t)img * hw);
    emit_pixel(S, img, rem / S.W, rem % S.W, gid, rays_o, rays_d, target, masks_out);
}

static int32_t check_source(int32_t H, int32_t W, float fx, float fy, const float* poses, int32_t pose_rows, int32_t pose_cols, int32_t n_images,
                            const float* rgbs, int32_t rgb_ch, const void* masks, int32_t mask_words, const float* rays_o,
                            const float* rays_d, const float* target, const void* masks_out) {
    NSOS_REQUIRE(H > 0 && W > 0 && n_images > 0 && fx != 0.0f && fy != 0.0f, NSOS_ERR_BAD_SHAPE);
    NSOS_REQUIRE((rays_o != nullptr) == (rays_d != nullptr), NSOS_ERR_NULL_POINTER);
    if (rays_o) {
        NSOS_REQUIRE(poses, NSOS_ERR_NULL_POINTER);
        NSOS_REQUIRE((pose_rows == 3 || pose_rows == 4) && pose_cols >= 4, NSOS_ERR_BAD_SHAPE);
    }
    if (target) {
        NSOS_REQUIRE(rgbs, NSOS_ERR_NULL_POINTER);
        NSOS_REQUIRE(rgb_ch >= 1, NSOS_ERR_BAD_SHAPE);
    }
    if (masks_out) {
        NSOS_REQUIRE(masks, NSOS_ERR_NULL_POINTER);
        NSOS_REQUIRE(mask_words >= 1, NSOS_ERR_BAD_SHAPE);
    }
    NSOS_REQUIRE(rays_o || target || masks_out, NSOS_ERR_NULL_POINTER);
    return NSOS_OK;
}

extern "C" int32_t nsos_patch_batch(int32_t H, int32_t W, float fx, float fy, float cx, float cy, const float* poses,
                                    int32_t pose_rows, int32_t pose_cols, int32_t n_images, const float* rgbs, int32_t rgb_ch, const void* masks,
                                    int32_t mask_words, const int32_t* sel_host, const int32_t* sel_dev, int32_t n_patches,
                                    int32_t patch, int32_t stride, float* rays_o, float* rays_d, float* target, void* masks_out,
                                    float* poses_out, float* start_out, void* stream) {
    NSOS_REQUIRE(n_patches >= 0 && patch >= 1 && stride >= 1, NSOS_ERR_BAD_SHAPE);
    if (n_patches == 0) return NSOS_OK;
    const int32_t rc = check_source(H, W, fx, fy, poses, pose_rows, pose_cols, n_images, rgbs, rgb_ch, masks, mask_words, rays_o, rays_d, target, masks_out);
    if (rc != NSOS_OK) return rc;
    NSOS_REQUIRE((sel_host != nullptr) != (sel_dev != nullptr), NSOS_ERR_NULL_POINTER);   // exactly one of the two
    NSOS_REQUIRE(!poses_out || (poses && (pose_rows == 3 || pose_rows == 4) && pose_cols >= 4), NSOS_ERR_NULL_POINTER);
    const int64_t last = (int64_t)(patch - 1) * stride;
    NSOS_REQUIRE(last < H && last < W, NSOS_ERR_BAD_SHAPE);
    if (sel_host)   // data/datasets.py:240-241: 0 <= h_idx <= H - crop_size; the last sampled row is h_idx + (P-1)*stride
        for (int b = 0; b < n_patches; ++b) {
            const int32_t* d = sel_host + 3 * b;
            NSOS_REQUIRE(d[0] >= 0 && d[0] < n_images && d[1] >= 0 && d[1] + last < H && d[2] >= 0 && d[2] + last < W, NSOS_ERR_BAD_SHAPE);
        }
    const PixelSource S = {H, W, fx, fy, cx, cy, poses, pose_rows, pose_cols, n_images, rgbs, rgb_ch, static_cast<const unsigned*>(masks), mask_words};
    const int64_t per = (int64_t)patch * patch;
    const int step = sel_dev ? n_patches : NSOS_PATCH_SEL_BY_VALUE;
    for (int b0 = 0; b0 < n_patches; b0 += step) {
        const int nb = n_patches - b0 < step ? n_patches - b0 : step;
        PatchSel sel = {};
        if (sel_host)
            for (int k = 0; k < nb * 3; ++k) sel.v[k] = sel_host[3 * b0 + k];
        const int64_t total = per * nb;
        NSOS_REQUIRE((total + 255) / 256 < (int64_t)1 << 31, NSOS_ERR_UNSUPPORTED);
        hipLaunchKernelGGL(patch_batch_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream, S, sel,
                           sel_dev ? sel_dev + 3 * b0 : nullptr, nb, patch, stride, rays_o, rays_d, target,
                           static_cast<unsigned*>(masks_out), poses_out, start_out, per * b0, b0);
    }
    return nsos_launch_status();
}

extern "C" int32_t nsos_pixel_batch(int32_t H, int32_t W, float fx, float fy, float cx, float cy, const float* poses,
                                    int32_t pose_rows, int32_t pose_cols, int32_t n_images, const float* rgbs, int32_t rgb_ch, const void* masks,
                                    int32_t mask_words, const int64_t* pix, int64_t n, float* rays_o, float* rays_d,
                                    float* target, void* masks_out, void* stream) {
    NSOS_REQUIRE(n >= 0, NSOS_ERR_BAD_SHAPE);
    if (n == 0) return NSOS_OK;
    const int32_t rc = check_source(H, W, fx, fy, poses, pose_rows, pose_cols, n_images, rgbs, rgb_ch, masks, mask_words, rays_o, rays_d, target, masks_out);
    if (rc != NSOS_OK) return rc;
    NSOS_REQUIRE(pix, NSOS_ERR_NULL_POINTER);
    NSOS_REQUIRE((n + 255) / 256 < (int64_t)1 << 31, NSOS_ERR_UNSUPPORTED);
    const PixelSource S = {H, W, fx, fy, cx, cy, poses, pose_rows, pose_cols, n_images, rgbs, rgb_ch, static_cast<const unsigned*>(masks), mask_words};
    hipLaunchKernelGGL(pixel_batch_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, S, pix, n, rays_o,
                       rays_d, target, static_cast<unsigned*>(masks_out));
    return nsos_launch_status();
}
