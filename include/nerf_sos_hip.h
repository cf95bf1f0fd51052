/* nerf_sos_hip.h -- C ABI of the MI355X (gfx950) NeRF-SOS volumetric-rendering path.
 *
 * The reference (VITA-Group/NeRF-SOS) is 100 % Python and has no FFI/plugin interface: its
 * boundary for this path is the nn.Module duck-type NeRFNet.forward()/render_rays()
 * (models/nerf_net.py:71-195).  `nerf-sos_amd/` mirrors that module in Python; THIS header is
 * what it binds underneath (ctypes), one entry point per fusion group of SURVEY.md section 2.3.
 * Each function cites the reference lines whose arithmetic it replaces.
 *
 * Conventions (all functions):
 *   - every pointer is a DEVICE pointer to fp32 (or int64 where stated), row-major, dense;
 *   - `stream` is a hipStream_t passed as void* (NULL = the null stream); calls only enqueue
 *     work: they never synchronise, never allocate, never free, never touch host copies of
 *     the data; the caller owns all buffers and keeps them alive until the stream drains;
 *   - return value: NSOS_OK (0), a negative NSOS_ERR_* validation code, or a positive
 *     hipError_t from the launch;
 *   - stateless and re-entrant; thread-safe as far as the HIP runtime is.
 *   - there is NO CPU fallback: without a gfx950 device every launch returns a hipError_t.
 */
#ifndef NERF_SOS_HIP_H
#define NERF_SOS_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* bumped whenever an existing entry point's arguments or buffer formats change (2: 16-bit / tile-major saved operands of
 * nsos_mlp_forward_rays_save16_lp and nsos_sem_head_wgrad_x3; 3: pose_rows in nsos_patch_batch / nsos_pixel_batch; 4: the tile-major
 * sem_hid16 of the default 16-bit kernel, nsos_mlp_save16_layout's NSOS_SEM_HID_TILED bit; 6: `scale` of nsos_mlp_input_grads_x3[_a16]
 * is three floats -- trunk scale, colour-branch factor, semantic-branch factor; 7: the generic kernels' packed program gained a field
 * (GenOp::ksplit_off: an older binding's buffer sizes still agree, but the two sides must match) + nsos_wgrad_batch) */
#define NSOS_ABI_VERSION 8

enum {
    NSOS_OK = 0,
    NSOS_ERR_NULL_POINTER = -1,   /* a required pointer argument is NULL */
    NSOS_ERR_BAD_SHAPE = -2,      /* negative / zero / inconsistent sizes */
    NSOS_ERR_UNSUPPORTED = -3,    /* shape outside what the kernels are specialised for */
    NSOS_ERR_BUFFER_TOO_SMALL = -4,
    NSOS_ERR_MISALIGNED = -5      /* a pointer that must be 16-byte aligned is not */
};

/* Fixed architecture of every shipped config (configs/ (all 34): N_samples 64, N_importance 128,
 * use_viewdirs; run_nerf.py:96-101 defaults netdepth 8 / netwidth 256; models/nerf_net.py:44 skips=[4];
 * run_nerf.py:109-112 multires 10 / multires_views 4). */
#define NSOS_NET_DEPTH 8
#define NSOS_NET_WIDTH 256
#define NSOS_XYZ_FREQS 10
#define NSOS_DIR_FREQS 4
#define NSOS_XYZ_DIM 63 /* 3 + 6*10 */
#define NSOS_DIR_DIM 27 /* 3 + 6*4  */

/* Semantic-head variants (models/nerf_mlp.py:53-64; run_nerf.py:180-190). */
enum { NSOS_SEM_NONE = 0, NSOS_SEM_PLAIN = 1, NSOS_SEM_COORD = 2 };

/* The tensors of ONE NeRFMLP.mlp in PyTorch nn.Linear layout ([out,in] row-major fp32, device
 * memory): exactly the state-dict entries `<net>.mlp.*` (models/nerf_mlp.py:40-64). */
typedef struct nsos_mlp_tensors {
    const float* pts_w[NSOS_NET_DEPTH]; /* pts_linears.i.weight: [256,63] [256,256]x4 [256,319] [256,256]x2 */
    const float* pts_b[NSOS_NET_DEPTH]; /* pts_linears.i.bias  : [256] */
    const float* alpha_w;               /* alpha_linear.weight  [1,256]   */
    const float* alpha_b;               /* alpha_linear.bias    [1]       */
    const float* feature_w;             /* feature_linear.weight[256,256] */
    const float* feature_b;             /* feature_linear.bias  [256]     */
    const float* views_w;               /* views_linears.0.weight [128,283] */
    const float* views_b;               /* views_linears.0.bias   [128]     */
    const float* rgb_w;                 /* rgb_linear.weight    [3,128]   */
    const float* rgb_b;                 /* rgb_linear.bias      [3]       */
    const float* sem0_w;                /* semantic_linear.0.weight [128,319] (COORD) / [128,256] (PLAIN) / NULL */
    const float* sem0_b;                /* semantic_linear.0.bias   [128] */
    const float* sem2_w;                /* semantic_linear.2.weight [2,128]   */
    const float* sem2_b;                /* semantic_linear.2.bias   [2]       */
} nsos_mlp_tensors;

int32_t nsos_abi_version(void);
/* First 16 hex digits of the sha256 over the kernel sources and headers under csrc, this header and the compiler flags the library was built
 * from ("unstamped" for a hand-run make): lets a CPU-only check refuse a stale binary (tests/test_abi.py). */
const char* nsos_source_hash(void);
const char* nsos_error_string(int32_t code);

/* ---- weight packing ---------------------------------------------------------------------
 * The fused MLP kernel streams its weights through LDS as 32 KiB chunks of MFMA A-operands in
 * issue order (DESIGN.md "HBM layout").  nsos_mlp_pack gathers one net's tensors into that
 * stream (call again whenever a parameter changes).  `packed` must be 16-byte aligned and hold
 * nsos_mlp_packed_bytes(sem_mode) bytes.  Replaces nothing in the reference: nn.Linear keeps
 * [out,in] tensors (models/nerf_mlp.py:40-64) and ATen re-reads them per addmm. */
size_t nsos_mlp_packed_bytes(int32_t sem_mode);
int32_t nsos_mlp_pack(const nsos_mlp_tensors* tensors, int32_t sem_mode, void* packed, size_t packed_bytes,
                      void* stream);

/* ---- K2-G: the MLP for any architecture the reference's constructors accept (csrc/mlp_generic.hip) -------------------------
 * nn.Linear tensors ([out,in] row-major fp32, device memory) of ONE NeRFMLP.mlp, described as the reference builds it
 * (models/nerf_mlp.py:40-64; NeRFMLP.__init__ :136-166): depth D, width W, the skip set as a bit mask (bit i: "i in skips", the
 * output of pts_linears.i is concatenated with input_pts, :73-74), multires / multires_views (xyz_freqs / dir_freqs; -1 = no
 * embedding: the raw 3-vector), use_viewdirs (0: output_linear only, :97-98), the semantic head as its Linear modules in order
 * (sem_layers of them: 2 for sem_layer <= 2, sem_layer otherwise, :58-63; ReLU between them), sem_with_coord (cat([h,
 * input_pts]), :79), sem_with_geo (geo[0..1] = geo_map_sem's two Linears on alpha; semantics *= mapping, :60,:81-83).
 * Exact-fp32 MFMA arithmetic: every Linear is an fmaf chain over its inputs in k order starting from the bias (the accumulators'
 * initial value) -- EXCEPT single-tile output ops (alpha 1 row, rgb 3, output_linear 4, the last Linear of the semantic chain), whose
 * K range is split into n_waves (4 or 8, chosen from the architecture's LDS footprint: two workgroups per CU -> 4, one -> 8) partial
 * chains that wave 0 adds in a fixed order: the same Linear therefore rounds differently under different architectures (always
 * deterministic; every case is held to the reference goldens at 1e-4, tests/test_generic_arch.py, re-validated after the change of
 * the reduction order in round 5).  Wave-wide sums / scans of the compositing, loss and eval kernels are DPP / permlane trees in a
 * fixed association (csrc/common.h nsos_wave_sum, nsos_wave_excl_scan).  Training: the K7-G entries below.
 * raw: [n, 4 + sem_dim] (4 without view directions).
 * The shipped architecture (8 x 256, skips {4}, 10 / 4 octaves, view directions, two-Linear head) should use nsos_mlp_forward_*:
 * this path is ~2x slower there.  Limits: depth <= 16, sem_layers <= 8, sem_dim <= 8, and the per-tile
 * activation buffers (ceil(W / 32) * 32 rows each) within 160 KiB of LDS: 32-point tiles up to W = 576 (384 with a deep semantic
 * head), 16-point tiles (half the matrix rate; chosen automatically) up to W = 800; NSOS_ERR_UNSUPPORTED otherwise. */
#define NSOS_GENERIC_MAX_DEPTH 16
#define NSOS_GENERIC_MAX_SEM 8
typedef struct nsos_generic_linear {
    const float* weight;   /* [out_dim, in_dim] */
    const float* bias;     /* [out_dim] */
    int32_t out_dim, in_dim;
} nsos_generic_linear;
typedef struct nsos_generic_mlp {
    int32_t depth, width, skip_mask;
    int32_t xyz_freqs, dir_freqs;              /* octaves; -1: use_embed = False */
    int32_t use_viewdirs, use_semantics, sem_dim, sem_with_coord, sem_with_geo, sem_layers;
    nsos_generic_linear pts[NSOS_GENERIC_MAX_DEPTH];
    nsos_generic_linear alpha, feature, views, rgb;   /* use_viewdirs */
    nsos_generic_linear output;                        /* !use_viewdirs */
    nsos_generic_linear sem[NSOS_GENERIC_MAX_SEM];
    nsos_generic_linear geo[2];
} nsos_generic_mlp;
size_t nsos_mlp_generic_packed_bytes(const nsos_generic_mlp* mlp);      /* 0: unsupported description */
int32_t nsos_mlp_generic_out_channels(const nsos_generic_mlp* mlp);    /* 0: unsupported description */
int32_t nsos_mlp_generic_pack(const nsos_generic_mlp* mlp, void* packed, size_t packed_bytes, void* stream);
/* Weights only, into a buffer nsos_mlp_generic_pack filled before for the same architecture (the program at its head depends on the
 * architecture alone): kernels only -- no host-to-device copy, so a training step that re-packs can be captured in a HIP graph. */
int32_t nsos_mlp_generic_repack(const nsos_generic_mlp* mlp, void* packed, size_t packed_bytes, void* stream);
/* `mlp` must describe the same architecture `packed` was packed for (its tensor pointers are not read here). */
int32_t nsos_mlp_generic_forward_rays(const nsos_generic_mlp* mlp, const void* packed, const float* rays_o, const float* rays_d,
                                      const float* viewdirs, const float* z_vals, int64_t n_rays, int32_t n_samples,
                                      float* raw, void* stream);
int32_t nsos_mlp_generic_forward_points(const nsos_generic_mlp* mlp, const void* packed, const float* pts, const float* dirs,
                                        int64_t n_pts, float* raw, void* stream);

/* ---- K7-G: training a generic-architecture net (every parameter; autograd of models/nerf_mlp.py:67-100 over the points of a ray
 * batch, the model the reference trains in engines/trainer.py:201-203 when its constructor arguments are not the shipped ones).
 *   forward_rays_save  = forward_rays + `acts` [n_rays * n_samples, ld]: per point the two encodings and every Linear's
 *                        post-activation output, each in a column block padded to 32 with zeros (ld and the blocks: save_layout),
 *                        and behind the blocks (ABI 7) the ReLU patterns as BITS: one 32-bit word per output tile of every ReLU
 *                        Linear, in program order (bit r + 16 h = [feature 32 t + (r & 3) + 8 (r >> 2) + 4 h of tile t > 0]); the row
 *                        is padded to a multiple of 4 floats.  The chain reads its masks from these words, never from the blocks.
 *   pack_bwd           the transposed weight streams + the reversed program, into nsos_mlp_generic_bwd_packed_bytes(mlp) bytes
 *                        (re-pack whenever a weight changed, like nsos_mlp_generic_pack).
 *   input_grads        one kernel: g_raw [n_pts, out_channels] (d loss / d raw, e.g. from nsos_composite_backward) -> `gbuf`
 *                        [n_pts, ld]: every Linear's pre-activation gradient in its column block (ReLU masks from `acts`' bit words).
 *   save_layout        table[0] = ld, table[1] = number of Linear ops, then NSOS_GENERIC_LAYOUT_STRIDE ints per op in forward order:
 *                        position of the Linear in nsos_generic_mlp (pts 0..15, alpha 16, feature 17, views 18, rgb 19, output 20,
 *                        sem 21..28, geo 29..30), its column block, out_dim, number of input segments, then per segment (3 slots)
 *                        {column block of the segment's rows in `acts`, rows, first column of the Linear's weight they multiply}.
 *                        Returns the number of ints written.  dW = gbuf[:, block]^T acts[:, segment block] and db = column sums of
 *                        gbuf[:, block] are nsos_wgrad calls (M, N in 32-multiples: the blocks' zero padding makes them exact).
 *   input_grads != 0 at pack time: the chain also carries the gradient into the two positional encodings, and
 *   input_grads_rays   (same kernel) writes d loss / d point [n_pts, 3] and / d view direction [n_pts, 3] (NULL without view
 *                        directions) through them (autograd of models/embedder.py:34-48); nsos_ray_grad_reduce folds them back onto
 *                        the rays: pts = o + d z (models/sampler.py:70,166), viewdirs = d / |d| (models/nerf_net.py:160-163) and the
 *                        renderer's dists * |d| (models/renderer.py:41; from the compositing's own d loss / d sigma in `g_raw` and
 *                        `raw`, + noise * noise_std as the forward added it) -> g_rays_o, g_rays_d [n_rays, 3].
 * Exact-fp32 MFMA like the forward. */
#define NSOS_GENERIC_LAYOUT_STRIDE 13
int32_t nsos_mlp_generic_save_layout(const nsos_generic_mlp* mlp, int32_t* table, int32_t capacity);
int32_t nsos_mlp_generic_forward_rays_save(const nsos_generic_mlp* mlp, const void* packed, const float* rays_o, const float* rays_d,
                                           const float* viewdirs, const float* z_vals, int64_t n_rays, int32_t n_samples,
                                           float* raw, float* acts, void* stream);
/* forward_rays_save for a trainable subset (the mask of nsos_mlp_generic_pack_bwd_subset, bit 31 included): only the blocks that subset's
 * backward reads are stored (the inputs of the trainable Linears, the factors of semantics * mapping) + the bit words, always; the other
 * columns of `acts` stay unwritten.  trainable = 1u << 31 (pose refinement against a frozen net): the bit words alone. */
int32_t nsos_mlp_generic_forward_rays_save_subset(const nsos_generic_mlp* mlp, const void* packed, const float* rays_o, const float* rays_d,
                                                  const float* viewdirs, const float* z_vals, int64_t n_rays, int32_t n_samples,
                                                  float* raw, float* acts, uint32_t trainable, void* stream);
size_t nsos_mlp_generic_bwd_packed_bytes(const nsos_generic_mlp* mlp, int32_t input_grads);   /* 0: unsupported description */
int32_t nsos_mlp_generic_pack_bwd(const nsos_generic_mlp* mlp, void* packed_bwd, size_t packed_bytes, int32_t input_grads, void* stream);
/* The chain for a SUBSET of trainable Linears (bit p of `trainable` = position p of the Linear in nsos_generic_mlp, as in save_layout):
 * only the pre-activation gradients of those Linears and of everything downstream of them are formed -- with a frozen backbone (the
 * shipped recipe, run_nerf.py:307-318) the chain stops at the semantic head.  gbuf blocks of the other Linears are left unwritten.
 * with_header = 0: weights only, as nsos_mlp_generic_repack (same subset as the pack that wrote the header).
 * Bit 31 of `trainable` (ABI 7): the program ALSO carries the gradient into the encodings (as input_grads != 0 of nsos_mlp_generic_pack_bwd;
 * packed_bytes of nsos_mlp_generic_bwd_packed_bytes(mlp, 1)): every gradient of the chain is formed, and only the trainable Linears'
 * are written to gbuf -- a frozen net's chain (trainable = 1u << 31) touches neither the activation blocks nor gbuf. */
int32_t nsos_mlp_generic_pack_bwd_subset(const nsos_generic_mlp* mlp, void* packed_bwd, size_t packed_bytes, uint32_t trainable,
                                         int32_t with_header, void* stream);
int32_t nsos_mlp_generic_repack_bwd(const nsos_generic_mlp* mlp, void* packed_bwd, size_t packed_bytes, int32_t input_grads, void* stream);  /* as nsos_mlp_generic_repack */
int32_t nsos_mlp_generic_input_grads(const nsos_generic_mlp* mlp, const void* packed_bwd, const float* g_raw, const float* acts,
                                     float* gbuf, int64_t n_pts, void* stream);
int32_t nsos_mlp_generic_input_grads_rays(const nsos_generic_mlp* mlp, const void* packed_bwd, const float* g_raw, const float* acts,
                                          float* gbuf, const float* rays_o, const float* rays_d, const float* viewdirs,
                                          const float* z_vals, int64_t n_rays, int32_t n_samples, float* g_pts, float* g_dirs,
                                          void* stream);
/* The same two kernels for the point-query entry (NeRFMLP.forward, models/nerf_mlp.py:179-215) and for MLP.forward's own input, the
 * PRE-ENCODED row [n, input_ch + input_ch_views] (models/nerf_mlp.py:67-68; `encoded` non-NULL: pts / dirs are not read, encodings
 * are taken as given), with saved activations (acts may be NULL: inference) -- and their input gradients (packed with input_grads):
 * d loss / d point and / d direction through the encodings, or (g_encoded non-NULL) d loss / d the encoded inputs themselves. */
int32_t nsos_mlp_generic_forward_points_save(const nsos_generic_mlp* mlp, const void* packed, const float* pts, const float* dirs,
                                             const float* encoded, int64_t n_pts, float* raw, float* acts, void* stream);
int32_t nsos_mlp_generic_input_grads_points(const nsos_generic_mlp* mlp, const void* packed_bwd, const float* g_raw, const float* acts,
                                            float* gbuf, const float* pts, const float* dirs, int64_t n_pts, float* g_pts,
                                            float* g_dirs, float* g_encoded, void* stream);
int32_t nsos_ray_grad_reduce(const float* g_pts, const float* g_dirs, const float* z_vals, const float* rays_d, const float* raw,
                             const float* g_raw, const float* noise, float noise_std, int64_t n_rays, int32_t n_samples,
                             int32_t n_ch, float* g_rays_o, float* g_rays_d, void* stream);

/* ---- K0: pinhole ray generation (SURVEY.md section 8f "next", rank 1) ------------------------------------
 * get_persp_rays (utils/ray.py:12-22; callers data/gen_dataset.py:189,202) for the pixels
 * [pix_begin, pix_end) of an H x W image in row-major order (pixel = j*W + i):
 *   dirs = [(i-cx)/fx, -(j-cy)/fy, -1];  rays_d = dirs @ c2w[:3,:3]^T (unnormalised);  rays_o = c2w[:3,3].
 * c2w_host: HOST pointer to the 12 floats of c2w[:3,:4] row-major (poses are tiny and live on the host);
 * rays_o, rays_d out: device [pix_end-pix_begin, 3].  Removes the [N,H,W,2,3] ray tensors from disk / PCIe. */
int32_t nsos_generate_rays(int32_t H, int32_t W, float fx, float fy, float cx, float cy, const float* c2w_host,
                           int64_t pix_begin, int64_t pix_end, float* rays_o, float* rays_d, void* stream);

/* ---- training-batch assembly on the device ---------------------------------
 * The scene's images (rgbs [n_images,H,W,rgb_ch] fp32), masks ([n_images,H,W,*]: `mask_words` 4-byte words per pixel --
 * int64 labels after the reference's thresholding, data/datasets.py:66-69, are 2 words per channel, float masks 1) and
 * poses ([n_images,pose_rows,pose_cols] fp32 exactly as poses_<split>.npy holds them: LLFF [3,5], blender / toydesk / tankstemple
 * [4,4], data/gen_dataset.py:228-233; rays use [:3,:4]) stay resident in device memory; a batch is gathered per
 * step, with the rays generated from the poses by K0's arithmetic (bit-identical to the reference's stored ray files).
 * Any of (rays_o+rays_d), target, masks_out may be NULL (that output is skipped).
 *
 * nsos_patch_batch: PatchNeRFDataset.__getitem__ (data/datasets.py:240-254) for n_patches items + PatchBatchCollater
 *   (data/collater.py:31-61).  Item b = (image, h_idx, w_idx) = sel[3b..3b+2] -- the caller draws the origins (the reference
 *   uses Python's random.randint(0, H - crop_size), crop_size = patch*stride) and passes them EITHER as a HOST array
 *   (sel_host: validated, travels in the kernel arguments, no copy) OR as a DEVICE array (sel_dev: out-of-range values are
 *   clamped; for captured graphs).  Output pixel (a, c) of item b is image pixel (h_idx + a*stride, w_idx + c*stride):
 *   rays_o / rays_d [n_patches, patch*patch, 3], target [n_patches, patch*patch, rgb_ch], masks_out [.., mask_words words],
 *   poses_out [n_patches, pose_rows, pose_cols] (the items' poses, :251) and start_out [n_patches, 2] = (h_idx, w_idx) as floats
 *   (:252); both optional.
 * nsos_pixel_batch: the same records for an explicit device list of flat pixel indices pix[k] = (image*H + y)*W + x:
 *   RayNeRFDataset items + RayBatchCollater (data/datasets.py:149-171, data/collater.py:7-29) and ViewNeRFDataset's
 *   np.random.choice pixels of one view + ViewBatchCollater (data/datasets.py:279-300, data/collater.py:63-84). */
int32_t nsos_patch_batch(int32_t H, int32_t W, float fx, float fy, float cx, float cy, const float* poses,
                         int32_t pose_rows, int32_t pose_cols, int32_t n_images, const float* rgbs, int32_t rgb_ch, const void* masks,
                         int32_t mask_words, const int32_t* sel_host, const int32_t* sel_dev, int32_t n_patches,
                         int32_t patch, int32_t stride, float* rays_o, float* rays_d, float* target, void* masks_out,
                         float* poses_out, float* start_out, void* stream);
int32_t nsos_pixel_batch(int32_t H, int32_t W, float fx, float fy, float cx, float cy, const float* poses,
                         int32_t pose_rows, int32_t pose_cols, int32_t n_images, const float* rgbs, int32_t rgb_ch, const void* masks,
                         int32_t mask_words, const int64_t* pix, int64_t n, float* rays_o, float* rays_d, float* target,
                         void* masks_out, void* stream);

/* ---- K1: ray set-up -----------------------------------------------------------------------
 * viewdirs = d/|d| (models/nerf_net.py:163-166) and the stratified depths z
 * (StratifiedSampler.forward, models/sampler.py:46-68): z = near(1-t)+far*t, t=linspace(0,1,S);
 * if t_rand != NULL (perturb > 0) jitter inside the mid-point intervals with t_rand [R,S].
 *   rays_d [R,3]; near, far [R]; z_vals out [R,S]; viewdirs out [R,3] (may be NULL). */
int32_t nsos_ray_setup(const float* rays_d, const float* near, const float* far, const float* t_rand,
                       int64_t n_rays, int32_t n_samples, float* z_vals, float* viewdirs, void* stream);

/* pts = o + d*z (models/sampler.py:70,166) -- only for callers that ask for `retpts`; the MLP
 * kernel forms the points on the fly and never reads this tensor.  pts out [R,S,3]. */
int32_t nsos_ray_points(const float* rays_o, const float* rays_d, const float* z_vals, int64_t n_rays,
                        int32_t n_samples, float* pts, void* stream);

/* ---- K2: positional encoding + MLP, fused ---------------------------------------------------
 * PositionEncoder.forward x2 (models/embedder.py:34-48), the encoder join (models/nerf_mlp.py:208),
 * MLP.forward (models/nerf_mlp.py:67-100) and the point-chunk loop (models/nerf_mlp.py:190-210)
 * for R*S points x = o + d*z of R rays, with the per-ray view direction broadcast
 * (models/nerf_net.py:94,111).  raw out [R,S,C], C = 4 (NSOS_SEM_NONE) or 6: [r,g,b,sigma,(sem0,sem1)]. */
int32_t nsos_mlp_forward_rays(const void* packed, int32_t sem_mode, const float* rays_o, const float* rays_d,
                              const float* viewdirs, const float* z_vals, int64_t n_rays, int32_t n_samples,
                              float* raw, void* stream);

/* Same network on explicit points, the NeRFMLP.__call__(pts, viewdirs) form used by density
 * export (engines/eval.py:297; models/nerf_mlp.py:179-215).  pts, dirs [P,3]; raw out [P,C]. */
int32_t nsos_mlp_forward_points(const void* packed, int32_t sem_mode, const float* pts, const float* dirs,
                                int64_t n_pts, float* raw, void* stream);

/* ---- K5: training with a frozen backbone (run_nerf.py:307-318, --fix_backbone) -------------------------------
 * nsos_mlp_forward_rays_save = nsos_mlp_forward_rays that also stores, per point, what the semantic head's
 * backward needs: sem_in out [R*S,320] = [relu(h7) (256) | x63 (63) | 1.0] (the input of semantic_linear.0,
 * models/nerf_mlp.py:79-80, plus a ones column that turns the bias gradient into the same GEMM) and
 * sem_hid out [R*S,128] = relu(semantic_linear.0(...)).  sem_mode must be PLAIN or COORD.
 * nsos_sem_head_backward: the element-wise part of d(semantics)/d(semantic_linear.*) (models/renderer.py:64-66,
 * models/nerf_mlp.py:61): g_logits out [R*S,2] = weights * g_semantics[ray], g_hid out [R*S,128] =
 * (sem_hid > 0) * (g_logits @ semantic_linear.2.weight).  The weight gradients are then plain GEMMs of these
 * with sem_hid / sem_in (done by the host with the BLAS library). */
int32_t nsos_mlp_forward_rays_save(const void* packed, int32_t sem_mode, const float* rays_o, const float* rays_d,
                                   const float* viewdirs, const float* z_vals, int64_t n_rays, int32_t n_samples,
                                   float* raw, float* sem_in, float* sem_hid, void* stream);
/* ---- K7: full backward (every parameter trainable) --------------------------------------------------
 * nsos_mlp_forward_rays_save_all = nsos_mlp_forward_rays that also stores every layer's activations, row-major
 * acts out [R*S, NSOS_ACTS_DIM] fp32, so that the backward is GEMMs + masks over saved data:
 *   [256 l, 256 l + 256)  relu(pts_linears.l(...)), l = 0..7        NSOS_ACTS_FEAT   feature_linear output (no activation)
 *   NSOS_ACTS_VIEWS  relu(views_linears.0(...)) [128]              NSOS_ACTS_SEM    relu(semantic_linear.0(...)) [128]
 *   NSOS_ACTS_X      encoded xyz [63] then 1.0                     NSOS_ACTS_D      encoded view direction [27], zero pad
 * (10.4 KB per point; training only).  Outputs are bit-identical to nsos_mlp_forward_rays. */
#define NSOS_ACTS_FEAT 2048
#define NSOS_ACTS_VIEWS 2304
#define NSOS_ACTS_SEM 2432
#define NSOS_ACTS_X 2560
#define NSOS_ACTS_D 2624
#define NSOS_ACTS_DIM 2656
int32_t nsos_mlp_forward_rays_save_all(const void* packed, int32_t sem_mode, const float* rays_o, const float* rays_d,
                                       const float* viewdirs, const float* z_vals, int64_t n_rays, int32_t n_samples,
                                       float* raw, float* acts, void* stream);
/* Building blocks of the full backward over saved activations (the input-gradient GEMMs g_in = g_out W are plain
 * [P,256]x[256,256] products and go through the BLAS library):
 * nsos_wgrad: dW [M, N] (row stride ldw) = sum_p G[p, 0:M]^T X[p, 0:N], db [M] = sum_p G[p, 0:M] (db may be NULL);
 *   G, X row-major with row strides ldg, ldx (slices of larger buffers); M, N in {32, 64, 128, 256}.  Exact-fp32 MFMA
 *   with K = points; deterministic.  workspace: nsos_wgrad_workspace_bytes().
 * nsos_relu_mask: g[p, c] = h[p, c] > 0 ? g[p, c] : 0 in place over n_cols (multiple of 4) columns. */
size_t nsos_wgrad_workspace_bytes(void);
int32_t nsos_wgrad(const float* G, int32_t ldg, const float* X, int32_t ldx, int64_t n_pts, int32_t M, int32_t N,
                   float* dW, int32_t ldw, float* db, void* workspace, size_t workspace_bytes, void* stream);
int32_t nsos_relu_mask(float* g, int32_t ldg, const float* h, int32_t ldh, int64_t n_pts, int32_t n_cols, void* stream);
/* nsos_wgrad_batch (ABI 7): a list of nsos_wgrad reductions over column blocks of ONE (G, X) pair -- all weight gradients of a
 * generic-architecture net (autograd of every Linear of models/nerf_mlp.py:40-64) in one call.  Item i: dW = out + w_off [M, N] with row
 * stride ldw = sum_p G[p, g_col : g_col + M]^T X[p, x_col : x_col + N], db = out + b_off [M] (b_off < 0: none); offsets in floats.  Items
 * run in list order and share the workspace; each is exactly one nsos_wgrad call (same numbers). */
typedef struct nsos_wgrad_item {
    int64_t w_off, b_off;
    int32_t g_col, x_col, M, N, ldw, reserved;
} nsos_wgrad_item;
int32_t nsos_wgrad_batch(const nsos_wgrad_item* items, int32_t n_items, const float* G, int32_t ldg, const float* X, int32_t ldx,
                         int64_t n_pts, float* out, void* workspace, size_t workspace_bytes, void* stream);
int32_t nsos_sem_head_backward(const float* weights, const float* g_semantics, const float* sem2_w,
                               const float* sem_hid, int64_t n_rays, int32_t n_samples, float* g_hid,
                               float* g_logits, void* stream);
/* The whole backward of the semantic head in one pass over the points (what NeRFNet uses): the element-wise part
 * above fused into the weight-gradient reductions, on the exact-fp32 MFMA.  Out: gw1_aug [128,320] =
 * [d semantic_linear.0.weight (first in_dim columns) | unused | d semantic_linear.0.bias (column 319)],
 * gw2 [2,128] = d semantic_linear.2.weight, gb2 [2] = d semantic_linear.2.bias.  g_hid / g_logits are never
 * materialised.  workspace: nsos_sem_head_wgrad_workspace_bytes() bytes.  Deterministic (block-ordered reduction).
 * gb1 != NULL (round 4): the first output is written as the CONTIGUOUS [128, in_dim] weight gradient instead (in_dim = 256 or
 * 319: semantic_linear.0's fan-in) and the bias gradient goes to gb1 [128] -- the tensors autograd takes, without slicing copies. */
size_t nsos_sem_head_wgrad_workspace_bytes(void);
int32_t nsos_sem_head_wgrad(const float* weights, const float* g_semantics, const float* sem2_w, const float* sem_hid,
                            const float* sem_in, int64_t n_rays, int32_t n_samples, float* gw1_aug, float* gw2,
                            float* gb2, void* workspace, size_t workspace_bytes, float* gb1, int32_t in_dim, void* stream);
/* The same on the 16-bit matrix pipe with split-fp16 operands (fp32 accumulate): HBM-bound instead of MFMA-bound.
 * g_hid = (g_logits @ W_sem2) * mask is brought into fp16 range by a power of two and gw1_aug divided by it again on
 * the way out: `scale` = that power of two as a device scalar, or NULL to have it derived on the device from
 * max |g_semantics| * max_m (|W_sem2[0,m]| + |W_sem2[1,m]|) (one extra small launch).  gw2 / gb2 are plain fp32 sums.
 * sem_in_dtype: NSOS_DTYPE_F32 for the fp32 matrix (three MFMAs per product), NSOS_DTYPE_F16 / NSOS_DTYPE_BF16 for the
 * compact 16-bit matrices of nsos_mlp_forward_rays_save16_lp -- sem_hid is then 16-bit [P,128] in the same format too, fp32
 * [P,128] otherwise -- (half the operand traffic; a 16-bit operand has no lo part:
 * two MFMAs per product on the format's own MFMA; csrc/sem_wgrad16.hip), optionally OR-ed with NSOS_SEM_IN_TILED (below) when
 * sem_in is in the tile-major layout.  n_samples >= 8, n_rays * n_samples < 2^31. */
int32_t nsos_sem_head_wgrad_x3(const float* weights, const float* g_semantics, const float* sem2_w, const void* sem_hid,
                               const void* sem_in, int32_t sem_in_dtype, int64_t n_rays, int32_t n_samples,
                               const float* scale, float* gw1_aug, float* gw2, float* gb2, void* workspace,
                               size_t workspace_bytes, float* gb1, int32_t in_dim, void* stream);

/* ---- K2-LP: the same fused network with 16-bit MFMA inputs and fp32 accumulation (reduced-precision configs) ----
 * For BASELINE configs C3 (bf16) and C5 (fp16, eval-only).  NOT bit/1e-4-comparable with the reference's fp32
 * arithmetic (fp16: ~1e-3 relative, bf16: ~1e-2); never used by the fp32 parity path.  Weights are packed to 16
 * bit by nsos_mlp_pack_lp into their own stream layout (nsos_mlp_packed_bytes_lp bytes).  raw out is fp32. */
enum { NSOS_DTYPE_F32 = 0, NSOS_DTYPE_F16 = 1, NSOS_DTYPE_BF16 = 2 };
size_t nsos_mlp_packed_bytes_lp(int32_t sem_mode);
int32_t nsos_mlp_pack_lp(const nsos_mlp_tensors* tensors, int32_t sem_mode, int32_t dtype, void* packed,
                         size_t packed_bytes, void* stream);
/* Re-pack only what depends on semantic_linear.* (the head's chunks of every stream + the vector-ALU heads' block) into a buffer
 * that already holds a full nsos_mlp_pack_lp of the same trunk: the shipped recipe trains the semantic heads alone
 * (run_nerf.py:307-318), so a training step re-packs 3 chunks per stream instead of 37-40.  sem_mode PLAIN / COORD only. */
int32_t nsos_mlp_pack_lp_heads(const nsos_mlp_tensors* tensors, int32_t sem_mode, int32_t dtype, void* packed,
                               size_t packed_bytes, void* stream);
int32_t nsos_mlp_forward_rays_lp(const void* packed, int32_t sem_mode, int32_t dtype, const float* rays_o,
                                 const float* rays_d, const float* viewdirs, const float* z_vals, int64_t n_rays,
                                 int32_t n_samples, float* raw, void* stream);
/* Training with a frozen backbone at reduced precision (config C3): as nsos_mlp_forward_rays_save, with sem_in
 * holding the 16-bit values the semantic head actually consumed (widened to fp32) and sem_hid its fp32 hidden
 * activations; nsos_sem_head_backward and the weight-gradient GEMMs are the same as for the fp32 path. */
int32_t nsos_mlp_forward_rays_save_lp(const void* packed, int32_t sem_mode, int32_t dtype, const float* rays_o,
                                      const float* rays_d, const float* viewdirs, const float* z_vals, int64_t n_rays,
                                      int32_t n_samples, float* raw, float* sem_in, float* sem_hid, void* stream);
/* The same with BOTH saved matrices kept in the 16-bit format `dtype`: sem_in16 [P,320] halves (640 B per point instead of
 * 1280: the values are 16-bit anyway) and sem_hid16 [P,128] halves (256 B instead of 512: relu of the fp32 accumulators,
 * rounded to nearest even -- the ReLU pattern is unchanged, the values feed only d semantic_linear.2.weight, at the
 * format's precision like everything else on this path).  896 B per point in all (round 2: 1152).
 * Consumer: nsos_sem_head_wgrad_x3 with the matching sem_in_dtype.
 * LAYOUT of sem_in16 -- nsos_mlp_save16_layout(n_points) says which one the call will write:
 *   NSOS_SEM_IN_ROWS  (0):  [P, 320] row-major;
 *   NSOS_SEM_IN_TILED (16): tile-major, the layout the two-waves-per-SIMD kernel (the default) stores without touching 32
 *     different rows per instruction: groups of 32 consecutive points, [group][K 0..19][kg 0..1][point 0..31][8 channels] --
 *     channel 16 K + 8 kg + c of point 32 g + i is element ((g * 20 + K) * 64 + kg * 32 + i) * 8 + c.  The buffer must hold
 *     ceil(P / 32) * 32 rows of 320 elements; rows past P are not written.
 * OR the layout value into nsos_sem_head_wgrad_x3's sem_in_dtype.
 * sem_hid16: [P, 128] row-major, or -- NSOS_SEM_HID_TILED (32) set in the layout value: the 16x16x32 kernel, the default since
 * round 4 -- tile-major like sem_in: [group of 32 points][octet 0..15][point][8 channels] (8 KiB per group; rows past P of the
 * last group are never written): every store instruction of the kernel then writes four runs of 256 contiguous bytes instead of
 * 64 pieces of 16 bytes in as many 256-byte rows (partial-line writes are filled from memory first: the training variant's
 * excess read traffic of round 3). */
enum { NSOS_SEM_IN_ROWS = 0, NSOS_SEM_IN_TILED = 16, NSOS_SEM_HID_TILED = 32 };
int32_t nsos_mlp_save16_layout(int64_t n_points);
int32_t nsos_mlp_forward_rays_save16_lp(const void* packed, int32_t sem_mode, int32_t dtype, const float* rays_o,
                                        const float* rays_d, const float* viewdirs, const float* z_vals, int64_t n_rays,
                                        int32_t n_samples, float* raw, void* sem_in16, void* sem_hid16, void* stream);

/* ---- K2-X3: the same fused network on the 16-bit matrix pipe with SPLIT-fp16 operands (fp32-grade results) -------
 * Every weight and activation is carried as hi = fp16(v), lo = fp16(v - hi) and every product as the three MFMAs
 * hi.hi + hi.lo + lo.hi with fp32 accumulation (the weights' lo parts scaled by 2^11 so that they stay normal fp16
 * numbers): max / rms error against an fp64 evaluation equal to plain fp32 arithmetic's (~1e-7 .. 2e-6 max), i.e.
 * three orders of magnitude inside the 1e-4 parity tolerance, at ~2.8x the exact-fp32 kernel's speed.
 * Limits: |activation| must stay below 65504 (fp16 range).  Opt-in; inference and frozen-backbone training.
 * Weights are packed by nsos_mlp_pack_x3 into their own stream layout (nsos_mlp_packed_bytes_x3 bytes).
 * ABI 8 (round 6): the buffer holds TWO streams -- [aux | the 32x32x16 kernel's (csrc/mlp_x3.hip: training variants, backward
 * chain) | the 16x16x32 kernel's (csrc/mlp_x316.hip: inference; mlp_lp16's workgroup and chunk schedule, three MFMAs per product)];
 * nsos_mlp_pack_x3 writes both, nsos_mlp_forward_rays_x3 runs the selected one (default: 16x16x32).  The two kernels agree to the
 * split format's accuracy (other contraction order inside the MFMAs), not bit for bit. */
size_t nsos_mlp_packed_bytes_x3(int32_t sem_mode);
/* Diagnostics / A-B: 1 = mlp_x3_kernel (32x32x16, one wave per SIMD), 2 = mlp_x316_kernel (16x16x32, two waves per SIMD; default;
 * NSOS_X3_KERNEL=1|2 in the environment chooses at first use).  Applies to nsos_mlp_forward_rays_x3 / nsos_mlp_profile_rays_x3. */
int32_t nsos_mlp_x3_select_kernel(int32_t kernel);
int32_t nsos_mlp_x3_selected_kernel(void);
int32_t nsos_mlp_pack_x3(const nsos_mlp_tensors* tensors, int32_t sem_mode, void* packed, size_t packed_bytes,
                         void* stream);
int32_t nsos_mlp_forward_rays_x3(const void* packed, int32_t sem_mode, const float* rays_o, const float* rays_d,
                                 const float* viewdirs, const float* z_vals, int64_t n_rays, int32_t n_samples,
                                 float* raw, void* stream);
/* Training with a frozen backbone (--fix_backbone, run_nerf.py:307-318) on the split-fp16 kernel: as
 * nsos_mlp_forward_rays_save, with sem_in holding the fp32 values hi + lo the semantic head consumed and sem_hid its
 * fp32 hidden activations; the backward kernels are the fp32 path's.  sem_mode must be 1 or 2. */
int32_t nsos_mlp_forward_rays_save_x3(const void* packed, int32_t sem_mode, const float* rays_o, const float* rays_d,
                                      const float* viewdirs, const float* z_vals, int64_t n_rays, int32_t n_samples,
                                      float* raw, float* sem_in, float* sem_hid, void* stream);
/* Full training (every parameter trainable) on the split-fp16 kernel: as nsos_mlp_forward_rays_save_all, acts holding
 * the fp32 values hi + lo each layer handed to the next.  relu_masks (nsos_mlp_relu_masks_bytes_x3(n_pts) bytes, 16-byte
 * aligned) receives the ReLU patterns of the 8 trunk layers as bits in the kernel's own lane order (32 B per point per
 * layer instead of 1 KB of fp32 reads): the input of nsos_mlp_input_grads_x3. */
size_t nsos_mlp_relu_masks_bytes_x3(int64_t n_pts);
int32_t nsos_mlp_forward_rays_save_all_x3(const void* packed, int32_t sem_mode, const float* rays_o, const float* rays_d,
                                          const float* viewdirs, const float* z_vals, int64_t n_rays, int32_t n_samples,
                                          float* raw, float* acts, void* relu_masks, void* stream);
/* Full training with 16-bit saved activations (round 4): the same kernels with `acts` [P, NSOS_ACTS_DIM] stored as IEEE half floats
 * -- the hi parts of the split-fp16 activations as the MFMAs consumed them -- 5.3 KB per point instead of 10.6.  The chain reads
 * only the two 128-wide heads' ReLU patterns from it (the trunk's come as the forward's bit masks, which are required here); the
 * weight-gradient reductions take it as their X operand (ldx in ELEMENTS): exact widening, G stays fp32. */
int32_t nsos_mlp_forward_rays_save_all16_x3(const void* packed, int32_t sem_mode, const float* rays_o, const float* rays_d,
                                            const float* viewdirs, const float* z_vals, int64_t n_rays, int32_t n_samples,
                                            float* raw, void* acts_f16, void* relu_masks, void* stream);
int32_t nsos_mlp_input_grads_x3_a16(const void* packed, int32_t sem_mode, const float* g_raw, const void* acts_f16,
                                    const void* relu_masks, int64_t n_pts, const float* scale, float* gbuf, void* stream);
int32_t nsos_wgrad_xh(const float* G, int32_t ldg, const void* X_f16, int32_t ldx, int64_t n_pts, int32_t M, int32_t N, float* dW,
                      int32_t ldw, float* db, void* workspace, size_t workspace_bytes, void* stream);
int32_t nsos_wgrad_x3_xh(const float* G, int32_t ldg, const void* X_f16, int32_t ldx, int64_t n_pts, float* dW, int32_t ldw, float* db,
                         void* workspace, size_t workspace_bytes, void* stream);

/* nsos_wgrad for M = N = 256 on the 16-bit matrix pipe: both operands split on the fly into fp16 hi + lo, three MFMAs per
 * product, fp32 accumulation (K7-X3).  |G| and |X| must be within fp16 range (G from nsos_mlp_input_grads_x3 is, by its
 * scale); same workspace, same deterministic reduction as nsos_wgrad. */
int32_t nsos_wgrad_x3(const float* G, int32_t ldg, const float* X, int32_t ldx, int64_t n_pts, float* dW, int32_t ldw,
                      float* db, void* workspace, size_t workspace_bytes, void* stream);

/* ---- K7-X3: the input-gradient chain of the full backward as one fused split-fp16 kernel --------------------------
 * Replaces the library GEMMs g_in = g_out @ W and the ReLU-mask passes of the full backward (autograd of
 * models/nerf_mlp.py:67-100): from g_raw [P,C] (C = 4 or 6: d loss / d [r,g,b,sigma,(sem0,sem1)]) and the activations
 * acts [P,NSOS_ACTS_DIM] saved by nsos_mlp_forward_rays_save_all[_x3] it writes gbuf [P,NSOS_GBUF_DIM], the gradients
 * with respect to every layer's pre-activation in the column map of acts (256 l: pts_linears.l, NSOS_ACTS_FEAT:
 * feature_linear output, NSOS_ACTS_VIEWS: views_linears.0, NSOS_ACTS_SEM: semantic_linear.0), each the GEMM input of
 * nsos_wgrad.  `scale`: THREE powers of two in device memory, applied to g_raw on load -- scale[0] brings max |g_raw| over all
 * channels to ~2^4 (the trunk's columns 256 l of gbuf carry it); scale[1] >= 1 is the colour branch's extra factor (blocks
 * NSOS_ACTS_VIEWS and NSOS_ACTS_FEAT carry scale[0] scale[1]); scale[2] >= 1 the semantic branch's (block NSOS_ACTS_SEM carries
 * scale[0] scale[2]): a branch whose upstream gradient is decades under the largest one keeps fp16's full precision (ABI 6).  The
 * weight gradients must be divided by the scale their gbuf block carries.  relu_masks: the bit masks written by
 * nsos_mlp_forward_rays_save_all_x3 for the same points, or NULL to derive the trunk masks from acts (fp32 reads).
 * Weights are packed (transposed, split fp16) by
 * nsos_mlp_bwd_pack_x3 into nsos_mlp_bwd_packed_bytes_x3 bytes. */
#define NSOS_GBUF_DIM 2560
size_t nsos_mlp_bwd_packed_bytes_x3(int32_t sem_mode);
int32_t nsos_mlp_bwd_pack_x3(const nsos_mlp_tensors* tensors, int32_t sem_mode, void* packed, size_t packed_bytes,
                             void* stream);
int32_t nsos_mlp_input_grads_x3(const void* packed, int32_t sem_mode, const float* g_raw, const float* acts,
                                const void* relu_masks, int64_t n_pts, const float* scale, float* gbuf, void* stream);

/* Diagnostics: nsos_mlp_forward_rays plus per-phase shader-clock stamps (s_memtime) of the first tile of
 * workgroups 0..3: stamps out uint64 [16 waves][64 slots] (slot meaning: scripts/phase_profile.py).
 * Not on the product path; used to attribute the kernel's non-MFMA cycles (profiles/). */
int32_t nsos_mlp_profile_rays(const void* packed, int32_t sem_mode, const float* rays_o, const float* rays_d,
                              const float* viewdirs, const float* z_vals, int64_t n_rays, int32_t n_samples,
                              float* raw, uint64_t* stamps, void* stream);
/* Same for the reduced-precision kernel; it stamps the SECOND tile of workgroups 0..3 (steady state), so give it
 * more than 2 x 256 x (CU count) points.  Slot meaning: scripts/phase_profile_lp.py. */
/* Diagnostics: stamp EVERY following 16-bit MLP launch (the training variants included) into `stamps` (layout of
 * nsos_mlp_profile_rays_lp), or stop with NULL.  scripts/phase_profile_lp.py --save. */
int32_t nsos_mlp_lp_set_stamp_buffer(uint64_t* stamps);

/* Diagnostics: which kernel serves the 16-bit entry points above.  2 (default) = two 256-register waves per SIMD, 32 points
 * each (mlp_lp8.hip); 1 = the round-1 kernel, one 512-register wave per SIMD with 64 points (mlp_lp.hip).  Same packed
 * stream, bit-identical results; exists for A/B measurements (models/nerf_mlp.py:67-100 is what both replace). */
int32_t nsos_mlp_lp_select_kernel(int32_t waves_per_simd);
int32_t nsos_mlp_lp_selected_kernel(void);   /* 3 = mlp_lp16_kernel (default), 2 = mlp_lp8_kernel, 1 = mlp_lp_kernel */

/* ... and for the split-fp16 kernel (128-point tiles: more than 2 x 128 x (CU count) points).  scripts/phase_profile_x3.py. */
int32_t nsos_mlp_profile_rays_x3(const void* packed, int32_t sem_mode, const float* rays_o, const float* rays_d,
                                 const float* viewdirs, const float* z_vals, int64_t n_rays, int32_t n_samples,
                                 float* raw, uint64_t* stamps, void* stream);
int32_t nsos_mlp_profile_rays_lp(const void* packed, int32_t sem_mode, int32_t dtype, const float* rays_o,
                                 const float* rays_d, const float* viewdirs, const float* z_vals, int64_t n_rays,
                                 int32_t n_samples, float* raw, uint64_t* stamps, void* stream);

/* ---- K3: compositing ------------------------------------------------------------------------
 * VolumetricRenderer.forward (models/renderer.py:35-85): sigma->alpha, exclusive transmittance
 * product, weights, and the weighted sums.  noise (may be NULL) is the raw randn [R,S]; it is
 * scaled by noise_std in-kernel (models/renderer.py:47).  n_ch = 4 or 6.
 *   weights out [R,S]; rgb out [R,3]; sem out [R,n_ch-4] (NULL iff n_ch == 4); depth/acc/disp out [R]. */
int32_t nsos_composite(const float* raw, const float* z_vals, const float* rays_d, const float* noise,
                       float noise_std, int64_t n_rays, int32_t n_samples, int32_t n_ch, int32_t white_bkgd,
                       float* weights, float* rgb, float* sem, float* depth, float* acc, float* disp,
                       void* stream);

/* Backward of nsos_composite w.r.t. raw (autograd of models/renderer.py:35-85; z_vals, rays_d and the noise carry no
 * gradient in the reference).  Upstream gradients g_rgb [R,3], g_sem [R,C-4], g_depth [R], g_acc [R], g_disp [R],
 * g_weights [R,S] may each be NULL (= zero).  g_raw out [R,S,C], C = 4 .. 12 (4 + sem_dim).  alpha / transmittance / weights
 * are recomputed. */
int32_t nsos_composite_backward(const float* raw, const float* z_vals, const float* rays_d, const float* noise,
                                float noise_std, int64_t n_rays, int32_t n_samples, int32_t n_ch, int32_t white_bkgd,
                                const float* g_rgb, const float* g_sem, const float* g_depth, const float* g_acc,
                                const float* g_disp, const float* g_weights, float* g_raw, void* stream);

/* nsos_composite on the coarse samples followed by nsos_importance_sample on its weights, in ONE launch (models/nerf_net.py:98-113:
 * renderer -> importance_sampler): same device code, bit-identical outputs, one launch and one round trip of the weights
 * less per step.  n_coarse in [2, 64]. */
int32_t nsos_composite_importance(const float* raw, const float* z_vals, const float* rays_d, const float* noise, float noise_std,
                                  int64_t n_rays, int32_t n_coarse, int32_t n_ch, int32_t white_bkgd, float* weights, float* rgb,
                                  float* sem, float* depth, float* acc, float* disp, const float* u, int32_t n_importance,
                                  float* z_fine, float* z_samples, float* z_std, void* stream);

/* Train-mode random tensors of one ray chunk in ONE launch (optional; NeRFNet.rng = "philox").  The reference draws, in this
 * order, rand[R,S] (stratified jitter, models/sampler.py:61), randn[R,S] (coarse sigma noise, models/renderer.py:47),
 * rand[R,N] (importance u, models/sampler.py:103), randn[R,S+N] (fine sigma noise) from torch's global generator: four
 * launches.  Here a counter-based Philox4x32-10 stream keyed by `seed`, advanced by `call` (one value per chunk / step),
 * fills whichever of the four buffers is not NULL.  NOT torch's values: the default path keeps torch's generator so that
 * the reference's captured draws can be injected (tests/golden/end_to_end.npz). */
int32_t nsos_render_draws(uint64_t seed, uint64_t call, int64_t n_rays, int32_t n_coarse, int32_t n_importance,
                          float* t_rand, float* noise0, float* u, float* noise1, void* stream);
/* The same draws with the call counter in DEVICE memory: uses call = *calls_so_far + 1 and then advances *calls_so_far by one
 * (a second, one-thread launch), so that a captured HIP graph of a training step draws fresh numbers on every replay -- and
 * the same numbers as eager calls of nsos_render_draws with call = 1, 2, 3, ... under the same seed. */
int32_t nsos_render_draws_counted(uint64_t seed, uint64_t* calls_so_far, int64_t n_rays, int32_t n_coarse, int32_t n_importance,
                                  float* t_rand, float* noise0, float* u, float* noise1, void* stream);

/* ---- K4: hierarchical sampling ----------------------------------------------------------------
 * ImportanceSampler.forward / sample_pdf (models/sampler.py:91-167) + z_std (models/nerf_net.py:124):
 * pdf over the inner 62 coarse weights, cdf (fp64-accumulated), right-bisect search of u,
 * lerp inside the bin, merge-sort with the coarse depths, population std of the new samples.
 *   z_vals, weights [R,S] (2 <= S <= 512; S <= 64 -- every shipped config uses 64 -- is the tuned one-sample-per-lane kernel); u [R,N] or NULL (= det: linspace(0,1,N), perturb == 0);
 *   cdf_in [R,S-1] or NULL: if given it REPLACES the computed cdf (stage-wise index pinning);
 *   outputs: z_fine [R,S+N] ascending; z_samples [R,N]; z_std [R];
 *   optional outputs (NULL to skip): cdf_out [R,S-1], inds_out int64 [R,N] (the searchsorted result). */
int32_t nsos_importance_sample(const float* z_vals, const float* weights, const float* u, const float* cdf_in,
                               int64_t n_rays, int32_t n_coarse, int32_t n_importance, float* z_fine,
                               float* z_samples, float* z_std, float* cdf_out, int64_t* inds_out, void* stream);

/* ---- evaluation post-processing ---------------------------------------------
 * What engines/eval.py:44-57,79-86 computes on the host after copying every output off the device:
 *   sem_prob = softmax(semantics, -1); sem_pred = argmax(sem_prob, -1)   (first maximal index, int32)
 *   metrics[0] = img2mse(rgb, target) (utils/image.py:125-128), metrics[1] = mse2psnr(.) (utils/image.py:134-137)
 * semantics [R,sem_dim] may be NULL (no semantic outputs); rgb/target [R,3] may be NULL (no metrics);
 * sem_prob [R,sem_dim] / sem_pred [R] may each be NULL.  workspace: nsos_eval_workspace_bytes() bytes, 8-byte
 * aligned, contents undefined on return except ws[0] = the fp64 sum of per-ray mean squared errors.
 * The reduction order is fixed (no atomics): results are bit-identical run to run. */
size_t nsos_eval_workspace_bytes(void);
int32_t nsos_eval_postprocess(const float* semantics, const float* rgb, const float* target, int64_t n_rays,
                              int32_t sem_dim, float* sem_prob, int32_t* sem_pred, float* metrics, void* workspace,
                              void* stream);

/* Row-partitioned GeoCorrelationLoss for the ray/patch-sharded multi-GPU step (utils/image.py:448-487; call site
 * engines/trainer.py:159-160): every rank holds the WHOLE batch's depth / code / rays (sharding.all_gather_patches) but
 * evaluates the O(P^4) pair sets only for ITS OWN row patches `rows[0..n_rows)` (device int32, global patch ids).  The patches
 * are coupled through four global sums (mean(fd), mean(fd1) per pair set, utils/image.py:316-319) and through the gradient a
 * patch receives as the NEGATIVE of another rank's patch, so the call is split into phases with TWO sum-all-reduces in between:
 *   phase 0: prep + pass 1 + the row-mean residual of the own rows   -> all-reduce means[0..3]  (4 doubles)
 *   phase 1: passes 3, 4 + role sums of the gradient + the loss sums -> all-reduce sums (fp32: batch * n_points * 4 role sums, then the
 *            two loss sums as three fp32 terms each -- an exact split of the rank's fp64 partial -- + 2 spare: nsos_corr_exchange_floats)
 *   phase 2: loss (the batch-wide value, identical on every rank) and grad_code [B,C,H,W] for every patch.
 *   phase 3: phases 0..2 in one call, for a single process (nothing to reduce in between): fewer, merged finishing launches.
 * `means` / `sums` are the workspace's own slots (offsets from nsos_corr_workspace_slots; the workspace is
 * nsos_corr_workspace_bytes(1, ...)) unless exchange_means (8 doubles, 8-byte aligned; all 8 may be reduced) / exchange_sums
 * (nsos_corr_exchange_floats(batch, n_points) floats) are given: a training step hands every loss evaluation of the step a slice
 * of ONE buffer per reduction and issues one all-reduce per phase for all of them (sharding._losses_direct).
 * With rows = all patches and no reductions the phases equal nsos_geo_correlation_loss up to summation order (the loss sums'
 * split is exact; summed over ranks it carries fp32 rounding of each term, <= 1e-7 of the fp32 loss value). */
int64_t nsos_corr_exchange_floats(int32_t batch, int32_t n_points);
int32_t nsos_corr_workspace_slots(int32_t batch, int32_t n_points, int64_t* scal_offset_bytes, int64_t* gsum_offset_bytes,
                                  int64_t* gsum_floats);
int32_t nsos_geo_correlation_loss_rows(int32_t phase, float* depth, const float* code, const float* ray_o, const float* ray_d,
                                       const int64_t* neg_indx, const int32_t* rows, int32_t n_rows, int32_t batch,
                                       int32_t code_dim, int32_t height, int32_t width, float self_shift, float self_weight,
                                       float neg_shift, float neg_weight, float max_depth, int32_t filter_in_place, float* loss,
                                       float* grad_code, void* workspace, size_t workspace_bytes, double* exchange_means,
                                       float* exchange_sums, void* stream);
/* The training step's form of the row-partitioned call (engines/trainer.py:147-166 scores the coarse AND the fine semantic map
 * against the same geometry): TWO codes over ONE set of `batch` geometry patches, evaluated as the stacked batch of 2 * batch
 * patches [code0; code1] -- every term of the loss is a mean over the batch and the batch-wide quantities it subtracts depend
 * on the geometry only, so the result is (L(code0) + L(code1)) / 2 -- without materialising the stack (patch n reads geometry
 * n % batch and code n / batch).  neg_indx [2*batch] and rows index the STACKED batch (neg of patch batch+b = batch + neg of b).
 * channel_last != 0: code0/1 and grad_code0/1 are [batch,H,W,C], ray_o / ray_d [batch,H,W,3] -- the renderer's own tensors;
 * 0: [batch,C,H,W] / [batch,3,H,W].  depth [batch,H*W] is read only (values > max_depth are filtered on the fly).
 * workspace: nsos_corr_workspace_bytes(1, 2*batch, H*W, 0).  Phases and reductions as nsos_geo_correlation_loss_rows. */
int32_t nsos_geo_correlation_loss_pair(int32_t phase, const float* depth, const float* code0, const float* code1,
                                       const float* ray_o, const float* ray_d, const int64_t* neg_indx, const int32_t* rows,
                                       int32_t n_rows, int32_t batch, int32_t channel_last, int32_t code_dim, int32_t height,
                                       int32_t width, float self_shift, float self_weight, float neg_shift, float neg_weight,
                                       float max_depth, float* loss, float* grad_code0, float* grad_code1, void* workspace,
                                       size_t workspace_bytes, double* exchange_means, float* exchange_sums, void* stream);
/* CorrelationLoss (utils/image.py:335-370) row-partitioned the same way: phases 0..2 and the two reductions of
 * nsos_geo_correlation_loss_rows (one process: nsos_app_correlation_loss[_nhwc]).  A row patch needs only its own samples (coords1
 * of n, coords2 of neg[n]); its column gradient lands in patch neg[n], which another rank may own: `sums` carries that gradient for
 * every row (batch * S*S * 4 floats + the split loss sums), and phase 2 writes grad_code for the patches in `rows` only (zeros for
 * the others -- no rank needs them).  channel_last as nsos_app_correlation_loss_nhwc.  workspace: nsos_corr_workspace_bytes(0, ...);
 * rand1 / rand2 must hold the same draws on every rank. */
int32_t nsos_app_correlation_loss_rows(int32_t phase, const float* feats, const float* code, const int64_t* neg_indx,
                                       const float* rand1, const float* rand2, const int32_t* rows, int32_t n_rows, int32_t batch,
                                       int32_t channel_last, int32_t feat_dim, int32_t feat_h, int32_t feat_w, int32_t code_dim,
                                       int32_t code_h, int32_t code_w, int32_t feature_samples, float self_shift, float self_weight,
                                       float neg_shift, float neg_weight, float* loss, float* grad_code, void* workspace,
                                       size_t workspace_bytes, double* exchange_means, float* exchange_sums, void* stream);

/* ---- contrastive loss on the batch's class tokens (BASELINE configs[2]: "contrastive loss") ---------------
 * NeRFContrastive.forward with min_max_contrast=True (utils/image.py:192-218; call site engines/trainer.py:168-170,
 * `contrast_loss(cls_)`): sim = cosine_similarity of embeddings [n_tokens, dim] (each vector divided by max(|e|, 1e-8)
 * first), the off-diagonal entries' minimum and maximum (first occurrence in row-major order, like torch.argmin / argmax),
 * loss [1] = -log(max / (max + min)) -- NaN when max + min < 0, as in the reference.  grad_embeddings (may be NULL):
 * d loss / d embeddings [n_tokens, dim] (only the two picked pairs carry gradient).  2 <= n_tokens <= 120.  One
 * single-workgroup launch, deterministic. */
int32_t nsos_contrastive_loss(const float* embeddings, int32_t n_tokens, int32_t dim, float* loss, float* grad_embeddings,
                              void* stream);

/* The negatives of the correlation losses: similarity = F.cosine_similarity(x[None], x[:, None], dim=2) of the batch's class
 * tokens [n_tokens, dim] (get_similarity_matrix, utils/image.py:186-189; call site engines/trainer.py:125) and
 * negatives[j] = torch.min(similarity, dim=0)[1][j] (utils/image.py:354; first occurrence, a NaN wins) in ONE launch (torch:
 * thirteen).  similarity float [n_tokens, n_tokens] may be NULL.  negatives int64 [copies * n_tokens]: copy c holds
 * negatives + c * n_tokens (the stacked two-map evaluation of nsos_geo_correlation_loss_pair takes copies = 2).
 * 1 <= n_tokens <= 120. */
int32_t nsos_similarity_negatives(const float* tokens, int32_t n_tokens, int32_t dim, float* similarity, int64_t* negatives,
                                  int32_t copies, void* stream);

/* ---- correlation losses on the rendered patches --------------------------------
 * CorrelationLoss.forward (utils/image.py:335-370) and GeoCorrelationLoss.forward (utils/image.py:448-487) for one
 * batch of B patches, with the random choices made by the caller:
 *   neg_indx int64 [B]   the negative patch of each patch (reference: torch.min(sim_matrix, 0)[1], or a permutation)
 *   rand1, rand2 [B,S,S,2] the two torch.rand draws (values in [0,1); the kernel applies the reference's *2-1)
 * loss out [1].  grad_code out (may be NULL): d loss / d code, same shape as code -- the only input that carries
 * gradient in the reference (feats / depth sides are under no_grad).  Nothing of size (H*W)^2 is materialised for the
 * geometric loss.  workspace: nsos_corr_workspace_bytes(kind, B, n_points, feat_dim) bytes, 16-byte aligned
 * (kind 0 = appearance: n_points = S*S; kind 1 = geometric: n_points = H*W <= 4096, feat_dim ignored).
 * code_dim (sem_dim) <= 4.  Deterministic (fixed-order fp64 reductions, no atomics).
 * nsos_geo_correlation_loss: depth [B,1,H,W] is filtered as the reference does (values > max_depth become the
 * largest value < max_depth); with filter_in_place != 0 the filtered values are also written back, which is what
 * the reference does to the caller's tensor (utils/image.py:455). */
size_t nsos_corr_workspace_bytes(int32_t kind, int32_t batch, int32_t n_points, int32_t feat_dim);
int32_t nsos_app_correlation_loss(const float* feats, const float* code, const int64_t* neg_indx, const float* rand1,
                                  const float* rand2, int32_t batch, int32_t feat_dim, int32_t feat_h, int32_t feat_w,
                                  int32_t code_dim, int32_t code_h, int32_t code_w, int32_t feature_samples,
                                  float self_shift, float self_weight, float neg_shift, float neg_weight, float* loss,
                                  float* grad_code, void* workspace, size_t workspace_bytes, void* stream);
/* The same with code [B,Hc,Wc,C] and grad_code in that layout: the renderer's own `semantics` tensor as it is (no
 * permute().contiguous() copy on the way in, none of the gradient on the way out). */
int32_t nsos_app_correlation_loss_nhwc(const float* feats, const float* code, const int64_t* neg_indx, const float* rand1,
                                       const float* rand2, int32_t batch, int32_t feat_dim, int32_t feat_h, int32_t feat_w,
                                       int32_t code_dim, int32_t code_h, int32_t code_w, int32_t feature_samples,
                                       float self_shift, float self_weight, float neg_shift, float neg_weight, float* loss,
                                       float* grad_code, void* workspace, size_t workspace_bytes, void* stream);
int32_t nsos_geo_correlation_loss(float* depth, const float* code, const float* ray_o, const float* ray_d,
                                  const int64_t* neg_indx, int32_t batch, int32_t code_dim, int32_t height,
                                  int32_t width, float self_shift, float self_weight, float neg_shift,
                                  float neg_weight, float max_depth, int32_t filter_in_place, float* loss,
                                  float* grad_code, void* workspace, size_t workspace_bytes, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* NERF_SOS_HIP_H */
