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
