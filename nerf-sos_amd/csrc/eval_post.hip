// Evaluation post-processing on device: what engines/eval.py:44-57,79-86 does after moving
// every output to the host -- softmax + argmax of the rendered semantic logits, and img2mse / mse2psnr
// (utils/image.py:125-137) of the rendered colours against the target view.
//
// HBM-bound and tiny: reads 4*(C+6) B and writes 4*(C+1) B per ray; one thread per ray, grid-stride.  The squared
// error is reduced deterministically: per-ray mean in fp32 (torch.mean over the 3 channels), fp64 partial per
// workgroup into ws[1 + block], then ONE wave adds the partials in a fixed order (no atomics).
#include "common.h"

namespace {
constexpr int kPostBlocks = 256;

__global__ __launch_bounds__(256) void eval_post_kernel(const float* __restrict__ semantics, const float* __restrict__ rgb,
                                                        const float* __restrict__ target, int64_t n_rays, int C,
                                                        float* __restrict__ sem_prob, int32_t* __restrict__ sem_pred,
                                                        double* __restrict__ ws) {
    __shared__ double wave_part[4];
    double part = 0.0;
    for (int64_t r = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; r < n_rays; r += (int64_t)gridDim.x * blockDim.x) {
        if (semantics) {
            const float* s = semantics + r * C;
            float mx = s[0];
            int arg = 0;
            for (int c = 1; c < C; ++c) mx = fmaxf(mx, s[c]);
            float den = 0.0f;
            for (int c = 0; c < C; ++c) den += expf(s[c] - mx);  // softmax(dim=-1), engines/eval.py:55
            float best = -1.0f;
            for (int c = 0; c < C; ++c) {
                const float p = expf(s[c] - mx) / den;
                if (sem_prob) sem_prob[r * C + c] = p;
                if (p > best) { best = p; arg = c; }  // torch.argmax: first maximal index, :56
            }
            if (sem_pred) sem_pred[r] = arg;
        }
        if (rgb && target) {
            const float d0 = rgb[3 * r] - target[3 * r], d1 = rgb[3 * r + 1] - target[3 * r + 1], d2 = rgb[3 * r + 2] - target[3 * r + 2];
            part += (double)(((d0 * d0 + d1 * d1) + d2 * d2) / 3.0f);  // utils/image.py:126
        }
    }
    part = nsos_wave_sum(part);
    if ((threadIdx.x & 63) == 0) wave_part[threadIdx.x >> 6] = part;
    __syncthreads();
    if (threadIdx.x == 0) ws[1 + blockIdx.x] = ((wave_part[0] + wave_part[1]) + wave_part[2]) + wave_part[3];
}

__global__ __launch_bounds__(64) void eval_post_finish_kernel(double* __restrict__ ws, int n_blocks, int64_t n_rays,
                                                              float* __restrict__ metrics) {
    double s = 0.0;
    for (int b = threadIdx.x; b < n_blocks; b += 64) s += ws[1 + b];
    s = nsos_wave_sum(s);
    if (threadIdx.x == 0) {
        ws[0] = s;
        const float mse = (float)(s / (double)n_rays);          // img2mse(..., 'mean'), utils/image.py:128
        metrics[0] = mse;
        metrics[1] = -10.0f * logf(mse) / logf(10.0f);          // mse2psnr, utils/image.py:137
    }
}
}  // namespace

extern "C" size_t nsos_eval_workspace_bytes(void) { return (size_t)(1 + kPostBlocks) * sizeof(double); }

extern "C" int32_t nsos_eval_postprocess(const float* semantics, const float* rgb, const float* target, int64_t n_rays,
                                         int32_t sem_dim, float* sem_prob, int32_t* sem_pred, float* metrics,
                                         void* workspace, void* stream) {
    if (n_rays == 0) return NSOS_OK;
    NSOS_REQUIRE(n_rays > 0, NSOS_ERR_BAD_SHAPE);
    NSOS_REQUIRE(semantics || (rgb && target), NSOS_ERR_NULL_POINTER);
    NSOS_REQUIRE(!semantics || (sem_dim >= 1 && sem_dim <= 64), NSOS_ERR_BAD_SHAPE);
    NSOS_REQUIRE(!semantics || sem_prob || sem_pred, NSOS_ERR_NULL_POINTER);
    const bool want_mse = rgb && target;
    NSOS_REQUIRE(workspace && (!want_mse || metrics), NSOS_ERR_NULL_POINTER);
    NSOS_REQUIRE(((uintptr_t)workspace & 7) == 0, NSOS_ERR_MISALIGNED);
    const int64_t need = (n_rays + 255) / 256;
    const int blocks = (int)(need < kPostBlocks ? need : kPostBlocks);
    double* ws = static_cast<double*>(workspace);
    hipLaunchKernelGGL(eval_post_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, semantics, want_mse ? rgb : nullptr,
                       want_mse ? target : nullptr, n_rays, (int)sem_dim, sem_prob, sem_pred, ws);
    if (want_mse)
        hipLaunchKernelGGL(eval_post_finish_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, ws, blocks, n_rays, metrics);
    return nsos_launch_status();
}
