// Interface of mlp_x316.hip (the split-fp16 forward kernel on v_mfma_f32_16x16x32) towards mlp_x3.hip, which owns the "fp16x3"
// entry points of the C ABI: one packed buffer holds [aux | mlp_x3_kernel's stream | this kernel's stream], nsos_mlp_pack_x3
// writes all of it, and nsos_mlp_forward_rays_x3 dispatches to the selected forward kernel.
#pragma once
#include <cstddef>
#include <cstdint>
#include <hip/hip_runtime.h>

namespace nsos {
namespace x316 {

__host__ __device__ constexpr int x316_chunks(int sem) {
    // L0 (2) + 7 pair layers x 8 + L5 h (8) + L5 x63 (2) + [sem0 h (4) (+ x63 1) + tail (1) | sigma (1)] + views (4)
    return 2 + 56 + 10 + (sem ? 5 + (sem == 2 ? 1 : 0) : 1) + 4;
}
constexpr int kX316TailBytes = 8192;        // behind the chunks: rgb_linear's eight A operands (4 slices x hi, lo), resident in LDS
constexpr size_t kX316SlotBytes = 36 * 1024;
__host__ __device__ constexpr size_t stream_bytes(int sem) { return (size_t)x316_chunks(sem) * kX316SlotBytes + kX316TailBytes; }

int32_t pack(const void* tensors, int32_t sem_mode, unsigned char* chunks, hipStream_t stream);
int32_t launch(const unsigned char* chunks, int32_t sem_mode, const float* rays_o, const float* rays_d, const float* viewdirs,
               const float* z_vals, long long n_pts, int32_t n_samples, float* raw, unsigned long long* prof, hipStream_t stream);

}  // namespace x316
}  // namespace nsos
