// K1 (ray set-up + stratified depths), ray points, and K4 (hierarchical sampling) for gfx950.
//
// HBM-bound, tiny next to the MLP (< 1 % of the path) -- the design goal here
// is bit-level agreement with the reference's arithmetic, not throughput:
//   * every fp32 expression is evaluated exactly as the reference writes it (this file is
//     compiled with -ffp-contract=off; the only fused ops are the explicit fmaf of linspace),
//   * sums / scans are accumulated in fp64 and rounded once, which is what ATen's CPU
//     cumsum / std do and within 1 ulp of its sum / norm,
//   * one 64-lane wave owns one ray: the 64 coarse samples map 1:1 onto lanes, so the cdf scan,
//     the bisection and the merge never leave the wave (no block barriers).
#include "common.h"

// ------------------------------------------------------------------------------------------ K0
// Pinhole ray generation on device: utils/ray.py:12-22 get_persp_rays.
//   dirs = [(i-cx)/fx, -(j-cy)/fy, -1];  rays_d[c] = (dirs0*R[c][0] + dirs1*R[c][1]) + dirs2*R[c][2];  rays_o = t
// (ATen sums the three products left to right: verified bitwise).  One thread per pixel of [pix_begin, pix_end).
struct Pose { float m[12]; };  // c2w[:3,:4] row-major, by value in the kernel arguments
__global__ __launch_bounds__(256) void generate_rays_kernel(int W, float fx, float fy, float cx, float cy,
                                                            const Pose c2w, int64_t pix_begin, int64_t n,
                                                            float* __restrict__ rays_o, float* __restrict__ rays_d) {
    const int64_t gid = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (gid >= n) return;
    const int64_t pix = pix_begin + gid;
    const float i = (float)(pix % W), j = (float)(pix / W);
    const float d0 = (i - cx) / fx, d1 = -((j - cy) / fy), d2 = -1.0f;
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        const float p0 = d0 * c2w.m[4 * c], p1 = d1 * c2w.m[4 * c + 1], p2 = d2 * c2w.m[4 * c + 2];
        rays_d[3 * gid + c] = (p0 + p1) + p2;
        rays_o[3 * gid + c] = c2w.m[4 * c + 3];
    }
}

// ------------------------------------------------------------------------------------------ K1
// models/nerf_net.py:163-166 + models/sampler.py:46-68.   grid: one thread per (ray, sample).
__global__ __launch_bounds__(256) void ray_setup_kernel(const float* __restrict__ rays_d,
                                                        const float* __restrict__ near,
                                                        const float* __restrict__ far,
                                                        const float* __restrict__ t_rand, int64_t n_rays, int S,
                                                        float* __restrict__ z_vals, float* __restrict__ viewdirs) {
    const int64_t gid = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (gid >= n_rays * S) return;
    const int64_t r = gid / S;
    const int s = (int)(gid - r * S);
    const float n = near[r], f = far[r];
    auto zlin = [&](int i) {  // models/sampler.py:48   near*(1-t) + far*t
        const float t = nsos_linspace01(i, S);
        return n * (1.0f - t) + f * t;
    };
    float z = zlin(s);
    if (t_rand) {  // models/sampler.py:54-68
        const float zl = zlin(0), zh = zlin(S - 1);
        const float lower = (s == 0) ? zl : 0.5f * (z + zlin(s - 1));
        const float upper = (s == S - 1) ? zh : 0.5f * (zlin(s + 1) + z);
        z = lower + (upper - lower) * t_rand[gid];
    }
    z_vals[gid] = z;
    if (viewdirs && s == 0) {
        const float dx = rays_d[3 * r], dy = rays_d[3 * r + 1], dz = rays_d[3 * r + 2];
        const float nrm = (float)sqrt((double)dx * dx + (double)dy * dy + (double)dz * dz);
        viewdirs[3 * r] = dx / nrm;
        viewdirs[3 * r + 1] = dy / nrm;
        viewdirs[3 * r + 2] = dz / nrm;
    }
}

// models/sampler.py:70,166.   grid: one thread per output float (coalesced stores).
__global__ __launch_bounds__(256) void ray_points_kernel(const float* __restrict__ rays_o,
                                                         const float* __restrict__ rays_d,
                                                         const float* __restrict__ z_vals, int64_t n_rays, int S,
                                                         float* __restrict__ pts) {
    const int64_t gid = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (gid >= n_rays * S * 3) return;
    const int64_t ps = gid / 3;
    const int c = (int)(gid - ps * 3);
    const int64_t r = ps / S;
    const float m = rays_d[3 * r + c] * z_vals[ps];
    pts[gid] = rays_o[3 * r + c] + m;
}

#include "importance_device.h"

__global__ __launch_bounds__(256) void importance_kernel(const float* __restrict__ z_vals,
                                                         const float* __restrict__ weights,
                                                         const float* __restrict__ u_in,
                                                         const float* __restrict__ cdf_in, int64_t n_rays, int S, int N,
                                                         float* __restrict__ z_fine, float* __restrict__ z_samples,
                                                         float* __restrict__ z_std, float* __restrict__ cdf_out,
                                                         int64_t* __restrict__ inds_out) {
    __shared__ ImportanceLds lds_all[4];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int64_t r = (int64_t)blockIdx.x * 4 + wave;
    if (r >= n_rays) return;  // whole wave exits together; no block-level barrier below
    const float z = lane < S ? z_vals[r * S + lane] : 0.0f;
    const float w = (weights && lane < S) ? weights[r * S + lane] : 0.0f;
    importance_ray(lds_all[wave], r, lane, z, w, u_in, cdf_in, S, N, z_fine, z_samples, z_std, cdf_out, inds_out);
}

// 64 < S <= NSOS_MAX_COARSE_WIDE coarse samples per ray (a per-call N_samples override): one wave per ray, two rays per block
__global__ __launch_bounds__(128) void importance_wide_kernel(const float* __restrict__ z_vals, const float* __restrict__ weights,
                                                              const float* __restrict__ u_in, const float* __restrict__ cdf_in,
                                                              int64_t n_rays, int S, int N, float* __restrict__ z_fine,
                                                              float* __restrict__ z_samples, float* __restrict__ z_std,
                                                              float* __restrict__ cdf_out, int64_t* __restrict__ inds_out) {
    __shared__ ImportanceLdsWide lds_all[2];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int64_t r = (int64_t)blockIdx.x * 2 + wave;
    if (r >= n_rays) return;
    importance_ray_wide(lds_all[wave], r, lane, z_vals + r * S, weights ? weights + r * S : nullptr, u_in, cdf_in, S, N, z_fine,
                        z_samples, z_std, cdf_out, inds_out);
}

// ------------------------------------------------------------------------------------------ train-mode draws
// The reference draws four random tensors per ray chunk in train mode (rand[R,S] jitter, randn[R,S] coarse
// sigma noise, rand[R,N] importance u, randn[R,S+N] fine sigma noise: models/sampler.py:61,103, models/renderer.py:47) --
// four generator launches plus their bookkeeping on the host.  This kernel fills all four in ONE launch from a
// counter-based generator (Philox4x32-10, the generator behind torch's own CUDA/HIP streams): element e of the
// concatenated stream is word (e & 3) of block e >> 2 under key = seed, counter = (block, call).  Uniforms are 24-bit
// (k + 0.5) 2^-24 in (0,1); normals are Box-Muller pairs.  Opt-in (NeRFNet.rng = "philox"): the values are NOT torch's, so
// the parity tests (which inject the reference's captured draws) keep the default path.
struct Philox {
    unsigned key[2];
    __device__ __forceinline__ static void round(unsigned (&c)[4], unsigned k0, unsigned k1) {
        const unsigned long long p0 = 0xD2511F53ull * c[0], p1 = 0xCD9E8D57ull * c[2];
        const unsigned h0 = (unsigned)(p0 >> 32), l0 = (unsigned)p0, h1 = (unsigned)(p1 >> 32), l1 = (unsigned)p1;
        c[0] = h1 ^ c[1] ^ k0; c[1] = l1; c[2] = h0 ^ c[3] ^ k1; c[3] = l0;
    }
    __device__ __forceinline__ void operator()(unsigned (&c)[4]) const {
        unsigned k0 = key[0], k1 = key[1];
#pragma unroll
        for (int i = 0; i < 10; ++i) { round(c, k0, k1); k0 += 0x9E3779B9u; k1 += 0xBB67AE85u; }
    }
};
__device__ __forceinline__ float u01(unsigned x) { return ((float)(x >> 8) + 0.5f) * 5.9604644775390625e-08f; }   // (0,1)

__global__ __launch_bounds__(256) void render_draws_kernel(unsigned long long seed, unsigned long long call,
                                                           long long n_uniform0, long long n_normal0, long long n_uniform1,
                                                           long long n_normal1, float* __restrict__ t_rand,
                                                           float* __restrict__ noise0, float* __restrict__ u,
                                                           float* __restrict__ noise1,
                                                           const unsigned long long* __restrict__ calls_so_far) {
    // a device-resident call counter (captured graphs: a by-value `call` would be baked into the graph and every replay
    // would draw the same numbers); the kernel after this one advances it
    if (calls_so_far) call = *calls_so_far + 1ull;
    // one thread = one Philox block = 4 consecutive elements of one of the four tensors (each padded to a multiple of 4)
    const long long b = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    const long long q0 = (n_uniform0 + 3) >> 2, q1 = (n_normal0 + 3) >> 2, q2 = (n_uniform1 + 3) >> 2, q3 = (n_normal1 + 3) >> 2;
    if (b >= q0 + q1 + q2 + q3) return;
    const Philox gen = {{(unsigned)seed, (unsigned)(seed >> 32)}};
    unsigned c[4] = {(unsigned)b, (unsigned)(b >> 32), (unsigned)call, (unsigned)(call >> 32)};
    gen(c);
    float* dst; long long e, n; bool normal;
    if (b < q0) { dst = t_rand; e = b * 4; n = n_uniform0; normal = false; }
    else if (b < q0 + q1) { dst = noise0; e = (b - q0) * 4; n = n_normal0; normal = true; }
    else if (b < q0 + q1 + q2) { dst = u; e = (b - q0 - q1) * 4; n = n_uniform1; normal = false; }
    else { dst = noise1; e = (b - q0 - q1 - q2) * 4; n = n_normal1; normal = true; }
    float v[4];
    if (normal) {
#pragma unroll
        for (int k = 0; k < 2; ++k) {
            const float r = sqrtf(-2.0f * __logf(u01(c[2 * k]))), th = 6.283185307179586f * u01(c[2 * k + 1]);
            v[2 * k] = r * __cosf(th);
            v[2 * k + 1] = r * __sinf(th);
        }
    } else {
#pragma unroll
        for (int k = 0; k < 4; ++k) v[k] = u01(c[k]);
    }
#pragma unroll
    for (int k = 0; k < 4; ++k)
        if (e + k < n) dst[e + k] = v[k];
}

__global__ void counter_add_kernel(unsigned long long* counter, unsigned long long inc) { *counter += inc; }

// ------------------------------------------------------------------------------------------ C ABI
extern "C" int32_t nsos_generate_rays(int32_t H, int32_t W, float fx, float fy, float cx, float cy,
                                      const float* c2w_host, int64_t pix_begin, int64_t pix_end, float* rays_o,
                                      float* rays_d, void* stream) {
    NSOS_REQUIRE(c2w_host, NSOS_ERR_NULL_POINTER);
    NSOS_REQUIRE(H > 0 && W > 0 && pix_begin >= 0 && pix_end >= pix_begin && pix_end <= (int64_t)H * W, NSOS_ERR_BAD_SHAPE);
    NSOS_REQUIRE(fx != 0.0f && fy != 0.0f, NSOS_ERR_BAD_SHAPE);
    const int64_t n = pix_end - pix_begin;
    if (n == 0) return NSOS_OK;
    NSOS_REQUIRE(rays_o && rays_d, NSOS_ERR_NULL_POINTER);
    NSOS_REQUIRE((n + 255) / 256 < (int64_t)1 << 31, NSOS_ERR_UNSUPPORTED);
    Pose pose;
    for (int k = 0; k < 12; ++k) pose.m[k] = c2w_host[k];
    hipLaunchKernelGGL(generate_rays_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, W,
                       fx, fy, cx, cy, pose, pix_begin, n, rays_o, rays_d);
    return nsos_launch_status();
}

extern "C" int32_t nsos_ray_setup(const float* rays_d, const float* near, const float* far, const float* t_rand,
                                  int64_t n_rays, int32_t n_samples, float* z_vals, float* viewdirs,
                                  void* stream) {
    if (n_rays == 0) return NSOS_OK;  // empty batch: nothing to launch (empty tensors have NULL data pointers)
    NSOS_REQUIRE(rays_d && near && far && z_vals, NSOS_ERR_NULL_POINTER);
    NSOS_REQUIRE(n_rays >= 0 && n_samples >= 2, NSOS_ERR_BAD_SHAPE);
    if (n_rays == 0) return NSOS_OK;
    const int64_t total = n_rays * n_samples;
    NSOS_REQUIRE((total + 255) / 256 < (int64_t)1 << 31, NSOS_ERR_UNSUPPORTED);
    hipLaunchKernelGGL(ray_setup_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream,
                       rays_d, near, far, t_rand, n_rays, n_samples, z_vals, viewdirs);
    return nsos_launch_status();
}

extern "C" int32_t nsos_ray_points(const float* rays_o, const float* rays_d, const float* z_vals, int64_t n_rays,
                                   int32_t n_samples, float* pts, void* stream) {
    if (n_rays == 0) return NSOS_OK;  // empty batch: nothing to launch (empty tensors have NULL data pointers)
    NSOS_REQUIRE(rays_o && rays_d && z_vals && pts, NSOS_ERR_NULL_POINTER);
    NSOS_REQUIRE(n_rays >= 0 && n_samples >= 1, NSOS_ERR_BAD_SHAPE);
    if (n_rays == 0) return NSOS_OK;
    const int64_t total = n_rays * n_samples * 3;
    NSOS_REQUIRE((total + 255) / 256 < (int64_t)1 << 31, NSOS_ERR_UNSUPPORTED);
    hipLaunchKernelGGL(ray_points_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream,
                       rays_o, rays_d, z_vals, n_rays, n_samples, pts);
    return nsos_launch_status();
}

extern "C" int32_t nsos_importance_sample(const float* z_vals, const float* weights, const float* u,
                                          const float* cdf_in, int64_t n_rays, int32_t n_coarse,
                                          int32_t n_importance, float* z_fine, float* z_samples, float* z_std,
                                          float* cdf_out, int64_t* inds_out, void* stream) {
    if (n_rays == 0) return NSOS_OK;  // empty batch: nothing to launch (empty tensors have NULL data pointers)
    NSOS_REQUIRE(z_vals && (weights || cdf_in) && z_fine && z_samples && z_std, NSOS_ERR_NULL_POINTER);
    NSOS_REQUIRE(n_rays >= 0 && n_importance >= 1, NSOS_ERR_BAD_SHAPE);
    NSOS_REQUIRE(n_coarse >= 2 && n_coarse <= NSOS_MAX_COARSE_WIDE && n_importance <= NSOS_MAX_IMPORTANCE, NSOS_ERR_UNSUPPORTED);
    NSOS_REQUIRE((n_rays + 1) / 2 < (int64_t)1 << 31, NSOS_ERR_UNSUPPORTED);
    if (n_coarse > 64) {   // more than one coarse sample per lane: the general (untuned) kernel
        hipLaunchKernelGGL(importance_wide_kernel, dim3((unsigned)((n_rays + 1) / 2)), dim3(128), 0, (hipStream_t)stream,
                           z_vals, weights, u, cdf_in, n_rays, n_coarse, n_importance, z_fine, z_samples, z_std, cdf_out, inds_out);
        return nsos_launch_status();
    }
    hipLaunchKernelGGL(importance_kernel, dim3((unsigned)((n_rays + 3) / 4)), dim3(256), 0, (hipStream_t)stream,
                       z_vals, weights, u, cdf_in, n_rays, n_coarse, n_importance, z_fine, z_samples, z_std, cdf_out,
                       inds_out);
    return nsos_launch_status();
}

static int32_t render_draws(uint64_t seed, uint64_t call, uint64_t* counter, int64_t n_rays, int32_t n_coarse, int32_t n_importance,
                            float* t_rand, float* noise0, float* u, float* noise1, void* stream) {
    if (n_rays == 0) return NSOS_OK;
    NSOS_REQUIRE(n_rays > 0 && n_coarse >= 1 && n_importance >= 0, NSOS_ERR_BAD_SHAPE);
    const long long n0 = t_rand ? (long long)n_rays * n_coarse : 0, n1 = noise0 ? (long long)n_rays * n_coarse : 0;
    const long long n2 = u ? (long long)n_rays * n_importance : 0, n3 = noise1 ? (long long)n_rays * (n_coarse + n_importance) : 0;
    const long long blocks4 = ((n0 + 3) >> 2) + ((n1 + 3) >> 2) + ((n2 + 3) >> 2) + ((n3 + 3) >> 2);
    if (blocks4 != 0) {
        NSOS_REQUIRE((blocks4 + 255) / 256 < (1ll << 31), NSOS_ERR_UNSUPPORTED);
        hipLaunchKernelGGL(render_draws_kernel, dim3((unsigned)((blocks4 + 255) / 256)), dim3(256), 0, (hipStream_t)stream,
                           (unsigned long long)seed, (unsigned long long)call, n0, n1, n2, n3, t_rand, noise0, u, noise1,
                           reinterpret_cast<const unsigned long long*>(counter));
    }
    if (counter) hipLaunchKernelGGL(counter_add_kernel, dim3(1), dim3(1), 0, (hipStream_t)stream, reinterpret_cast<unsigned long long*>(counter), 1ull);
    return nsos_launch_status();
}

extern "C" int32_t nsos_render_draws(uint64_t seed, uint64_t call, int64_t n_rays, int32_t n_coarse, int32_t n_importance,
                                     float* t_rand, float* noise0, float* u, float* noise1, void* stream) {
    return render_draws(seed, call, nullptr, n_rays, n_coarse, n_importance, t_rand, noise0, u, noise1, stream);
}

extern "C" int32_t nsos_render_draws_counted(uint64_t seed, uint64_t* calls_so_far, int64_t n_rays, int32_t n_coarse,
                                             int32_t n_importance, float* t_rand, float* noise0, float* u, float* noise1,
                                             void* stream) {
    NSOS_REQUIRE(calls_so_far, NSOS_ERR_NULL_POINTER);
    return render_draws(seed, 0, calls_so_far, n_rays, n_coarse, n_importance, t_rand, noise0, u, noise1, stream);
}
