// K4 device code shared by the stand-alone importance kernel (sampling.hip) and the fused coarse-compositing +
// importance-sampling kernel (composite.hip).  models/sampler.py:91-167 + models/nerf_net.py:124.
#pragma once
#include "common.h"

// a > b in the total order that puts NaNs last (what torch.sort does)
__device__ __forceinline__ bool nan_last_gt(float a, float b) { return (a > b) || (a != a && b == b); }

// ------------------------------------------------------------------------------------------ K4
// models/sampler.py:91-167 + models/nerf_net.py:124.   One wave per ray, 4 rays per block.
// n_coarse is fixed at 64 (= one sample per lane); n_importance <= NSOS_MAX_IMPORTANCE.
#define NSOS_MAX_IMPORTANCE 448  // 64 + 448 = 512 merged samples per ray at most

template <int CAP_S, int CAP_M>
struct ImportanceLdsT {
    static constexpr int kCapS = CAP_S, kCapM = CAP_M;
    float cdf[CAP_S];    // S - 1 used
    float bins[CAP_S];   // S - 1 used
    float vals[CAP_M];   // coarse z (S) followed by the new samples (N)
};
typedef ImportanceLdsT<64, 512> ImportanceLds;            // the shipped shape: one coarse sample per lane
#define NSOS_MAX_COARSE_WIDE 512
typedef ImportanceLdsT<NSOS_MAX_COARSE_WIDE, NSOS_MAX_COARSE_WIDE + NSOS_MAX_IMPORTANCE + 64> ImportanceLdsWide;   // per-call N_samples > 64

// Everything after the cdf: L.cdf[0..S-1), L.bins[0..S-1) and L.vals[0..S) (the coarse depths) are in LDS.
template <typename LDS>
__device__ __forceinline__ void importance_tail(LDS& L, const int64_t r, const int lane, const float* __restrict__ u_in, int S, int N,
                                                float* __restrict__ z_fine, float* __restrict__ z_samples,
                                                float* __restrict__ z_std, int64_t* __restrict__ inds_out) {
    const int NB = S - 1;
    // invert the cdf (models/sampler.py:116-132): each lane owns samples i = lane, lane+64, ...
    double s1 = 0.0;
    for (int i = lane; i < N; i += 64) {
        const float u = u_in ? u_in[r * N + i] : nsos_linspace01(i, N);
        // searchsorted(right=True): count of entries <= u.  Branch-free descent over powers of two (round 5: the while-loop form cost
        // ~12 instructions and one exec-mask round trip per step; the kernel is issue-bound at one wave per ray): the same count for
        // every nondecreasing cdf.
        int lo = 0;
#pragma unroll
        for (int step = LDS::kCapS / 2; step >= 1; step >>= 1) {
            const int t = lo + step;
            const float c = L.cdf[(t <= NB ? t : NB) - 1];
            lo = (t <= NB && c <= u) ? t : lo;
        }
        if (inds_out) inds_out[r * N + i] = lo;
        const int below = lo - 1 > 0 ? lo - 1 : 0;
        const int above = lo < NB - 1 ? lo : NB - 1;
        const float c0 = L.cdf[below], c1 = L.cdf[above];
        float denom = c1 - c0;
        if (denom < 1e-5f) denom = 1.0f;
        const float t = (u - c0) / denom;
        const float b0 = L.bins[below], b1 = L.bins[above];
        const float span = b1 - b0;
        const float smp = b0 + t * span;
        L.vals[S + i] = smp;
        z_samples[r * N + i] = smp;
        s1 += (double)smp;
    }
    // z_std: population std of the N new samples, two-pass in fp64 (models/nerf_net.py:124)
    const double mean = nsos_wave_sum(s1) / (double)N;
    double s2 = 0.0;
    for (int i = lane; i < N; i += 64) {
        const double dlt = (double)L.vals[S + i] - mean;
        s2 += dlt * dlt;
    }
    s2 = nsos_wave_sum(s2);
    if (lane == 0) z_std[r] = (float)sqrt(s2 / (double)N);
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");

    // merge (models/sampler.py:161: sort(cat([z, samples])), values only).  Both lists are sorted -- the coarse z by
    // construction; the new samples because the inverse cdf is monotone in u: already in order for the deterministic
    // u = linspace (eval), and put in order by an in-wave bitonic sort when u was drawn at random (train) -- so the rank
    // of an element in the union is its position in its own list plus a binary-search count in the other one:
    //   rank(z_i) = i + #{samples <  z_i},     rank(s_k) = k + #{z <= s_k}
    // (ties: only VALUES are returned, so any consistent tie rule gives the reference's output; NaNs order last, as in
    // torch.sort, so a poisoned ray still gets every slot of its row written exactly once).  O(M log M) instead of
    // the O(M^2) rank sort of round 1 (192 x 192 compares per ray: 34.6 us of the 4096-ray step).
    const int M = S + N;
    float* const smp = L.vals + S;
    if (u_in) {
        // bitonic sort of NP = 2^k >= N values (padded), element e = q * 64 + lane lives in register q of `lane`
        constexpr int QMAX = (NSOS_MAX_IMPORTANCE + 63) / 64 + 1;   // 8 registers cover 512 >= 448
        int np = 64;
        while (np < N) np <<= 1;
        const int nq = np >> 6;
        // Sorted as unsigned KEYS (round 5): key(x) is monotone in x for every non-NaN float (sign bit flipped for x >= 0, all bits
        // for x < 0), NaNs and the padding share the largest key -- both order after everything, as torch.sort orders NaNs, and a
        // poisoned ray's NaN samples stay inside the first N slots because only padding can tie with them.  A compare-exchange is then
        // v_min_u32 / v_max_u32 (the float comparator with its NaN and padding cases was ~15 instructions per element and stage:
        // 180 of the 266 us of a 65 536-ray train-mode launch).  A NaN comes back as the canonical quiet NaN.
        unsigned v[QMAX];
        auto key = [](float x) -> unsigned {
            const unsigned b = __float_as_uint(x);
            return (x != x) ? 0xffffffffu : (b ^ ((b >> 31) ? 0xffffffffu : 0x80000000u));
        };
#pragma unroll
        for (int q = 0; q < QMAX; ++q) v[q] = (q < nq && q * 64 + lane < N) ? key(smp[q * 64 + lane]) : 0xffffffffu;
        for (int k = 2; k <= np; k <<= 1) {
            for (int j = k >> 1; j > 0; j >>= 1) {
                if (j >= 64) {                       // partner element lives in another register of the same lane
                    const int dq = j >> 6;
#pragma unroll
                    for (int q = 0; q < QMAX; ++q) {
                        if (q < nq && (q & dq) == 0) {
                            const bool up = ((q * 64 + lane) & k) == 0;
                            const unsigned a = v[q], b = v[q | dq];
                            const unsigned mn = a < b ? a : b, mx = a < b ? b : a;
                            v[q] = up ? mn : mx;
                            v[q | dq] = up ? mx : mn;
                        }
                    }
                } else {                             // partner element lives in lane ^ j, same register
#pragma unroll
                    for (int q = 0; q < QMAX; ++q) {
                        if (q < nq) {
                            const bool up = ((q * 64 + lane) & k) == 0, lower = (lane & j) == 0;
                            const unsigned o = (unsigned)__shfl_xor((int)v[q], j, NSOS_WAVE);
                            const unsigned mn = v[q] < o ? v[q] : o, mx = v[q] < o ? o : v[q];
                            v[q] = (up == lower) ? mn : mx;      // this lane keeps the smaller of the pair iff (up == lower)
                        }
                    }
                }
            }
        }
#pragma unroll
        for (int q = 0; q < QMAX; ++q)
            if (q < nq && q * 64 + lane < N) {
                const unsigned kq = v[q];
                smp[q * 64 + lane] = __uint_as_float(kq ^ ((kq >> 31) ? 0x80000000u : 0xffffffffu));   // (key 0xffffffff -> 0x7fffffff: NaN)
            }
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    }
    // The merge needs both lists in order.  The reference sorts the concatenation and assumes nothing (models/sampler.py:161):
    // caller-supplied depths, near > far, or bins that make the inverse cdf non-monotone all reach this point.  Checked per
    // wave (a handful of LDS reads); an out-of-order ray takes the general rank sort -- O(M^2), correct for any input
    // (ties and NaNs in index order, NaNs last) -- so every slot of its row is still written exactly once.
    bool unsorted = false;
    for (int j = lane; j + 1 < S; j += 64) unsorted |= nan_last_gt(L.vals[j], L.vals[j + 1]);
    for (int i = lane; i + 1 < N; i += 64) unsorted |= nan_last_gt(smp[i], smp[i + 1]);
    if (__builtin_amdgcn_ballot_w64(unsorted) != 0ull) {
        for (int e = lane; e < M; e += 64) {
            const float v = L.vals[e];
            int rank = 0;
            for (int f = 0; f < M; ++f) {
                const float o = L.vals[f];
                rank += (nan_last_gt(v, o) || (!nan_last_gt(o, v) && f < e)) ? 1 : 0;
            }
            z_fine[r * M + rank] = v;
        }
        return;
    }
    for (int j = lane; j < S; j += 64) {             // coarse z_j: count of samples strictly below it (both lists are in order here)
        const float z = L.vals[j];
        int lo = 0;
#pragma unroll
        for (int step = 256; step >= 1; step >>= 1) {        // N <= NSOS_MAX_IMPORTANCE = 448 < 512
            const int t = lo + step;
            const float o = smp[(t <= N ? t : N) - 1];
            lo = (t <= N && nan_last_gt(z, o)) ? t : lo;
        }
        z_fine[r * M + j + lo] = z;
    }
    for (int i = lane; i < N; i += 64) {             // sample s_k: count of coarse z <= s_k
        const float sv = smp[i];
        int lo = 0;
#pragma unroll
        for (int step = LDS::kCapS; step >= 1; step >>= 1) {     // (the count reaches S itself: one more step than the cdf search)
            const int t = lo + step;
            const float o = L.vals[(t <= S ? t : S) - 1];
            lo = (t <= S && !nan_last_gt(o, sv)) ? t : lo;
        }
        z_fine[r * M + i + lo] = sv;
    }
}

// One ray per wave: everything after the coarse compositing.  `wlane` is lane j's coarse weight w[j] (from memory in the
// stand-alone kernel, straight from the compositing registers in the fused one), `z` lane j's coarse depth.
__device__ __forceinline__ void importance_ray(ImportanceLds& L, const int64_t r, const int lane, const float z, const float wlane,
                                               const float* __restrict__ u_in, const float* __restrict__ cdf_in, int S, int N,
                                               float* __restrict__ z_fine, float* __restrict__ z_samples,
                                               float* __restrict__ z_std, float* __restrict__ cdf_out,
                                               int64_t* __restrict__ inds_out) {
    const int NB = S - 1;   // 2 <= S <= 64 coarse samples: one per lane, one cdf entry per lane

    // bins = mid-points (models/sampler.py:155): lane j holds .5*(z[j+1]+z[j]), j < 63
    const float z_next = __shfl_down(z, 1, NSOS_WAVE);
    if (lane < S) L.vals[lane] = z;
    if (lane < NB) L.bins[lane] = 0.5f * (z_next + z);

    // cdf (models/sampler.py:93-97): entry k lives in lane k; entry 0 = 0, entry k>=1 = inclusive
    // fp64 prefix sum of pdf over the inner weights w[1..k]
    float cdf;
    if (cdf_in) {
        cdf = (lane < NB) ? cdf_in[r * NB + lane] : 0.0f;
    } else {
        const bool inner = (lane >= 1 && lane <= NB - 1);
        const float w = inner ? (wlane + 1e-5f) : 0.0f;
        const float fsum = (float)nsos_wave_sum((double)w);
        const float pdf = inner ? (w / fsum) : 0.0f;
        const double run = nsos_wave_scan_incl<false>((double)pdf);   // (DPP scan: common.h)
        cdf = (float)run;  // lane 0: pdf 0 -> 0
    }
    if (lane < NB) {
        L.cdf[lane] = cdf;
        if (cdf_out) cdf_out[r * NB + lane] = cdf;
    }
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    importance_tail(L, r, lane, u_in, S, N, z_fine, z_samples, z_std, inds_out);
}

// The same for 64 < S <= NSOS_MAX_COARSE_WIDE coarse samples (a per-call N_samples override, models/sampler.py:41; no shipped
// config): lane l owns the contiguous entries [l K, (l+1) K), K = ceil(S / 64); the prefix sum is a local fp64 run plus a
// wave scan of the lanes' totals.  A correctness path, not a tuned one.
__device__ __forceinline__ void importance_ray_wide(ImportanceLdsWide& L, const int64_t r, const int lane,
                                                    const float* __restrict__ z_row, const float* __restrict__ w_row,
                                                    const float* __restrict__ u_in, const float* __restrict__ cdf_in, int S, int N,
                                                    float* __restrict__ z_fine, float* __restrict__ z_samples,
                                                    float* __restrict__ z_std, float* __restrict__ cdf_out,
                                                    int64_t* __restrict__ inds_out) {
    const int NB = S - 1;
    for (int j = lane; j < S; j += 64) L.vals[j] = z_row[j];
    for (int j = lane; j < NB; j += 64) L.bins[j] = 0.5f * (z_row[j + 1] + z_row[j]);
    if (cdf_in) {
        for (int j = lane; j < NB; j += 64) L.cdf[j] = cdf_in[r * NB + j];
    } else {
        const int K = (S + 63) >> 6, j0 = lane * K;
        double part = 0.0;
        for (int t = 0; t < K; ++t) {
            const int j = j0 + t;
            if (j >= 1 && j <= NB - 1) part += (double)(w_row[j] + 1e-5f);
        }
        const float fsum = (float)nsos_wave_sum(part);
        double local = 0.0;                                   // this lane's pdf total, then the exclusive scan over the lanes
        for (int t = 0; t < K; ++t) {
            const int j = j0 + t;
            if (j >= 1 && j <= NB - 1) local += (double)((w_row[j] + 1e-5f) / fsum);
        }
        double run = local;
#pragma unroll
        for (int off = 1; off < 64; off <<= 1) {
            const double o = __shfl_up(run, off, NSOS_WAVE);
            if (lane >= off) run += o;
        }
        run -= local;
        for (int t = 0; t < K; ++t) {
            const int j = j0 + t;
            if (j >= 1 && j <= NB - 1) run += (double)((w_row[j] + 1e-5f) / fsum);
            if (j < NB) L.cdf[j] = (float)run;                // entry 0 = 0; entry j = inclusive sum over the inner weights w[1..j]
        }
    }
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    if (cdf_out)
        for (int j = lane; j < NB; j += 64) cdf_out[r * NB + j] = L.cdf[j];
    importance_tail(L, r, lane, u_in, S, N, z_fine, z_samples, z_std, inds_out);
}

