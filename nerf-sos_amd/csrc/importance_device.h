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
        int lo = 0, hi = NB;  // searchsorted(right=True): count of entries <= u
        while (lo < hi) {
            const int mid = (lo + hi) >> 1;
            if (L.cdf[mid] <= u) lo = mid + 1; else hi = mid;
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
        float v[QMAX];
        // padding orders after EVERYTHING, NaNs included (a +inf pad would sort in front of a poisoned ray's NaN samples and
        // push them out of the first N slots): a NaN of its own bit pattern, recognised by the comparator.  Should a
        // sample carry the same pattern it is a NaN like any other, and a NaN is what comes back.
        const float kPad = __uint_as_float(0x7fffffffu);
        auto is_pad = [](float x) { return __float_as_uint(x) == 0x7fffffffu; };
        auto gt = [&](float a, float b) { return is_pad(b) ? false : (is_pad(a) ? true : nan_last_gt(a, b)); };
#pragma unroll
        for (int q = 0; q < QMAX; ++q) v[q] = (q < nq && q * 64 + lane < N) ? smp[q * 64 + lane] : kPad;
        for (int k = 2; k <= np; k <<= 1) {
            for (int j = k >> 1; j > 0; j >>= 1) {
                if (j >= 64) {                       // partner element lives in another register of the same lane
                    const int dq = j >> 6;
#pragma unroll
                    for (int q = 0; q < QMAX; ++q) {
                        if (q < nq && (q & dq) == 0) {
                            const int e = q * 64 + lane;
                            const bool up = (e & k) == 0;
                            const float a = v[q], b = v[q | dq];
                            const bool swap = up ? gt(a, b) : gt(b, a);
                            v[q] = swap ? b : a;
                            v[q | dq] = swap ? a : b;
                        }
                    }
                } else {                             // partner element lives in lane ^ j, same register
#pragma unroll
                    for (int q = 0; q < QMAX; ++q) {
                        if (q < nq) {
                            const int e = q * 64 + lane;
                            const bool up = (e & k) == 0, lower = (lane & j) == 0;
                            const float o = __shfl_xor(v[q], j, NSOS_WAVE);
                            // this lane keeps the smaller of the pair iff (up == lower); select, never fmin/fmax: those drop NaNs
                            const bool take = (up == lower) ? gt(v[q], o) : gt(o, v[q]);
                            v[q] = take ? o : v[q];
                        }
                    }
                }
            }
        }
#pragma unroll
        for (int q = 0; q < QMAX; ++q)
            if (q < nq && q * 64 + lane < N) smp[q * 64 + lane] = v[q];
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
    for (int j = lane; j < S; j += 64) {             // coarse z_j: count of samples strictly below it
        const float z = L.vals[j];
        int lo = 0, hi = N;
        while (lo < hi) {
            const int mid = (lo + hi) >> 1;
            if (nan_last_gt(z, smp[mid])) lo = mid + 1; else hi = mid;
        }
        z_fine[r * M + j + lo] = z;
    }
    for (int i = lane; i < N; i += 64) {             // sample s_k: count of coarse z <= s_k
        const float sv = smp[i];
        int lo = 0, hi = S;
        while (lo < hi) {
            const int mid = (lo + hi) >> 1;
            if (!nan_last_gt(L.vals[mid], sv)) lo = mid + 1; else hi = mid;
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
        double run = (double)pdf;
#pragma unroll
        for (int off = 1; off < 64; off <<= 1) {
            const double o = __shfl_up(run, off, NSOS_WAVE);
            if (lane >= off) run += o;
        }
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

