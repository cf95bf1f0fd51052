// NeRFContrastive (utils/image.py:192-218; call site engines/trainer.py:168-170): the contrastive loss on the B class tokens
// of a patch batch -- cosine-similarity matrix, its off-diagonal minimum and maximum, loss = -log(max / (max + min)) -- and
// d loss / d embeddings, in ONE single-workgroup launch (B x 384 values: latency-bound; the reference spends ~12 launches
// on it forward and as many backward).  Deterministic: fixed reduction orders, first-occurrence arg-min / arg-max.
#include "common.h"

#define NSOS_CONTRASTIVE_MAX_B 120   // (B*B + B) floats + the reduction scratch must fit 64 KiB of LDS

struct Pick { float v; int idx; };

// torch.argmax / argmin semantics: a NaN wins (and the first one at that); ties go to the smaller flat index
__device__ __forceinline__ bool better_max(Pick a, Pick b) { return (a.v > b.v) || (a.v != a.v && b.v == b.v) || ((a.v == b.v || (a.v != a.v && b.v != b.v)) && a.idx < b.idx); }
__device__ __forceinline__ bool better_min(Pick a, Pick b) { return (a.v < b.v) || (a.v != a.v && b.v == b.v) || ((a.v == b.v || (a.v != a.v && b.v != b.v)) && a.idx < b.idx); }

__global__ __launch_bounds__(256) void contrastive_kernel(const float* __restrict__ emb, int B, int D, float* __restrict__ loss,
                                                          float* __restrict__ grad) {
    extern __shared__ float lds[];
    float* sim = lds;                       // [B,B]
    float* nrm = lds + B * B;               // [B]   max(|e_i|, eps)
    __shared__ Pick red[2][256];
    __shared__ float coef[2];
    __shared__ int pick[2];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;

    // |e_i| (fp64 accumulation, rounded once), clamped like F.cosine_similarity's eps = 1e-8
    for (int i = wave; i < B; i += 4) {
        double s = 0.0;
        for (int k = lane; k < D; k += 64) { const double x = emb[(size_t)i * D + k]; s += x * x; }
        s = nsos_wave_sum(s);
        if (lane == 0) nrm[i] = fmaxf((float)sqrt(s), 1e-8f);
    }
    __syncthreads();
    // sim[i][j] = sum_k (e_i[k] / n_i) * (e_j[k] / n_j): both vectors are normalised FIRST (ATen does so for stability), each
    // quotient and product rounded to fp32 as there, the sum in fp64.  One wave per pair (i <= j), mirrored.
    const int n_pairs = B * (B + 1) / 2;
    for (int p = wave; p < n_pairs; p += 4) {
        int i = 0, rem = p;                 // row-major upper triangle: row i holds B - i entries
        while (rem >= B - i) { rem -= B - i; ++i; }
        const int j = i + rem;
        const float ni = nrm[i], nj = nrm[j];
        double s = 0.0;
        for (int k = lane; k < D; k += 64) s += (double)((emb[(size_t)i * D + k] / ni) * (emb[(size_t)j * D + k] / nj));
        s = nsos_wave_sum(s);
        if (lane == 0) { sim[i * B + j] = (float)s; sim[j * B + i] = (float)s; }
    }
    __syncthreads();
    // arg-min / arg-max over the off-diagonal entries in row-major order (similarity_matrix[~mask], :205-208)
    Pick bmax = {-__builtin_inff(), 0x7fffffff}, bmin = {__builtin_inff(), 0x7fffffff};
    bool any = false;
    for (int e = tid; e < B * B; e += 256) {
        const int i = e / B, j = e - i * B;
        if (i == j) continue;
        const Pick c = {sim[e], e};
        if (!any || better_max(c, bmax)) bmax = c;
        if (!any || better_min(c, bmin)) bmin = c;
        any = true;
    }
    if (!any) { bmax.idx = bmin.idx = 0x7fffffff; }
    red[0][tid] = bmax; red[1][tid] = bmin;
    __syncthreads();
    for (int off = 128; off >= 1; off >>= 1) {
        if (tid < off) {
            const Pick a0 = red[0][tid], b0 = red[0][tid + off], a1 = red[1][tid], b1 = red[1][tid + off];
            if (b0.idx != 0x7fffffff && (a0.idx == 0x7fffffff || better_max(b0, a0))) red[0][tid] = b0;
            if (b1.idx != 0x7fffffff && (a1.idx == 0x7fffffff || better_min(b1, a1))) red[1][tid] = b1;
        }
        __syncthreads();
    }
    if (tid == 0) {
        const float mx = red[0][0].v, mn = red[1][0].v;
        const float sum = mx + mn, q = mx / sum;                 // :209  -log(max / (max + min)), fp32 like the reference
        *loss = -(float)log((double)q);
        // autograd of the same expression: dL/dq = -1/q; dq/dmax = 1/sum - max/sum^2; dq/dmin = -max/sum^2
        const float dq = -1.0f / q;
        coef[0] = dq * (1.0f / sum - mx / (sum * sum));
        coef[1] = dq * (-(mx / (sum * sum)));
        pick[0] = red[0][0].idx;
        pick[1] = red[1][0].idx;
    }
    __syncthreads();
    if (!grad) return;
    // d sim_ij / d e_i = (e^_j - sim_ij e^_i) / n_i (and i <-> j); only the two picked entries carry gradient.  Thread k owns
    // column k of every row, so the four row updates need no atomics.
    for (int k = tid; k < D; k += 256) {
        for (int i = 0; i < B; ++i) grad[(size_t)i * D + k] = 0.0f;
        for (int t = 0; t < 2; ++t) {
            const int i = pick[t] / B, j = pick[t] - i * B;
            const float s = sim[pick[t]], c = coef[t], ni = nrm[i], nj = nrm[j];
            const float ei = emb[(size_t)i * D + k] / ni, ej = emb[(size_t)j * D + k] / nj;
            grad[(size_t)i * D + k] += c * ((ej - s * ei) / ni);
            grad[(size_t)j * D + k] += c * ((ei - s * ej) / nj);
        }
    }
}

extern "C" int32_t nsos_contrastive_loss(const float* embeddings, int32_t n_tokens, int32_t dim, float* loss, float* grad_embeddings,
                                         void* stream) {
    NSOS_REQUIRE(embeddings && loss, NSOS_ERR_NULL_POINTER);
    NSOS_REQUIRE(n_tokens >= 2 && dim >= 1, NSOS_ERR_BAD_SHAPE);      // B = 1 has no off-diagonal entry (the reference fails on an empty argmin)
    NSOS_REQUIRE(n_tokens <= NSOS_CONTRASTIVE_MAX_B, NSOS_ERR_UNSUPPORTED);
    const size_t lds = ((size_t)n_tokens * n_tokens + n_tokens) * sizeof(float);
    hipLaunchKernelGGL(contrastive_kernel, dim3(1), dim3(256), lds, (hipStream_t)stream, embeddings, n_tokens, dim, loss, grad_embeddings);
    return nsos_launch_status();
}

// The negatives of the correlation losses for a patch batch: similarity_matrix = F.cosine_similarity(x[None], x[:, None], dim=2)
// of the B class tokens (utils/image.py:186-189, engines/trainer.py:125) and neg[j] = torch.min(similarity_matrix, dim=0)[1][j]
// (utils/image.py:354) -- thirteen element-wise / reduction launches of B x B values in torch, one single-workgroup launch here.
// Same arithmetic as contrastive_kernel (both vectors normalised first, fp64 sums); first occurrence wins, a NaN wins.
// neg is written `copies` times, copy c offset by c * B: the stacked two-map evaluation of the geometric loss wants [neg, neg + B].
__global__ __launch_bounds__(256) void similarity_negatives_kernel(const float* __restrict__ emb, int B, int D, float* __restrict__ sim_out,
                                                                   long long* __restrict__ neg, int copies) {
    extern __shared__ float lds[];
    float* sim = lds;                       // [B,B]
    float* nrm = lds + B * B;               // [B]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    for (int i = wave; i < B; i += 4) {
        double s = 0.0;
        for (int k = lane; k < D; k += 64) { const double x = emb[(size_t)i * D + k]; s += x * x; }
        s = nsos_wave_sum(s);
        if (lane == 0) nrm[i] = fmaxf((float)sqrt(s), 1e-8f);
    }
    __syncthreads();
    const int n_pairs = B * (B + 1) / 2;
    for (int p = wave; p < n_pairs; p += 4) {
        int i = 0, rem = p;
        while (rem >= B - i) { rem -= B - i; ++i; }
        const int j = i + rem;
        const float ni = nrm[i], nj = nrm[j];
        double s = 0.0;
        for (int k = lane; k < D; k += 64) s += (double)((emb[(size_t)i * D + k] / ni) * (emb[(size_t)j * D + k] / nj));
        s = nsos_wave_sum(s);
        if (lane == 0) { sim[i * B + j] = (float)s; sim[j * B + i] = (float)s; }
    }
    __syncthreads();
    if (sim_out)
        for (int e = tid; e < B * B; e += 256) sim_out[e] = sim[e];
    for (int j = tid; j < B; j += 256) {
        Pick best = {sim[j], 0};
        for (int i = 1; i < B; ++i) {
            const Pick c = {sim[i * B + j], i};
            if (better_min(c, best)) best = c;
        }
        for (int c = 0; c < copies; ++c) neg[(size_t)c * B + j] = (long long)best.idx + (long long)c * B;
    }
}

extern "C" int32_t nsos_similarity_negatives(const float* tokens, int32_t n_tokens, int32_t dim, float* similarity, int64_t* negatives,
                                             int32_t copies, void* stream) {
    NSOS_REQUIRE(tokens && negatives, NSOS_ERR_NULL_POINTER);
    NSOS_REQUIRE(n_tokens >= 1 && dim >= 1 && copies >= 1, NSOS_ERR_BAD_SHAPE);
    NSOS_REQUIRE(n_tokens <= NSOS_CONTRASTIVE_MAX_B, NSOS_ERR_UNSUPPORTED);
    const size_t lds = ((size_t)n_tokens * n_tokens + n_tokens) * sizeof(float);
    hipLaunchKernelGGL(similarity_negatives_kernel, dim3(1), dim3(256), lds, (hipStream_t)stream, tokens, n_tokens, dim, similarity,
                       reinterpret_cast<long long*>(negatives), copies);
    return nsos_launch_status();
}
