// Correlation losses on the render path's outputs for gfx950:
//   CorrelationLoss.forward     utils/image.py:335-370  (appearance: DINO features vs rendered semantic logits)
//   GeoCorrelationLoss.forward  utils/image.py:448-487  (geometry: back-projected depth vs rendered semantic logits)
// Both reduce, for a "self" pair set (patch n with itself) and a "negative" pair set (patch n with patch neg[n]),
//   fd[n,p,q]  feature affinity of row point p and column point q      (no gradient)
//   cd[n,p,q]  affinity of the L2-normalised semantic codes of p and q (gradient -> semantics)
//   pointwise (utils/image.py:316-319): fd1 = fd - mean_q fd;  fd2 = (fd1 - mean(fd1)) + mean(fd)
//   loss = mean( -max(cd, 0) * (fd2 - shift) );  total = neg_weight * L_neg + self_weight * L_self.
// The reference materialises every [B,N,N] tensor (N = P*P = 4096 for the geometric loss: 537 MB each, an O(P^4)
// HBM-bound chain of ~20 element-wise kernels).  Here nothing of size N^2 exists for the geometric loss: fd and cd
// are recomputed from LDS-resident points in each of four passes
//   1 row sums of fd            -> row means, mean(fd)
//   2 sum of fl32(fd - rowmean) -> mean(fd1)
//   3 loss sum + gradient w.r.t. the row codes       (thread = row point p, loop over q)
//   4 gradient w.r.t. the column codes               (thread = column point q, loop over p)
// so the kernel is VALU-bound (~25 flop per pair per pass) and reads O(N) bytes.  All reductions run in a fixed order
// (fp32 over 32 consecutive pairs, fp64 across those blocks; no atomics): results are bit-identical run to run.  The appearance loss has N = 121 sample points; its fd
// (a 384-channel dot product) is materialised once ([2,B,121,121]) and the same four passes read it back.
// The forward call also produces d(loss)/d(code) (the loss is only ever back-propagated with a scalar upstream
// gradient), so autograd's backward is one scaling.
// Compiled with -ffp-contract=off: element-wise fp32 expressions follow the reference's op order.
#include <type_traits>
#include "common.h"

namespace {

constexpr int kMaxC = 4;        // semantic code channels supported (sem_dim; the shipped recipes use 2)
constexpr int kRedBlocks = 8192;  // upper bound on partial sums per reduction (64 row blocks x 128 patches)

constexpr int kSumTail = 8;     // floats behind the exchanged gradient sums: 2 loss sums x 3 fp32 terms (+ 2 spare)

struct CorrParams {
    float self_shift, self_weight, neg_shift, neg_weight;
};

// ---- workspace layout (doubles first, then floats), shared by both losses ----------------------------------------
struct Ws {
    double* rowsum;    // [2][B][N]
    double* partial;   // [2][kRedBlocks]  scratch for block partials
    double* scal;      // [16]: 0,1 sum fd (neg,self)  2,3 sum fd1  4,5 loss sums  6 depth max (as double)
    float* pts;        // geo: xyz [B][N][4]           app: unused
    float* cn;         // normalised codes, row side    [B][N][kMaxC]
    float* cn2;        // app only: column-side codes of the negative set [B][N][kMaxC]
    float* dinv;       // 1 / max(||c||, eps) and the norm itself, row side [B][N][2]
    float* dinv2;      // app only, negative column side
    float* grow;       // [2][B][N][kMaxC] gradient w.r.t. normalised row codes
    float* gcol;       // [2][B][N][kMaxC] gradient w.r.t. normalised column codes
    float* fdmat;      // app only: [2][B][N][N]
    float* fn;         // app only: normalised sampled features [2][B][N][Cf]  (0: coords1 of n, 1: coords2 of neg[n])
    float* gsum;       // row-partitioned calls, the slot the ranks sum: geo [B][N][kMaxC] gradient w.r.t. the normalised codes, summed
                       // over roles; app [B][N][kMaxC] gradient w.r.t. the coords2 samples of neg[n] (row n's);  + kSumTail floats:
                       // the two loss sums as three fp32 terms each (exact split of the fp64 partials)
    float* gcolp;      // geo only: [2][B][ceil(N / 64)][kMaxC][N] column-code gradient partials, one per block of 64 rows (fused pass 3)
};

__host__ __device__ inline size_t align16(size_t x) { return (x + 15) & ~(size_t)15; }

inline size_t ws_layout(Ws* w, void* base, int B, int N, int Cf, bool app) {
    size_t off = 0;
    auto take = [&](size_t bytes) { size_t o = off; off = align16(off + bytes); return base ? (char*)base + o : (char*)nullptr; };
    double* rowsum = (double*)take(sizeof(double) * 2 * B * N);
    double* partial = (double*)take(sizeof(double) * 2 * kRedBlocks);
    double* scal = (double*)take(sizeof(double) * 16);
    float* pts = (float*)take(app ? 0 : sizeof(float) * B * N * 4);
    float* cn = (float*)take(sizeof(float) * B * N * kMaxC);
    float* cn2 = (float*)take(app ? sizeof(float) * B * N * kMaxC : 0);
    float* dinv = (float*)take(sizeof(float) * B * N * 2);
    float* dinv2 = (float*)take(app ? sizeof(float) * B * N * 2 : 0);
    float* grow = (float*)take(sizeof(float) * 2 * B * N * kMaxC);
    float* gcol = (float*)take(sizeof(float) * 2 * B * N * kMaxC);
    float* fdmat = (float*)take(app ? sizeof(float) * 2 * B * N * N : 0);
    float* fn = (float*)take(app ? sizeof(float) * 2 * B * N * Cf : 0);
    float* gsum = (float*)take(sizeof(float) * ((size_t)B * N * kMaxC + kSumTail));
    float* gcolp = (float*)take(app ? 0 : sizeof(float) * 2 * B * ((N + 63) / 64) * kMaxC * N);
    if (w) *w = Ws{rowsum, partial, scal, pts, cn, cn2, dinv, dinv2, grow, gcol, fdmat, fn, gsum, gcolp};
    return off;
}

// block-wide fp64 sum in a fixed order; valid in thread 0
__device__ __forceinline__ double block_sum(double v, double* smem) {
    v = nsos_wave_sum(v);
    const int w = threadIdx.x >> 6, nw = (blockDim.x + 63) >> 6;
    if ((threadIdx.x & 63) == 0) smem[w] = v;
    __syncthreads();
    double s = 0.0;
    if (threadIdx.x == 0)
        for (int i = 0; i < nw; ++i) s += smem[i];
    __syncthreads();
    return s;
}

// 1/x for x in [0.05, 1e11]: hardware reciprocal + one Newton step (<= 1 ulp from the correctly rounded quotient the
// reference computes; the IEEE division sequence costs 3x more and the pair passes are VALU-bound)
__device__ __forceinline__ float recip(float x) {
    const float r = __builtin_amdgcn_rcpf(x);
    return __fmaf_rn(r, __fmaf_rn(-x, r, 1.0f), r);
}

// clamped inverse L1 distance, GeoCorrelationLoss.tensor_correlation (utils/image.py:404-413).  The clamp is one v_min_f32
// (a compare + select costs two VALU slots and a wait state in a VALU-bound loop); v_min drops a NaN operand, so NaN inputs
// are caught per POINT in geo_prep_kernel instead (scal[7]) and poison the loss there.
template <int D>
__device__ __forceinline__ float inv_l1(const float (&a)[D], const float* b, float max_depth) {
    float s = fabsf(a[0] - b[0]);
#pragma unroll
    for (int k = 1; k < D; ++k) s = s + fabsf(a[k] - b[k]);   // torch.sum over dim 1, in order
    return __builtin_fminf(recip(s + 5e-2f), max_depth);
}

// The code side of one pair of the geometric loss: cd = clamped inverse L1 affinity of row code c1 and column code c2, and
// v[c] = gt * d cd / d c1[c]  (gt = d total / d cd for this pair; 0 where the clamp is active) -- the term the ROW code's
// gradient collects; the column code's is -v[c].  d cd / d c1[c] = -cd^2 sign(c1[c] - c2[c]) with sign(0) = 0 (torch.abs):
// the sign is med3(df * 2^126, -1, 1) -- exact -1 / 0 / +1 for every normal df -- instead of compare + select + two bit ops.
template <int C>
__device__ __forceinline__ float geo_code_pair(const float (&c1)[C], const float* c2, float max_depth, float gt, float (&v)[C]) {
    float df[C];
#pragma unroll
    for (int c = 0; c < C; ++c) df[c] = c1[c] - c2[c];
    float s = fabsf(df[0]);
#pragma unroll
    for (int c = 1; c < C; ++c) s = s + fabsf(df[c]);
    const float r = recip(s + 5e-2f);
    const float cd = __builtin_fminf(r, max_depth);
    const float k = (r > max_depth ? 0.0f : gt) * (cd * cd);
#pragma unroll
    for (int c = 0; c < C; ++c) v[c] = k * -__builtin_amdgcn_fmed3f(df[c] * 0x1p126f, -1.0f, 1.0f);
    return cd;
}

// ------------------------------------------------------------------------------------------ geometric loss: prep
__global__ __launch_bounds__(256) void depth_max_kernel(const float* __restrict__ depth, long long n, float max_depth,
                                                        double* __restrict__ partial, double* __restrict__ scal) {
    __shared__ double sm[4];
    double m = -1.0e300;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) {
        const float d = depth[i];
        if (d < max_depth && (double)d > m) m = (double)d;   // orig_feats[orig_feats < max_depth].max(), :455
    }
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) { const double o = __shfl_xor(m, off, NSOS_WAVE); m = o > m ? o : m; }
    if ((threadIdx.x & 63) == 0) sm[threadIdx.x >> 6] = m;
    __syncthreads();
    if (threadIdx.x == 0) {
        for (int i = 1; i < (int)(blockDim.x >> 6); ++i) m = sm[i] > m ? sm[i] : m;
        partial[blockIdx.x] = m;
        if (blockIdx.x == 0) {
            scal[7] = 0.0;  // set by geo_prep_kernel when a point's depth / ray / code holds a NaN: the loss is NaN then, as the reference's is
            scal[8] = 0.0;  // the last-arriver tickets of the pair passes (nsos_last_block: two 32-bit counters)
        }
    }
}
// the maximum over depth_max_kernel's block partials (nb <= 256 = the block size of its callers), valid in every thread: what
// depth_max_finish_kernel computed in a launch of its own until round 6 (-1e300 if no element was below max_depth: torch raises
// on the empty max; here the filter yields NaN-free -inf)
__device__ __forceinline__ double depth_max_of(const double* __restrict__ partial, int nb) {
    __shared__ double dm[4];
    double m = (int)threadIdx.x < nb ? partial[threadIdx.x] : -1.0e300;
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) { const double o = __shfl_xor(m, off, NSOS_WAVE); m = o > m ? o : m; }
    if ((threadIdx.x & 63) == 0) dm[threadIdx.x >> 6] = m;
    __syncthreads();
    m = dm[0];
    for (int i = 1; i < (int)(blockDim.x >> 6); ++i) m = dm[i] > m ? dm[i] : m;
    return m;
}

// Exactly ONE workgroup of a grid of `total` gets true (in all its threads): the one that arrives last -- every other workgroup's
// global writes made before its call are visible to it (agent-scope release by each arriver, acquire by the last).  The ticket
// resets itself for the next launch.  Folds a reduction's finishing step into the kernel that produced the partials: a launch of
// its own costs ~4.8 us on the step's critical path (profiles/r05/h_c3_step_timeline.txt), seven of them per training step.
// Round 6, second form (the first one released with __threadfence(): buffer_wbl2 + buffer_inv per workgroup, 66-196 us per step
// slower than the launches it saved, profiles/r06/e_loss_finish_fold_ab.txt): NO cache-wide fence.  What the last workgroup reads --
// the block partials and the row sums -- is written with agent-scope (sc1, write-through) stores and read with agent-scope loads
// (dev_store / dev_load below), so only those few kilobytes travel through memory; "s_waitcnt vmcnt(0)" before the ticket makes each
// wave's own stores complete first, and the ticket itself is a device-scope atomic.
__device__ __forceinline__ void dev_store(double* p, double v) {
    __hip_atomic_store(reinterpret_cast<unsigned long long*>(p), __builtin_bit_cast(unsigned long long, v), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ double dev_load(const double* p) {
    return __builtin_bit_cast(double, __hip_atomic_load(reinterpret_cast<const unsigned long long*>(p), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
}
__device__ __forceinline__ bool nsos_last_block(unsigned* ticket, unsigned total) {
    __shared__ int last_s;
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (threadIdx.x == 0) {
        const unsigned t = __hip_atomic_fetch_add(ticket, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        last_s = t == total - 1u;
        if (last_s) __hip_atomic_store(ticket, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    __syncthreads();
    return last_s != 0;
}

// How a kernel finds the inputs of patch n of a (possibly stacked) batch.  The training step scores TWO semantic maps (coarse
// and fine) against the SAME geometry (engines/trainer.py:147-166): the stacked batch of n_codes * Bg patches is never
// materialised -- patch n takes its depth / rays from geometry patch n % Bg and its code from tensor n / Bg -- and the renderer's
// own channel-last tensors ([B,P,P,C]: `semantics`, rays [B,P,P,3]) are read as they are (channel_last), so neither the
// permute().contiguous() copies nor the repeat() / cat() of the stacked evaluation exist.
struct GeoInputs {
    const float* code[2];     // n_codes tensors [Bg, C, N] (NCHW) or [Bg, N, C] (channel_last)
    float* grad[2];           // d loss / d code, same layout (may be NULL)
    const float* ray_o;       // [Bg, 3, N] or [Bg, N, 3]
    const float* ray_d;
    int Bg, n_codes, channel_last;
    __device__ __forceinline__ size_t code_at(int g, int c, int p, int C, int N) const {
        return channel_last ? ((size_t)g * N + p) * C + c : ((size_t)g * C + c) * N + p;
    }
    __device__ __forceinline__ size_t ray_at(int g, int k, int p, int N) const {
        return channel_last ? ((size_t)g * N + p) * 3 + k : ((size_t)g * 3 + k) * N + p;
    }
};

template <int C>
__global__ __launch_bounds__(256) void geo_prep_kernel(float* __restrict__ depth, const GeoInputs in,
                                                       int B, int N, float max_depth, int write_back,
                                                       const double* __restrict__ scal, float* __restrict__ pts,
                                                       float* __restrict__ cn, float* __restrict__ dinv,
                                                       const double* __restrict__ dmax_partial, int dmax_blocks) {
    const double dmax = depth_max_of(dmax_partial, dmax_blocks);      // (before the early return: a block-wide reduction)
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (long long)B * N) return;
    const int n = (int)(i / N), p = (int)(i % N);
    const int g = n % in.Bg, which = n / in.Bg;
    float d = depth[(size_t)g * N + p];
    if (d > max_depth) {  // :455
        d = (float)dmax;
        if (write_back && which == 0) depth[(size_t)g * N + p] = d;
    }
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        const size_t j = in.ray_at(g, k, p, N);
        const float m = in.ray_d[j] * d;
        pts[i * 4 + k] = in.ray_o[j] + m;   // depth2pts, :443
    }
    pts[i * 4 + 3] = 0.0f;
    float v[C], ss = 0.0f;
    const float* code = in.code[which];
#pragma unroll
    for (int c = 0; c < C; ++c) {
        v[c] = code[in.code_at(g, c, p, C, N)];
        ss = c == 0 ? v[c] * v[c] : ss + v[c] * v[c];
    }
    const float nrm = sqrtf(ss), den = fmaxf(nrm, 1e-10f);  // F.normalize(dim=1, eps=1e-10), :301
#pragma unroll
    for (int c = 0; c < kMaxC; ++c) cn[i * kMaxC + c] = c < C ? v[c] / den : 0.0f;
    dinv[i * 2] = den;
    dinv[i * 2 + 1] = nrm;
    float probe = d + ss;                                         // NaN anywhere in this point's inputs? (the pair passes clamp with
#pragma unroll                                                    //  v_min_f32, which would drop it)
    for (int k = 0; k < 3; ++k) probe += pts[i * 4 + k];
    if (probe != probe) const_cast<double*>(scal)[7] = 1.0;       // same value from every writer
}

// ------------------------------------------------------------------------------------------ the four pair passes
// grid = (ceil(N / 256), B, 2 sets); set 0 = negative pairs (patch n rows, patch neg[n] columns), set 1 = self.
// GEO: fd from LDS-resident xyz; otherwise fd from the materialised matrix.
struct PairArgs {
    int B, N, C;
    const long long* neg;
    const float* pts;      // [B][N][4]
    const float* cn;       // row codes [B][N][kMaxC]
    const float* cn2;      // app: negative-set column codes (indexed by n); geo: NULL (columns are cn[neg[n]])
    const float* fdmat;    // app
    double* rowsum;
    double* partial;
    double* scal;
    float* grow;
    float* gcol;
    float max_depth;
    CorrParams prm;
    const int* rows;       // row patches this call evaluates (blockIdx.y indexes it), or NULL = all B patches
    int n_rows;
    float* gcolp;          // geo, fused pass 3: column-gradient partials [2][row slots][row blocks][C][N]
    int Bg;                // geo: geometry patches behind the B code patches (patch n -> geometry n % Bg); 0 = B
    // single-process calls (round 6): the passes' finishing reductions run in the LAST workgroup of the pass itself (nsos_last_block)
    // instead of in launches of their own.  tickets = two zeroed 32-bit counters (workspace scal[8]), nb = block partials per set.
    unsigned* tickets;     // NULL: finish kernels are launched (the row-partitioned multi-GPU phases)
    int nb;
    double cnt;            // pass 3: B N N
    float* loss;           // pass 3: where the loss goes
    const double* flags;   // pass 3, geo: scal of the workspace ([7] = NaN flag)
    int geo;
};

__device__ __forceinline__ int row_patch(const PairArgs& A) { return A.rows ? A.rows[blockIdx.y] : (int)blockIdx.y; }

// Pass 1 of the geometric loss (row sums of fd) depends on the GEOMETRY of the row and the column patch only.  When several
// code maps are scored against one geometry (GeoInputs: B = n_codes * Bg), pair sets with the same (row geometry, column
// geometry) have bit-identical row sums: only the first such (set, row slot) in launch order computes them, the others copy
// (pair_rowsum_copy_kernel).  In the C3 step all four pair sets share one fd matrix.
__device__ __forceinline__ bool first_with_same_geometry(const PairArgs& A, int set, int slot, int* cset, int* cslot) {
    if (A.Bg <= 0 || A.Bg >= A.B) return false;
    const int nr = A.rows ? A.n_rows : A.B;
    const int n = A.rows ? A.rows[slot] : slot, m = set == 0 ? (int)A.neg[n] : n;
    const int kr = n % A.Bg, kc = m % A.Bg;
    for (int z = 0; z <= set; ++z)
        for (int y = 0; y < (z < set ? nr : slot); ++y) {
            const int n2 = A.rows ? A.rows[y] : y, m2 = z == 0 ? (int)A.neg[n2] : n2;
            if (n2 % A.Bg == kr && m2 % A.Bg == kc) { *cset = z; *cslot = y; return true; }
        }
    return false;
}

template <bool GEO>
__device__ __forceinline__ const float* col_codes(const PairArgs& A, int set, int n) {
    if (GEO) return A.cn + (size_t)(set == 0 ? (int)A.neg[n] : n) * A.N * kMaxC;
    return (set == 0 ? A.cn2 : A.cn) + (size_t)n * A.N * kMaxC;
}

// PASS 1: row sums.  PASS 3: loss + row-code gradient.  (PASS 2, sum fl32(fd - rowmean), is no longer launched: rowmean_residual_kernel.)
// GEO: a workgroup = kRows row points x kSlabs column slabs, 1024 threads = four waves per SIMD (thread (row, slab) loops over
// the slab's N / kSlabs columns); the slabs' fp64 partials are folded in slab order through LDS -- the dynamic region, reused
// once every wave is done with the column image.  The column image (up to 144 KiB) allows ONE workgroup per CU, so the
// threads of that workgroup are all the latency hiding there is: 256 threads (one wave per SIMD) ran the C4 step's pair
// kernels in 0.80 ms, 1024 in 0.45 ms.  Rows per workgroup: 64, or 32 (NARROW) when 64 would leave CUs without a workgroup
// (one or two row patches per GPU in the sharded step; each workgroup re-stages the image, so narrow only when needed).
template <bool GEO, bool NARROW = false>
struct PairShape {
    // appearance loss (N = 121 sample points): 32 rows x 4 column slabs -- one workgroup of 128 rows looping over all 121
    // columns was a grid of two workgroups per patch and a chain of 121 dependent iterations (10-18 us per pass for 15 k pairs)
    static constexpr int kThreads = GEO ? 1024 : 128;
    static constexpr int kRows = GEO ? (NARROW ? 32 : 64) : 32;
    static constexpr int kSlabs = kThreads / kRows;
};

// Sum over the wave's 64 lanes (= 64 row points) of R registers (= R consecutive columns), leaving the R totals in v[0]:
// a transposing reduction -- each stage pairs register i with register i + n/2, lanes of one class keep the first and hand
// the second to a lane of the other class (v_permlane32_swap / v_permlane16_swap exchange whole halves / 16-lane rows of a
// register pair; quad permutes for lane bits 0 and 1), so the register count halves while every sum stays complete -- then
// rotations inside the 16-lane rows add the lanes that hold the same column.  ~2.2 VALU per summed value instead of the 6
// of a butterfly per register; fixed order.  Column j ends in EVERY lane with  col_of_lane(lane) == j.
template <int R>
__device__ __forceinline__ void wave_transpose_sum(float (&v)[R]) {
    static_assert(R == 8 || R == 16, "8 or 16 columns per group");
    const unsigned lane = threadIdx.x & 63;
    // (asm: through the builtin every swap cost two register copies and a wait state; the operands here were written
    //  many instructions earlier, and the leading s_nop covers the first one)
    asm volatile("s_nop 1");
#pragma unroll
    for (int i = 0; i < R / 2; ++i) {                      // lane bit 5: lanes 0-31 keep i, lanes 32-63 keep i + R/2
        asm volatile("v_permlane32_swap_b32 %0, %1" : "+v"(v[i]), "+v"(v[i + R / 2]));
    }
#pragma unroll
    for (int i = 0; i < R / 2; ++i) v[i] = v[i] + v[i + R / 2];
    asm volatile("s_nop 1");
#pragma unroll
    for (int i = 0; i < R / 4; ++i) {                      // lane bit 4: even rows of 16 keep i, odd rows keep i + R/4
        asm volatile("v_permlane16_swap_b32 %0, %1" : "+v"(v[i]), "+v"(v[i + R / 4]));
    }
#pragma unroll
    for (int i = 0; i < R / 4; ++i) v[i] = v[i] + v[i + R / 4];
    const bool b0 = lane & 1, b1 = lane & 2;
#pragma unroll
    for (int i = 0; i < R / 8; ++i) {                      // lane bit 0 (quad_perm [1,0,3,2])
        const float keep = b0 ? v[i + R / 8] : v[i], give = b0 ? v[i] : v[i + R / 8];
        v[i] = keep + __uint_as_float(__builtin_amdgcn_update_dpp(0u, __float_as_uint(give), 0xB1, 0xF, 0xF, false));
    }
    if (R == 16) {                                         // lane bit 1 (quad_perm [2,3,0,1])
        const float keep = b1 ? v[1] : v[0], give = b1 ? v[0] : v[1];
        v[0] = keep + __uint_as_float(__builtin_amdgcn_update_dpp(0u, __float_as_uint(give), 0x4E, 0xF, 0xF, false));
    } else {
        v[0] = v[0] + __uint_as_float(__builtin_amdgcn_update_dpp(0u, __float_as_uint(v[0]), 0x4E, 0xF, 0xF, false));
    }
    v[0] = v[0] + __uint_as_float(__builtin_amdgcn_update_dpp(0u, __float_as_uint(v[0]), 0x124, 0xF, 0xF, false));   // row_ror:4
    v[0] = v[0] + __uint_as_float(__builtin_amdgcn_update_dpp(0u, __float_as_uint(v[0]), 0x128, 0xF, 0xF, false));   // row_ror:8
}
template <int R>
__device__ __forceinline__ int col_of_lane(unsigned lane) {
    const int b5 = (lane >> 5) & 1, b4 = (lane >> 4) & 1, b0 = lane & 1, b1 = (lane >> 1) & 1;
    return R == 16 ? 8 * b5 + 4 * b4 + 2 * b0 + b1 : 4 * b5 + 2 * b4 + b0;
}

// FUSE (GEO, PASS 3, 64 rows): the gradient w.r.t. the COLUMN codes comes out of this pass too.  For the clamped inverse L1
// affinity d cd / d c2 = -d cd / d c1, so the column-side term of pair (p, q) is minus the row-side term the thread just
// formed: every group of R columns is summed over the wave's 64 rows (wave_transpose_sum) and written as one partial per
// (block of 64 rows, column); pair_cols_fold_kernel adds the row blocks in fp64, in order.  That replaces the fourth pass
// over all N^2 pairs (pair_cols_kernel: 88 of the 220 us the geometric loss took per C3 step) by ~4 VALU per pair here.
__host__ __device__ inline int pair_padded_columns(int N, int slabs) { return ((N + slabs * 32 - 1) / (slabs * 32)) * 32 * slabs; }

__device__ __forceinline__ void pass1_finish_in_block(const PairArgs& A, double* red);
__device__ __forceinline__ void pass3_finish_in_block(const PairArgs& A);

template <bool GEO, int C, int PASS, bool NARROW = false, bool FUSE = false>
__global__ __launch_bounds__((PairShape<GEO, NARROW>::kThreads)) void pair_rows_kernel(const PairArgs A) {
    static_assert(!FUSE || (GEO && PASS == 3 && !NARROW), "the fused column gradient needs 64-row waves of the geometric pass 3");
    extern __shared__ __attribute__((aligned(16))) float lds[];   // columns: GEO xyz [N][4] then codes [N][kMaxC]
    __shared__ double red[4];
    constexpr int kSlabs = PairShape<GEO, NARROW>::kSlabs, kRows = PairShape<GEO, NARROW>::kRows;
    double* const slab = reinterpret_cast<double*>(lds);   // the slab fold reuses the column image once every wave is done with it
    const int set = blockIdx.z, n = row_patch(A), N = A.N;
    const int m = set == 0 ? (int)A.neg[n] : n;
    if constexpr (GEO && PASS == 1) {
        int cset, cslot;
        if (first_with_same_geometry(A, set, blockIdx.y, &cset, &cslot)) return;   // whole workgroup: copied afterwards
    }
    // FUSE: the image is padded with zeros to whole slabs (<= 4096 columns whenever N <= 4096), so that the pair loop reads
    // column q + j unconditionally at  scalar base + immediate  (a pair past N contributes through a zeroed t only)
    const int Nimg = FUSE ? pair_padded_columns(N, kSlabs) : N;
    float* lx = lds;
    float* lc = lds + (GEO ? (size_t)Nimg * 4 : 0);
    if (GEO)
        for (int i = threadIdx.x; i < Nimg * 4; i += blockDim.x) lx[i] = (!FUSE || i < N * 4) ? A.pts[(size_t)m * N * 4 + i] : 0.0f;
    if (PASS == 3) {
        const float* cc = col_codes<GEO>(A, set, n);
        for (int i = threadIdx.x; i < Nimg * kMaxC; i += blockDim.x) lc[i] = (!FUSE || i < N * kMaxC) ? cc[i] : 0.0f;
    }
    __syncthreads();
    const int rl = threadIdx.x % kRows, sl = threadIdx.x / kRows;       // row within the block, column slab
    const int p = blockIdx.x * kRows + rl;
    const bool live = p < N;
    const int pc = live ? p : N - 1;
    const size_t row = (size_t)n * N + pc;
    float x[3] = {0, 0, 0}, c1[C];
    if (GEO) { x[0] = A.pts[row * 4]; x[1] = A.pts[row * 4 + 1]; x[2] = A.pts[row * 4 + 2]; }
#pragma unroll
    for (int c = 0; c < C; ++c) c1[c] = A.cn[row * kMaxC + c];
    const float* fdrow = GEO ? nullptr : A.fdmat + (((size_t)set * A.B + n) * N + pc) * N;
    const double cnt = (double)A.B * N * N;
    float rm = 0.0f, m1 = 0.0f, old_mean = 0.0f;
    if (PASS >= 2) rm = (float)(A.rowsum[((size_t)set * A.B + n) * N + pc] / (double)N);   // fd.mean([3,4]), :318
    if (PASS == 3) { old_mean = (float)(A.scal[set] / cnt); m1 = (float)(A.scal[2 + set] / cnt); }
    const float shift = set == 0 ? A.prm.neg_shift : A.prm.self_shift;
    const float gscale = -(set == 0 ? A.prm.neg_weight : A.prm.self_weight) / (float)cnt;
    // sums: fp32 over blocks of kBlk consecutive q (fixed order), blocks folded into fp64
    constexpr int kBlk = 32;
    const int per = ((N + kSlabs * kBlk - 1) / (kSlabs * kBlk)) * kBlk;   // columns per slab, a multiple of kBlk
    const int qb = sl * per, qend = qb + per < N ? qb + per : N;
    double acc = 0.0;
    double g[C];
#pragma unroll
    for (int c = 0; c < C; ++c) g[c] = 0.0;
    if constexpr (FUSE) {
        constexpr int R = C <= 2 ? 16 : 8;                 // columns per transposed group: R * C values live per lane
        const unsigned lane = threadIdx.x & 63;
        const bool store_lane = (lane & 12) == 0 && (R == 16 || (lane & 2) == 0);
        const int jcol = col_of_lane<R>(lane);
        float* const gp = A.gcolp + (((size_t)set * gridDim.y + blockIdx.y) * gridDim.x + blockIdx.x) * C * N;
        const float gs = live ? gscale : 0.0f;             // rows past N are copies of row N - 1: no column terms from them
        const int qb_u = __builtin_amdgcn_readfirstlane(qb), qend_u = __builtin_amdgcn_readfirstlane(qend);   // 64 rows: slab = wave,
        for (int q0 = qb_u; q0 < qend_u; q0 += kBlk) {                 // so q is wave-uniform: LDS addresses are scalar + immediate
            const int qe = q0 + kBlk < qend_u ? q0 + kBlk : qend_u;
            float acc32 = 0.0f, g32[C];
#pragma unroll
            for (int c = 0; c < C; ++c) g32[c] = 0.0f;
#pragma unroll 1
            for (int q = q0; q < qe; q += R) {
                float val[C][R];
#pragma unroll
                for (int j = 0; j < R; ++j) {
                    const bool ok = q + j < qe;             // ragged last block: the pair does not exist (scalar test)
                    const int qq = q + j;
                    const float fd = inv_l1<3>(x, lx + qq * 4, A.max_depth);
                    const float fd1 = fd - rm;
                    const float fd2 = (fd1 - m1) + old_mean;
                    const float t = ok ? fd2 - shift : 0.0f;
                    float v[C];
                    const float cd = geo_code_pair<C>(c1, lc + qq * kMaxC, A.max_depth, gs * t, v);
                    acc32 += -cd * t;                       // cd > 0: the reference's clamp(0) never acts on this affinity
#pragma unroll
                    for (int c = 0; c < C; ++c) {
                        val[c][j] = v[c];
                        g32[c] += v[c];
                    }
                    if ((j & 3) == 3) __builtin_amdgcn_sched_barrier(0);   // four pairs in flight, not sixteen (VGPR budget: 128; measured: no slower)
                }
#pragma unroll
                for (int c = 0; c < C; ++c) {
                    wave_transpose_sum<R>(val[c]);
                    if (store_lane && q + jcol < qe) gp[(size_t)c * N + q + jcol] = val[c][0];
                }
            }
            acc += (double)acc32;
#pragma unroll
            for (int c = 0; c < C; ++c) g[c] += (double)g32[c];
        }
    } else
    for (int q0 = qb; q0 < qend; q0 += kBlk) {
        const int qe = q0 + kBlk < qend ? q0 + kBlk : qend;
        float acc32 = 0.0f, g32[C];
#pragma unroll
        for (int c = 0; c < C; ++c) g32[c] = 0.0f;
#pragma unroll 4
        for (int q = q0; q < qe; ++q) {
            const float fd = GEO ? inv_l1<3>(x, lx + q * 4, A.max_depth) : fdrow[q];
            if (PASS == 1) { acc32 += fd; continue; }
            const float fd1 = fd - rm;                         // fd -= fd.mean([3,4]), :318
            if (PASS == 2) { acc32 += fd1; continue; }
            const float fd2 = (fd1 - m1) + old_mean;           // fd - fd.mean() + old_mean, :319
            const float t = fd2 - shift;
            if constexpr (GEO) {
                float v[C];
                const float cd = geo_code_pair<C>(c1, lc + q * kMaxC, A.max_depth, gscale * t, v);   // overridden tensor_correlation, :427
                acc32 += -cd * t;                              // cd.clamp(0) is the identity here, :330
#pragma unroll
                for (int c = 0; c < C; ++c) g32[c] += v[c];
            } else {
                float cd = c1[0] * lc[q * kMaxC];
#pragma unroll
                for (int c = 1; c < C; ++c) cd = cd + c1[c] * lc[q * kMaxC + c];
                const float cdc = cd < 0.0f ? 0.0f : cd;       // cd.clamp(0), :330
                acc32 += -cdc * t;
                const float gcd = cd >= 0.0f ? gscale * t : 0.0f;   // d total / d cd
#pragma unroll
                for (int c = 0; c < C; ++c) g32[c] += gcd * lc[q * kMaxC + c];
            }
        }
        acc += (double)acc32;
#pragma unroll
        for (int c = 0; c < C; ++c) g[c] += (double)g32[c];
    }
    if (kSlabs > 1) {   // fold the slabs in slab order: slab 0's lanes own the row
        __syncthreads();
        if (sl > 0) {
            slab[((sl - 1) * kRows + rl) * (1 + kMaxC)] = acc;
#pragma unroll
            for (int c = 0; c < C; ++c) slab[((sl - 1) * kRows + rl) * (1 + kMaxC) + 1 + c] = g[c];
        }
        __syncthreads();
        if (sl == 0)
            for (int k = 1; k < kSlabs; ++k) {
                acc += slab[((k - 1) * kRows + rl) * (1 + kMaxC)];
#pragma unroll
                for (int c = 0; c < C; ++c) g[c] += slab[((k - 1) * kRows + rl) * (1 + kMaxC) + 1 + c];
            }
    }
    const bool owner = live && sl == 0;
    if (PASS == 1 && owner) {
        if (A.tickets) dev_store(&A.rowsum[((size_t)set * A.B + n) * N + p], acc);     // (read by this launch's last workgroup)
        else A.rowsum[((size_t)set * A.B + n) * N + p] = acc;
    }
    if (PASS == 3 && owner)
#pragma unroll
        for (int c = 0; c < kMaxC; ++c) A.grow[(((size_t)set * A.B + n) * N + p) * kMaxC + c] = c < C ? (float)g[c] : 0.0f;
    const double s = block_sum(owner ? acc : 0.0, red);
    if (threadIdx.x == 0) {
        if (A.tickets) dev_store(&A.partial[(size_t)set * kRedBlocks + blockIdx.y * gridDim.x + blockIdx.x], s);
        else A.partial[(size_t)set * kRedBlocks + blockIdx.y * gridDim.x + blockIdx.x] = s;
    }
    // single-process calls: the pass's finishing reduction in its last workgroup (PASS 1 of a stacked geometric batch finishes in
    // pair_rowsum_copy_kernel instead, which runs between the two)
    if (A.tickets != nullptr && (PASS == 3 || !(GEO && A.Bg > 0 && A.Bg < A.B))) {
        if (nsos_last_block(A.tickets + (PASS == 1 ? 0 : 1), gridDim.x * gridDim.y * gridDim.z)) {
            if constexpr (PASS == 1) {
                __shared__ double fin[16];
                pass1_finish_in_block(A, fin);
            } else {
                pass3_finish_in_block(A);
            }
        }
    }
}

__global__ __launch_bounds__(256) void pair_rowsum_copy_kernel(const PairArgs A, int row_blocks) {
    int cset, cslot;
    const int set = blockIdx.z, N = A.N;
    if (!first_with_same_geometry(A, set, blockIdx.y, &cset, &cslot)) return;
    const int n = row_patch(A), cn = A.rows ? A.rows[cslot] : cslot;
    const int p = blockIdx.x * blockDim.x + threadIdx.x;
    if (p < N) A.rowsum[((size_t)set * A.B + n) * N + p] = A.rowsum[((size_t)cset * A.B + cn) * N + p];
    if (blockIdx.x == 0)
        for (int k = threadIdx.x; k < row_blocks; k += blockDim.x)
            A.partial[(size_t)set * kRedBlocks + blockIdx.y * row_blocks + k] = A.partial[(size_t)cset * kRedBlocks + cslot * row_blocks + k];
}
// the same with pass 1's finishing reduction in the last workgroup (single-process calls; blocks that copy nothing take part too)
__global__ __launch_bounds__(256) void pair_rowsum_copy_finish_kernel(const PairArgs A, int row_blocks) {
    int cset, cslot;
    const int set = blockIdx.z, N = A.N;
    if (first_with_same_geometry(A, set, blockIdx.y, &cset, &cslot)) {
        const int n = row_patch(A), cn = A.rows ? A.rows[cslot] : cslot;
        const int p = blockIdx.x * blockDim.x + threadIdx.x;
        if (p < N) dev_store(&A.rowsum[((size_t)set * A.B + n) * N + p], A.rowsum[((size_t)cset * A.B + cn) * N + p]);
        if (blockIdx.x == 0)
            for (int k = threadIdx.x; k < row_blocks; k += blockDim.x)
                dev_store(&A.partial[(size_t)set * kRedBlocks + blockIdx.y * row_blocks + k], A.partial[(size_t)cset * kRedBlocks + cslot * row_blocks + k]);
    }
    if (nsos_last_block(A.tickets, gridDim.x * gridDim.y * gridDim.z)) {
        __shared__ double fin[16];
        pass1_finish_in_block(A, fin);
    }
}

// What used to be PASS 2: scal[2 + set] = sum over the evaluated pairs of (fd - rowmean), the numerator of the mean the
// reference subtracts after centring (fd - fd.mean() + old_mean, utils/image.py:319).  That mean is what is left of a sum that
// is zero in exact arithmetic: per row, rowsum - N * fl32(rowsum / N) (the rounding of the row mean) plus per-element rounding
// noise of order 1e-11 of fd.  The first part is available from the row sums -- one thread block per set instead of a fourth
// pass over all N^2 pairs (69 of 530 us in the C4 step); the second is below anything the loss can see (|m1| <~ 1e-8 |fd|
// against a parity bar of 1e-4) and differs between the reference's own CPU and GPU reductions anyway.
__device__ __forceinline__ double rowmean_residual_sum(const PairArgs& A, int set) {
    const int N = A.N, nr = A.rows ? A.n_rows : A.B;
    double s = 0.0;
    for (long long k = threadIdx.x; k < (long long)nr * N; k += blockDim.x) {
        const int n = A.rows ? A.rows[k / N] : (int)(k / N), p = (int)(k % N);
        const double rs = A.rowsum[((size_t)set * A.B + n) * N + p];
        const float rm = (float)(rs / (double)N);                       // fd.mean([3,4]) as the row passes use it
        s += rs - (double)N * (double)rm;
    }
    return s;
}
__global__ __launch_bounds__(256) void rowmean_residual_kernel(const PairArgs A) {
    __shared__ double red[4];
    const double s = block_sum(rowmean_residual_sum(A, blockIdx.x), red);
    if (threadIdx.x == 0) A.scal[2 + blockIdx.x] = s;
}

// scal[slot + set] = sum of the block partials, fixed order (64 lanes, each a strided sub-sum, then a fixed tree)
__device__ __forceinline__ double partial_sum(const double* __restrict__ partial, int nb, int set) {   // wave 0; valid in every lane
    double s = 0.0;
    for (int i = threadIdx.x; i < nb; i += 64) s += partial[(size_t)set * kRedBlocks + i];
    return nsos_wave_sum(s);
}
__global__ void pair_finish_kernel(const double* __restrict__ partial, int nb, double* __restrict__ scal, int slot) {
    const double s = partial_sum(partial, nb, blockIdx.x);
    if (threadIdx.x == 0) scal[slot + blockIdx.x] = s;
}
// The same two finishing steps as device functions for the LAST workgroup of a pass (any block size that is a multiple of 64):
// identical summation orders, so the folded and the launched forms give the same bits.
// partial_sum with agent-scope loads (the partials were written by other workgroups of the SAME launch)
__device__ __forceinline__ double partial_sum_dev(const double* __restrict__ partial, int nb, int set) {
    double s = 0.0;
    for (int i = threadIdx.x; i < nb; i += 64) s += dev_load(&partial[(size_t)set * kRedBlocks + i]);
    return nsos_wave_sum(s);
}
__device__ __forceinline__ void pass1_finish_in_block(const PairArgs& A, double* red /* [16] */) {
    for (int set = 0; set < 2; ++set) {
        if (threadIdx.x < 64) {
            const double s = partial_sum_dev(A.partial, A.nb, set);
            if (threadIdx.x == 0) A.scal[set] = s;
        }
        // rowmean_residual_sum strides by blockDim.x: pass1_finish_kernel runs it with 1024 threads -- keep that order
        double r = 0.0;
        {
            const int N = A.N, nr = A.rows ? A.n_rows : A.B;
            for (int v = threadIdx.x; v < 1024; v += blockDim.x) {          // virtual thread v of the 1024-thread form
                double sv = 0.0;
                for (long long k = v; k < (long long)nr * N; k += 1024) {
                    const int n = A.rows ? A.rows[k / N] : (int)(k / N), p = (int)(k % N);
                    const double rs = dev_load(&A.rowsum[((size_t)set * A.B + n) * N + p]);
                    const float rm = (float)(rs / (double)N);
                    sv += rs - (double)N * (double)rm;
                }
                // fold the virtual wave (64 consecutive virtual threads = this wave's lanes) exactly as block_sum does
                sv = nsos_wave_sum(sv);
                if ((threadIdx.x & 63) == 0) red[v >> 6] = sv;
            }
        }
        __syncthreads();
        if (threadIdx.x == 0) {
            for (int i = 0; i < 16; ++i) r += red[i];
            A.scal[2 + set] = r;
        }
        __syncthreads();
    }
}
__device__ __forceinline__ void pass3_finish_in_block(const PairArgs& A) {
    if (threadIdx.x >= 64) return;
    const double s0 = partial_sum_dev(A.partial, A.nb, 0), s1 = partial_sum_dev(A.partial, A.nb, 1);
    if (threadIdx.x != 0) return;
    A.scal[4] = s0;
    A.scal[5] = s1;
    const float l_neg = (float)(s0 / A.cnt), l_self = (float)(s1 / A.cnt);
    A.loss[0] = A.prm.neg_weight * l_neg + A.prm.self_weight * l_self;
    if (A.geo && A.flags[7] != 0.0) A.loss[0] = __builtin_nanf("");
}

// single-process call: what follows pass 1 in ONE launch (block = set): scal[set] = sum of the pass's block partials (wave 0,
// the same order as pair_finish_kernel) and scal[2 + set] = the row-mean residual (all 1024 threads)
__global__ __launch_bounds__(1024) void pass1_finish_kernel(const PairArgs A, int nb) {
    __shared__ double red[16];
    const int set = blockIdx.x;
    if (threadIdx.x < 64) {
        const double s = partial_sum(A.partial, nb, set);
        if (threadIdx.x == 0) A.scal[set] = s;
    }
    const double r = block_sum(rowmean_residual_sum(A, set), red);
    if (threadIdx.x == 0) A.scal[2 + set] = r;
}

// PASS 4: gradient w.r.t. the column codes: thread = column point q of pair (n -> m), loop over the row points p
// (GEO: split into kSlabs row slabs per workgroup like the row passes, folded in slab order).
template <bool GEO, int C, bool NARROW = false>
__global__ __launch_bounds__((PairShape<GEO, NARROW>::kThreads)) void pair_cols_kernel(const PairArgs A) {
    extern __shared__ __attribute__((aligned(16))) float lds[];   // rows: xyz [N][4] (GEO), codes [N][kMaxC], rowmean [N]
    constexpr int kSlabs = PairShape<GEO, NARROW>::kSlabs, kRows = PairShape<GEO, NARROW>::kRows;
    double* const slab = reinterpret_cast<double*>(lds);   // the slab fold reuses the row image once every wave is done with it
    const int set = blockIdx.z, n = row_patch(A), N = A.N;
    const int m = set == 0 ? (int)A.neg[n] : n;
    float* lx = lds;
    float* lc = lds + (GEO ? (size_t)N * 4 : 0);
    float* lr = lc + (size_t)N * kMaxC;
    if (GEO)
        for (int i = threadIdx.x; i < N * 4; i += blockDim.x) lx[i] = A.pts[(size_t)n * N * 4 + i];
    for (int i = threadIdx.x; i < N * kMaxC; i += blockDim.x) lc[i] = A.cn[(size_t)n * N * kMaxC + i];
    for (int i = threadIdx.x; i < N; i += blockDim.x) lr[i] = (float)(A.rowsum[((size_t)set * A.B + n) * N + i] / (double)N);
    __syncthreads();
    const int ql = threadIdx.x % kRows, sl = threadIdx.x / kRows;
    const int q = blockIdx.x * kRows + ql;
    const bool live = q < N;
    const int qc = live ? q : N - 1;
    float y[3] = {0, 0, 0}, c2[C];
    if (GEO) { const size_t col = (size_t)m * N + qc; y[0] = A.pts[col * 4]; y[1] = A.pts[col * 4 + 1]; y[2] = A.pts[col * 4 + 2]; }
    const float* cc = col_codes<GEO>(A, set, n);
#pragma unroll
    for (int c = 0; c < C; ++c) c2[c] = cc[(size_t)qc * kMaxC + c];
    const double cnt = (double)A.B * N * N;
    const float old_mean = (float)(A.scal[set] / cnt), m1 = (float)(A.scal[2 + set] / cnt);
    const float shift = set == 0 ? A.prm.neg_shift : A.prm.self_shift;
    const float gscale = -(set == 0 ? A.prm.neg_weight : A.prm.self_weight) / (float)cnt;
    const float* fdcol = GEO ? nullptr : A.fdmat + ((size_t)set * A.B + n) * N * N + qc;
    constexpr int kBlk = 32;
    const int per = ((N + kSlabs * kBlk - 1) / (kSlabs * kBlk)) * kBlk;
    const int pb = sl * per, pend = pb + per < N ? pb + per : N;
    double g[C];
#pragma unroll
    for (int c = 0; c < C; ++c) g[c] = 0.0;
    for (int p0 = pb; p0 < pend; p0 += kBlk) {
        const int pe = p0 + kBlk < pend ? p0 + kBlk : pend;
        float g32[C];
#pragma unroll
        for (int c = 0; c < C; ++c) g32[c] = 0.0f;
#pragma unroll 4
        for (int p = p0; p < pe; ++p) {
            // fd(p, q) with the ROW point first, exactly as the row passes evaluate it
            float fd;
            if (GEO) { const float xp[3] = {lx[p * 4], lx[p * 4 + 1], lx[p * 4 + 2]}; fd = inv_l1<3>(xp, y, A.max_depth); }
            else fd = fdcol[(size_t)p * N];
            const float fd2 = ((fd - lr[p]) - m1) + old_mean;
            const float t = fd2 - shift;
            float c1[C];
#pragma unroll
            for (int c = 0; c < C; ++c) c1[c] = lc[p * kMaxC + c];
            if constexpr (GEO) {
                float v[C];
                geo_code_pair<C>(c1, c2, A.max_depth, gscale * t, v);
#pragma unroll
                for (int c = 0; c < C; ++c) g32[c] += -v[c];   // d cd / d c2 = -d cd / d c1
            } else {
                float cd = c1[0] * c2[0];
#pragma unroll
                for (int c = 1; c < C; ++c) cd = cd + c1[c] * c2[c];
                const float gcd = cd >= 0.0f ? gscale * t : 0.0f;
#pragma unroll
                for (int c = 0; c < C; ++c) g32[c] += gcd * c1[c];
            }
        }
#pragma unroll
        for (int c = 0; c < C; ++c) g[c] += (double)g32[c];
    }
    if (kSlabs > 1) {
        __syncthreads();
        if (sl > 0)
#pragma unroll
            for (int c = 0; c < C; ++c) slab[((sl - 1) * kRows + ql) * kMaxC + c] = g[c];
        __syncthreads();
        if (sl == 0)
            for (int k = 1; k < kSlabs; ++k)
#pragma unroll
                for (int c = 0; c < C; ++c) g[c] += slab[((k - 1) * kRows + ql) * kMaxC + c];
    }
    if (live && sl == 0)
#pragma unroll
        for (int c = 0; c < kMaxC; ++c) A.gcol[(((size_t)set * A.B + n) * N + q) * kMaxC + c] = c < C ? (float)g[c] : 0.0f;
}

// gcol[set][n][q][c] = - sum over the row blocks of the fused pass's partials: fp64, block order (each of the four thread
// groups of a workgroup adds a contiguous quarter of the row blocks, the quarters are added in order)
template <int C>
__global__ __launch_bounds__(256) void pair_cols_fold_kernel(const PairArgs A, int n_row_blocks) {
    __shared__ double part[3][64][C];
    const int ql = threadIdx.x & 63, k = threadIdx.x >> 6;
    const int q = blockIdx.x * 64 + ql, set = blockIdx.z, n = row_patch(A), N = A.N;
    const float* gp = A.gcolp + ((size_t)set * gridDim.y + blockIdx.y) * n_row_blocks * C * N;
    const int per = (n_row_blocks + 3) / 4, r0 = k * per, r1 = r0 + per < n_row_blocks ? r0 + per : n_row_blocks;
    double s[C];
#pragma unroll
    for (int c = 0; c < C; ++c) s[c] = 0.0;
    if (q < N)
#pragma unroll 4
        for (int rb = r0; rb < r1; ++rb)
#pragma unroll
            for (int c = 0; c < C; ++c) s[c] += (double)gp[((size_t)rb * C + c) * N + q];
    if (k > 0)
#pragma unroll
        for (int c = 0; c < C; ++c) part[k - 1][ql][c] = s[c];
    __syncthreads();
    if (k > 0 || q >= N) return;
#pragma unroll
    for (int j = 0; j < 3; ++j)
#pragma unroll
        for (int c = 0; c < C; ++c) s[c] += part[j][ql][c];
#pragma unroll
    for (int c = 0; c < kMaxC; ++c) A.gcol[(((size_t)set * A.B + n) * N + q) * kMaxC + c] = c < C ? (float)-s[c] : 0.0f;
}

__global__ void loss_finish_kernel(const double* __restrict__ scal, double cnt, CorrParams prm, float* __restrict__ loss, int geo) {
    const float l_neg = (float)(scal[4] / cnt), l_self = (float)(scal[5] / cnt);   // .mean(), :370
    loss[0] = prm.neg_weight * l_neg + prm.self_weight * l_self;
    if (geo && scal[7] != 0.0) loss[0] = __builtin_nanf("");                      // NaN inputs (geo_prep_kernel)
}
// Row-partitioned calls: the loss sums travel with the fp32 gradient sums (ONE all-reduce for both) as three fp32 terms each --
// an exact split of the rank's fp64 partial (3 x 24 bits >= 53); the sum over the ranks then carries fp32 rounding of each term
// (<= 1e-7 of the loss, whose value is fp32 anyway).  With nothing to add (one rank) the round trip is exact.
__device__ __forceinline__ void split_sum(double v, float* out) {
    const float a = (float)v;
    const float b = (float)(v - (double)a);
    out[0] = a;
    out[1] = b;
    out[2] = (float)(v - (double)a - (double)b);
}
__global__ void split_sums_kernel(const double* __restrict__ scal, float* __restrict__ tail) {
    split_sum(scal[4], tail);
    split_sum(scal[5], tail + 3);
    tail[6] = tail[7] = 0.0f;
}
__global__ void loss_finish_split_kernel(const float* __restrict__ tail, const double* __restrict__ flags, double cnt, CorrParams prm,
                                         float* __restrict__ loss, int geo) {
    const double s0 = (double)tail[0] + (double)tail[1] + (double)tail[2], s1 = (double)tail[3] + (double)tail[4] + (double)tail[5];
    const float l_neg = (float)(s0 / cnt), l_self = (float)(s1 / cnt);
    loss[0] = prm.neg_weight * l_neg + prm.self_weight * l_self;
    if (geo && flags[7] != 0.0) loss[0] = __builtin_nanf("");
}
// single-process call: pair_finish_kernel(slot 4) for both sets + loss_finish_kernel in one launch of 64 threads
__global__ void pass3_finish_kernel(const double* __restrict__ partial, int nb, double* __restrict__ scal, double cnt, CorrParams prm,
                                    float* __restrict__ loss, int geo, const double* __restrict__ flags) {
    const double s0 = partial_sum(partial, nb, 0), s1 = partial_sum(partial, nb, 1);
    if (threadIdx.x != 0) return;
    scal[4] = s0;
    scal[5] = s1;
    const float l_neg = (float)(s0 / cnt), l_self = (float)(s1 / cnt);
    loss[0] = prm.neg_weight * l_neg + prm.self_weight * l_self;
    if (geo && flags[7] != 0.0) loss[0] = __builtin_nanf("");
}

// backward of F.normalize for one point: g_v = (g - y (g.y)) / d  if ||v|| >= eps, else g / eps
template <int C>
__device__ __forceinline__ void normalize_backward(const float (&gy)[C], const float* y, float den, float nrm, float (&gv)[C]) {
    if (nrm >= 1e-10f) {
        float dot = 0.0f;
#pragma unroll
        for (int c = 0; c < C; ++c) dot += gy[c] * y[c];
#pragma unroll
        for (int c = 0; c < C; ++c) gv[c] = (gy[c] - y[c] * dot) / den;
    } else {
#pragma unroll
        for (int c = 0; c < C; ++c) gv[c] = gy[c] / den;
    }
}

// geo: gradient w.r.t. code[m,:,pixel] = normalize_backward( rows(neg,m) + rows(self,m) + cols(self,m) + sum_{n: neg[n]=m} cols(neg,n) )
template <int C>
__global__ __launch_bounds__(256) void geo_grad_kernel(int B, int N, const long long* __restrict__ neg, const float* __restrict__ cn,
                                                       const float* __restrict__ dinv, const float* __restrict__ grow,
                                                       const float* __restrict__ gcol, const GeoInputs in) {
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (long long)B * N) return;
    const int m = (int)(i / N), p = (int)(i % N);
    float gy[C];
#pragma unroll
    for (int c = 0; c < C; ++c) {
        float s = grow[((size_t)(0 * B + m) * N + p) * kMaxC + c];
        s += grow[((size_t)(1 * B + m) * N + p) * kMaxC + c];
        s += gcol[((size_t)(1 * B + m) * N + p) * kMaxC + c];
        gy[c] = s;
    }
    for (int n = 0; n < B; ++n)
        if ((int)neg[n] == m)
#pragma unroll
            for (int c = 0; c < C; ++c) gy[c] += gcol[((size_t)(0 * B + n) * N + p) * kMaxC + c];
    float gv[C];
    normalize_backward<C>(gy, cn + i * kMaxC, dinv[i * 2], dinv[i * 2 + 1], gv);
    float* grad_code = in.grad[m / in.Bg];
    if (!grad_code) return;
#pragma unroll
    for (int c = 0; c < C; ++c) grad_code[in.code_at(m % in.Bg, c, p, C, N)] = gv[c];
}

// ------------------------------------------------------------------------------------------ appearance loss pieces
// F.grid_sample(bilinear, padding_mode='border', align_corners=True) geometry for one sample (utils/image.py:303-304)
struct Bilinear {
    int x0, y0;           // north-west corner
    float w[4];           // nw, ne, sw, se
};
__device__ __forceinline__ Bilinear bilinear_of(float gx, float gy, int W, int H) {
    float ix = ((gx + 1.0f) / 2.0f) * (float)(W - 1);
    float iy = ((gy + 1.0f) / 2.0f) * (float)(H - 1);
    ix = fminf((float)(W - 1), fmaxf(ix, 0.0f));
    iy = fminf((float)(H - 1), fmaxf(iy, 0.0f));
    const float fx = floorf(ix), fy = floorf(iy);
    Bilinear b;
    b.x0 = (int)fx; b.y0 = (int)fy;
    const float ex = fx + 1.0f, ey = fy + 1.0f;   // south-east corner
    b.w[0] = (ex - ix) * (ey - iy);
    b.w[1] = (ix - fx) * (ey - iy);
    b.w[2] = (ex - ix) * (iy - fy);
    b.w[3] = (ix - fx) * (iy - fy);
    return b;
}
// sample point p = i*S + j of patch n reads coords[n, j, i, :] (coords.permute(0,2,1,3)); coords = rand*2-1 (:343-344)
__device__ __forceinline__ void sample_coord(const float* rnd, int n, int p, int S, float& gx, float& gy) {
    const int i = p / S, j = p % S;
    const float* r = rnd + (((size_t)n * S + j) * S + i) * 2;
    gx = r[0] * 2.0f - 1.0f;
    gy = r[1] * 2.0f - 1.0f;
}
__device__ __forceinline__ float bilinear_fetch(const float* img, int W, int H, const Bilinear& b, int px = 1) {   // px: floats between pixels
    float out = 0.0f;
    const bool xin = b.x0 + 1 <= W - 1, yin = b.y0 + 1 <= H - 1;
    out = img[((size_t)b.y0 * W + b.x0) * px] * b.w[0];
    if (xin) out = out + img[((size_t)b.y0 * W + b.x0 + 1) * px] * b.w[1];
    if (yin) out = out + img[((size_t)(b.y0 + 1) * W + b.x0) * px] * b.w[2];
    if (xin && yin) out = out + img[((size_t)(b.y0 + 1) * W + b.x0 + 1) * px] * b.w[3];
    return out;
}

// one workgroup per (sample point p, patch n, side): side 0 = patch n at coords1, side 1 = patch neg[n] at coords2.
// Samples all Cf feature channels, L2-normalises them, and (thread 0) samples + normalises the code.
template <int C>
__global__ __launch_bounds__(128) void app_sample_kernel(const float* __restrict__ feats, const float* __restrict__ code,
                                                         const long long* __restrict__ neg, const float* __restrict__ rnd1,
                                                         const float* __restrict__ rnd2, int B, int Cf, int Hf, int Wf, int Hc,
                                                         int Wc, int S, float* __restrict__ fn, float* __restrict__ cn,
                                                         float* __restrict__ cn2, float* __restrict__ dinv, float* __restrict__ dinv2,
                                                         int channel_last, const int* __restrict__ rows, double* __restrict__ scal) {
    __shared__ double red[2];
    if (blockIdx.x == 0 && blockIdx.y == 0 && blockIdx.z == 0 && threadIdx.x == 0) scal[8] = 0.0;   // the pair passes' last-arriver tickets
    const int p = blockIdx.x, n = rows ? rows[blockIdx.y] : (int)blockIdx.y, side = blockIdx.z, N = S * S;
    const int src = side == 0 ? n : (int)neg[n];
    float gx, gy;
    sample_coord(side == 0 ? rnd1 : rnd2, n, p, S, gx, gy);
    const Bilinear bf = bilinear_of(gx, gy, Wf, Hf);
    float* out = fn + (((size_t)side * B + n) * N + p) * Cf;
    double ss = 0.0;
    for (int c = threadIdx.x; c < Cf; c += blockDim.x) {
        const float v = bilinear_fetch(feats + ((size_t)src * Cf + c) * Hf * Wf, Wf, Hf, bf);
        out[c] = v;
        ss += (double)v * v;
    }
    const double tot = block_sum(ss, red);
    __shared__ float den_s;
    if (threadIdx.x == 0) den_s = fmaxf((float)sqrt(tot), 1e-10f);
    __syncthreads();
    for (int c = threadIdx.x; c < Cf; c += blockDim.x) out[c] = out[c] / den_s;
    if (threadIdx.x == 0) {
        const Bilinear bc = bilinear_of(gx, gy, Wc, Hc);
        float v[C], s2 = 0.0f;
#pragma unroll
        for (int c = 0; c < C; ++c) {
            v[c] = channel_last ? bilinear_fetch(code + (size_t)src * C * Hc * Wc + c, Wc, Hc, bc, C)     // [B,Hc,Wc,C]: the renderer's own layout
                                : bilinear_fetch(code + ((size_t)src * C + c) * Hc * Wc, Wc, Hc, bc);
            s2 = c == 0 ? v[c] * v[c] : s2 + v[c] * v[c];
        }
        const float nrm = sqrtf(s2), den = fmaxf(nrm, 1e-10f);
        float* co = (side == 0 ? cn : cn2) + ((size_t)n * N + p) * kMaxC;
        float* dv = (side == 0 ? dinv : dinv2) + ((size_t)n * N + p) * 2;
#pragma unroll
        for (int c = 0; c < kMaxC; ++c) co[c] = c < C ? v[c] / den : 0.0f;
        dv[0] = den;
        dv[1] = nrm;
    }
}

// fd[set][n][p][q] = <f1n[n][p], f2n[q]>, f2n = side 1 for the negative set, side 0 (the same patch) for the self set.
// One workgroup per row point p; a wave takes four column points at a time, lanes across channels (coalesced), so four
// independent row fetches are in flight per lane (one column per iteration was a chain of dependent L2 round trips: 63 us
// for 14 MFLOP).  Products are rounded to fp32 and summed in fp64, as before.
__global__ __launch_bounds__(1024) void app_fd_kernel(const float* __restrict__ fn, int B, int N, int Cf, float* __restrict__ fdmat,
                                                      const int* __restrict__ rows) {
    const int set = blockIdx.z, n = rows ? rows[blockIdx.y] : (int)blockIdx.y, p = blockIdx.x;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nw = blockDim.x >> 6;
    const float* a = fn + (((size_t)0 * B + n) * N + p) * Cf;
    const float* bb = fn + (((size_t)(set == 0 ? 1 : 0) * B + n) * N) * Cf;
    for (int q = 4 * wave; q < N; q += 4 * nw) {
        const float* b[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) b[u] = bb + (size_t)(q + u < N ? q + u : N - 1) * Cf;
        double s[4] = {0.0, 0.0, 0.0, 0.0};
        for (int c = lane; c < Cf; c += 64) {
            const float av = a[c];
#pragma unroll
            for (int u = 0; u < 4; ++u) s[u] += (double)(av * b[u][c]);
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const double t = nsos_wave_sum(s[u]);
            if (lane == 0 && q + u < N) fdmat[(((size_t)set * B + n) * N + p) * N + q + u] = (float)t;
        }
    }
}

// gradient w.r.t. the SAMPLED codes, back through F.normalize:  out g1[n][p][C] (coords1 of n), g2[n][p][C] (coords2)
template <int C>
__global__ __launch_bounds__(128) void app_point_grad_kernel(int B, int N, const float* __restrict__ cn, const float* __restrict__ cn2,
                                                             const float* __restrict__ dinv, const float* __restrict__ dinv2,
                                                             const float* __restrict__ grow, const float* __restrict__ gcol,
                                                             float* __restrict__ g1, float* __restrict__ g2,
                                                             const int* __restrict__ rows, int n_rows) {
    long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (long long)(rows ? n_rows : B) * N) return;
    if (rows) i = (long long)rows[i / N] * N + i % N;          // the row patches of this call; every index below is patch-major
    float gy[C], gv[C];
#pragma unroll
    for (int c = 0; c < C; ++c)   // rows of both sets + columns of the self set all are code(coords1) of patch n
        gy[c] = (grow[((size_t)0 * B * N + i) * kMaxC + c] + grow[((size_t)1 * B * N + i) * kMaxC + c]) + gcol[((size_t)1 * B * N + i) * kMaxC + c];
    normalize_backward<C>(gy, cn + i * kMaxC, dinv[i * 2], dinv[i * 2 + 1], gv);
#pragma unroll
    for (int c = 0; c < C; ++c) g1[i * kMaxC + c] = gv[c];
#pragma unroll
    for (int c = 0; c < C; ++c) gy[c] = gcol[((size_t)0 * B * N + i) * kMaxC + c];
    normalize_backward<C>(gy, cn2 + i * kMaxC, dinv2[i * 2], dinv2[i * 2 + 1], gv);
#pragma unroll
    for (int c = 0; c < C; ++c) g2[i * kMaxC + c] = gv[c];
}

// grid_sample backward as a gather (deterministic): thread = one pixel of patch m; it collects the bilinear weights
// of the coords1 samples of m and of the coords2 samples of every n whose negative is m.  The geometry of a sample set
// (corner + four weights per sample) is worked out once per workgroup into LDS; every thread then walks the N samples with
// broadcast reads (each thread re-deriving it, divisions and all, for all N samples: 70 us per call on 32 CUs).
template <int C>
__global__ __launch_bounds__(256) void app_scatter_kernel(int B, int N, int S, int Hc, int Wc, const long long* __restrict__ neg,
                                                          const float* __restrict__ rnd1, const float* __restrict__ rnd2,
                                                          const float* __restrict__ g1, const float* __restrict__ g2,
                                                          float* __restrict__ grad_code, int channel_last,
                                                          const int* __restrict__ rows, int n_rows) {
    constexpr int kMaxSamples = 1024;
    __shared__ __attribute__((aligned(16))) int corner[kMaxSamples];          // x0 | y0 << 16
    __shared__ float weight[kMaxSamples][4];
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    const long long px = (long long)Hc * Wc;
    // a workgroup's pixels may straddle two patches when Hc*Wc is not a multiple of 256: geometry is staged per patch below
    // `rows`: the patches whose gradient this call forms (the rank's own in the sharded step); pixel index i runs over THEIR slots
    const int n_out = rows ? n_rows : B;
    const bool live = i < (long long)n_out * px;
    const int slot = live ? (int)(i / px) : -1, y = live ? (int)((i / Wc) % Hc) : 0, x = live ? (int)(i % Wc) : 0;
    const int m = !live ? -1 : rows ? rows[slot] : slot;
    const int s_first = (int)(((long long)blockIdx.x * blockDim.x) / px);
    const long long last_i = (long long)blockIdx.x * blockDim.x + blockDim.x - 1;
    const int s_last = (int)((last_i < (long long)n_out * px ? last_i : (long long)n_out * px - 1) / px);
    float acc[C];
#pragma unroll
    for (int c = 0; c < C; ++c) acc[c] = 0.0f;
    for (int ss = s_first; ss <= s_last; ++ss) {
        const int mm = rows ? rows[ss] : ss;
        for (int side = 0; side < 2; ++side)
            for (int n = 0; n < B; ++n) {
                if (side == 0 ? n != mm : (int)neg[n] != mm) continue;        // workgroup-uniform
                const float* gsrc = (side == 0 ? g1 : g2) + (size_t)n * N * kMaxC;
                for (int p0 = 0; p0 < N; p0 += kMaxSamples) {
                    const int cnt = N - p0 < kMaxSamples ? N - p0 : kMaxSamples;
                    __syncthreads();
                    for (int p = threadIdx.x; p < ((cnt + 3) & ~3); p += blockDim.x) {
                        if (p >= cnt) { corner[p] = 0x7fff7fff; continue; }      // padding of the last quad: a corner no pixel is near
                        float gx, gy;
                        sample_coord(side == 0 ? rnd1 : rnd2, n, p0 + p, S, gx, gy);
                        const Bilinear b = bilinear_of(gx, gy, Wc, Hc);
                        corner[p] = b.x0 | (b.y0 << 16);
#pragma unroll
                        for (int k = 0; k < 4; ++k) weight[p][k] = b.w[k];
                    }
                    __syncthreads();
                    if (m != mm) continue;
                    // four corners per LDS read: a sample touches 4 of the patch's pixels, so nearly every test fails, and one
                    // dependent LDS round trip per sample was what this loop cost (24 us per call for 121 samples)
                    for (int p = 0; p < cnt; p += 4) {
                        const int4 cr4 = *reinterpret_cast<const int4*>(&corner[p]);
                        const int crs[4] = {cr4.x, cr4.y, cr4.z, cr4.w};
#pragma unroll
                        for (int k = 0; k < 4; ++k) {
                            const int dx = x - (crs[k] & 0xffff), dy = y - (crs[k] >> 16);
                            if (dx < 0 || dx > 1 || dy < 0 || dy > 1) continue;
                            const float w = weight[p + k][dy * 2 + dx];
#pragma unroll
                            for (int c = 0; c < C; ++c) acc[c] += gsrc[(size_t)(p0 + p + k) * kMaxC + c] * w;
                        }
                    }
                }
            }
    }
    if (!live) return;
#pragma unroll
    for (int c = 0; c < C; ++c)
        grad_code[channel_last ? (((size_t)m * Hc + y) * Wc + x) * C + c : ((size_t)m * C + c) * Hc * Wc + (size_t)y * Wc + x] = acc[c];
}

// ------------------------------------------------------------------------------------------ host side
template <bool GEO, int C, bool NARROW>
int32_t run_pair_passes_shape(const PairArgs& A_in, bool want_grad, float* loss, hipStream_t st, int phases, const double* flags) {
    PairArgs A = A_in;
    const int N = A.N, B = A.B;
    const int tb = PairShape<GEO, NARROW>::kThreads, rows_per_block = PairShape<GEO, NARROW>::kRows;
    const dim3 grid((N + rows_per_block - 1) / rows_per_block, A.rows ? A.n_rows : B, 2);
    const int nb = (int)(grid.x * grid.y);
    if (nb > kRedBlocks) return NSOS_ERR_UNSUPPORTED;
    const size_t lds_rows12_ = GEO ? (size_t)N * 4 * 4 : 0;
    const size_t lds_rows3_ = GEO && !NARROW ? (size_t)pair_padded_columns(N, PairShape<GEO, NARROW>::kSlabs) * (4 + kMaxC) * 4   // the fused pass pads
                                             : lds_rows12_ + (size_t)N * kMaxC * 4;
    const size_t lds_cols_ = lds_rows3_ + (size_t)N * 4;
    // the slab folds reuse the dynamic region after the pair loops: it must hold them even when N is small
    constexpr size_t kLdsTotal = 160 * 1024 - 256, kFold = (size_t)(PairShape<GEO, NARROW>::kSlabs - 1) * PairShape<GEO, NARROW>::kRows * (1 + kMaxC) * 8;
    const size_t lds_rows12 = lds_rows12_ > kFold ? lds_rows12_ : kFold, lds_rows3 = lds_rows3_ > kFold ? lds_rows3_ : kFold,
                 lds_cols = lds_cols_ > kFold ? lds_cols_ : kFold;
    if (lds_rows3 > kLdsTotal || lds_cols > kLdsTotal) return NSOS_ERR_UNSUPPORTED;
    static size_t configured_rows_on[NSOS_MAX_DEVICES], configured_cols_on[NSOS_MAX_DEVICES];   // per device
    size_t& configured_rows = configured_rows_on[nsos_current_device()];   // the largest dynamic size each kernel has been
    size_t& configured_cols = configured_cols_on[nsos_current_device()];   // allowed so far on this device
    if (lds_rows3 > configured_rows || lds_cols > configured_cols) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&pair_rows_kernel<GEO, C, 1, NARROW>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_rows3);
        if (e == hipSuccess) e = hipFuncSetAttribute(reinterpret_cast<const void*>(&pair_rows_kernel<GEO, C, 3, NARROW>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_rows3);
        if constexpr (GEO && !NARROW)
            if (e == hipSuccess) e = hipFuncSetAttribute(reinterpret_cast<const void*>(&pair_rows_kernel<GEO, C, 3, NARROW, true>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_rows3);
        if (e == hipSuccess) e = hipFuncSetAttribute(reinterpret_cast<const void*>(&pair_cols_kernel<GEO, C, NARROW>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_cols);
        if (e != hipSuccess) return (int32_t)e;
        configured_rows = lds_rows3;
        configured_cols = lds_cols;
    }
    // phases (bit mask): 1 = pass 1, 2 = pass 2, 4 = passes 3 (+4).  A single-process call runs all of them; the row-partitioned
    // multi-GPU call runs them one at a time and all-reduces scal[0..1], scal[2..3] over the ranks in between (the global
    // means of fd and fd1 couple every patch of the batch, utils/image.py:316-319).
    // single-process call (all phases, a loss to write): fold the finishing reductions into the passes' last workgroups
    // MEASURED, NOT KEPT AS THE DEFAULT (round 6, profiles/r06/e_loss_finish_fold_ab.txt).  First form (__threadfence() per
    // workgroup): the replayed C3 step 1.583 ms folded against 1.517 ms with the finish launches (C4: 2.917 against 2.721) -- an
    // agent-scope release is an L2 write-back + invalidate on a chip whose eight XCD L2s are not coherent with each other, 256-1024 of
    // them per pass.  Second form (no fence: sc1 stores / loads of exactly the values the last workgroup reads, nsos_last_block):
    // 1.586 against 1.581 ms, C4 2.889 against 2.866 -- no slower any more and no faster either: the appearance loss's finish
    // launches run on the side stream under the geometric loss, and ONE workgroup summing 8 k row sums through memory takes what the
    // two 1024-thread finish blocks took.  NSOS_LOSS_FOLD_FINISH=1 selects the folded form (same bits: tests/test_gpu_losses.py).
    const bool fold = phases == 7 && loss != nullptr && nsos_env_flag("NSOS_LOSS_FOLD_FINISH");
    A.tickets = fold ? reinterpret_cast<unsigned*>(A.scal + 8) : nullptr;
    A.nb = nb;
    A.cnt = (double)B * N * N;
    A.loss = loss;
    A.flags = flags ? flags : A.scal;
    A.geo = GEO ? 1 : 0;
    if (phases & 1) {
        hipLaunchKernelGGL((pair_rows_kernel<GEO, C, 1, NARROW>), grid, dim3(tb), lds_rows12, st, A);
        if (GEO && A.Bg > 0 && A.Bg < B) {
            if (fold) hipLaunchKernelGGL(pair_rowsum_copy_finish_kernel, dim3((N + 255) / 256, grid.y, 2), dim3(256), 0, st, A, (int)grid.x);
            else hipLaunchKernelGGL(pair_rowsum_copy_kernel, dim3((N + 255) / 256, grid.y, 2), dim3(256), 0, st, A, (int)grid.x);
        }
        if (fold) {}
        else if (phases & 2) hipLaunchKernelGGL(pass1_finish_kernel, dim3(2), dim3(1024), 0, st, A, nb);
        else hipLaunchKernelGGL(pair_finish_kernel, dim3(2), dim3(64), 0, st, A.partial, nb, A.scal, 0);
    } else if (phases & 2) hipLaunchKernelGGL(rowmean_residual_kernel, dim3(2), dim3(256), 0, st, A);
    if (phases & 4) {
        constexpr bool kCanFuse = GEO && !NARROW;
        const bool fuse = kCanFuse && want_grad && A.gcolp && !nsos_env_flag("NSOS_GEO_SEPARATE_COLS");
        if constexpr (kCanFuse) {
            if (fuse) hipLaunchKernelGGL((pair_rows_kernel<GEO, C, 3, NARROW, true>), grid, dim3(tb), lds_rows3, st, A);
        }
        if (!fuse) hipLaunchKernelGGL((pair_rows_kernel<GEO, C, 3, NARROW>), grid, dim3(tb), lds_rows3, st, A);
        if (fold) {}
        else if (loss) hipLaunchKernelGGL(pass3_finish_kernel, dim3(1), dim3(64), 0, st, A.partial, nb, A.scal, (double)B * N * N, A.prm, loss, GEO ? 1 : 0, flags ? flags : A.scal);
        else hipLaunchKernelGGL(pair_finish_kernel, dim3(2), dim3(64), 0, st, A.partial, nb, A.scal, 4);
        if constexpr (kCanFuse) {
            if (fuse) hipLaunchKernelGGL((pair_cols_fold_kernel<C>), dim3((N + 63) / 64, grid.y, 2), dim3(256), 0, st, A, (int)grid.x);
        }
        if (want_grad && !fuse) hipLaunchKernelGGL((pair_cols_kernel<GEO, C, NARROW>), grid, dim3(tb), lds_cols, st, A);
    }
    return nsos_launch_status();
}

template <bool GEO, int C>
int32_t run_pair_passes(const PairArgs& A, bool want_grad, float* loss, hipStream_t st, int phases = 7, const double* flags = nullptr) {
    if constexpr (GEO) {
        const int cus = nsos_device_cus();
        const long long wide = (long long)((A.N + 63) / 64) * (A.rows ? A.n_rows : A.B) * 2;   // workgroups at 64 rows each
        if (wide < cus && !nsos_env_flag("NSOS_GEO_FORCE_WIDE")) return run_pair_passes_shape<true, C, true>(A, want_grad, loss, st, phases, flags);
    }
    return run_pair_passes_shape<GEO, C, false>(A, want_grad, loss, st, phases, flags);
}

template <int C>
int32_t geo_impl(float* depth, const GeoInputs in, const long long* neg, int B, int N,
                 CorrParams prm, float max_depth, int write_back, float* loss, void* workspace, hipStream_t st) {
    const bool want_grad = in.grad[0] != nullptr || in.grad[1] != nullptr;
    Ws w;
    ws_layout(&w, workspace, B, N, 0, false);
    const long long tot = (long long)B * N, tot_geo = (long long)in.Bg * N;      // depth: one map per GEOMETRY patch
    const int rb = (int)((tot_geo + 255) / 256 < 256 ? (tot_geo + 255) / 256 : 256);
    hipLaunchKernelGGL(depth_max_kernel, dim3(rb), dim3(256), 0, st, depth, tot_geo, max_depth, w.partial, w.scal);
    hipLaunchKernelGGL((geo_prep_kernel<C>), dim3((unsigned)((tot + 255) / 256)), dim3(256), 0, st, depth, in, B, N,
                       max_depth, write_back, w.scal, w.pts, w.cn, w.dinv, w.partial, rb);
    PairArgs A = {B, N, C, neg, w.pts, w.cn, nullptr, nullptr, w.rowsum, w.partial, w.scal, w.grow, w.gcol, max_depth, prm, nullptr, 0, w.gcolp, in.Bg};
    const int32_t rc = run_pair_passes<true, C>(A, want_grad, loss, st);
    if (rc != NSOS_OK) return rc;
    if (want_grad)
        hipLaunchKernelGGL((geo_grad_kernel<C>), dim3((unsigned)((tot + 255) / 256)), dim3(256), 0, st, B, N, neg, w.cn, w.dinv, w.grow,
                           w.gcol, in);
    return nsos_launch_status();
}

// ---- row-partitioned geometric loss (multi-GPU: each rank evaluates the pair sets of ITS OWN row patches) -----------
// gsum[m] = sum over the roles patch m plays in the row patches `rows`:  rows(neg,m) + rows(self,m) + cols(self,m) if m is one
// of them, + cols(neg,n) for every n in `rows` whose negative is m.  Summed over the ranks (all-reduce) it is the argument of
// normalize_backward in geo_grad_kernel.
template <int C>
__global__ __launch_bounds__(256) void geo_gsum_kernel(int B, int N, const long long* __restrict__ neg, const int* __restrict__ rows,
                                                       int n_rows, const float* __restrict__ grow, const float* __restrict__ gcol,
                                                       float* __restrict__ gsum) {
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (long long)B * N) return;
    const int m = (int)(i / N), p = (int)(i % N);
    float gy[C];
#pragma unroll
    for (int c = 0; c < C; ++c) gy[c] = 0.0f;
    for (int r = 0; r < n_rows; ++r) {
        const int n = rows[r];
        if (n == m) {
#pragma unroll
            for (int c = 0; c < C; ++c) {
                float s = grow[((size_t)(0 * B + m) * N + p) * kMaxC + c];
                s += grow[((size_t)(1 * B + m) * N + p) * kMaxC + c];
                s += gcol[((size_t)(1 * B + m) * N + p) * kMaxC + c];
                gy[c] += s;
            }
        }
        if ((int)neg[n] == m)
#pragma unroll
            for (int c = 0; c < C; ++c) gy[c] += gcol[((size_t)(0 * B + n) * N + p) * kMaxC + c];
    }
#pragma unroll
    for (int c = 0; c < kMaxC; ++c) gsum[i * kMaxC + c] = c < C ? gy[c] : 0.0f;
}

template <int C>
__global__ __launch_bounds__(256) void geo_finish_kernel(int B, int N, const float* __restrict__ cn, const float* __restrict__ dinv,
                                                         const float* __restrict__ gsum, const GeoInputs in) {
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (long long)B * N) return;
    const int m = (int)(i / N), p = (int)(i % N);
    float gy[C], gv[C];
#pragma unroll
    for (int c = 0; c < C; ++c) gy[c] = gsum[i * kMaxC + c];
    normalize_backward<C>(gy, cn + i * kMaxC, dinv[i * 2], dinv[i * 2 + 1], gv);
    float* grad_code = in.grad[m / in.Bg];
    if (!grad_code) return;
#pragma unroll
    for (int c = 0; c < C; ++c) grad_code[in.code_at(m % in.Bg, c, p, C, N)] = gv[c];
}

// Row-partitioned evaluation (multi-GPU: each rank evaluates the pair sets of ITS OWN row patches), two reductions per call:
// phase 0: depth filter + points + normalised codes of the WHOLE batch (cheap, every rank), pass 1 over own rows and the row-mean
//          residual of those rows -> means[0..3]                                   [means[0..3] summed over the ranks]
// phase 1: passes 3 and 4 over own rows -> sums = the role sums of the gradient + the split loss sums    [sums summed over the ranks]
// phase 2: loss value and d loss / d code for every patch of the batch
// phase 3: all of it in one call (single process: nothing to reduce)
// `means` (8 doubles) / `sums` (B N kMaxC + kSumTail floats) default to the workspace's own slots; the sharded training step passes
// slices of ONE buffer per reduction shared by all of its loss evaluations, so that a step issues one all-reduce per phase.
template <int C>
int32_t geo_rows_impl(int phase, float* depth, const GeoInputs in, const long long* neg,
                      const int* rows, int n_rows, int B, int N, CorrParams prm, float max_depth, int write_back, float* loss,
                      void* workspace, double* xmeans, float* xsums, hipStream_t st) {
    Ws w;
    ws_layout(&w, workspace, B, N, 0, false);
    const long long tot = (long long)B * N;
    const unsigned gb = (unsigned)((tot + 255) / 256);
    double* means = xmeans ? xmeans : w.scal;       // [0..5]; the depth maximum and the NaN flag ([6], [7]) stay in the workspace
    float* sums = xsums ? xsums : w.gsum;
    PairArgs A = {B, N, C, neg, w.pts, w.cn, nullptr, nullptr, w.rowsum, w.partial, means, w.grow, w.gcol, max_depth, prm, rows, n_rows, w.gcolp, in.Bg};
    if (phase == 0 || phase == 3) {
        const long long tot_geo = (long long)in.Bg * N;                            // depth: one map per GEOMETRY patch
        const int rb = (int)((tot_geo + 255) / 256 < 256 ? (tot_geo + 255) / 256 : 256);
        hipLaunchKernelGGL(depth_max_kernel, dim3(rb), dim3(256), 0, st, depth, tot_geo, max_depth, w.partial, w.scal);
        hipLaunchKernelGGL((geo_prep_kernel<C>), dim3(gb), dim3(256), 0, st, depth, in, B, N, max_depth, write_back,
                           w.scal, w.pts, w.cn, w.dinv, w.partial, rb);
    }
    if (phase == 3) {   // single process: nothing to reduce between the phases -- every launch of the loss from one call
        if (n_rows == 0) {
            hipError_t e = hipMemsetAsync(means, 0, 6 * sizeof(double), st);
            if (e == hipSuccess) e = hipMemsetAsync(sums, 0, sizeof(float) * (tot * kMaxC + kSumTail), st);
            if (e != hipSuccess) return (int32_t)e;
            hipLaunchKernelGGL(loss_finish_split_kernel, dim3(1), dim3(1), 0, st, sums + tot * kMaxC, w.scal, (double)B * N * N, prm, loss, 1);
        } else {
            const bool want_grad = in.grad[0] || in.grad[1];
            const int32_t rc = run_pair_passes<true, C>(A, want_grad, loss, st, 7, w.scal);
            if (rc != NSOS_OK) return rc;
            if (want_grad) hipLaunchKernelGGL((geo_gsum_kernel<C>), dim3(gb), dim3(256), 0, st, B, N, neg, rows, n_rows, w.grow, w.gcol, sums);
        }
        if (in.grad[0] || in.grad[1]) hipLaunchKernelGGL((geo_finish_kernel<C>), dim3(gb), dim3(256), 0, st, B, N, w.cn, w.dinv, sums, in);
        return nsos_launch_status();
    }
    if (phase == 0) {
        hipError_t e = hipMemsetAsync(means, 0, (xmeans ? 8 : 6) * sizeof(double), st);   // a rank without patches contributes zeros
        if (e != hipSuccess) return (int32_t)e;
        if (n_rows == 0) return nsos_launch_status();
        return run_pair_passes<true, C>(A, true, nullptr, st, 3, w.scal);
    }
    if (phase == 1) {
        if (n_rows == 0) {
            const hipError_t e = hipMemsetAsync(sums, 0, sizeof(float) * (tot * kMaxC + kSumTail), st);
            return e == hipSuccess ? nsos_launch_status() : (int32_t)e;
        }
        const int32_t rc = run_pair_passes<true, C>(A, true, nullptr, st, 4, w.scal);
        if (rc != NSOS_OK) return rc;
        hipLaunchKernelGGL((geo_gsum_kernel<C>), dim3(gb), dim3(256), 0, st, B, N, neg, rows, n_rows, w.grow, w.gcol, sums);
        hipLaunchKernelGGL(split_sums_kernel, dim3(1), dim3(1), 0, st, means, sums + tot * kMaxC);
        return nsos_launch_status();
    }
    hipLaunchKernelGGL(loss_finish_split_kernel, dim3(1), dim3(1), 0, st, sums + tot * kMaxC, w.scal, (double)B * N * N, prm, loss, 1);
    if (in.grad[0] || in.grad[1]) hipLaunchKernelGGL((geo_finish_kernel<C>), dim3(gb), dim3(256), 0, st, B, N, w.cn, w.dinv, sums, in);
    return nsos_launch_status();
}

template <int C>
int32_t app_impl(const float* feats, const float* code, const long long* neg, const float* rnd1, const float* rnd2, int B, int Cf,
                 int Hf, int Wf, int Hc, int Wc, int S, CorrParams prm, float* loss, float* grad_code, void* workspace, hipStream_t st,
                 int channel_last) {
    const int N = S * S;
    Ws w;
    ws_layout(&w, workspace, B, N, Cf, true);
    hipLaunchKernelGGL((app_sample_kernel<C>), dim3(N, B, 2), dim3(128), 0, st, feats, code, neg, rnd1, rnd2, B, Cf, Hf, Wf, Hc, Wc, S,
                       w.fn, w.cn, w.cn2, w.dinv, w.dinv2, channel_last, nullptr, w.scal);
    hipLaunchKernelGGL(app_fd_kernel, dim3(N, B, 2), dim3(1024), 0, st, w.fn, B, N, Cf, w.fdmat, nullptr);   // 16 waves x 2 column quads each: a short dependent chain
    PairArgs A = {B, N, C, neg, nullptr, w.cn, w.cn2, w.fdmat, w.rowsum, w.partial, w.scal, w.grow, w.gcol, 0.0f, prm, nullptr, 0, nullptr, 0};
    const int32_t rc = run_pair_passes<false, C>(A, grad_code != nullptr, loss, st);
    if (rc != NSOS_OK) return rc;
    if (grad_code) {
        // g1/g2 reuse the (now consumed) feature buffer
        float* g1 = w.fn;
        float* g2 = w.fn + (size_t)B * N * kMaxC;
        const long long tot = (long long)B * N;
        hipLaunchKernelGGL((app_point_grad_kernel<C>), dim3((unsigned)((tot + 127) / 128)), dim3(128), 0, st, B, N, w.cn, w.cn2, w.dinv,
                           w.dinv2, w.grow, w.gcol, g1, g2, nullptr, 0);
        const long long px = (long long)B * Hc * Wc;
        hipLaunchKernelGGL((app_scatter_kernel<C>), dim3((unsigned)((px + 63) / 64)), dim3(64), 0, st, B, N, S, Hc, Wc, neg, rnd1, rnd2,   // one wave per workgroup: 64 CUs busy per patch instead of 16
                           g1, g2, grad_code, channel_last, nullptr, 0);
    }
    return nsos_launch_status();
}

// Row-partitioned appearance loss: the phases and the two reductions of geo_rows_impl.  Everything a row patch n needs is its own
// (the coords1 samples of n, the coords2 samples of neg[n], both fd matrices), except where its column gradient lands: g2[n] is
// scattered into patch neg[n], which another rank may own -- `sums` carries g2 of every row (zeros for rows of other ranks), so
// after the reduction each rank scatters into ITS patches (grad_code of the others: zeros; no rank needs them).
template <int C>
int32_t app_rows_impl(int phase, const float* feats, const float* code, const long long* neg, const float* rnd1, const float* rnd2,
                      const int* rows, int n_rows, int B, int Cf, int Hf, int Wf, int Hc, int Wc, int S, CorrParams prm, float* loss,
                      float* grad_code, void* workspace, double* xmeans, float* xsums, hipStream_t st, int channel_last) {
    const int N = S * S;
    Ws w;
    ws_layout(&w, workspace, B, N, Cf, true);
    const long long tot = (long long)B * N;
    double* means = xmeans ? xmeans : w.scal;
    float* sums = xsums ? xsums : w.gsum;
    PairArgs A = {B, N, C, neg, nullptr, w.cn, w.cn2, w.fdmat, w.rowsum, w.partial, means, w.grow, w.gcol, 0.0f, prm, rows, n_rows, nullptr, 0};
    if (phase == 0) {
        hipError_t e = hipMemsetAsync(means, 0, (xmeans ? 8 : 6) * sizeof(double), st);
        if (e != hipSuccess) return (int32_t)e;
        if (n_rows == 0) return nsos_launch_status();
        hipLaunchKernelGGL((app_sample_kernel<C>), dim3(N, n_rows, 2), dim3(128), 0, st, feats, code, neg, rnd1, rnd2, B, Cf, Hf, Wf, Hc, Wc, S,
                           w.fn, w.cn, w.cn2, w.dinv, w.dinv2, channel_last, rows, w.scal);
        hipLaunchKernelGGL(app_fd_kernel, dim3(N, n_rows, 2), dim3(1024), 0, st, w.fn, B, N, Cf, w.fdmat, rows);
        return run_pair_passes<false, C>(A, true, nullptr, st, 3);
    }
    if (phase == 1) {
        hipError_t e = hipMemsetAsync(sums, 0, sizeof(float) * (tot * kMaxC + kSumTail), st);
        if (e != hipSuccess) return (int32_t)e;
        if (n_rows == 0) return nsos_launch_status();
        const int32_t rc = run_pair_passes<false, C>(A, true, nullptr, st, 4);
        if (rc != NSOS_OK) return rc;
        hipLaunchKernelGGL((app_point_grad_kernel<C>), dim3((unsigned)(((long long)n_rows * N + 127) / 128)), dim3(128), 0, st, B, N, w.cn, w.cn2,
                           w.dinv, w.dinv2, w.grow, w.gcol, w.fn, sums, rows, n_rows);       // g1 -> the (consumed) feature buffer
        hipLaunchKernelGGL(split_sums_kernel, dim3(1), dim3(1), 0, st, means, sums + tot * kMaxC);
        return nsos_launch_status();
    }
    hipLaunchKernelGGL(loss_finish_split_kernel, dim3(1), dim3(1), 0, st, sums + tot * kMaxC, w.scal, (double)B * N * N, prm, loss, 0);
    if (grad_code) {
        const long long px_all = (long long)B * Hc * Wc;
        hipError_t e = hipMemsetAsync(grad_code, 0, sizeof(float) * px_all * C, st);
        if (e != hipSuccess) return (int32_t)e;
        if (n_rows > 0) {
            const long long px = (long long)n_rows * Hc * Wc;
            hipLaunchKernelGGL((app_scatter_kernel<C>), dim3((unsigned)((px + 63) / 64)), dim3(64), 0, st, B, N, S, Hc, Wc, neg, rnd1, rnd2,
                               w.fn, sums, grad_code, channel_last, rows, n_rows);
        }
    }
    return nsos_launch_status();
}

}  // namespace

// ------------------------------------------------------------------------------------------ C ABI
extern "C" size_t nsos_corr_workspace_bytes(int32_t kind, int32_t batch, int32_t n_points, int32_t feat_dim) {
    if (batch <= 0 || n_points <= 0 || (kind != 0 && kind != 1)) return 0;
    const int cf = kind == 0 ? (feat_dim < 2 * kMaxC ? 2 * kMaxC : feat_dim) : 0;
    return ws_layout(nullptr, nullptr, batch, n_points, cf, kind == 0);
}

extern "C" int32_t nsos_geo_correlation_loss(float* depth, const float* code, const float* ray_o, const float* ray_d,
                                             const int64_t* neg_indx, int32_t batch, int32_t code_dim, int32_t height,
                                             int32_t width, float self_shift, float self_weight, float neg_shift,
                                             float neg_weight, float max_depth, int32_t filter_in_place, float* loss,
                                             float* grad_code, void* workspace, size_t workspace_bytes, void* stream) {
    if (batch == 0) return NSOS_OK;
    NSOS_REQUIRE(depth && code && ray_o && ray_d && neg_indx && loss && workspace, NSOS_ERR_NULL_POINTER);
    NSOS_REQUIRE(batch > 0 && height > 0 && width > 0, NSOS_ERR_BAD_SHAPE);
    NSOS_REQUIRE(code_dim >= 1 && code_dim <= kMaxC, NSOS_ERR_UNSUPPORTED);
    const long long N = (long long)height * width;
    NSOS_REQUIRE(N <= 4096, NSOS_ERR_UNSUPPORTED);   // one patch's points + codes + row means stay resident in LDS
    NSOS_REQUIRE(((uintptr_t)workspace & 15) == 0, NSOS_ERR_MISALIGNED);
    NSOS_REQUIRE(workspace_bytes >= nsos_corr_workspace_bytes(1, batch, (int32_t)N, 0), NSOS_ERR_BUFFER_TOO_SMALL);
    const CorrParams prm = {self_shift, self_weight, neg_shift, neg_weight};
    const long long* neg = reinterpret_cast<const long long*>(neg_indx);
    const hipStream_t st = (hipStream_t)stream;
    const GeoInputs in = {{code, nullptr}, {grad_code, nullptr}, ray_o, ray_d, batch, 1, 0};
    switch (code_dim) {
        case 1: return geo_impl<1>(depth, in, neg, batch, (int)N, prm, max_depth, filter_in_place, loss, workspace, st);
        case 2: return geo_impl<2>(depth, in, neg, batch, (int)N, prm, max_depth, filter_in_place, loss, workspace, st);
        case 3: return geo_impl<3>(depth, in, neg, batch, (int)N, prm, max_depth, filter_in_place, loss, workspace, st);
        default: return geo_impl<4>(depth, in, neg, batch, (int)N, prm, max_depth, filter_in_place, loss, workspace, st);
    }
}

extern "C" int32_t nsos_corr_workspace_slots(int32_t batch, int32_t n_points, int64_t* scal_offset_bytes,
                                             int64_t* gsum_offset_bytes, int64_t* gsum_floats) {
    NSOS_REQUIRE(batch > 0 && n_points > 0 && scal_offset_bytes && gsum_offset_bytes && gsum_floats, NSOS_ERR_BAD_SHAPE);
    Ws w;
    ws_layout(&w, reinterpret_cast<void*>((uintptr_t)4096), batch, n_points, 0, false);   // offsets relative to a fake base
    *scal_offset_bytes = (int64_t)((uintptr_t)w.scal - 4096);
    *gsum_offset_bytes = (int64_t)((uintptr_t)w.gsum - 4096);
    *gsum_floats = (int64_t)batch * n_points * kMaxC + kSumTail;
    return NSOS_OK;
}

extern "C" int64_t nsos_corr_exchange_floats(int32_t batch, int32_t n_points) {
    return batch > 0 && n_points > 0 ? (int64_t)batch * n_points * kMaxC + kSumTail : 0;
}

static int32_t geo_rows_entry(int32_t phase, float* depth, const GeoInputs in, const int64_t* neg_indx, const int32_t* rows, int32_t n_rows,
                              int32_t code_dim, int32_t height, int32_t width, float self_shift, float self_weight, float neg_shift,
                              float neg_weight, float max_depth, int32_t filter_in_place, float* loss, void* workspace,
                              size_t workspace_bytes, double* xmeans, float* xsums, void* stream) {
    const int batch = in.Bg * in.n_codes;
    NSOS_REQUIRE(phase >= 0 && phase <= 3, NSOS_ERR_UNSUPPORTED);
    NSOS_REQUIRE((((uintptr_t)xmeans) & 7) == 0 && (((uintptr_t)xsums) & 3) == 0, NSOS_ERR_MISALIGNED);
    NSOS_REQUIRE(depth && in.code[0] && (in.n_codes == 1 || in.code[1]) && in.ray_o && in.ray_d && neg_indx && workspace && (n_rows == 0 || rows),
                 NSOS_ERR_NULL_POINTER);
    NSOS_REQUIRE(phase < 2 || loss, NSOS_ERR_NULL_POINTER);   // phases 2 and 3 write the loss
    NSOS_REQUIRE(batch > 0 && height > 0 && width > 0 && n_rows >= 0 && n_rows <= batch, NSOS_ERR_BAD_SHAPE);
    NSOS_REQUIRE(code_dim >= 1 && code_dim <= kMaxC, NSOS_ERR_UNSUPPORTED);
    const long long N = (long long)height * width;
    NSOS_REQUIRE(N <= 4096, NSOS_ERR_UNSUPPORTED);
    NSOS_REQUIRE(((uintptr_t)workspace & 15) == 0, NSOS_ERR_MISALIGNED);
    NSOS_REQUIRE(workspace_bytes >= nsos_corr_workspace_bytes(1, batch, (int32_t)N, 0), NSOS_ERR_BUFFER_TOO_SMALL);
    const CorrParams prm = {self_shift, self_weight, neg_shift, neg_weight};
    const long long* neg = reinterpret_cast<const long long*>(neg_indx);
    const hipStream_t st = (hipStream_t)stream;
    switch (code_dim) {
        case 1: return geo_rows_impl<1>(phase, depth, in, neg, rows, n_rows, batch, (int)N, prm, max_depth, filter_in_place, loss, workspace, xmeans, xsums, st);
        case 2: return geo_rows_impl<2>(phase, depth, in, neg, rows, n_rows, batch, (int)N, prm, max_depth, filter_in_place, loss, workspace, xmeans, xsums, st);
        case 3: return geo_rows_impl<3>(phase, depth, in, neg, rows, n_rows, batch, (int)N, prm, max_depth, filter_in_place, loss, workspace, xmeans, xsums, st);
        default: return geo_rows_impl<4>(phase, depth, in, neg, rows, n_rows, batch, (int)N, prm, max_depth, filter_in_place, loss, workspace, xmeans, xsums, st);
    }
}

extern "C" int32_t nsos_geo_correlation_loss_rows(int32_t phase, float* depth, const float* code, const float* ray_o,
                                                  const float* ray_d, const int64_t* neg_indx, const int32_t* rows, int32_t n_rows,
                                                  int32_t batch, int32_t code_dim, int32_t height, int32_t width,
                                                  float self_shift, float self_weight, float neg_shift, float neg_weight,
                                                  float max_depth, int32_t filter_in_place, float* loss, float* grad_code,
                                                  void* workspace, size_t workspace_bytes, double* exchange_means,
                                                  float* exchange_sums, void* stream) {
    if (batch == 0) return NSOS_OK;
    const GeoInputs in = {{code, nullptr}, {grad_code, nullptr}, ray_o, ray_d, batch, 1, 0};
    return geo_rows_entry(phase, depth, in, neg_indx, rows, n_rows, code_dim, height, width, self_shift, self_weight, neg_shift,
                          neg_weight, max_depth, filter_in_place, loss, workspace, workspace_bytes, exchange_means, exchange_sums, stream);
}

extern "C" int32_t nsos_geo_correlation_loss_pair(int32_t phase, const float* depth, const float* code0, const float* code1,
                                                  const float* ray_o, const float* ray_d, const int64_t* neg_indx,
                                                  const int32_t* rows, int32_t n_rows, int32_t batch, int32_t channel_last,
                                                  int32_t code_dim, int32_t height, int32_t width, float self_shift,
                                                  float self_weight, float neg_shift, float neg_weight, float max_depth, float* loss,
                                                  float* grad_code0, float* grad_code1, void* workspace, size_t workspace_bytes,
                                                  double* exchange_means, float* exchange_sums, void* stream) {
    if (batch == 0) return NSOS_OK;
    NSOS_REQUIRE(batch > 0, NSOS_ERR_BAD_SHAPE);
    const GeoInputs in = {{code0, code1}, {grad_code0, grad_code1}, ray_o, ray_d, batch, 2, channel_last != 0};
    // the depth filter (values > max_depth -> the largest value below it) is applied on the fly, never written back
    return geo_rows_entry(phase, const_cast<float*>(depth), in, neg_indx, rows, n_rows, code_dim, height, width, self_shift, self_weight,
                          neg_shift, neg_weight, max_depth, 0, loss, workspace, workspace_bytes, exchange_means, exchange_sums, stream);
}

static int32_t app_entry(const float* feats, const float* code, const int64_t* neg_indx, const float* rand1, const float* rand2,
                         int32_t batch, int32_t feat_dim, int32_t feat_h, int32_t feat_w, int32_t code_dim, int32_t code_h, int32_t code_w,
                         int32_t feature_samples, float self_shift, float self_weight, float neg_shift, float neg_weight, float* loss,
                         float* grad_code, void* workspace, size_t workspace_bytes, void* stream, int channel_last) {
    if (batch == 0) return NSOS_OK;
    NSOS_REQUIRE(feats && code && neg_indx && rand1 && rand2 && loss && workspace, NSOS_ERR_NULL_POINTER);
    NSOS_REQUIRE(batch > 0 && feat_dim > 0 && feat_h > 0 && feat_w > 0 && code_h > 0 && code_w > 0 && feature_samples > 0,
                 NSOS_ERR_BAD_SHAPE);
    NSOS_REQUIRE(code_dim >= 1 && code_dim <= kMaxC, NSOS_ERR_UNSUPPORTED);
    const int N = feature_samples * feature_samples;
    NSOS_REQUIRE(N <= 1024, NSOS_ERR_UNSUPPORTED);
    NSOS_REQUIRE(((uintptr_t)workspace & 15) == 0, NSOS_ERR_MISALIGNED);
    NSOS_REQUIRE(workspace_bytes >= nsos_corr_workspace_bytes(0, batch, N, feat_dim), NSOS_ERR_BUFFER_TOO_SMALL);
    const CorrParams prm = {self_shift, self_weight, neg_shift, neg_weight};
    const long long* neg = reinterpret_cast<const long long*>(neg_indx);
    const hipStream_t st = (hipStream_t)stream;
    switch (code_dim) {
        case 1: return app_impl<1>(feats, code, neg, rand1, rand2, batch, feat_dim, feat_h, feat_w, code_h, code_w, feature_samples, prm, loss, grad_code, workspace, st, channel_last);
        case 2: return app_impl<2>(feats, code, neg, rand1, rand2, batch, feat_dim, feat_h, feat_w, code_h, code_w, feature_samples, prm, loss, grad_code, workspace, st, channel_last);
        case 3: return app_impl<3>(feats, code, neg, rand1, rand2, batch, feat_dim, feat_h, feat_w, code_h, code_w, feature_samples, prm, loss, grad_code, workspace, st, channel_last);
        default: return app_impl<4>(feats, code, neg, rand1, rand2, batch, feat_dim, feat_h, feat_w, code_h, code_w, feature_samples, prm, loss, grad_code, workspace, st, channel_last);
    }
}

extern "C" int32_t nsos_app_correlation_loss(const float* feats, const float* code, const int64_t* neg_indx,
                                             const float* rand1, const float* rand2, int32_t batch, int32_t feat_dim,
                                             int32_t feat_h, int32_t feat_w, int32_t code_dim, int32_t code_h,
                                             int32_t code_w, int32_t feature_samples, float self_shift,
                                             float self_weight, float neg_shift, float neg_weight, float* loss,
                                             float* grad_code, void* workspace, size_t workspace_bytes, void* stream) {
    return app_entry(feats, code, neg_indx, rand1, rand2, batch, feat_dim, feat_h, feat_w, code_dim, code_h, code_w, feature_samples,
                     self_shift, self_weight, neg_shift, neg_weight, loss, grad_code, workspace, workspace_bytes, stream, 0);
}

extern "C" int32_t nsos_app_correlation_loss_nhwc(const float* feats, const float* code, const int64_t* neg_indx,
                                                  const float* rand1, const float* rand2, int32_t batch, int32_t feat_dim,
                                                  int32_t feat_h, int32_t feat_w, int32_t code_dim, int32_t code_h,
                                                  int32_t code_w, int32_t feature_samples, float self_shift,
                                                  float self_weight, float neg_shift, float neg_weight, float* loss,
                                                  float* grad_code, void* workspace, size_t workspace_bytes, void* stream) {
    return app_entry(feats, code, neg_indx, rand1, rand2, batch, feat_dim, feat_h, feat_w, code_dim, code_h, code_w, feature_samples,
                     self_shift, self_weight, neg_shift, neg_weight, loss, grad_code, workspace, workspace_bytes, stream, 1);
}

extern "C" int32_t nsos_app_correlation_loss_rows(int32_t phase, const float* feats, const float* code, const int64_t* neg_indx,
                                                  const float* rand1, const float* rand2, const int32_t* rows, int32_t n_rows,
                                                  int32_t batch, int32_t channel_last, int32_t feat_dim, int32_t feat_h, int32_t feat_w,
                                                  int32_t code_dim, int32_t code_h, int32_t code_w, int32_t feature_samples,
                                                  float self_shift, float self_weight, float neg_shift, float neg_weight, float* loss,
                                                  float* grad_code, void* workspace, size_t workspace_bytes, double* exchange_means,
                                                  float* exchange_sums, void* stream) {
    if (batch == 0) return NSOS_OK;
    NSOS_REQUIRE(phase >= 0 && phase <= 2, NSOS_ERR_UNSUPPORTED);      // (one process: nsos_app_correlation_loss[_nhwc])
    NSOS_REQUIRE(feats && code && neg_indx && rand1 && rand2 && workspace && (n_rows == 0 || rows) && (phase < 2 || loss), NSOS_ERR_NULL_POINTER);
    NSOS_REQUIRE(batch > 0 && feat_dim > 0 && feat_h > 0 && feat_w > 0 && code_h > 0 && code_w > 0 && feature_samples > 0 && n_rows >= 0 &&
                 n_rows <= batch, NSOS_ERR_BAD_SHAPE);
    NSOS_REQUIRE(code_dim >= 1 && code_dim <= kMaxC, NSOS_ERR_UNSUPPORTED);
    const int N = feature_samples * feature_samples;
    NSOS_REQUIRE(N <= 1024, NSOS_ERR_UNSUPPORTED);
    NSOS_REQUIRE(((uintptr_t)workspace & 15) == 0 && (((uintptr_t)exchange_means) & 7) == 0 && (((uintptr_t)exchange_sums) & 3) == 0, NSOS_ERR_MISALIGNED);
    NSOS_REQUIRE(workspace_bytes >= nsos_corr_workspace_bytes(0, batch, N, feat_dim), NSOS_ERR_BUFFER_TOO_SMALL);
    const CorrParams prm = {self_shift, self_weight, neg_shift, neg_weight};
    const long long* neg = reinterpret_cast<const long long*>(neg_indx);
    const hipStream_t st = (hipStream_t)stream;
#define NSOS_APP_ROWS(CC)                                                                                                                    \
    return app_rows_impl<CC>(phase, feats, code, neg, rand1, rand2, rows, n_rows, batch, feat_dim, feat_h, feat_w, code_h, code_w, feature_samples, \
                             prm, loss, grad_code, workspace, exchange_means, exchange_sums, st, channel_last != 0)
    switch (code_dim) {
        case 1: NSOS_APP_ROWS(1);
        case 2: NSOS_APP_ROWS(2);
        case 3: NSOS_APP_ROWS(3);
        default: NSOS_APP_ROWS(4);
    }
#undef NSOS_APP_ROWS
}
