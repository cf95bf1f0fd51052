// K2-G: positional encoding + MLP for ANY architecture the reference's constructors accept (models/nerf_mlp.py:40-64,
// models/nerf_net.py:22-56): depth, width, skip set, multires / multires_views (or no embedding), use_viewdirs = False
// (output_linear), the deep semantic head (sem_layer > 2), sem_dim, sem_with_geo.  The shipped architecture (8 x 256, skip 4,
// multires 10 / 4, view directions, the two-layer head) has its own hand-scheduled kernels (mlp_fused.hip and the 16-bit
// family); everything else renders through this one, on the same exact-fp32 matrix instruction (v_mfma_f32_32x32x2_f32:
// bitwise an fmaf chain); the backward of the same program (training, ray gradients) is further down.
//
// Mapping.  A workgroup = one tile of 32 points = 4 waves (one per SIMD) where the net's buffers leave room for TWO workgroups per CU,
// 8 waves (two per SIMD) where only one fits (template parameter NW, GenProgram::n_waves).  The network is a PROGRAM of dense ops
// (nsos_generic_mlp -> build_program below, mirroring MLP.forward line by line); every op reads its inputs from and writes
// its output to per-tile activation buffers in LDS, [feature][32 points] fp32 (a feature row = 128 B).  An op's output tiles
// (32 features each) are dealt to the waves round-robin, two at a time per wave (tiles t and t + NW: one B operand feeds both
// accumulators); the contraction runs over the op's input SEGMENTS (the concatenations of the reference: [input_pts, h] of
// the skip layer, [h, input_pts] of the semantic head, [feature, input_views]) in groups of 4 k-steps = 8 input rows:
//   B operand  lane (point j, hi): row 2 ks + hi of the segment, one ds_read_b32 per k-step (conflict-free: 64 consecutive words)
//   A operand  one global_load_dwordx4 per group and tile from the packed stream [tile][group][lane][4] (L2-resident: the
//              whole net is 0.3-3 MB; every wave reads only ITS tiles' stream), prefetched four groups ahead -- no LDS staging, no
//              barrier inside an op
//   bias       the accumulators' initial value, from a per-tile table in the packed stream (round 5).  Until then the bias was a leading
//              GROUP on a constant buffer [1, 0, ...]: 4 k-steps for one useful row, and it pushed a 256-wide layer from 32 groups to 33,
//              padded to 36 -- 11 % of every trunk layer's matrix time (20 % of a 128-wide head's) spent on zeros
// One __syncthreads per op; ops with ONE output tile of a few rows (alpha, rgb, output_linear) are split over K across the waves
// (ksplit_off) and cost a second one.  The op descriptor lives in the constant address space: every scalar of it that an epilogue uses is
// read once and pinned in an SGPR (left to itself hipcc re-issued the s_load in front of every store, see the kernel).
// Measured (round 5, profiles/r05): 0.79-0.86 of the fp32-MFMA peak at widths >= 256, a 256 x 256 op 33-34 k cycles against 32.8 k of
// MFMA issue with two workgroups sharing the pipe.
// Ceiling: 2 x 64-cycle MFMAs per k-step and wave against 16 B/lane of A operand per 4 k-steps: ~8 B/clk/CU from L2.
#include "mlp_common.h"

using namespace nsos;

#ifdef NSOS_GEN_PROF   // diagnostic build (scripts/diag/gen_prof.py): s_memtime stamps per op and phase, workgroup 1, every wave
__device__ unsigned long long nsos_gen_prof[4][2][64];   // [wave][0: work, 1: barrier wait][slot: 0 = tiles, 1 = encode, 2 + op, 62 = output]
#define GEN_PROF_T() __builtin_amdgcn_s_memtime()
#endif
namespace {

constexpr int kGenMaxOps = 64, kGenMaxGroups = 112, kGenMaxSeg = 3;   // (ops: the backward program of a 16-deep net with the deepest head has 61)
// One LDS row = one feature of the tile's points.  32 points (128 B) while the activation buffers fit the LDS; nets too wide for that
// (W > 256 with the deep semantic head, > 320 without) run on 16-POINT tiles: the buffers halve, lanes 16..31 of a half-wave read the
// same 16 points again (the matrix instruction's B operand has 32 columns: half of them are duplicates, results of duplicates are
// dropped) -- half the matrix rate, the same arithmetic per point.  Template parameter RF of the kernels, `row_floats` of the program.

enum { kGenDense = 0, kGenMul = 1, kGenBwdHead = 2, kGenBwdMul = 3 };
struct GenOp {
    int kind;        // kGenDense / kGenMul; the backward program (build_bwd_program) adds kGenBwdHead / kGenBwdMul
    int out_off;     // LDS float offset of the op's output row 0
    int out_dim;     // rows written (dense: out features; mul: rows multiplied)
    int out_tiles;
    int relu;        // bit 0: ReLU; bit 1: also write zeros into the pad rows of the last tile (a whole buffer is this op's output);
                     // bit 2 (backward dense): add to what the buffer holds (a second consumer's contribution)
    int n_groups;    // groups of 4 k-steps over all segments (a multiple of 4: padded with zero-weight groups on the constant buffer)
    int w_off;       // float offset of the op's A stream in the packed weights
    int src_off;     // mul: LDS float offset of the factor rows
    int act_col;     // first column of the op's block (pad32(out_dim) columns) in the saved-activation / gradient rows (training)
    int aux_col;     // kGenBwdMul: the column block of geo_map_sem's output
    int mask_col;    // ReLU ops (forward dense op / its backward head step), -1 = none: the column of the point's saved row that holds the op's
                     // ReLU pattern as BITS -- one 32-bit word per output tile: bit r + 16 hi = [accumulator register r of lane (point, hi) > 0]
                     // (round 5: the chain read 4 bytes of saved activation per mask bit, and the backward of a frozen net stored them for nothing else)
    int ksplit_off;  // forward dense op, 0 = none: a ONE-tile op with <= 8 outputs (alpha, rgb, output_linear) whose K range is split over the
                     // four waves -- LDS float offset of 24 dead rows (3 slabs x 8) for the partial sums of waves 1..3; n_groups is then a multiple of 16
    int b_off;       // forward dense op: float offset of the bias table [out_tiles][2 (hi)][16] in the packed weights -- the accumulators'
                     // initial values (acc = bias, then the fmaf chain: the same numbers as a leading fma(bias, 1, 0)); 0: none (backward ops)
    int grp_off[kGenMaxGroups + 8];   // LDS float offset of the first input row of group g (entries past n_groups: the constant buffer)
};
struct GenProgram {              // at the head of the packed buffer (device memory); identical on the host (build_program)
    int n_ops, lds_floats, n_out, out_off;
    int x_off, x_rows, x_dim, x_freqs;      // encoded xyz: buffer, padded rows, real rows, octaves (-1: raw coordinates)
    int v_off, v_rows, v_dim, v_freqs;      // encoded view direction (v_dim = 0 without view directions)
    int n_waves;     // waves per workgroup the program is laid out for (4 or 8: see mlp_generic_kernel)
    int ones_off, w_floats, act_ld, input_grads;   // input_grads (backward program): the chain also fills the encodings' gradient rows   // act_ld: floats per point of the saved activations = columns of X | V | every dense op
    int x_col, v_col, row_floats, out_rows;
    GenOp ops[kGenMaxOps];
};

struct GenParams {
    const GenProgram* prog;
    const float* wts;
    const float* rays_o; const float* rays_d; const float* viewdirs; const float* z_vals;   // ray mode
    const float* pts; const float* dirs;                                                      // point mode
    const float* enc;     // pre-encoded mode (MLP.forward's own input, models/nerf_mlp.py:67-68): [n_pts, x_dim + v_dim], encodings as given
    float* raw;
    float* acts;          // training variant: [n_pts, act_ld] post-activation outputs of every dense op + both encodings
    unsigned long long save_mask;   // ... bit oi: op oi's block is stored; bit 62 / 63: the xyz / direction encoding (all ones: everything)
    long long n_pts;
    int n_samples;
    int n_tiles;
};

__device__ __forceinline__ f32x16 mfma_f32(float a, float b, f32x16 c) { return __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, c, 0, 0, 0); }

// encoded feature f of a 3-vector (models/embedder.py:34-48): [x y z | sin(2^0 x..z) cos(2^0 x..z) | sin(2^1 ..) ...]
__device__ __forceinline__ float gen_feature(const float (&x)[3], int f, int dim, int freqs) {
    if (f >= dim) return 0.0f;
    if (f < 3 || freqs < 0) return f == 0 ? x[0] : (f == 1 ? x[1] : x[2]);   // (freqs < 0: use_embed = False, dim = 3)
    const int m = f - 3, k = m / 6, r = m - 6 * k, c = r >= 3 ? r - 3 : r;
    const float xv = c == 0 ? x[0] : (c == 1 ? x[1] : x[2]);
    const float a = xv * __builtin_bit_cast(float, (unsigned)(127 + k) << 23);   // 2^k, exact (freq_bands = 2 ** linspace(0, L-1, L))
    float sn, cs;
    sincos_pe(a, sn, cs);
    return r >= 3 ? cs : sn;
}

typedef const __attribute__((address_space(4))) GenOp& GenOpRef;

// The stores of saved activations and gbuf.  (Round 5 tried non-temporal stores here, as the tuned 16-bit kernel's training variant uses
// for its 700 MB per launch: no gain on these kernels -- training steps 26.7 -> 28.3 / 46.0 -> 45.4 / 113.8 -> 115.1 ms across two boxes --
// so they stay plain.)
__device__ __forceinline__ void gen_store4(float* p, const f32x4 v) { *reinterpret_cast<f32x4*>(p) = v; }

template <int OFF>
__device__ __forceinline__ void gen_ld_off(f32x4& v, unsigned voff, unsigned long long base) {
    asm volatile("global_load_dwordx4 %0, %1, %2 offset:%3" : "=v"(v) : "v"(voff), "s"(base), "i"(OFF) : "memory");
}
__device__ __forceinline__ unsigned long long gen_uniform64(const void* p) {   // the pointer is wave-uniform: say so (see sem_wgrad16.hip)
    const unsigned long long b = (unsigned long long)p;
    unsigned long long u = ((unsigned long long)(unsigned)__builtin_amdgcn_readfirstlane((unsigned)(b >> 32)) << 32) |
                           (unsigned long long)(unsigned)__builtin_amdgcn_readfirstlane((unsigned)b);
    asm volatile("s_nop 4" : "+s"(u));     // readfirstlane -> SGPR base of a vector-memory instruction: five wait states
    return u;
}

// The contraction of one op for output tiles t0 (and t0 + 4 if TWO): groups of 4 k-steps, A operands prefetched FOUR groups
// ahead through a register ring with hand-counted waits (the loads are asm: left to itself hipcc rotates the ring with copies
// and drains the queue at every group).  n_groups is a multiple of 4 (the packer pads with zero-weight groups on the constant
// buffer); the ring's last four loads run up to 4 KiB past the tile's stream (into the next tile's, or the buffer's tail pad).
// BIAS: the accumulators start from the op's bias table (forward; b_off = its float offset): the table's loads are issued BEHIND the
// ring's first eight and everything is awaited once -- one L2 round trip in front of an op's first MFMA instead of two (the table
// first, waited for, then the ring: what hipcc made of loads written in front of the call).
template <bool TWO, int RF, bool BIAS = false, int NW = 4>
__device__ __forceinline__ void dense_tiles(GenOpRef op, const float* wts, const float* lds, int t0, int lane, int pt, int hi,
                                            f32x16& acc0, f32x16& acc1, int b_off = 0, int g0 = 0, int ng_part = 0) {
    constexpr int kGenRowFloats = RF;
    pt &= RF - 1;                          // 16-point tiles: columns 16..31 of the B operand repeat columns 0..15
    const int ng_all = op.n_groups;
    const int ng = ng_part ? ng_part : ng_all;          // (K-split ops: this wave's groups g0 .. g0 + ng_part - 1 of the tile's stream)
    const unsigned long long base0 = gen_uniform64(wts + op.w_off + (size_t)t0 * ng_all * 256 + (size_t)g0 * 256);
    const unsigned long long base1 = gen_uniform64(wts + op.w_off + (size_t)(t0 + (TWO ? NW : 0)) * ng_all * 256);
    unsigned voff = (unsigned)lane * 16u;
    f32x4 r0[4], r1[4];
    gen_ld_off<0>(r0[0], voff, base0);
    if constexpr (TWO) gen_ld_off<0>(r1[0], voff, base1);
    gen_ld_off<1024>(r0[1], voff, base0);
    if constexpr (TWO) gen_ld_off<1024>(r1[1], voff, base1);
    gen_ld_off<2048>(r0[2], voff, base0);
    if constexpr (TWO) gen_ld_off<2048>(r1[2], voff, base1);
    gen_ld_off<3072>(r0[3], voff, base0);
    if constexpr (TWO) gen_ld_off<3072>(r1[3], voff, base1);
    voff += 4096u;
    if constexpr (BIAS) {   // register r of lane (j, hi) is output feature 32 t + (r & 3) + 8 (r >> 2) + 4 hi: table [tile][hi][16], tile t0 + 4 is 512 B on
        const unsigned long long bb = gen_uniform64(wts + b_off + (size_t)t0 * 32);
        const unsigned boff = (unsigned)hi * 64u;
        f32x4 v0[4], v1[4];
        gen_ld_off<0>(v0[0], boff, bb); gen_ld_off<16>(v0[1], boff, bb); gen_ld_off<32>(v0[2], boff, bb); gen_ld_off<48>(v0[3], boff, bb);
        if constexpr (TWO) { gen_ld_off<128 * NW>(v1[0], boff, bb); gen_ld_off<128 * NW + 16>(v1[1], boff, bb); gen_ld_off<128 * NW + 32>(v1[2], boff, bb); gen_ld_off<128 * NW + 48>(v1[3], boff, bb); }
        asm volatile("s_waitcnt vmcnt(0)" : "+v"(v0[0]), "+v"(v0[1]), "+v"(v0[2]), "+v"(v0[3]), "+v"(r0[0]), "+v"(r0[1]), "+v"(r0[2]), "+v"(r0[3]) : : "memory");
        if constexpr (TWO) asm volatile("" : "+v"(v1[0]), "+v"(v1[1]), "+v"(v1[2]), "+v"(v1[3]), "+v"(r1[0]), "+v"(r1[1]), "+v"(r1[2]), "+v"(r1[3]));
#pragma unroll
        for (int q = 0; q < 4; ++q)
#pragma unroll
            for (int j = 0; j < 4; ++j) { acc0[4 * q + j] = v0[q][j]; if constexpr (TWO) acc1[4 * q + j] = v1[q][j]; }
    }
    // B operands one group ahead (LDS latency under the previous group's MFMAs); the last prefetch reads group `ng`: a valid
    // table entry (the table is padded past kGenMaxGroups) pointing at the constant buffer
    const float* bp0 = lds + op.grp_off[g0] + hi * kGenRowFloats + pt;
    float bn0 = bp0[0], bn1 = bp0[2 * kGenRowFloats], bn2 = bp0[4 * kGenRowFloats], bn3 = bp0[6 * kGenRowFloats];
    for (int g = 0; g < ng; g += 4) {
        static_for<0, 4>([&](auto uc) {
            constexpr int u = decltype(uc)::value;
            const float b0 = bn0, b1 = bn1, b2 = bn2, b3 = bn3;
            const float* bp = lds + op.grp_off[g0 + g + u + 1] + hi * kGenRowFloats + pt;
            bn0 = bp[0]; bn1 = bp[2 * kGenRowFloats]; bn2 = bp[4 * kGenRowFloats]; bn3 = bp[6 * kGenRowFloats];
            // group g + u has landed when at most the loads of the three younger groups are outstanding
            if constexpr (TWO) asm volatile("s_waitcnt vmcnt(6)" : "+v"(r0[u]), "+v"(r1[u]) : : "memory");
            else asm volatile("s_waitcnt vmcnt(3)" : "+v"(r0[u]) : : "memory");
            acc0 = mfma_f32(r0[u][0], b0, acc0);
            if constexpr (TWO) acc1 = mfma_f32(r1[u][0], b0, acc1);
            acc0 = mfma_f32(r0[u][1], b1, acc0);
            if constexpr (TWO) acc1 = mfma_f32(r1[u][1], b1, acc1);
            acc0 = mfma_f32(r0[u][2], b2, acc0);
            if constexpr (TWO) acc1 = mfma_f32(r1[u][2], b2, acc1);
            acc0 = mfma_f32(r0[u][3], b3, acc0);
            if constexpr (TWO) acc1 = mfma_f32(r1[u][3], b3, acc1);
            NSOS_PIN();
            gen_ld_off<u * 1024>(r0[u], voff, base0);     // group g + u + 4
            if constexpr (TWO) gen_ld_off<u * 1024>(r1[u], voff, base1);
            NSOS_PIN();
        });
        voff += 4096u;
    }
    asm volatile("s_waitcnt vmcnt(0)" : "+v"(r0[0]), "+v"(r0[1]), "+v"(r0[2]), "+v"(r0[3]) : : "memory");   // the overshoot loads: drained, unused
    if constexpr (TWO) asm volatile("" : "+v"(r1[0]), "+v"(r1[1]), "+v"(r1[2]), "+v"(r1[3]));
}

// SAVE: the training variant -- every dense op's post-activation output tile and both encodings also go to P.acts (point-major
// rows of act_ld floats, one 32-aligned column block per op: the layout nsos_wgrad reads its X operand in), straight from the
// accumulators: lane (point, hi) holds features 8 q + 4 hi + (0..3) of its point = one 16-byte store per q.
// NW: waves per workgroup.  4: one per SIMD -- nets whose buffers leave room for TWO workgroups per CU (one's barriers, epilogues and
// load latencies run under the other's MFMAs).  8: two per SIMD inside ONE workgroup -- nets whose buffers (three 256-row buffers:
// deep semantic heads, sem_with_geo; wide nets) let only one workgroup live on a CU: with four waves every exposed cycle was the CU's.
// A wave then owns tiles t and t + NW of an op (a 256-wide layer: one tile per wave).
template <bool SAVE, int RF, int NW>
__global__ __launch_bounds__(64 * NW) void mlp_generic_kernel(const GenParams P) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    constexpr int kGenRowFloats = RF, NP = 64 * NW / RF;      // points per tile; thread (point p, part): NP parts share a point's rows
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int pt = lane & 31, hi = lane >> 5;
    const bool own = pt < RF;              // (16-point tiles: lanes 16..31 of a half-wave hold duplicates)
    // the program is read-only and every access is wave-uniform: through the CONSTANT address space the loads are scalar (s_load,
    // lgkm counter) -- as vector-memory loads they shared the vm counter with the A-operand prefetches, and waiting for a group's
    // LDS offset drained the whole prefetch queue
    const __attribute__((address_space(4))) GenProgram& G = *(const __attribute__((address_space(4))) GenProgram*)P.prog;
    const int n_ops = G.n_ops, n_out = G.n_out, out_off = G.out_off;
    {   // a constant buffer (8 finite rows): the B operand of the zero-weight pad groups
        const int row = tid / RF, col = tid % RF;
        if (row < 8) lds[G.ones_off + row * kGenRowFloats + col] = row == 0 ? 1.0f : 0.0f;
    }
#ifdef NSOS_GEN_PROF
    unsigned long long pw[64], pb[64];
    for (int k = 0; k < 64; ++k) pw[k] = pb[k] = 0;
#endif
    for (int tile = blockIdx.x; tile < P.n_tiles; tile += gridDim.x) {
#ifdef NSOS_GEN_PROF
        unsigned long long tp = GEN_PROF_T(), tq;
        pw[0] += 1;
#define GEN_PROF_WORK(k) do { tq = GEN_PROF_T(); pw[k] += tq - tp; tp = tq; } while (0)
#define GEN_PROF_BAR(k) do { tq = GEN_PROF_T(); pb[k] += tq - tp; tp = tq; } while (0)
#else
#define GEN_PROF_WORK(k)
#define GEN_PROF_BAR(k)
#endif
        // ---- inputs and encodings: thread (point p, part): features part, part + 8, ...
        const int p = tid % RF, part = tid / RF;
        const long long gp = (long long)tile * RF + p;
        const bool valid = gp < P.n_pts;
        const long long gc = valid ? gp : P.n_pts - 1;
        float x[3] = {0.0f, 0.0f, 0.0f}, dv[3] = {0.0f, 0.0f, 0.0f};
        const float* erow = P.enc ? P.enc + gc * (G.x_dim + G.v_dim) : nullptr;
        if (P.enc) {
        } else if (P.pts) {
#pragma unroll
            for (int c = 0; c < 3; ++c) {
                x[c] = P.pts[3 * gc + c];
                if (G.v_dim) dv[c] = P.dirs[3 * gc + c];
            }
        } else {
            const long long ray = gc / P.n_samples;
            const float z = P.z_vals[gc];
#pragma unroll
            for (int c = 0; c < 3; ++c) {
                const float m = P.rays_d[3 * ray + c] * z;      // models/sampler.py:70,166 (mul, then add)
                x[c] = P.rays_o[3 * ray + c] + m;
                if (G.v_dim) dv[c] = P.viewdirs[3 * ray + c];
            }
        }
        // NaN / Inf in a point's inputs must come out as NaN (v_max(0, NaN) = 0 would launder them at the first ReLU)
        float poison = ((x[0] - x[0]) + (x[1] - x[1])) + ((x[2] - x[2]) + (dv[0] - dv[0])) + ((dv[1] - dv[1]) + (dv[2] - dv[2]));
        // encoded feature f of the point: evaluated here, or (pre-encoded mode) read as given
        auto x_feat = [&](int f) { return erow ? (f < G.x_dim ? erow[f] : 0.0f) : gen_feature(x, f, G.x_dim, G.x_freqs); };
        auto v_feat = [&](int f) { return erow ? (f < G.v_dim ? erow[G.x_dim + f] : 0.0f) : gen_feature(dv, f, G.v_dim, G.v_freqs); };
        if (erow)      // (every part of a point scans the whole row: each writes some of the point's output channels)
            for (int f = 0; f < G.x_dim + G.v_dim; ++f) poison += erow[f] - erow[f];
        // One sincos per (octave, coordinate) and point: it yields TWO rows (sin: 3 + 6 k + c, cos: + 3), written to LDS and -- SAVE -- to the
        // point's saved row in the same pass.  (Until round 5 every ROW was evaluated on its own, both functions computed and one dropped, and
        // the SAVE variant evaluated everything a second time: the encode phase was 6.6 % of an 8 x 256 tile, profiles/r05.)
        auto encode = [&](const float (&xv)[3], int dim, int rows, int freqs, int off, bool save, int col) {
            float* arow = (SAVE && save && valid) ? P.acts + gp * G.act_ld + col : nullptr;
            const int n_units = freqs < 0 ? 3 : 3 + 3 * freqs;
            for (int u = part; u < n_units; u += NP) {
                if (u < 3) {
                    const float v = u == 0 ? xv[0] : (u == 1 ? xv[1] : xv[2]);
                    lds[off + u * kGenRowFloats + p] = v;
                    if (SAVE && arow) arow[u] = v;
                } else {
                    const int m = u - 3, k = m / 3, c = m - 3 * k, r = 3 + 6 * k + c;
                    const float a = (c == 0 ? xv[0] : (c == 1 ? xv[1] : xv[2])) * __builtin_bit_cast(float, (unsigned)(127 + k) << 23);   // 2^k, exact
                    float sn, cs;
                    sincos_pe(a, sn, cs);
                    lds[off + r * kGenRowFloats + p] = sn;
                    lds[off + (r + 3) * kGenRowFloats + p] = cs;
                    if (SAVE && arow) { arow[r] = sn; arow[r + 3] = cs; }
                }
            }
            for (int f = dim + part; f < rows; f += NP) lds[off + f * kGenRowFloats + p] = 0.0f;                 // the buffer's pad rows
            if (SAVE && arow) for (int f = dim + part; f < ((dim + 31) & ~31); f += NP) arow[f] = 0.0f;        // the block's pad columns
        };
        if (erow) {
            for (int f = part; f < G.x_rows; f += NP) lds[G.x_off + f * kGenRowFloats + p] = x_feat(f);
            for (int f = part; f < G.v_rows; f += NP) lds[G.v_off + f * kGenRowFloats + p] = v_feat(f);
            if constexpr (SAVE) {
                if (valid) {       // (blocks are padded to 32 columns with zeros)
                    float* arow = P.acts + gp * G.act_ld;
                    if ((P.save_mask >> 62) & 1) for (int f = part; f < ((G.x_dim + 31) & ~31); f += NP) arow[G.x_col + f] = x_feat(f);
                    if ((P.save_mask >> 63) & 1) for (int f = part; f < ((G.v_dim + 31) & ~31); f += NP) arow[G.v_col + f] = v_feat(f);
                }
            }
        } else {
            encode(x, G.x_dim, G.x_rows, G.x_freqs, G.x_off, (P.save_mask >> 62) & 1, G.x_col);
            if (G.v_dim) encode(dv, G.v_dim, G.v_rows, G.v_freqs, G.v_off, (P.save_mask >> 63) & 1, G.v_col);
        }
        for (int r = part; r < G.out_rows; r += NP) lds[out_off + r * kGenRowFloats + p] = 0.0f;
        GEN_PROF_WORK(1);
        __syncthreads();
        GEN_PROF_BAR(1);

        for (int oi = 0; oi < n_ops; ++oi) {
            const __attribute__((address_space(4))) GenOp& op = G.ops[oi];
            if (op.kind == kGenMul) {       // semantics * geo_map_sem(alpha) (models/nerf_mlp.py:81-83)
                for (int r = part; r < op.out_dim; r += NP) lds[op.out_off + r * kGenRowFloats + p] *= lds[op.src_off + r * kGenRowFloats + p];
                __syncthreads();
                continue;
            }
            // the op's scalars, read ONCE from the program and pinned in SGPRs: as plain constant-address-space reads hipcc re-issued the
            // s_load of op.out_off (and waited for it with lgkmcnt(0)) in front of EVERY accumulator store of the epilogue -- 32 exposed
            // scalar-cache round trips per op and wave, about a quarter of a 256 x 256 layer's time (round 5, found in the ISA)
            int out_tiles = op.out_tiles, o_off = op.out_off, o_dim = op.out_dim, o_flags = op.relu, o_act = op.act_col, o_boff = op.b_off, o_mcol = op.mask_col;
            asm volatile("" : "+s"(out_tiles), "+s"(o_off), "+s"(o_dim), "+s"(o_flags), "+s"(o_act), "+s"(o_boff), "+s"(o_mcol));
            const bool relu = o_flags & 1, padw = o_flags & 2;
            int o_ks = op.ksplit_off;
            asm volatile("" : "+s"(o_ks));
            if (o_ks) {
                // ONE output tile with a few real rows (alpha 1, rgb 3, output_linear 4 ..): as a plain op wave 0 ran its whole K range while three
                // waves waited at the barrier -- 6.4 % + 4.3 % of an 8 x 256 tile's time (profiles/r05).  Split over K instead: wave w
                // contracts groups w ng/4 .. (w+1) ng/4 - 1 (wave 0 starts from the bias), waves 1..3 park rows 0..7 of their partial tiles in
                // dead LDS rows, wave 0 adds them in a fixed order ((own + w1) + w2) + w3 and finishes the op as before.
                f32x16 acc0, acc1;
#pragma unroll
                for (int r = 0; r < 16; ++r) acc0[r] = acc1[r] = 0.0f;
                const int gq = op.n_groups / NW;              // (a multiple of 4: the builder pads a K-split op's groups to 4 NW)
                if (wave == 0) dense_tiles<false, RF, true, NW>(op, P.wts, lds, 0, lane, pt, hi, acc0, acc1, o_boff, 0, gq);
                else dense_tiles<false, RF, false, NW>(op, P.wts, lds, 0, lane, pt, hi, acc0, acc1, 0, wave * gq, gq);
                float* slab = lds + o_ks + (4 * hi) * kGenRowFloats + pt;       // accumulator register r < 4 of lane (j, hi) = row r + 4 hi
                if (wave != 0 && (RF == 32 || own)) {
#pragma unroll
                    for (int r = 0; r < 4; ++r) slab[((wave - 1) * 8 + r) * kGenRowFloats] = acc0[r];
                }
                __syncthreads();
                if (wave == 0) {
                    if (RF == 32 || own) {
#pragma unroll
                        for (int r = 0; r < 4; ++r) {
                            float v = acc0[r];
#pragma unroll
                            for (int w = 0; w < NW - 1; ++w) v += slab[(8 * w + r) * kGenRowFloats];      // fixed order: ((own + w1) + w2) + ...
                            acc0[r] = v;
                        }
                    }
                    if (relu) {
#pragma unroll
                        for (int r = 0; r < 4; ++r) acc0[r] = fmaxf(acc0[r], 0.0f);
                    }
                    if (RF == 32 || own) {
#pragma unroll
                        for (int r = 0; r < 4; ++r)
                            if (r + 4 * hi < o_dim) lds[o_off + (r + 4 * hi) * kGenRowFloats + pt] = acc0[r];
                    }
                    if constexpr (SAVE) {
                        const long long gpl = (long long)tile * RF + pt;
                        if (own && gpl < P.n_pts && ((P.save_mask >> oi) & 1)) {
                            float* dst = P.acts + gpl * G.act_ld + o_act + 4 * hi;
#pragma unroll
                            for (int q = 0; q < 4; ++q) {
                                f32x4 v0;
#pragma unroll
                                for (int j = 0; j < 4; ++j) v0[j] = acc0[4 * q + j];       // (rows >= 8: zero weights and zero bias, exact zeros)
                                gen_store4(dst + 8 * q, v0);
                            }
                        }
                    }
                }
                GEN_PROF_WORK(2 + (oi < 58 ? oi : 58));
                __syncthreads();
                GEN_PROF_BAR(2 + (oi < 58 ? oi : 58));
                continue;
            }
            for (int t0 = wave; t0 < out_tiles; t0 += 2 * NW) {
                f32x16 acc0, acc1;
                const bool two = t0 + NW < out_tiles;         // wave-uniform
#pragma unroll
                for (int r = 0; r < 16; ++r) acc1[r] = 0.0f;       // (one-tile items: acc1 is stored nowhere, but read by the ReLU pass)
                if (two) dense_tiles<true, RF, true, NW>(op, P.wts, lds, t0, lane, pt, hi, acc0, acc1, o_boff);
                else dense_tiles<false, RF, true, NW>(op, P.wts, lds, t0, lane, pt, hi, acc0, acc1, o_boff);
                if (relu) {     // (asm: fmaxf is two instructions -- a canonicalising v_max x, x in front of the v_max 0, x; same values, a NaN -> 0 either way)
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        float a = acc0[r], b = acc1[r];
                        asm("v_max_f32 %0, 0, %0" : "+v"(a));
                        asm("v_max_f32 %0, 0, %0" : "+v"(b));
                        acc0[r] = a; acc1[r] = b;
                    }
                }
                // accumulator register r of lane (j, hi) = output feature 32 t + (r & 3) + 8 (r >> 2) + 4 hi of point j: row stride RF floats, so
                // the 16 stores of a tile are one base address + compile-time offsets.  (Rows past out_dim have all-zero weights: their
                // accumulators are exactly 0 -- written only where the buffer's pad rows are read by a later op as zero-weighted inputs, never
                // into the shared OUT buffer.)  A tile wholly inside the op's rows (every tile of a 32-multiple width) takes the branch-free path.
                if (RF == 32 || own) {
                    float* d0 = lds + o_off + (32 * t0 + 4 * hi) * kGenRowFloats + pt;
                    const int rb = 32 * t0 + 4 * hi;
                    if (padw || 32 * t0 + 32 <= o_dim) {
#pragma unroll
                        for (int r = 0; r < 16; ++r) d0[((r & 3) + 8 * (r >> 2)) * kGenRowFloats] = acc0[r];
                    } else {
#pragma unroll
                        for (int r = 0; r < 16; ++r)
                            if (rb + (r & 3) + 8 * (r >> 2) < o_dim) d0[((r & 3) + 8 * (r >> 2)) * kGenRowFloats] = acc0[r];
                    }
                    if (two) {
                        float* d1 = d0 + 32 * NW * kGenRowFloats;
                        if (padw || 32 * (t0 + NW) + 32 <= o_dim) {
#pragma unroll
                            for (int r = 0; r < 16; ++r) d1[((r & 3) + 8 * (r >> 2)) * kGenRowFloats] = acc1[r];
                        } else {
#pragma unroll
                            for (int r = 0; r < 16; ++r)
                                if (rb + 32 * NW + (r & 3) + 8 * (r >> 2) < o_dim) d1[((r & 3) + 8 * (r >> 2)) * kGenRowFloats] = acc1[r];
                        }
                    }
                }
                if constexpr (SAVE) {
                    const long long gpl = (long long)tile * RF + pt;
                    if (o_mcol >= 0 && own && gpl < P.n_pts) {      // the ReLU pattern as bits: 2 bytes per lane and tile (always stored: the chain's masks)
                        unsigned m0 = 0, m1 = 0;
#pragma unroll
                        for (int r = 0; r < 16; ++r) { m0 |= (acc0[r] > 0.0f ? 1u : 0u) << r; m1 |= (acc1[r] > 0.0f ? 1u : 0u) << r; }
                        unsigned short* mrow = reinterpret_cast<unsigned short*>(P.acts + gpl * G.act_ld + o_mcol);
                        mrow[2 * t0 + hi] = (unsigned short)m0;
                        if (two) mrow[2 * (t0 + NW) + hi] = (unsigned short)m1;
                    }
                    if (own && gpl < P.n_pts && ((P.save_mask >> oi) & 1)) {
                        float* dst = P.acts + gpl * G.act_ld + o_act + 32 * t0 + 4 * hi;
#pragma unroll
                        for (int q = 0; q < 4; ++q) {
                            f32x4 v0, v1;
#pragma unroll
                            for (int j = 0; j < 4; ++j) { v0[j] = acc0[4 * q + j]; v1[j] = acc1[4 * q + j]; }
                            gen_store4(dst + 8 * q, v0);                          // (rows past out_dim: exact zeros, the block's padding)
                            if (two) gen_store4(dst + 32 * NW + 8 * q, v1);
                        }
                    }
                }
            }
            GEN_PROF_WORK(2 + (oi < 58 ? oi : 58));
            __syncthreads();
            GEN_PROF_BAR(2 + (oi < 58 ? oi : 58));
        }
        // ---- raw[p, c] = OUT[c][p]   (models/nerf_mlp.py:93-98: cat([rgb, alpha, semantics]) / output_linear)
        if (valid)
            for (int c = part; c < n_out; c += NP) P.raw[gp * n_out + c] = poison != poison ? __builtin_nanf("") : lds[out_off + c * kGenRowFloats + p];
        GEN_PROF_WORK(62);
        __syncthreads();
        GEN_PROF_BAR(62);
    }
#ifdef NSOS_GEN_PROF
    if (blockIdx.x == 1 && lane == 0 && wave < 4)
        for (int k = 0; k < 64; ++k) { nsos_gen_prof[wave][0][k] = pw[k]; nsos_gen_prof[wave][1][k] = pb[k]; }
#endif
}


// ---------------------------------------------------------------------------------------------- backward (training)
// The input-gradient chain of the same program, reversed: gradient buffers live in LDS at the FORWARD buffers' offsets (a
// buffer's forward lifetime [producer, last consumer] is its gradient's lifetime read backwards, so the forward's ping-pong
// allocation is valid as it stands).  Per forward dense op o, in reverse order:
//   kGenBwdHead   dY_o = (gradient buffer of o's output) (x) [saved output > 0]  (ReLU ops; models/nerf_mlp.py:71-72,90), written back
//                 to LDS and -- point-major, zero-padded to the op's 32-column block -- to gbuf: the G operand of nsos_wgrad
//   kGenDense     for every input segment that carries a gradient (not the encodings): d(segment) (+)= W_o[:, segment]^T dY_o, the
//                 same dense_tiles contraction over a TRANSPOSED weight stream (gen_pack_t_kernel), K = o's output features
//   kGenBwdMul    semantics * mapping (models/nerf_mlp.py:81-83): d sem = g * mapping, d mapping = g * sem (saved values)
// The weight gradients dW_o = dY_o^T [segments], db_o = column sums are nsos_wgrad calls over (gbuf, acts) column blocks, sequenced
// by the host (ops.py) from nsos_mlp_generic_save_layout.  Exact fp32 MFMA throughout: the reference's autograd arithmetic up to
// summation order.
struct GenBwdParams {
    const GenProgram* prog;
    const float* wts;
    const float* g_raw;       // [n_pts, n_out]
    const float* acts;        // [n_pts, ld]
    float* gbuf;              // [n_pts, ld]
    // gradients w.r.t. the points and view directions (programs packed with input_grads; ray mode): the rays of the forward call
    const float* rays_o; const float* rays_d; const float* viewdirs; const float* z_vals;
    const float* pts; const float* dirs;      // ... or the points / per-point directions of a point query
    float* g_pts;             // [n_pts, 3]  d loss / d (o + d z)
    float* g_dirs;            // [n_pts, 3]  d loss / d viewdirs, per point (NULL without view directions)
    float* g_enc;             // pre-encoded mode: [n_pts, x_dim + v_dim]  d loss / d the encoded inputs themselves
    int n_samples;
    long long n_pts;
    int n_tiles;
};

template <int RF, int NW>
__global__ __launch_bounds__(64 * NW) void mlp_generic_bwd_kernel(const GenBwdParams P) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    constexpr int kGenRowFloats = RF, NP = 64 * NW / RF;
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int pt = lane & 31, hi = lane >> 5;
    const bool own = pt < RF;
    const __attribute__((address_space(4))) GenProgram& G = *(const __attribute__((address_space(4))) GenProgram*)P.prog;
    const int n_ops = G.n_ops, n_out = G.n_out, out_off = G.out_off, ld = G.act_ld;
    {
        const int row = tid / RF, col = tid % RF;
        if (row < 8) lds[G.ones_off + row * kGenRowFloats + col] = row == 0 ? 1.0f : 0.0f;     // (pad groups carry zero weights: any finite rows do)
    }
#ifdef NSOS_GEN_PROF
    unsigned long long pw[64], pb[64];
    for (int k = 0; k < 64; ++k) pw[k] = pb[k] = 0;
#endif
    for (int tile = blockIdx.x; tile < P.n_tiles; tile += gridDim.x) {
#ifdef NSOS_GEN_PROF
        unsigned long long tp = GEN_PROF_T(), tq;
        pw[0] += 1;
#endif
        const int p = tid % RF, part = tid / RF;
        const long long gp = (long long)tile * RF + p;
        const bool valid = gp < P.n_pts;
        const long long gc = valid ? gp : P.n_pts - 1;
        // d loss / d raw into the OUT rows (points past the end: zero gradient, so nothing of theirs reaches gbuf or a neighbour)
        for (int r = part; r < G.out_rows; r += NP) lds[out_off + r * kGenRowFloats + p] = (r < n_out && valid) ? P.g_raw[gp * n_out + r] : 0.0f;
        GEN_PROF_WORK(1);
        __syncthreads();
        GEN_PROF_BAR(1);
        for (int oi = 0; oi < n_ops; ++oi) {
            const __attribute__((address_space(4))) GenOp& op = G.ops[oi];
            if (op.kind == kGenBwdMul) {
                for (int r = part; r < op.out_dim; r += NP) {
                    const float g = lds[op.out_off + r * kGenRowFloats + p];
                    const float sv = P.acts[gc * ld + op.act_col + r], mv = P.acts[gc * ld + op.aux_col + r];
                    lds[op.out_off + r * kGenRowFloats + p] = g * mv;
                    lds[op.src_off + r * kGenRowFloats + p] = g * sv;
                }
                GEN_PROF_WORK(2 + (oi < 58 ? oi : 58));
                __syncthreads();
                GEN_PROF_BAR(2 + (oi < 58 ? oi : 58));
                continue;
            }
            if (op.kind == kGenBwdHead) {
                int h_tiles = op.out_tiles, h_off = op.out_off, h_dim = op.out_dim, h_flags = op.relu, h_act = op.act_col, h_mcol = op.mask_col;     // pinned: see the forward kernel
                asm volatile("" : "+s"(h_tiles), "+s"(h_off), "+s"(h_dim), "+s"(h_flags), "+s"(h_act), "+s"(h_mcol));
                const long long gpl = (long long)tile * RF + pt;
                const bool vpt = own && gpl < P.n_pts;
                // the ReLU pattern comes from the forward's bit words (2 bytes per lane and tile; bit r of half hi = accumulator register r =
                // feature 32 t + (r & 3) + 8 (r >> 2) + 4 hi: exactly the four-feature runs 8 q + 4 hi + j, r = 4 q + j, this step walks);
                // bit 3 of the flags: no weight-gradient reduction reads this op's gradient (a frozen Linear) -- nothing goes to gbuf
                const unsigned short* mrow = reinterpret_cast<const unsigned short*>(P.acts + (vpt ? gpl : P.n_pts - 1) * ld + (h_mcol >= 0 ? h_mcol : 0));
                float* grow = P.gbuf + gpl * ld + h_act;
                const bool relu = (h_flags & 1) && h_mcol >= 0, store = !(h_flags & 8);
                for (int t = wave; t < h_tiles; t += NW) {
                    float* lrow = lds + h_off + (32 * t + 4 * hi) * kGenRowFloats + pt;
                    const unsigned mw = relu ? (unsigned)mrow[2 * t + hi] : 0xffffu;
                    if (32 * t + 32 <= h_dim && (RF == 32 || own)) {       // a tile wholly inside the op's rows: no per-element tests
#pragma unroll
                        for (int q = 0; q < 4; ++q) {
                            const int f0 = 32 * t + 8 * q + 4 * hi;
                            f32x4 g;
#pragma unroll
                            for (int j = 0; j < 4; ++j) {
                                const float v = lrow[(8 * q + j) * kGenRowFloats];
                                g[j] = ((mw >> (4 * q + j)) & 1u) ? v : 0.0f;          // relu'(x) = [x > 0] (ATen threshold_backward)
                                if (relu) lrow[(8 * q + j) * kGenRowFloats] = g[j];   // (ReLU outputs own whole pad32 buffers)
                            }
                            if (vpt && store) gen_store4(grow + f0, g);
                        }
                        continue;
                    }
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        const int f0 = 32 * t + 8 * q + 4 * hi;
                        f32x4 g;
#pragma unroll
                        for (int j = 0; j < 4; ++j) {
                            const float v = (own && f0 + j < h_dim) ? lds[h_off + (f0 + j) * kGenRowFloats + pt] : 0.0f;
                            g[j] = ((mw >> (4 * q + j)) & 1u) ? v : 0.0f;
                            if (relu && own) lds[h_off + (f0 + j) * kGenRowFloats + pt] = g[j];
                        }
                        if (vpt && store) gen_store4(grow + f0, g);
                    }
                }
                GEN_PROF_WORK(2 + (oi < 58 ? oi : 58));
                __syncthreads();
                GEN_PROF_BAR(2 + (oi < 58 ? oi : 58));
                continue;
            }
            int out_tiles = op.out_tiles, o_off = op.out_off, o_dim = op.out_dim, o_flags = op.relu, o_fuse = op.aux_col, o_mcol = op.mask_col, o_act = op.act_col;   // pinned: see the forward kernel
            asm volatile("" : "+s"(out_tiles), "+s"(o_off), "+s"(o_dim), "+s"(o_flags), "+s"(o_fuse), "+s"(o_mcol), "+s"(o_act));
            const bool padw = o_flags & 2, add = o_flags & 4;
            // o_fuse != 0 (build_bwd_program): this op is the LAST contribution to the gradient of a forward op's output, and that op's head
            // step (ReLU mask from the bit words, the gradient's copy to gbuf) happens HERE, on the values in registers, instead of in a step of
            // its own behind one more barrier and one more pass over the buffer.  bit 0: mask; bit 3: nothing goes to gbuf (a frozen Linear)
            const bool f_relu = (o_fuse & 1) && o_mcol >= 0, f_store = o_fuse && !(o_fuse & 8);
            const long long gpl = (long long)tile * RF + pt;
            const bool vpt = own && gpl < P.n_pts;
            for (int t0 = wave; t0 < out_tiles; t0 += 2 * NW) {
                f32x16 acc0, acc1;
#pragma unroll
                for (int r = 0; r < 16; ++r) acc0[r] = acc1[r] = 0.0f;
                const bool two = t0 + NW < out_tiles;
                unsigned mw0 = 0xffffu, mw1 = 0xffffu;
                if (f_relu) {       // requested ahead of the contraction (older than every load of the ring: its counted waits still hold)
                    const unsigned short* mrow = reinterpret_cast<const unsigned short*>(P.acts + (vpt ? gpl : P.n_pts - 1) * ld + o_mcol);
                    mw0 = mrow[2 * t0 + hi];
                    if (two) mw1 = mrow[2 * (t0 + NW) + hi];
                }
                if (two) dense_tiles<true, RF, false, NW>(op, P.wts, lds, t0, lane, pt, hi, acc0, acc1);
                else dense_tiles<false, RF, false, NW>(op, P.wts, lds, t0, lane, pt, hi, acc0, acc1);
                if (RF == 32 || own) {
                    float* d0 = lds + o_off + (32 * t0 + 4 * hi) * kGenRowFloats + pt;
                    const int rb = 32 * t0 + 4 * hi;
                    auto put = [&](float* d, f32x16& acc, int rbase, bool full, unsigned mw, int t) {
                        if (full) {
                            if (add) {
                                float old[16];
#pragma unroll
                                for (int r = 0; r < 16; ++r) old[r] = d[((r & 3) + 8 * (r >> 2)) * kGenRowFloats];
#pragma unroll
                                for (int r = 0; r < 16; ++r) acc[r] = old[r] + acc[r];
                            }
                            if (f_relu) {
#pragma unroll
                                for (int r = 0; r < 16; ++r) acc[r] = ((mw >> r) & 1u) ? acc[r] : 0.0f;      // relu'(x) = [x > 0] (ATen threshold_backward)
                            }
#pragma unroll
                            for (int r = 0; r < 16; ++r) d[((r & 3) + 8 * (r >> 2)) * kGenRowFloats] = acc[r];
                            if (f_store && vpt) {       // the head step's store: register r = 4 q + j is feature 32 t + 8 q + 4 hi + j
                                float* grow = P.gbuf + gpl * ld + o_act + 32 * t + 4 * hi;
#pragma unroll
                                for (int q = 0; q < 4; ++q) gen_store4(grow + 8 * q, f32x4{acc[4 * q], acc[4 * q + 1], acc[4 * q + 2], acc[4 * q + 3]});
                            }
                        } else {
#pragma unroll
                            for (int r = 0; r < 16; ++r)
                                if (rbase + (r & 3) + 8 * (r >> 2) < o_dim) {
                                    float* e = d + ((r & 3) + 8 * (r >> 2)) * kGenRowFloats;
                                    *e = add ? *e + acc[r] : acc[r];
                                }
                        }
                    };
                    put(d0, acc0, rb, padw || 32 * t0 + 32 <= o_dim, mw0, t0);
                    if (two) put(d0 + 32 * NW * kGenRowFloats, acc1, rb + 32 * NW, padw || 32 * (t0 + NW) + 32 <= o_dim, mw1, t0 + NW);
                }
            }
            GEN_PROF_WORK(2 + (oi < 58 ? oi : 58));
            __syncthreads();
            GEN_PROF_BAR(2 + (oi < 58 ? oi : 58));
        }
        // ---- d loss / d point and / d view direction through the positional encodings (models/embedder.py:34-48):
        //      d/dx [x, sin(2^k x), cos(2^k x)] = [1, 2^k cos(2^k x), -2^k sin(2^k x)]; thread (point, part): part 0..2 = xyz, 3..5 = direction
        if (P.g_enc) {
            if (valid) {
                float* grow = P.g_enc + gp * (G.x_dim + G.v_dim);
                for (int f = part; f < G.x_dim; f += NP) grow[f] = lds[G.x_off + f * kGenRowFloats + p];
                for (int f = part; f < G.v_dim; f += NP) grow[G.x_dim + f] = lds[G.v_off + f * kGenRowFloats + p];
            }
        } else if (P.g_pts && part < 6) {
            const bool isdir = part >= 3;
            const int c = isdir ? part - 3 : part;
            if (!isdir || G.v_dim) {
                const long long ray = gc / P.n_samples;
                float xv;
                if (P.pts) xv = isdir ? P.dirs[3 * gc + c] : P.pts[3 * gc + c];
                else if (isdir) xv = P.viewdirs[3 * ray + c];
                else { const float m = P.rays_d[3 * ray + c] * P.z_vals[gc]; xv = P.rays_o[3 * ray + c] + m; }     // as the forward forms it
                const int off = isdir ? G.v_off : G.x_off, freqs = isdir ? G.v_freqs : G.x_freqs;
                double g = (double)lds[off + c * kGenRowFloats + p];         // (the octaves' terms carry factors up to 2^(L-1) and cancel: summed in fp64, rounded once)
                for (int k = 0; k < freqs; ++k) {
                    const float f = __builtin_bit_cast(float, (unsigned)(127 + k) << 23);
                    float sn, cs;
                    sincos_pe(xv * f, sn, cs);
                    const float gs = lds[off + (3 + 6 * k + c) * kGenRowFloats + p], gcs = lds[off + (6 + 6 * k + c) * kGenRowFloats + p];
                    g += (double)f * ((double)cs * (double)gs - (double)sn * (double)gcs);
                }
                if (valid) (isdir ? P.g_dirs : P.g_pts)[gp * 3 + c] = (float)g;
            }
        }
        GEN_PROF_WORK(62);
    }
#ifdef NSOS_GEN_PROF
    if (blockIdx.x == 1 && lane == 0 && wave < 4)
        for (int k = 0; k < 64; ++k) { nsos_gen_prof[wave][0][k] = pw[k]; nsos_gen_prof[wave][1][k] = pb[k]; }
#endif
}

// Per ray: the gradients of its points folded back onto the ray (autograd of pts = o + d z, models/sampler.py:70,166; of
// viewdirs = d / |d|, models/nerf_net.py:160-163; of dists * |d|, models/renderer.py:41).  One wave per ray.
//   g_o = sum_s g_pts,   g_d = sum_s z_s g_pts + (g_v - v (v . g_v)) / |d| + g_n d / |d|,
//   g_v = sum_s g_dirs,  g_n = sum_s g_sigma_s relu(sigma_s + noise_s std) / |d|   (d alpha / d |d| = d alpha / d sigma * relu(sigma) / |d|)
__global__ __launch_bounds__(256) void ray_grad_reduce_kernel(const float* __restrict__ g_pts, const float* __restrict__ g_dirs,
                                                              const float* __restrict__ z_vals, const float* __restrict__ rays_d,
                                                              const float* __restrict__ raw, const float* __restrict__ g_raw,
                                                              const float* __restrict__ noise, float noise_std, long long n_rays, int S, int C,
                                                              float* __restrict__ g_o, float* __restrict__ g_d) {
    const int lane = threadIdx.x & 63;
    const long long r = (long long)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (r >= n_rays) return;
    double so[3] = {0, 0, 0}, sz[3] = {0, 0, 0}, sv[3] = {0, 0, 0}, sn = 0;
    for (int s = lane; s < S; s += 64) {
        const long long q = r * S + s;
        const float z = z_vals[q];
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            const float g = g_pts[3 * q + c];
            so[c] += (double)g;
            sz[c] += (double)(g * z);
            if (g_dirs) sv[c] += (double)g_dirs[3 * q + c];
        }
        float sigma = raw[q * C + 3];
        if (noise) sigma = sigma + noise[q] * noise_std;
        sn += (double)(g_raw[q * C + 3] * fmaxf(sigma, 0.0f));
    }
#pragma unroll
    for (int c = 0; c < 3; ++c) { so[c] = nsos_wave_sum(so[c]); sz[c] = nsos_wave_sum(sz[c]); sv[c] = nsos_wave_sum(sv[c]); }
    sn = nsos_wave_sum(sn);
    if (lane == 0) {
        const double d[3] = {rays_d[3 * r], rays_d[3 * r + 1], rays_d[3 * r + 2]};
        const double n = sqrt(d[0] * d[0] + d[1] * d[1] + d[2] * d[2]);
        const double v[3] = {d[0] / n, d[1] / n, d[2] / n};
        const double vg = v[0] * sv[0] + v[1] * sv[1] + v[2] * sv[2];
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            g_o[3 * r + c] = (float)so[c];
            g_d[3 * r + c] = (float)(sz[c] + (sv[c] - v[c] * vg) / n + (sn / n) * v[c]);
        }
    }
}

// the transposed A stream of one (forward op, segment): [tile of segment rows][group of 8 output features][lane][4]
struct GenPackT {
    const float* w;
    int in_dim, col0, rows, k_dim, out_tiles, n_groups;
    float* out;
};
__global__ __launch_bounds__(256) void gen_pack_t_kernel(const GenPackT Q) {
    const long long gid = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    const long long total = (long long)Q.out_tiles * Q.n_groups * 256;
    if (gid >= total) return;
    const int j = (int)(gid & 3), lane = (int)((gid >> 2) & 63);
    const long long tg = gid >> 8;
    const int g = (int)(tg % Q.n_groups), t = (int)(tg / Q.n_groups);
    const int i = lane & 31, hi = lane >> 5, row = 32 * t + i, kr = 8 * g + 2 * j + hi;
    Q.out[gid] = (row < Q.rows && kr < Q.k_dim) ? Q.w[(long long)kr * Q.in_dim + Q.col0 + row] : 0.0f;
}

// one launch per op: the op's A stream [tile][group][lane][4]
struct GenPackOp {
    const float* w; const float* bias;
    int in_dim, out_dim, out_tiles, n_groups, n_seg;
    int seg_col0[kGenMaxSeg], seg_rows[kGenMaxSeg], seg_groups[kGenMaxSeg];
    float* out;
};
__global__ __launch_bounds__(256) void gen_pack_kernel(const GenPackOp Q) {
    const long long gid = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    const long long stream = (long long)Q.out_tiles * Q.n_groups * 256, total = stream + (long long)Q.out_tiles * 32;
    if (gid >= total) return;
    if (gid >= stream) {                                // the bias table behind the stream: [tile][hi][16] in accumulator-register order
        const int idx = (int)(gid - stream), t = idx >> 5, hi = (idx >> 4) & 1, r = idx & 15;
        const int row = 32 * t + (r & 3) + 8 * (r >> 2) + 4 * hi;
        Q.out[gid] = (Q.bias && row < Q.out_dim) ? Q.bias[row] : 0.0f;
        return;
    }
    const int j = (int)(gid & 3), lane = (int)((gid >> 2) & 63);
    const long long tg = gid >> 8;
    const int g = (int)(tg % Q.n_groups), t = (int)(tg / Q.n_groups);
    const int i = lane & 31, hi = lane >> 5, row = 32 * t + i;
    float v = 0.0f;
    if (row < Q.out_dim) {
        int gl = g, s = 0;
        while (s < Q.n_seg && gl >= Q.seg_groups[s]) { gl -= Q.seg_groups[s]; ++s; }
        const int kr = 8 * gl + 2 * j + hi;
        if (s < Q.n_seg && kr < Q.seg_rows[s]) v = Q.w[(long long)row * Q.in_dim + Q.seg_col0[s] + kr];     // (s == n_seg: a pad group)
    }
    Q.out[gid] = v;
}

// ---------------------------------------------------------------------------------------------- host: the program
struct HostSeg { int buf_off, rows, col0, src_col; };   // src_col: the column block of the saved activations holding the segment's rows
struct HostOp { GenOp op; const float* w; const float* bias; int in_dim; int n_seg; HostSeg seg[kGenMaxSeg]; int lin_id; int out_buf; };
struct HostProgram { GenProgram prog; HostOp hops[kGenMaxOps]; int32_t err; };

inline int pad_to(int v, int m) { return (v + m - 1) / m * m; }

// Mirrors MLP.__init__ / MLP.forward (models/nerf_mlp.py:40-100) and NeRFMLP.__init__ (:136-166) as a sequence of dense ops over
// LDS activation buffers.  Pure host arithmetic: pack and every launch rebuild it from the same description.
void build_program_rf(const nsos_generic_mlp& M, HostProgram& H, const int kGenRowFloats) {
    H = HostProgram{};
    GenProgram& G = H.prog;
    H.err = NSOS_OK;
    const int D = M.depth, W = M.width;
    if (D < 1 || D > NSOS_GENERIC_MAX_DEPTH || W < 1 || M.sem_layers < 0 || M.sem_layers > NSOS_GENERIC_MAX_SEM || M.xyz_freqs > 24 || M.dir_freqs > 24) { H.err = NSOS_ERR_UNSUPPORTED; return; }
    const int x_dim = M.xyz_freqs < 0 ? 3 : 3 + 6 * M.xyz_freqs;
    const int v_dim = M.use_viewdirs ? (M.dir_freqs < 0 ? 3 : 3 + 6 * M.dir_freqs) : 0;
    const int Wp = pad_to(W, 32), Hp = pad_to(W / 2 > 0 ? W / 2 : 1, 32);
    const bool sem = M.use_viewdirs && M.use_semantics && M.sem_layers > 0;     // without view directions the reference never runs the head (:97-98)
    const int sem_dim = sem ? M.sem_dim : 0;
    // sem_dim <= 8: OUT holds rows 4 .. 4 + sem_dim - 1 of logits (+ as many of geo_map_sem) in 16 (32) rows and the kernels read
    // them as ONE 8-row k-group from row 4 (9 .. 28 passed the old `<= 32 rows` check and wrote past the LDS allocation);
    // nsos_composite_backward's channel limit (4 + 8) is the same number.
    if (sem && (sem_dim < 1 || sem_dim > 8)) { H.err = NSOS_ERR_UNSUPPORTED; return; }
    // the skip set: layer i's output is concatenated with the input (i in skips) -- the LAST layer must not be one (the heads
    // take W inputs: the reference itself fails there)
    if (D >= 1 && ((M.skip_mask >> (D - 1)) & 1)) { H.err = NSOS_ERR_UNSUPPORTED; return; }
    // LDS buffers (float offsets): ONES | X | V | HA | HB | SA | OUT
    int off = 0;
    auto alloc = [&](int rows) { const int o = off; off += rows * kGenRowFloats; return o; };
    G.ones_off = alloc(8);
    G.x_dim = x_dim; G.x_rows = pad_to(x_dim, 8); G.x_freqs = M.xyz_freqs; G.x_off = alloc(G.x_rows);
    G.v_dim = v_dim; G.v_rows = pad_to(v_dim, 8); G.v_freqs = M.dir_freqs; G.v_off = alloc(G.v_rows > 0 ? G.v_rows : 0);
    const int HA = alloc(Wp), HB = alloc(Wp);
    // the head buffers: semantic chain (widths W ... W/2) and the view branch's hidden layer (W/2)
    const int sw = sem ? (M.sem_layers > 2 ? Wp : Hp) : 0;
    // Buffers of their own only where nothing dead can be borrowed.  The two-Linear semantic head keeps its hidden layer in the trunk's
    // free ping-pong buffer (the feature layer overwrites it afterwards: by then it is consumed), the view branch's hidden layer goes
    // into the trunk's OUTPUT buffer (dead once feature_linear has read it: alpha and the semantic head run before).  Only deep
    // semantic chains and geo_map_sem's hidden layer need a buffer of their own (SA).  A shipped-like net (W = 256, view directions, two-Linear head) then
    // takes 79 KiB per tile instead of 97: TWO workgroups per CU (one's barriers and single-tile heads under the other's MFMAs).
    // A deep semantic chain alternates between that free ping-pong buffer and ONE buffer of its own (SA); geo_map_sem's hidden layer
    // runs after the chain and takes SA as well.
    const bool own_sem_bufs = sem && (M.sem_layers > 2 || M.sem_with_geo);
    const int sa_rows = own_sem_bufs ? (M.sem_layers > 2 && sw > Hp ? sw : Hp) : 0;
    const int SA = alloc(sa_rows);
    G.out_rows = (sem && M.sem_with_geo) ? 32 : 16;      // rows 0..2 rgb, 3 sigma, 4.. logits (<= 8), then geo_map_sem's (<= 8); the widest read is 8 rows from row 4
    G.out_off = alloc(G.out_rows);
    G.lds_floats = off;
    // one workgroup per CU only (its buffers take more than half of the 160 KiB): two waves per SIMD inside it instead
    G.n_waves = (2 * G.lds_floats * 4 > 160 * 1024 && Wp >= 64) ? 8 : 4;
    const int NWp = G.n_waves;
    G.n_out = M.use_viewdirs ? 4 + sem_dim : 4;
    int n = 0, w_off = 0;
    // the saved-activation row of a point (training): [X pad32 | V pad32 | one pad32(out_dim) block per dense op, in program order]
    G.x_col = 0; G.v_col = pad_to(x_dim, 32);
    int next_col = G.v_col + pad_to(v_dim, 32);
    struct { int off, col; } produced[kGenMaxOps + 2];
    int n_produced = 0;
    produced[n_produced++] = {G.x_off, G.x_col};
    if (v_dim) produced[n_produced++] = {G.v_off, G.v_col};
    auto col_of = [&](int buf_off) { for (int k = n_produced - 1; k >= 0; --k) if (produced[k].off == buf_off) return produced[k].col; return -1; };
    // ks_scratch >= 0: a buffer that is dead while the op runs (>= 24 rows) -- the op's K range is split over the waves if it qualifies
    auto dense = [&](const nsos_generic_linear& L, int out_buf, int out_row0, bool relu, int n_seg, const HostSeg* segs, int ks_scratch = -1) {
        if (n >= kGenMaxOps || !L.weight || !L.bias) { H.err = H.err ? H.err : (n >= kGenMaxOps ? NSOS_ERR_UNSUPPORTED : NSOS_ERR_NULL_POINTER); return; }
        HostOp& ho = H.hops[n];
        GenOp& op = ho.op;
        op.kind = kGenDense; op.out_off = out_buf + out_row0 * kGenRowFloats; op.out_dim = L.out_dim; op.out_tiles = pad_to(L.out_dim, 32) / 32;
        op.relu = (relu ? 1 : 0) | (out_buf != G.out_off ? 2 : 0);
        int g = 0, in_dim = 0;
        for (int s = 0; s < n_seg; ++s) {
            ho.seg[s] = segs[s];
            ho.seg[s].col0 = in_dim;
            ho.seg[s].src_col = col_of(segs[s].buf_off);
            in_dim += segs[s].rows;
            const int ng = pad_to(segs[s].rows, 8) / 8;
            for (int k = 0; k < ng; ++k) {
                if (g >= kGenMaxGroups) { H.err = NSOS_ERR_UNSUPPORTED; return; }
                op.grp_off[g++] = segs[s].buf_off + 8 * k * kGenRowFloats;
            }
        }
        if (in_dim != L.in_dim) { H.err = NSOS_ERR_BAD_SHAPE; return; }
        const int ks_rows = ks_scratch == G.x_off ? G.x_rows : Wp;          // rows of the scratch buffer (a trunk buffer, or the encoded-xyz one)
        const bool ksplit = ks_scratch > 0 && out_buf == G.out_off && L.out_dim <= 8 && g >= 8 && pad_to(g, 4 * NWp) <= kGenMaxGroups && ks_rows >= 8 * (NWp - 1);
        op.ksplit_off = ksplit ? ks_scratch : 0;
        while (g % (ksplit ? 4 * NWp : 4)) {                   // the kernel's loop is unrolled by four groups (per wave, for a K-split op): pad with zero-weight groups
            if (g >= kGenMaxGroups) { H.err = NSOS_ERR_UNSUPPORTED; return; }
            op.grp_off[g++] = G.ones_off;
        }
        for (int k = g; k < kGenMaxGroups + 8; ++k) op.grp_off[k] = G.ones_off;
        op.n_groups = g; op.w_off = w_off;
        op.b_off = w_off + op.out_tiles * g * 256;        // the bias table sits behind the op's A stream
        w_off += op.out_tiles * g * 256 + op.out_tiles * 32;
        ho.w = L.weight; ho.bias = L.bias; ho.in_dim = in_dim; ho.n_seg = n_seg;
        ho.lin_id = (int)(&L - &M.pts[0]);        // position in the description: pts 0-15, alpha, feature, views, rgb, output, sem 21-28, geo 29-30
        ho.out_buf = out_buf;
        op.act_col = next_col;
        next_col += op.out_tiles * 32;
        produced[n_produced++] = {op.out_off, op.act_col};
        ++n;
    };
    const HostSeg Xs = {G.x_off, x_dim, 0, 0}, Vs = {G.v_off, v_dim, 0, 0};
    int cur = -1;                                                      // the trunk's current activation buffer
    for (int i = 0; i < D; ++i) {
        const int out = (i & 1) ? HB : HA;
        if (i == 0) dense(M.pts[0], out, 0, true, 1, &Xs);
        else {
            const HostSeg hs = {cur, W, 0, 0};
            if ((M.skip_mask >> (i - 1)) & 1) { const HostSeg two[2] = {Xs, hs}; dense(M.pts[i], out, 0, true, 2, two); }   // cat([input_pts, h]) (:73-74)
            else dense(M.pts[i], out, 0, true, 1, &hs);
        }
        cur = out;
    }
    const int other = cur == HA ? HB : HA;
    const HostSeg hs = {cur, W, 0, 0};
    if (!M.use_viewdirs) {
        dense(M.output, G.out_off, 0, false, 1, &hs, other);                                   // output_linear (:97-98); `other` is dead (it held h[D-2])
    } else {
        dense(M.alpha, G.out_off, 3, false, 1, &hs, other);                                    // alpha = alpha_linear(h) (:77); `other` is dead until the heads write it
        if (sem) {
            int src = -1, src_rows = 0;
            for (int k = 0; k < M.sem_layers; ++k) {                                           // semantic_linear (:58-64, :79-80)
                const bool last = k == M.sem_layers - 1;
                const int out = last ? G.out_off : ((k & 1) ? SA : other);
                // (the chain's last Linear -- one tile of logits -- is split over K like alpha and rgb: the encoded-xyz buffer is dead by then,
                //  every reader of it (layer 0, the skip layers, semantic_linear.0 with sem_with_coord) lies in front; not when it IS that reader)
                if (k == 0) {
                    if (M.sem_with_coord) { const HostSeg two[2] = {hs, Xs}; dense(M.sem[0], out, last ? 4 : 0, !last, 2, two); }   // cat([h, input_pts])
                    else dense(M.sem[0], out, last ? 4 : 0, !last, 1, &hs, last ? G.x_off : -1);
                } else {
                    const HostSeg ss = {src, src_rows, 0, 0};
                    dense(M.sem[k], out, last ? 4 : 0, !last, 1, &ss, last ? G.x_off : -1);
                }
                src = out; src_rows = M.sem[k].out_dim;
            }
            if (M.sem_with_geo) {                                                              // semantics *= geo_map_sem(alpha) (:60, :81-83)
                const int gbuf = SA;                                                           // (the chain is done with it)
                const HostSeg as = {G.out_off + 3 * kGenRowFloats, 1, 0, 0};
                dense(M.geo[0], gbuf, 0, true, 1, &as);
                const HostSeg gs = {gbuf, M.geo[0].out_dim, 0, 0};
                dense(M.geo[1], G.out_off, 4 + sem_dim, false, 1, &gs);
                if (n < kGenMaxOps) {
                    GenOp& op = H.hops[n].op;
                    op = GenOp{};
                    op.kind = kGenMul; op.out_off = G.out_off + 4 * kGenRowFloats; op.src_off = G.out_off + (4 + sem_dim) * kGenRowFloats; op.out_dim = sem_dim;
                    ++n;
                }
            }
        }
        dense(M.feature, other, 0, false, 1, &hs);                                             // feature = feature_linear(h) (:86)
        const HostSeg fv[2] = {{other, W, 0, 0}, Vs};
        dense(M.views, cur, 0, true, 2, fv);                                                   // relu(views_linears.0(cat([feature, input_views]))) (:87-90)
        const HostSeg vh = {cur, M.views.out_dim, 0, 0};
        dense(M.rgb, G.out_off, 0, false, 1, &vh, other);                                      // rgb_linear (:92); `other` held feature: consumed by the view branch
    }
    // the ReLU patterns as bits, behind the blocks: one 32-bit word per output tile of every ReLU op (two 16-bit halves: hi = 0, 1)
    for (int i = 0; i < n; ++i) {
        GenOp& op = H.hops[i].op;
        op.mask_col = -1;
        if (op.kind == kGenDense && (op.relu & 1)) { op.mask_col = next_col; next_col += op.out_tiles; }
    }
    next_col = pad_to(next_col, 4);                         // rows stay 16-byte aligned
    G.n_ops = n;
    G.w_floats = w_off;
    G.act_ld = next_col;
    G.row_floats = kGenRowFloats;
    for (int i = 0; i < n; ++i) G.ops[i] = H.hops[i].op;
    if (!H.err && G.lds_floats * 4 > 160 * 1024) H.err = NSOS_ERR_BUFFER_TOO_SMALL;      // (build_program: try narrower point tiles)
}

// 32-point tiles while the activation buffers fit the LDS, 16-point tiles for wider nets (see kGenRowFloats)
void build_program(const nsos_generic_mlp& M, HostProgram& H) {
    build_program_rf(M, H, 32);
    if (H.err == NSOS_ERR_BUFFER_TOO_SMALL) build_program_rf(M, H, 16);
    if (H.err == NSOS_ERR_BUFFER_TOO_SMALL) H.err = NSOS_ERR_UNSUPPORTED;
}


int producer_op(const HostProgram& H, int src_col) {
    for (int k = 0; k < H.prog.n_ops; ++k) if (H.hops[k].op.kind == kGenDense && H.hops[k].op.act_col == src_col) return k;
    return -1;
}
// need[o]: is the pre-activation gradient of forward op o formed at all (see build_bwd_program)
void compute_need(const nsos_generic_mlp& M, const HostProgram& H, bool input_grads, unsigned trainable, bool* need, int& sem_last_op, int& geo1_op) {
    const GenProgram& F = H.prog;
    sem_last_op = geo1_op = -1;
    for (int oi = 0; oi < F.n_ops; ++oi) {
        const HostOp& ho = H.hops[oi];
        need[oi] = input_grads;
        if (ho.op.kind != kGenDense) continue;
        if (ho.lin_id == 21 + M.sem_layers - 1) sem_last_op = oi;
        if (ho.lin_id == 30) geo1_op = oi;
        if ((trainable >> ho.lin_id) & 1u) need[oi] = true;
        for (int s = 0; s < ho.n_seg && !need[oi]; ++s) {
            const bool enc = ho.seg[s].buf_off == F.x_off || (F.v_dim && ho.seg[s].buf_off == F.v_off);
            const int pr = enc ? -1 : producer_op(H, ho.seg[s].src_col);
            if (pr >= 0 && pr < oi && need[pr]) need[oi] = true;
        }
    }
}
// Which blocks of the saved-activation row a backward for this trainable subset reads: the inputs of a trainable Linear (the X operand of
// its weight-gradient reduction), the outputs of ReLU ops whose gradient is formed (their masks), the two factors of semantics * mapping.
// (round 5: the ReLU patterns travel as bit words that every SAVE launch writes, so a block is stored only for the reductions and the
//  product's factors.)  Bit 31 of `trainable` (kGenInputGrads): the chain also reaches the inputs -- every op's gradient is formed.
constexpr unsigned kGenInputGrads = 1u << 31;
unsigned long long save_mask_for(const nsos_generic_mlp& M, const HostProgram& H, unsigned trainable) {
    if (trainable == ~0u) return ~0ull;
    const GenProgram& F = H.prog;
    bool need[kGenMaxOps];
    int sem_last_op, geo1_op;
    compute_need(M, H, (trainable & kGenInputGrads) != 0, trainable & ~kGenInputGrads, need, sem_last_op, geo1_op);
    trainable &= ~kGenInputGrads;
    unsigned long long m = 0;
    for (int oi = 0; oi < F.n_ops; ++oi) {
        const HostOp& ho = H.hops[oi];
        if (ho.op.kind != kGenDense) continue;
        if (!((trainable >> ho.lin_id) & 1u)) continue;
        for (int s = 0; s < ho.n_seg; ++s) {
            if (ho.seg[s].buf_off == F.x_off) m |= 1ull << 62;
            else if (F.v_dim && ho.seg[s].buf_off == F.v_off) m |= 1ull << 63;
            else { const int pr = producer_op(H, ho.seg[s].src_col); if (pr >= 0) m |= 1ull << pr; }
        }
    }
    if ((sem_last_op >= 0 && need[sem_last_op]) || (geo1_op >= 0 && need[geo1_op])) {
        if (sem_last_op >= 0) m |= 1ull << sem_last_op;
        if (geo1_op >= 0) m |= 1ull << geo1_op;
    }
    return m;
}

// The backward program of a forward program (see mlp_generic_bwd_kernel).  `T` gets the transposed streams' pack descriptors.
struct HostBwd { GenProgram prog; GenPackT packs[kGenMaxOps]; int pack_w_off[kGenMaxOps]; int n_packs; int32_t err; };
void build_bwd_program(const nsos_generic_mlp& M, const HostProgram& H, HostBwd& B, bool input_grads, unsigned trainable = ~0u) {
    B = HostBwd{};
    B.err = H.err;
    if (H.err) return;
    const GenProgram& F = H.prog;
    const int kGenRowFloats = F.row_floats;
    GenProgram& G = B.prog;
    G = F;
    G.n_ops = 0;
    int n = 0, w_off = 0;
    // which gradient buffers hold a contribution already (a second consumer adds).  OUT is preloaded with d loss / d raw.
    struct { int off; bool has; } state[16];
    int n_state = 0;
    auto has_grad = [&](int off) -> bool& {
        for (int k = 0; k < n_state; ++k) if (state[k].off == off) return state[k].has;
        state[n_state] = {off, false};
        return state[n_state++].has;
    };
    auto in_out = [&](int off) { return off >= F.out_off && off < F.out_off + F.out_rows * kGenRowFloats; };
    int sem_last_col = -1, geo_col = -1;
    for (int i = 0; i < F.n_ops; ++i) {
        if (H.hops[i].op.kind != kGenDense) continue;
        if (H.hops[i].lin_id == 21 + M.sem_layers - 1) sem_last_col = H.hops[i].op.act_col;
        if (H.hops[i].lin_id == 30) geo_col = H.hops[i].op.act_col;
    }
    // Which pre-activation gradients are needed at all: those of a trainable Linear (bit lin_id of `trainable`) and of everything
    // DOWNSTREAM of one (a gradient reaches a Linear through all of its consumers).  need[o] = trainable[o] or need[a producer of o's
    // inputs], in forward order.  With the backbone frozen (the shipped recipe) the chain stops at the semantic head: no trunk, view
    // branch or sigma steps.  Gradients to the inputs need everything.
    bool need[kGenMaxOps];
    auto producer_of = [&](int src_col) { return producer_op(H, src_col); };
    int sem_last_op = -1, geo1_op = -1;
    compute_need(M, H, input_grads, trainable, need, sem_last_op, geo1_op);
    for (int oi = F.n_ops - 1; oi >= 0; --oi) {
        const HostOp& ho = H.hops[oi];
        if (n + 1 + kGenMaxSeg > kGenMaxOps) { B.err = NSOS_ERR_UNSUPPORTED; return; }
        if (ho.op.kind == kGenMul) {
            if (!((sem_last_op >= 0 && need[sem_last_op]) || (geo1_op >= 0 && need[geo1_op]))) continue;
            GenOp& op = G.ops[n++];
            op = GenOp{};
            op.kind = kGenBwdMul; op.out_off = ho.op.out_off; op.src_off = ho.op.src_off; op.out_dim = ho.op.out_dim;
            op.act_col = sem_last_col; op.aux_col = geo_col;
            continue;
        }
        if (need[oi]) {
            GenOp& op = G.ops[n++];
            op = GenOp{};
            op.kind = kGenBwdHead; op.out_off = ho.op.out_off; op.out_dim = ho.op.out_dim; op.out_tiles = ho.op.out_tiles;
            op.relu = (ho.op.relu & 1) | (((trainable >> ho.lin_id) & 1u) ? 0 : 8);      // bit 3: a frozen Linear -- its gradient feeds no reduction, nothing goes to gbuf
            op.act_col = ho.op.act_col; op.mask_col = ho.op.mask_col;
        }
        for (int s = 0; s < ho.n_seg && need[oi]; ++s) {
            const HostSeg& sg = ho.seg[s];
            const bool enc = sg.buf_off == F.x_off || (F.v_dim && sg.buf_off == F.v_off);
            if (enc && !input_grads) continue;     // rays are data: no gradient flows to the encodings unless the caller asks for it
            if (!enc) { const int pr = producer_of(sg.src_col); if (pr < 0 || !need[pr]) continue; }     // nobody upstream wants it
            GenOp& op = G.ops[n++];
            op = GenOp{};
            op.kind = kGenDense; op.out_off = sg.buf_off; op.out_dim = sg.rows; op.out_tiles = pad_to(sg.rows, 32) / 32;
            const bool to_out = in_out(sg.buf_off);
            bool& has = has_grad(to_out ? F.out_off : sg.buf_off);
            // (the encodings' buffers hold pad8 rows, OUT is shared: write the real rows only; every other target is a whole pad32 buffer)
            op.relu = ((to_out || enc) ? 0 : 2) | ((has || to_out) ? 4 : 0);
            has = true;
            int g = 0;
            const int kg = pad_to(ho.op.out_dim, 8) / 8;
            if (kg > kGenMaxGroups - 3) { B.err = NSOS_ERR_UNSUPPORTED; return; }
            for (int k = 0; k < kg; ++k) op.grp_off[g++] = ho.op.out_off + 8 * k * kGenRowFloats;
            while (g % 4) op.grp_off[g++] = F.ones_off;
            for (int k = g; k < kGenMaxGroups + 8; ++k) op.grp_off[k] = F.ones_off;
            op.n_groups = g; op.w_off = w_off;
            GenPackT& Q = B.packs[B.n_packs++];
            Q = GenPackT{};
            Q.w = ho.w; Q.in_dim = ho.in_dim; Q.col0 = sg.col0; Q.rows = sg.rows; Q.k_dim = ho.op.out_dim; Q.out_tiles = op.out_tiles; Q.n_groups = g;
            B.pack_w_off[B.n_packs - 1] = w_off;
            w_off += op.out_tiles * g * 256;
        }
        if (!in_out(ho.op.out_off)) has_grad(ho.out_buf) = false;     // consumed: the buffer's next tenant starts afresh
    }
    // Fuse a head step into the transposed product right in front of it when that product is the last contribution to the head's buffer
    // and writes the whole of it (every hidden layer of a trunk or a head chain: its one consumer's product; the trunk's last layer: the
    // alpha head's): the mask and the copy to gbuf then happen in that product's epilogue (mlp_generic_bwd_kernel), one barrier and one
    // pass over the buffer less per layer.  aux_col of a dense op (unused there otherwise) carries the head's flags + 16.
    int m = 0;
    for (int i = 0; i < n; ++i) {
        GenOp& op = G.ops[i];
        if (i + 1 < n && op.kind == kGenDense && (op.relu & 2) && G.ops[i + 1].kind == kGenBwdHead && G.ops[i + 1].out_off == op.out_off &&
            G.ops[i + 1].out_tiles == op.out_tiles && G.ops[i + 1].out_dim == op.out_dim) {
            const GenOp& h = G.ops[i + 1];
            op.aux_col = 16 | (h.relu & 1) | (h.relu & 8);
            op.mask_col = h.mask_col; op.act_col = h.act_col;
            G.ops[m++] = op;
            ++i;                     // the head step is gone
            continue;
        }
        if (op.kind == kGenDense) op.aux_col = 0;
        G.ops[m++] = op;
    }
    n = m;
    G.n_ops = n;
    G.w_floats = w_off;
    G.input_grads = input_grads ? 1 : 0;
}

constexpr size_t kGenHeaderBytes = (sizeof(GenProgram) + 255) / 256 * 256;
constexpr size_t kGenTailBytes = 8192;     // the prefetch ring reads up to four groups (4 KiB) past the last tile's stream

}  // namespace

// ------------------------------------------------------------------------------------------ C ABI
extern "C" size_t nsos_mlp_generic_packed_bytes(const nsos_generic_mlp* mlp) {
    if (!mlp) return 0;
    static thread_local HostProgram H;
    build_program(*mlp, H);
    if (H.err) return 0;
    return kGenHeaderBytes + (size_t)H.prog.w_floats * 4 + kGenTailBytes;
}

extern "C" int32_t nsos_mlp_generic_out_channels(const nsos_generic_mlp* mlp) {
    if (!mlp) return 0;
    static thread_local HostProgram H;
    build_program(*mlp, H);
    return H.err ? 0 : H.prog.n_out;
}

static int32_t generic_pack(const nsos_generic_mlp* mlp, void* packed, size_t packed_bytes, void* stream, bool with_header) {
    NSOS_REQUIRE(mlp && packed, NSOS_ERR_NULL_POINTER);
    NSOS_REQUIRE(((uintptr_t)packed & 15) == 0, NSOS_ERR_MISALIGNED);
    static thread_local HostProgram H;
    build_program(*mlp, H);
    if (H.err) return H.err;
    NSOS_REQUIRE(packed_bytes >= kGenHeaderBytes + (size_t)H.prog.w_floats * 4 + kGenTailBytes, NSOS_ERR_BUFFER_TOO_SMALL);
    const hipStream_t st = (hipStream_t)stream;
    // the program travels with the weights (pageable host memory: the copy is staged by the runtime before the call returns).  It depends
    // on the architecture alone: a re-pack after a weight update (nsos_mlp_generic_repack) leaves it where it is -- kernels only, capturable
    if (with_header) {
        hipError_t e = hipMemcpyAsync(packed, &H.prog, sizeof(GenProgram), hipMemcpyHostToDevice, st);
        if (e != hipSuccess) return (int32_t)e;
    }
    float* wts = reinterpret_cast<float*>(static_cast<unsigned char*>(packed) + kGenHeaderBytes);
    for (int i = 0; i < H.prog.n_ops; ++i) {
        const HostOp& ho = H.hops[i];
        if (ho.op.kind != kGenDense) continue;
        GenPackOp Q = {};
        Q.w = ho.w; Q.bias = ho.bias; Q.in_dim = ho.in_dim; Q.out_dim = ho.op.out_dim; Q.out_tiles = ho.op.out_tiles;
        Q.n_groups = ho.op.n_groups; Q.n_seg = ho.n_seg;
        for (int s = 0; s < ho.n_seg; ++s) { Q.seg_col0[s] = ho.seg[s].col0; Q.seg_rows[s] = ho.seg[s].rows; Q.seg_groups[s] = pad_to(ho.seg[s].rows, 8) / 8; }
        Q.out = wts + ho.op.w_off;
        const long long total = (long long)Q.out_tiles * Q.n_groups * 256 + (long long)Q.out_tiles * 32;      // stream + bias table
        hipLaunchKernelGGL(gen_pack_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, st, Q);
    }
    return nsos_launch_status();
}

extern "C" int32_t nsos_mlp_generic_pack(const nsos_generic_mlp* mlp, void* packed, size_t packed_bytes, void* stream) {
    return generic_pack(mlp, packed, packed_bytes, stream, true);
}
extern "C" int32_t nsos_mlp_generic_repack(const nsos_generic_mlp* mlp, void* packed, size_t packed_bytes, void* stream) {
    return generic_pack(mlp, packed, packed_bytes, stream, false);
}

static int32_t generic_launch(const nsos_generic_mlp* mlp, const void* packed, GenParams p, int64_t n_pts, hipStream_t st) {
    static thread_local HostProgram H;
    build_program(*mlp, H);
    if (H.err) return H.err;
    const int rf = H.prog.row_floats;
    NSOS_REQUIRE((n_pts + rf - 1) / rf < (1ll << 31), NSOS_ERR_UNSUPPORTED);
    p.prog = static_cast<const GenProgram*>(packed);
    p.wts = reinterpret_cast<const float*>(static_cast<const unsigned char*>(packed) + kGenHeaderBytes);
    p.n_pts = n_pts;
    p.n_tiles = (int)((n_pts + rf - 1) / rf);
    const int lds_bytes = H.prog.lds_floats * 4;
    // (per call: the attribute is a property of the kernel on this device, the size a property of the architecture rendered)
    const int nw = H.prog.n_waves;
    const bool sv = p.acts != nullptr;
#define NSOS_GEN_FN(S, R, W) reinterpret_cast<const void*>(&mlp_generic_kernel<S, R, W>)
    const void* fn = rf == 32 ? (nw == 8 ? (sv ? NSOS_GEN_FN(true, 32, 8) : NSOS_GEN_FN(false, 32, 8)) : (sv ? NSOS_GEN_FN(true, 32, 4) : NSOS_GEN_FN(false, 32, 4)))
                              : (nw == 8 ? (sv ? NSOS_GEN_FN(true, 16, 8) : NSOS_GEN_FN(false, 16, 8)) : (sv ? NSOS_GEN_FN(true, 16, 4) : NSOS_GEN_FN(false, 16, 4)));
#undef NSOS_GEN_FN
    hipError_t e = hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    if (e != hipSuccess) return (int32_t)e;
    const int per_cu = lds_bytes > 0 ? (160 * 1024) / lds_bytes : 1;
    const int wgs = nsos_device_cus() * (per_cu < 1 ? 1 : (per_cu > 4 ? 4 : per_cu));
    const int grid = p.n_tiles < wgs ? p.n_tiles : wgs;
#define NSOS_GEN_GO(S, R, W) hipLaunchKernelGGL((mlp_generic_kernel<S, R, W>), dim3(grid), dim3(64 * W), lds_bytes, st, p)
    if (rf == 32) {
        if (nw == 8) { if (sv) NSOS_GEN_GO(true, 32, 8); else NSOS_GEN_GO(false, 32, 8); }
        else { if (sv) NSOS_GEN_GO(true, 32, 4); else NSOS_GEN_GO(false, 32, 4); }
    } else {
        if (nw == 8) { if (sv) NSOS_GEN_GO(true, 16, 8); else NSOS_GEN_GO(false, 16, 8); }
        else { if (sv) NSOS_GEN_GO(true, 16, 4); else NSOS_GEN_GO(false, 16, 4); }
    }
#undef NSOS_GEN_GO
    return nsos_launch_status();
}

extern "C" int32_t nsos_mlp_generic_forward_rays(const nsos_generic_mlp* mlp, const void* packed, const float* rays_o, const float* rays_d,
                                                 const float* viewdirs, const float* z_vals, int64_t n_rays, int32_t n_samples,
                                                 float* raw, void* stream) {
    if (n_rays == 0) return NSOS_OK;
    NSOS_REQUIRE(mlp && packed && rays_o && rays_d && z_vals && raw, NSOS_ERR_NULL_POINTER);
    NSOS_REQUIRE(!mlp->use_viewdirs || viewdirs, NSOS_ERR_NULL_POINTER);
    NSOS_REQUIRE(n_rays > 0 && n_samples >= 1, NSOS_ERR_BAD_SHAPE);
    GenParams p = {};
    p.rays_o = rays_o; p.rays_d = rays_d; p.viewdirs = viewdirs; p.z_vals = z_vals; p.raw = raw; p.n_samples = n_samples;
    return generic_launch(mlp, packed, p, n_rays * (int64_t)n_samples, (hipStream_t)stream);
}

extern "C" int32_t nsos_mlp_generic_forward_points(const nsos_generic_mlp* mlp, const void* packed, const float* pts, const float* dirs,
                                                   int64_t n_pts, float* raw, void* stream) {
    if (n_pts == 0) return NSOS_OK;
    NSOS_REQUIRE(mlp && packed && pts && raw, NSOS_ERR_NULL_POINTER);
    NSOS_REQUIRE(!mlp->use_viewdirs || dirs, NSOS_ERR_NULL_POINTER);
    NSOS_REQUIRE(n_pts > 0, NSOS_ERR_BAD_SHAPE);
    GenParams p = {};
    p.pts = pts; p.dirs = dirs; p.raw = raw; p.n_samples = 1;
    return generic_launch(mlp, packed, p, n_pts, (hipStream_t)stream);
}

// ---- training (every parameter of a generic-architecture net): saved activations, the input-gradient chain, the layout table
extern "C" int32_t nsos_mlp_generic_save_layout(const nsos_generic_mlp* mlp, int32_t* table, int32_t capacity) {
    NSOS_REQUIRE(mlp && table, NSOS_ERR_NULL_POINTER);
    static thread_local HostProgram H;
    build_program(*mlp, H);
    if (H.err) return H.err;
    int n_dense = 0;
    for (int i = 0; i < H.prog.n_ops; ++i) n_dense += H.hops[i].op.kind == kGenDense;
    const int need = 2 + n_dense * NSOS_GENERIC_LAYOUT_STRIDE;
    NSOS_REQUIRE(capacity >= need, NSOS_ERR_BUFFER_TOO_SMALL);
    int k = 0;
    table[k++] = H.prog.act_ld;
    table[k++] = n_dense;
    for (int i = 0; i < H.prog.n_ops; ++i) {
        const HostOp& ho = H.hops[i];
        if (ho.op.kind != kGenDense) continue;
        table[k++] = ho.lin_id; table[k++] = ho.op.act_col; table[k++] = ho.op.out_dim; table[k++] = ho.n_seg;
        for (int s = 0; s < kGenMaxSeg; ++s) {
            table[k++] = s < ho.n_seg ? ho.seg[s].src_col : -1;
            table[k++] = s < ho.n_seg ? ho.seg[s].rows : 0;
            table[k++] = s < ho.n_seg ? ho.seg[s].col0 : 0;
        }
    }
    return need;
}

extern "C" int32_t nsos_mlp_generic_forward_rays_save(const nsos_generic_mlp* mlp, const void* packed, const float* rays_o, const float* rays_d,
                                                      const float* viewdirs, const float* z_vals, int64_t n_rays, int32_t n_samples,
                                                      float* raw, float* acts, void* stream) {
    if (n_rays == 0) return NSOS_OK;
    NSOS_REQUIRE(mlp && packed && rays_o && rays_d && z_vals && raw && acts, NSOS_ERR_NULL_POINTER);
    NSOS_REQUIRE(!mlp->use_viewdirs || viewdirs, NSOS_ERR_NULL_POINTER);
    NSOS_REQUIRE(n_rays > 0 && n_samples >= 1, NSOS_ERR_BAD_SHAPE);
    NSOS_REQUIRE(((uintptr_t)acts & 15) == 0, NSOS_ERR_MISALIGNED);
    GenParams p = {};
    p.rays_o = rays_o; p.rays_d = rays_d; p.viewdirs = viewdirs; p.z_vals = z_vals; p.raw = raw; p.acts = acts; p.n_samples = n_samples;
    p.save_mask = ~0ull;
    return generic_launch(mlp, packed, p, n_rays * (int64_t)n_samples, (hipStream_t)stream);
}

extern "C" int32_t nsos_mlp_generic_forward_rays_save_subset(const nsos_generic_mlp* mlp, const void* packed, const float* rays_o, const float* rays_d,
                                                             const float* viewdirs, const float* z_vals, int64_t n_rays, int32_t n_samples,
                                                             float* raw, float* acts, uint32_t trainable, void* stream) {
    if (n_rays == 0) return NSOS_OK;
    NSOS_REQUIRE(mlp && packed && rays_o && rays_d && z_vals && raw && acts, NSOS_ERR_NULL_POINTER);
    NSOS_REQUIRE(!mlp->use_viewdirs || viewdirs, NSOS_ERR_NULL_POINTER);
    NSOS_REQUIRE(n_rays > 0 && n_samples >= 1, NSOS_ERR_BAD_SHAPE);
    NSOS_REQUIRE(((uintptr_t)acts & 15) == 0, NSOS_ERR_MISALIGNED);
    static thread_local HostProgram H;
    build_program(*mlp, H);
    if (H.err) return H.err;
    GenParams p = {};
    p.rays_o = rays_o; p.rays_d = rays_d; p.viewdirs = viewdirs; p.z_vals = z_vals; p.raw = raw; p.acts = acts; p.n_samples = n_samples;
    p.save_mask = save_mask_for(*mlp, H, trainable);
    return generic_launch(mlp, packed, p, n_rays * (int64_t)n_samples, (hipStream_t)stream);
}

extern "C" size_t nsos_mlp_generic_bwd_packed_bytes(const nsos_generic_mlp* mlp, int32_t input_grads) {   // (the full program: an upper bound for every trainable subset)
    if (!mlp) return 0;
    static thread_local HostProgram H;
    static thread_local HostBwd B;
    build_program(*mlp, H);
    build_bwd_program(*mlp, H, B, input_grads != 0);
    if (B.err) return 0;
    return kGenHeaderBytes + (size_t)B.prog.w_floats * 4 + kGenTailBytes;
}

static int32_t generic_pack_bwd(const nsos_generic_mlp* mlp, void* packed_bwd, size_t packed_bytes, int32_t input_grads, void* stream, bool with_header,
                                unsigned trainable = ~0u) {
    NSOS_REQUIRE(mlp && packed_bwd, NSOS_ERR_NULL_POINTER);
    NSOS_REQUIRE(((uintptr_t)packed_bwd & 15) == 0, NSOS_ERR_MISALIGNED);
    static thread_local HostProgram H;
    static thread_local HostBwd B;
    build_program(*mlp, H);
    build_bwd_program(*mlp, H, B, input_grads != 0, trainable);
    if (B.err) return B.err;
    NSOS_REQUIRE(packed_bytes >= kGenHeaderBytes + (size_t)B.prog.w_floats * 4 + kGenTailBytes, NSOS_ERR_BUFFER_TOO_SMALL);
    const hipStream_t st = (hipStream_t)stream;
    if (with_header) {
        hipError_t e = hipMemcpyAsync(packed_bwd, &B.prog, sizeof(GenProgram), hipMemcpyHostToDevice, st);
        if (e != hipSuccess) return (int32_t)e;
    }
    float* wts = reinterpret_cast<float*>(static_cast<unsigned char*>(packed_bwd) + kGenHeaderBytes);
    for (int i = 0; i < B.n_packs; ++i) {
        GenPackT Q = B.packs[i];
        NSOS_REQUIRE(Q.w, NSOS_ERR_NULL_POINTER);
        Q.out = wts + B.pack_w_off[i];
        const long long total = (long long)Q.out_tiles * Q.n_groups * 256;
        hipLaunchKernelGGL(gen_pack_t_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, st, Q);
    }
    return nsos_launch_status();
}

extern "C" int32_t nsos_mlp_generic_pack_bwd(const nsos_generic_mlp* mlp, void* packed_bwd, size_t packed_bytes, int32_t input_grads, void* stream) {
    return generic_pack_bwd(mlp, packed_bwd, packed_bytes, input_grads, stream, true);
}
extern "C" int32_t nsos_mlp_generic_repack_bwd(const nsos_generic_mlp* mlp, void* packed_bwd, size_t packed_bytes, int32_t input_grads, void* stream) {
    return generic_pack_bwd(mlp, packed_bwd, packed_bytes, input_grads, stream, false);
}
extern "C" int32_t nsos_mlp_generic_pack_bwd_subset(const nsos_generic_mlp* mlp, void* packed_bwd, size_t packed_bytes, uint32_t trainable,
                                                    int32_t with_header, void* stream) {
    // bit 31 of `trainable`: the program also reaches the encodings (ray / point gradients) -- e.g. pose refinement against a frozen net:
    // trainable = 1u << 31 forms every gradient of the chain and stores none of them
    return generic_pack_bwd(mlp, packed_bwd, packed_bytes, (trainable & kGenInputGrads) ? 1 : 0, stream, with_header != 0, trainable & ~kGenInputGrads);
}

static int32_t generic_bwd_launch(const nsos_generic_mlp* mlp, const void* packed_bwd, GenBwdParams p, int64_t n_pts, hipStream_t st) {
    static thread_local HostProgram H;
    build_program(*mlp, H);
    if (H.err) return H.err;
    p.prog = static_cast<const GenProgram*>(packed_bwd);
    p.wts = reinterpret_cast<const float*>(static_cast<const unsigned char*>(packed_bwd) + kGenHeaderBytes);
    const int rf = H.prog.row_floats;
    p.n_pts = n_pts; p.n_tiles = (int)((n_pts + rf - 1) / rf);
    const int lds_bytes = H.prog.lds_floats * 4;
    const int nw = H.prog.n_waves;
#define NSOS_GENB_FN(R, W) reinterpret_cast<const void*>(&mlp_generic_bwd_kernel<R, W>)
    hipError_t e = hipFuncSetAttribute(rf == 32 ? (nw == 8 ? NSOS_GENB_FN(32, 8) : NSOS_GENB_FN(32, 4)) : (nw == 8 ? NSOS_GENB_FN(16, 8) : NSOS_GENB_FN(16, 4)),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
#undef NSOS_GENB_FN
    if (e != hipSuccess) return (int32_t)e;
    const int per_cu = lds_bytes > 0 ? (160 * 1024) / lds_bytes : 1;
    const int wgs = nsos_device_cus() * (per_cu < 1 ? 1 : (per_cu > 4 ? 4 : per_cu));
    const int grid = p.n_tiles < wgs ? p.n_tiles : wgs;
#define NSOS_GENB_GO(R, W) hipLaunchKernelGGL((mlp_generic_bwd_kernel<R, W>), dim3(grid), dim3(64 * W), lds_bytes, st, p)
    if (rf == 32) { if (nw == 8) NSOS_GENB_GO(32, 8); else NSOS_GENB_GO(32, 4); }
    else { if (nw == 8) NSOS_GENB_GO(16, 8); else NSOS_GENB_GO(16, 4); }
#undef NSOS_GENB_GO
    return nsos_launch_status();
}

extern "C" int32_t nsos_mlp_generic_input_grads(const nsos_generic_mlp* mlp, const void* packed_bwd, const float* g_raw, const float* acts,
                                                float* gbuf, int64_t n_pts, void* stream) {
    if (n_pts == 0) return NSOS_OK;
    NSOS_REQUIRE(mlp && packed_bwd && g_raw && acts && gbuf, NSOS_ERR_NULL_POINTER);
    NSOS_REQUIRE(n_pts > 0 && (n_pts + 15) / 16 < (1ll << 31), NSOS_ERR_BAD_SHAPE);
    NSOS_REQUIRE((((uintptr_t)acts | (uintptr_t)gbuf) & 15) == 0, NSOS_ERR_MISALIGNED);
    GenBwdParams p = {};
    p.g_raw = g_raw; p.acts = acts; p.gbuf = gbuf; p.n_samples = 1;
    return generic_bwd_launch(mlp, packed_bwd, p, n_pts, (hipStream_t)stream);
}

extern "C" int32_t nsos_mlp_generic_input_grads_rays(const nsos_generic_mlp* mlp, const void* packed_bwd, const float* g_raw, const float* acts,
                                                     float* gbuf, const float* rays_o, const float* rays_d, const float* viewdirs,
                                                     const float* z_vals, int64_t n_rays, int32_t n_samples, float* g_pts, float* g_dirs,
                                                     void* stream) {
    if (n_rays == 0) return NSOS_OK;
    NSOS_REQUIRE(mlp && packed_bwd && g_raw && acts && gbuf && rays_o && rays_d && z_vals && g_pts, NSOS_ERR_NULL_POINTER);
    NSOS_REQUIRE(!mlp->use_viewdirs || (viewdirs && g_dirs), NSOS_ERR_NULL_POINTER);
    NSOS_REQUIRE(n_rays > 0 && n_samples >= 1 && (n_rays * (int64_t)n_samples + 15) / 16 < (1ll << 31), NSOS_ERR_BAD_SHAPE);
    NSOS_REQUIRE((((uintptr_t)acts | (uintptr_t)gbuf) & 15) == 0, NSOS_ERR_MISALIGNED);
    GenBwdParams p = {};
    p.g_raw = g_raw; p.acts = acts; p.gbuf = gbuf; p.rays_o = rays_o; p.rays_d = rays_d; p.viewdirs = viewdirs; p.z_vals = z_vals;
    p.g_pts = g_pts; p.g_dirs = mlp->use_viewdirs ? g_dirs : nullptr; p.n_samples = n_samples;
    return generic_bwd_launch(mlp, packed_bwd, p, n_rays * (int64_t)n_samples, (hipStream_t)stream);
}

extern "C" int32_t nsos_ray_grad_reduce(const float* g_pts, const float* g_dirs, const float* z_vals, const float* rays_d, const float* raw,
                                        const float* g_raw, const float* noise, float noise_std, int64_t n_rays, int32_t n_samples,
                                        int32_t n_ch, float* g_rays_o, float* g_rays_d, void* stream) {
    if (n_rays == 0) return NSOS_OK;
    NSOS_REQUIRE(g_pts && z_vals && rays_d && raw && g_raw && g_rays_o && g_rays_d, NSOS_ERR_NULL_POINTER);
    NSOS_REQUIRE(n_rays > 0 && n_samples >= 1 && n_ch >= 4 && (n_rays + 3) / 4 < (1ll << 31), NSOS_ERR_BAD_SHAPE);
    hipLaunchKernelGGL(ray_grad_reduce_kernel, dim3((unsigned)((n_rays + 3) / 4)), dim3(256), 0, (hipStream_t)stream, g_pts, g_dirs, z_vals,
                       rays_d, raw, g_raw, noise, noise_std, (long long)n_rays, (int)n_samples, (int)n_ch, g_rays_o, g_rays_d);
    return nsos_launch_status();
}

// ---- point queries and pre-encoded inputs under autograd (NeRFMLP.forward, MLP.forward: models/nerf_mlp.py:179-215, 67-100)
extern "C" int32_t nsos_mlp_generic_forward_points_save(const nsos_generic_mlp* mlp, const void* packed, const float* pts, const float* dirs,
                                                        const float* encoded, int64_t n_pts, float* raw, float* acts, void* stream) {
    if (n_pts == 0) return NSOS_OK;
    NSOS_REQUIRE(mlp && packed && raw && (encoded || pts), NSOS_ERR_NULL_POINTER);
    NSOS_REQUIRE(encoded || !mlp->use_viewdirs || dirs, NSOS_ERR_NULL_POINTER);
    NSOS_REQUIRE(n_pts > 0, NSOS_ERR_BAD_SHAPE);
    NSOS_REQUIRE(((uintptr_t)acts & 15) == 0, NSOS_ERR_MISALIGNED);
    GenParams p = {};
    p.pts = encoded ? nullptr : pts; p.dirs = encoded ? nullptr : dirs; p.enc = encoded; p.raw = raw; p.acts = acts; p.n_samples = 1;
    p.save_mask = ~0ull;
    return generic_launch(mlp, packed, p, n_pts, (hipStream_t)stream);
}

extern "C" int32_t nsos_mlp_generic_input_grads_points(const nsos_generic_mlp* mlp, const void* packed_bwd, const float* g_raw, const float* acts,
                                                       float* gbuf, const float* pts, const float* dirs, int64_t n_pts, float* g_pts,
                                                       float* g_dirs, float* g_encoded, void* stream) {
    if (n_pts == 0) return NSOS_OK;
    NSOS_REQUIRE(mlp && packed_bwd && g_raw && acts && gbuf, NSOS_ERR_NULL_POINTER);
    NSOS_REQUIRE(g_encoded || (pts && g_pts && (!mlp->use_viewdirs || (dirs && g_dirs))), NSOS_ERR_NULL_POINTER);
    NSOS_REQUIRE(n_pts > 0 && (n_pts + 15) / 16 < (1ll << 31), NSOS_ERR_BAD_SHAPE);
    NSOS_REQUIRE((((uintptr_t)acts | (uintptr_t)gbuf) & 15) == 0, NSOS_ERR_MISALIGNED);
    GenBwdParams p = {};
    p.g_raw = g_raw; p.acts = acts; p.gbuf = gbuf; p.n_samples = 1;
    if (g_encoded) p.g_enc = g_encoded;
    else { p.pts = pts; p.dirs = dirs; p.g_pts = g_pts; p.g_dirs = mlp->use_viewdirs ? g_dirs : nullptr; }
    return generic_bwd_launch(mlp, packed_bwd, p, n_pts, (hipStream_t)stream);
}

#ifdef NSOS_GEN_PROF
extern "C" int32_t nsos_gen_prof_read(unsigned long long* out) {
    return (int32_t)hipMemcpyFromSymbol(out, HIP_SYMBOL(nsos_gen_prof), sizeof(unsigned long long) * 4 * 2 * 64);
}
#endif
