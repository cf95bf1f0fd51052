"""Backward of one NeRF MLP over saved activations (K7): the orchestration of two HIP kernels -- the fused
input-gradient chain (`nsos_mlp_input_grads_x3`) and the weight-gradient reductions (`nsos_wgrad[_x3]`).  Mirrors the autograd
graph of MLP.forward (models/nerf_mlp.py:67-100): heads -> view branch -> feature -> [semantic head] -> 8 trunk layers with
the skip connection into layer 5.  No library GEMM: round 1 ran the nine [P,256]x[256,256] input-gradient products of
`mlp_precision="fp32"` training through the BLAS library plus separate ReLU-mask passes (14.3 ms per 4096-ray step); both
precisions now use the fused chain.

acts [P, ACTS_DIM] is the buffer nsos_mlp_forward_rays_save_all wrote (column map in include/nerf_sos_hip.h);
g_raw [P, C] = d loss / d [r, g, b, sigma, (sem0, sem1)] from nsos_composite_backward.
"""
from __future__ import annotations

from typing import Dict

import torch

from . import ops
from .ops import ACTS_D, ACTS_FEAT, ACTS_SEM, ACTS_VIEWS, ACTS_X, SEM_COORD, SEM_NONE

W = 256


def mlp_backward(mlp, sem_mode: int, acts: torch.Tensor, g_raw: torch.Tensor, packed_bwd: torch.Tensor,
                 masks=None, split_wgrad: bool = True) -> Dict[str, torch.Tensor]:
    """Gradients of every parameter of one MLP, by name.  The input-gradient chain (nine products g_in = g_out W with the
    ReLU masks between them) is ONE fused kernel on the 16-bit matrix pipe with split-fp16 operands
    (nsos_mlp_input_grads_x3, csrc/mlp_x3_bwd.hip: fp32-grade, <= 1.1e-6 of each block's scale against fp32 GEMMs); it
    writes every layer's pre-activation gradient once, and the weight-gradient reductions dW = G^T X read them:
    split_wgrad=True  -> the 256x256 reductions on the 16-bit pipe with split operands (nsos_wgrad_x3; mlp_precision="fp16x3"),
    split_wgrad=False -> all reductions on the exact-fp32 MFMA (nsos_wgrad; mlp_precision="fp32": exact forward, exact
                         reductions, fp32-grade chain).
    `packed_bwd` = the net's packed_weights("fp16x3_bwd"); `masks` = the ReLU bit masks of the split-fp16 forward, or None
    (then the trunk masks are read from the saved fp32 activations)."""
    P_, C = g_raw.shape
    dev = acts.device
    f32 = dict(device=dev, dtype=torch.float32)
    col = lambda a, n: acts[:, a:a + n]  # noqa: E731
    h = lambda l: col(W * l, W)          # noqa: E731
    # fp16 has 5 exponent bits: bring max |g_raw| to ~2^4 with an exact power of two (no host sync), undo on the results.
    # 2^4 leaves a factor 2^12 of growth along the chain before fp16 overflows and 2^-18 of shrinkage before the hi parts
    # go subnormal (layer gains of trained NeRFs are O(1)).
    # Round 5: per BRANCH.  The colour channels (0..2), sigma (3) and the logits (4..) of g_raw come from different losses -- img2mse's
    # batch mean next to per-logit gradients of the correlation losses -- and sit decades apart; a branch scaled by the common factor
    # alone loses its lo parts to fp16's subnormals (2-5e-4 of scale in feature_linear's gradient on a trained field).  scale3 =
    # [trunk scale, colour factor, semantic factor] (csrc/mlp_x3_bwd.hip); inv_* undo them on the blocks that carry them.
    am = g_raw.abs().amax(0)                                                   # [C]
    grp = torch.stack([am.max(), am[:3].max(), am[4:].max() if C > 4 else am.max()]).clamp_min(1e-30)
    lg = torch.floor(torch.log2(torch.cat([16.0 / grp[:1], grp[:1] / grp[1:]])))
    scale3 = torch.exp2(torch.cat([lg[:1].clamp(-60, 60), lg[1:].clamp(0, 40)]))
    inv = 1.0 / scale3[0:1]
    inv_rgb, inv_sem = inv / scale3[1:2], inv / scale3[2:3]
    gbuf = ops.mlp_input_grads_x3(packed_bwd, sem_mode, g_raw, acts, scale3, masks)   # masks: ReLU bit masks of the forward, or None
    G = lambda a, n: gbuf[:, a:a + n]    # noqa: E731
    out: Dict[str, torch.Tensor] = {}

    Gs = torch.zeros((P_, 32), **f32)     # the small output heads, straight from g_raw (unscaled), as in mlp_backward
    Gs[:, :C] = g_raw
    small, db_s = torch.empty((32, W), **f32), torch.empty((32,), **f32)
    ops.wgrad(Gs, col(ACTS_VIEWS, 128), small[:, :128], db_s)
    out["rgb_linear.weight"], out["rgb_linear.bias"] = small[0:3, :128].clone(), db_s[0:3].clone()
    out["alpha_linear.bias"] = db_s[3:4].clone()
    ops.wgrad(Gs, h(7), small)
    out["alpha_linear.weight"] = small[3:4].clone()
    if sem_mode != SEM_NONE:
        ops.wgrad(Gs, col(ACTS_SEM, 128), small[:, :128])
        out["semantic_linear.2.weight"], out["semantic_linear.2.bias"] = small[4:6, :128].clone(), db_s[4:6].clone()

    dWv, db = torch.empty((128, W + 32), **f32), torch.empty((128,), **f32)
    ops.wgrad(G(ACTS_VIEWS, 128), col(ACTS_FEAT, W), dWv[:, :W], db)
    ops.wgrad(G(ACTS_VIEWS, 128), col(ACTS_D, 32), dWv[:, W:])
    out["views_linears.0.weight"], out["views_linears.0.bias"] = dWv[:, :W + 27] * inv_rgb, db * inv_rgb
    dWf, dbf = torch.empty((W, W), **f32), torch.empty((W,), **f32)
    ops.wgrad(G(ACTS_FEAT, W), h(7), dWf, dbf, split_fp16=split_wgrad)
    out["feature_linear.weight"], out["feature_linear.bias"] = dWf * inv_rgb, dbf * inv_rgb
    if sem_mode != SEM_NONE:
        dWs, dbs = torch.empty((128, W + 64), **f32), torch.empty((128,), **f32)
        ops.wgrad(G(ACTS_SEM, 128), h(7), dWs[:, :W], dbs)
        n_in = W
        if sem_mode == SEM_COORD:
            ops.wgrad(G(ACTS_SEM, 128), col(ACTS_X, 64), dWs[:, W:])
            n_in = W + 63
        out["semantic_linear.0.weight"], out["semantic_linear.0.bias"] = dWs[:, :n_in] * inv_sem, dbs * inv_sem
    tmp_x = torch.empty((W, 64), **f32)
    for l in range(7, -1, -1):
        g = G(W * l, W)
        dbl = torch.empty((W,), **f32)
        wname, bname = f"pts_linears.{l}.weight", f"pts_linears.{l}.bias"
        if l == 0:
            ops.wgrad(g, col(ACTS_X, 64), tmp_x, dbl)
            out[wname] = tmp_x[:, :63] * inv
        elif l == 5:
            dW5 = torch.empty((W, 63 + W), **f32)
            ops.wgrad(g, col(ACTS_X, 64), tmp_x, dbl)
            dWh = torch.empty((W, W), **f32)
            ops.wgrad(g, h(4), dWh, split_fp16=split_wgrad)
            dW5[:, :63], dW5[:, 63:] = tmp_x[:, :63], dWh
            out[wname] = dW5 * inv
        else:
            dWl = torch.empty((W, W), **f32)
            ops.wgrad(g, h(l - 1), dWl, dbl, split_fp16=split_wgrad)
            out[wname] = dWl * inv
        out[bname] = dbl * inv
    return out


_CHUNKS = (256, 128, 64, 32)
_WG_PLANS: dict = {}


def _chunks(n: int):
    """A 32-multiple as a sum of nsos_wgrad's tile sizes: [(offset, size)]."""
    out, o = [], 0
    while n > 0:
        c = next(c for c in _CHUNKS if c <= n)
        out.append((o, c))
        o, n = o + c, n - c
    return out


def _cover(n: int, room: int):
    """Tiles of nsos_wgrad's sizes covering columns 0 .. n-1 of a block (n a 32-multiple) that has `room` readable columns from its
    start: whole 256s first, then ONE tile for the rest -- the smallest size that holds it, reading past the block where the row has
    room (what lies there is another block or the bit words: the products it feeds land in output rows / columns that are sliced
    away; every output element is a sum over its own pair of columns only).  96 = one 128-tile instead of 64 + 32: for a 96 x 96
    layer one reduction over 128 + 128 columns per point instead of four over 384 in all -- the reductions are HBM-bound.  Falls back
    to exact tiles where the row ends.  [(offset, size)]"""
    out, o = [], 0
    while n - o > 256:
        out.append((o, 256))
        o += 256
    r = n - o
    if r > 0:
        c = next(c for c in reversed(_CHUNKS) if c >= r)
        if o + c <= room:
            out.append((o, c))
        else:
            out += [(o + oo, cc) for oo, cc in _chunks(r)]
    return out


def _wgrad_plan(plan, wanted: tuple):
    """The reductions of one generic net as ONE nsos_wgrad_batch list (cached per plan and set of Linears): per wanted Linear a block
    [pad32(out_dim), sum of pad32(segment rows)] + a bias row in one flat output buffer, every (row tile, column tile) of every segment
    one item.  The saved blocks are zero-padded, so padded rows / columns come out as exact zeros and are sliced away afterwards.
    Returns (ctypes item array, n_items, floats of the flat buffer, [(name, out_dim, w_off, Mp, Kp, b_off, [(rows, wcol, column in the
    block)], aligned)])."""
    import ctypes as C
    from . import _lib
    key = (id(plan), wanted)
    hit = _WG_PLANS.get(key)
    if hit is not None and hit[0] is plan:
        return hit[1]
    pad = lambda n: (n + 31) // 32 * 32  # noqa: E731
    ld, layout = plan.layout()
    items, blocks, off = [], [], 0
    for name, col, out_dim, segs in layout:
        if name not in wanted:
            continue
        m_tiles = _cover(pad(out_dim), ld - col)
        n_tiles = [_cover(pad(rows), ld - src_col) for src_col, rows, _ in segs]
        Mp = m_tiles[-1][0] + m_tiles[-1][1]                           # rows / columns of the output block: what the tiles cover
        Nps = [t[-1][0] + t[-1][1] for t in n_tiles]
        Kp = sum(Nps)
        w_off, b_off = off, off + Mp * Kp
        off = b_off + Mp
        kc, placed, first = 0, [], True
        for (src_col, rows, wcol), tiles, Np in zip(segs, n_tiles, Nps):
            for mo, mc in m_tiles:
                for no, nc in tiles:
                    items.append((w_off + mo * Kp + kc + no, (b_off + mo) if (first and no == 0) else -1, col + mo, src_col + no, mc, nc, Kp))
            placed.append((rows, wcol, kc))
            kc += Np
            first = False
        aligned = Mp == out_dim and all(np_ == r for (r, _, _), np_ in zip(placed, Nps)) and all(w == k for _, w, k in placed)
        blocks.append((name, out_dim, w_off, Mp, Kp, b_off, placed, aligned))
    arr = (_lib.WgradItem * max(len(items), 1))()
    for i, (w, b, g, x, m, n, ldw) in enumerate(items):
        arr[i].w_off, arr[i].b_off, arr[i].g_col, arr[i].x_col, arr[i].M, arr[i].N, arr[i].ldw = w, b, g, x, m, n, ldw
    res = (arr, len(items), off, blocks)
    if len(_WG_PLANS) > 64:
        _WG_PLANS.clear()
    _WG_PLANS[key] = (plan, res)
    return res


def generic_weight_grads(mlp, plan, gbuf: torch.Tensor, acts: torch.Tensor, skip_frozen: bool = True) -> Dict[str, torch.Tensor]:
    """dW = gbuf[:, block]^T acts[:, segment], db = column sums for every (trainable) Linear of a generic net: one nsos_wgrad_batch call
    into one flat buffer, then views (32-aligned Linears: the gradient IS the block) or one slicing copy per ragged Linear (the
    encodings' 63 / 27 columns, 1-3-row heads).  Round 5: until then one Python-level nsos_wgrad call per tile with three tensor
    slicings each -- 40-110 calls per pass, most of a small net's backward time."""
    from . import _lib
    params = dict(mlp.named_parameters())
    wanted = tuple(name for name, _, _, _ in plan.layout()[1]
                   if not skip_frozen or params[name + ".weight"].requires_grad or params[name + ".bias"].requires_grad)
    arr, n_items, floats, blocks = _wgrad_plan(plan, wanted)
    flat = torch.empty((max(floats, 1),), device=acts.device, dtype=torch.float32)
    ops.wgrad_batch(arr, n_items, gbuf, acts, flat)
    out: Dict[str, torch.Tensor] = {}
    for name, out_dim, w_off, Mp, Kp, b_off, placed, aligned in blocks:
        blk = flat[w_off: w_off + Mp * Kp].view(Mp, Kp)
        if aligned:
            out[name + ".weight"] = blk
        elif len(placed) == 1:
            out[name + ".weight"] = blk[:out_dim, :placed[0][0]].contiguous()
        else:
            out[name + ".weight"] = torch.cat([blk[:out_dim, kc: kc + rows] for rows, _, kc in placed], dim=1)
        out[name + ".bias"] = flat[b_off: b_off + out_dim]
    return out


def generic_mlp_backward(mlp, plan, acts: torch.Tensor, g_raw: torch.Tensor, packed_bwd: torch.Tensor, rays=None, points=None,
                         skip_frozen: bool = True):
    """Gradients of every parameter of one generic-architecture MLP, by name (K7-G; autograd of models/nerf_mlp.py:67-100 for any
    depth / width / skip set / head shape).  One kernel runs the whole input-gradient chain over the saved activations
    (nsos_mlp_generic_input_grads: exact-fp32 MFMA over transposed weight streams, ReLU masks from `acts`) and leaves every
    Linear's pre-activation gradient in its column block of `gbuf`; each weight gradient is then dW = gbuf[:, block]^T acts[:, segment]
    on the exact-fp32 reduction kernel (nsos_wgrad), in 32-multiple tiles -- the blocks are zero-padded, so the padded rows and
    columns come out as exact zeros and are sliced away.  `plan` = the net's ops.GenericPlan (layout of the saved rows).
    rays = (rays_o, rays_d, viewdirs, z_vals) with a `packed_bwd` packed for input gradients: returns (gradients by name, g_pts [P,3],
    g_dirs [P,3] or None) -- the chain continued through the positional encodings.  points = (pts, dirs) / ("encoded",): the same for a
    point query / for MLP.forward's pre-encoded input (then g_pts is the gradient of the encoded row)."""
    g_pts = g_dirs = None
    if points is not None:         # a point query: points = (pts, dirs or None) or ("encoded",): gradients to the query's own inputs
        if isinstance(points[0], str):
            gbuf, g_pts, g_dirs = ops.mlp_generic_input_grads_points(plan, packed_bwd, g_raw, acts, None, None, encoded=True)
        else:
            gbuf, g_pts, g_dirs = ops.mlp_generic_input_grads_points(plan, packed_bwd, g_raw, acts, points[0], points[1])
        rays = points
    elif rays is None:
        gbuf = ops.mlp_generic_input_grads(plan, packed_bwd, g_raw, acts)
    else:
        gbuf, g_pts, g_dirs = ops.mlp_generic_input_grads(plan, packed_bwd, g_raw, acts, rays)
    out = generic_weight_grads(mlp, plan, gbuf, acts, skip_frozen)
    return out if rays is None else (out, g_pts, g_dirs)
