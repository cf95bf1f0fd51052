"""Backward of one NeRF MLP over saved activations (K7): the orchestration of HIP kernels (`nsos_wgrad`,
`nsos_relu_mask`) and plain library GEMMs for the input gradients.  Mirrors the autograd graph of
MLP.forward (models/nerf_mlp.py:67-100): heads -> view branch -> feature -> [semantic head] -> 8 trunk layers with
the skip connection into layer 5.

acts [P, ACTS_DIM] is the buffer nsos_mlp_forward_rays_save_all wrote (column map in include/nerf_sos_hip.h);
g_raw [P, C] = d loss / d [r, g, b, sigma, (sem0, sem1)] from nsos_composite_backward.
"""
from __future__ import annotations

from typing import Dict

import torch

from . import ops
from .ops import ACTS_D, ACTS_FEAT, ACTS_SEM, ACTS_VIEWS, ACTS_X, SEM_COORD, SEM_NONE

W = 256


def mlp_backward(mlp, sem_mode: int, acts: torch.Tensor, g_raw: torch.Tensor) -> Dict[str, torch.Tensor]:
    P_, C = g_raw.shape
    dev = acts.device
    prm = {n: p.detach() for n, p in mlp.named_parameters()}
    f32 = dict(device=dev, dtype=torch.float32)
    col = lambda a, n: acts[:, a:a + n]  # noqa: E731
    h = lambda l: col(W * l, W)          # relu(pts_linears.l)  # noqa: E731
    out: Dict[str, torch.Tensor] = {}

    # ---- the small output heads share one padded gradient matrix Gs [P,32] = [g_rgb(3) | g_sigma | g_sem(2) | 0...]
    Gs = torch.zeros((P_, 32), **f32)
    Gs[:, :C] = g_raw
    small, db_s = torch.empty((32, W), **f32), torch.empty((32,), **f32)
    ops.wgrad(Gs, col(ACTS_VIEWS, 128), small[:, :128], db_s)          # rgb = rgb_linear(relu(views))   (:92)
    out["rgb_linear.weight"], out["rgb_linear.bias"] = small[0:3, :128].clone(), db_s[0:3].clone()
    out["alpha_linear.bias"] = db_s[3:4].clone()
    ops.wgrad(Gs, h(7), small)                                           # sigma = alpha_linear(h7)         (:77)
    out["alpha_linear.weight"] = small[3:4].clone()
    if sem_mode != SEM_NONE:
        ops.wgrad(Gs, col(ACTS_SEM, 128), small[:, :128])               # logits = semantic_linear.2(hs)  (:61,80)
        out["semantic_linear.2.weight"], out["semantic_linear.2.bias"] = small[4:6, :128].clone(), db_s[4:6].clone()

    # ---- view branch: v = relu(views_linears.0(cat([feature, d27])))  (:87-91)
    g_v = g_raw[:, 0:3].contiguous() @ prm["rgb_linear.weight"]          # [P,128]
    ops.relu_mask_(g_v, col(ACTS_VIEWS, 128))
    dWv, db = torch.empty((128, W + 32), **f32), torch.empty((128,), **f32)
    ops.wgrad(g_v, col(ACTS_FEAT, W), dWv[:, :W], db)
    ops.wgrad(g_v, col(ACTS_D, 32), dWv[:, W:])
    out["views_linears.0.weight"], out["views_linears.0.bias"] = dWv[:, :W + 27].contiguous(), db
    g_feat = g_v @ prm["views_linears.0.weight"][:, :W].contiguous()    # [P,256]
    dWf, dbf = torch.empty((W, W), **f32), torch.empty((W,), **f32)
    ops.wgrad(g_feat, h(7), dWf, dbf)                                    # feature = feature_linear(h7)    (:86)
    out["feature_linear.weight"], out["feature_linear.bias"] = dWf, dbf
    g_h = g_feat @ prm["feature_linear.weight"]                          # d/d h7 ...
    g_h.addmm_(g_raw[:, 3:4].contiguous(), prm["alpha_linear.weight"])   # ... + sigma head

    # ---- semantic head: hs = relu(semantic_linear.0(cat([h7, x63])))  (:79-80; h first)
    if sem_mode != SEM_NONE:
        g_hs = g_raw[:, 4:6].contiguous() @ prm["semantic_linear.2.weight"]   # [P,128]
        ops.relu_mask_(g_hs, col(ACTS_SEM, 128))
        dWs, dbs = torch.empty((128, W + 64), **f32), torch.empty((128,), **f32)
        ops.wgrad(g_hs, h(7), dWs[:, :W], dbs)
        n_in = W
        if sem_mode == SEM_COORD:
            ops.wgrad(g_hs, col(ACTS_X, 64), dWs[:, W:])
            n_in = W + 63
        out["semantic_linear.0.weight"], out["semantic_linear.0.bias"] = dWs[:, :n_in].contiguous(), dbs
        g_h.addmm_(g_hs, prm["semantic_linear.0.weight"][:, :W].contiguous())

    # ---- trunk, layers 7..0  (:69-75): h_l = relu(pts_linears.l(in_l)), in_5 = cat([x63, h4])
    tmp_x = torch.empty((W, 64), **f32)
    for l in range(7, -1, -1):
        ops.relu_mask_(g_h, h(l))                                        # g_h is now d/d (pre-activation of layer l)
        dbl = torch.empty((W,), **f32)
        wname, bname = f"pts_linears.{l}.weight", f"pts_linears.{l}.bias"
        if l == 0:
            ops.wgrad(g_h, col(ACTS_X, 64), tmp_x, dbl)
            out[wname] = tmp_x[:, :63].clone()
        elif l == 5:
            dW5 = torch.empty((W, 63 + W), **f32)
            ops.wgrad(g_h, col(ACTS_X, 64), tmp_x, dbl)
            dWh = torch.empty((W, W), **f32)
            ops.wgrad(g_h, h(4), dWh)
            dW5[:, :63], dW5[:, 63:] = tmp_x[:, :63], dWh
            out[wname] = dW5
        else:
            dWl = torch.empty((W, W), **f32)
            ops.wgrad(g_h, h(l - 1), dWl, dbl)
            out[wname] = dWl
        out[bname] = dbl
        if l > 0:
            w_h = prm[wname][:, 63:].contiguous() if l == 5 else prm[wname]
            g_h = g_h @ w_h
    return out


def mlp_backward_x3(mlp, sem_mode: int, acts: torch.Tensor, g_raw: torch.Tensor, packed_bwd: torch.Tensor,
                    masks=None) -> Dict[str, torch.Tensor]:
    """Same result as mlp_backward with the input-gradient chain (nine GEMMs + ReLU masks) replaced by ONE fused
    split-fp16 kernel (nsos_mlp_input_grads_x3, csrc/mlp_x3_bwd.hip); the weight-gradient reductions are the same
    nsos_wgrad calls, fed from its output matrix.  `packed_bwd` = the net's packed_weights("fp16x3_bwd")."""
    P_, C = g_raw.shape
    dev = acts.device
    f32 = dict(device=dev, dtype=torch.float32)
    col = lambda a, n: acts[:, a:a + n]  # noqa: E731
    h = lambda l: col(W * l, W)          # noqa: E731
    # fp16 has 5 exponent bits: bring max |g_raw| to ~2^4 with an exact power of two (no host sync), undo on the results.
    # 2^4 leaves a factor 2^12 of growth along the chain before fp16 overflows and 2^-18 of shrinkage before the hi parts
    # go subnormal (layer gains of trained NeRFs are O(1)).
    amax = g_raw.abs().max().clamp_min(1e-30)
    scale = torch.exp2(torch.floor(torch.log2(16.0 / amax))).clamp(2.0 ** -60, 2.0 ** 60).reshape(1)
    inv = 1.0 / scale
    gbuf = ops.mlp_input_grads_x3(packed_bwd, sem_mode, g_raw, acts, scale, masks)   # masks: ReLU bit masks of the forward, or None
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
    out["views_linears.0.weight"], out["views_linears.0.bias"] = dWv[:, :W + 27] * inv, db * inv
    dWf, dbf = torch.empty((W, W), **f32), torch.empty((W,), **f32)
    ops.wgrad(G(ACTS_FEAT, W), h(7), dWf, dbf, split_fp16=True)
    out["feature_linear.weight"], out["feature_linear.bias"] = dWf * inv, dbf * inv
    if sem_mode != SEM_NONE:
        dWs, dbs = torch.empty((128, W + 64), **f32), torch.empty((128,), **f32)
        ops.wgrad(G(ACTS_SEM, 128), h(7), dWs[:, :W], dbs)
        n_in = W
        if sem_mode == SEM_COORD:
            ops.wgrad(G(ACTS_SEM, 128), col(ACTS_X, 64), dWs[:, W:])
            n_in = W + 63
        out["semantic_linear.0.weight"], out["semantic_linear.0.bias"] = dWs[:, :n_in] * inv, dbs * inv
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
            ops.wgrad(g, h(4), dWh, split_fp16=True)
            dW5[:, :63], dW5[:, 63:] = tmp_x[:, :63], dWh
            out[wname] = dW5 * inv
        else:
            dWl = torch.empty((W, W), **f32)
            ops.wgrad(g, h(l - 1), dWl, dbl, split_fp16=True)
            out[wname] = dWl * inv
        out[bname] = dbl * inv
    return out
