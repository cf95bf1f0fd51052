"""TEST INFRASTRUCTURE ONLY -- CPU restatement (torch ops, op for op) of the consumers right after the render path
(SURVEY 8f ranks 2 and 3): the two correlation losses and the evaluation post-processing.

Pinned: tests/golden/make_goldens_losses.py runs the REAL reference classes (utils/image.py, imported from
/root/reference in the build container) on the same inputs with the same random draws injected and asserts this
port reproduces them bit for bit before writing tests/golden/losses.npz.  Only tests/, __graft_entry__.smoke()
and bench.py's cpu_baseline leg may import this module; the product (nerf-sos_amd/) never does.

Randomness is an explicit input here (`coords1`, `coords2`, `neg_indx`): the reference draws
`coords1 = rand([B,11,11,2])*2-1` then `coords2` likewise (utils/image.py:343-344) from the global generator.
"""
from __future__ import annotations

from dataclasses import dataclass
from typing import Optional

import torch
import torch.nn.functional as F


# ------------------------------------------------------------------------------------------ metrics / eval post
def img2mse(x: torch.Tensor, y: torch.Tensor) -> torch.Tensor:
    """utils/image.py:125-128 (reduction='mean')."""
    return torch.mean(torch.mean((x - y) ** 2, -1))


def mse2psnr(x: torch.Tensor) -> torch.Tensor:
    """utils/image.py:134-137."""
    return -10. * torch.log(x) / torch.log(torch.tensor([10.]))


def eval_postprocess(semantics: Optional[torch.Tensor], rgb: Optional[torch.Tensor] = None,
                     target: Optional[torch.Tensor] = None) -> dict:
    """engines/eval.py:44-57 (clus_no_sfm False branch) and :79-86."""
    out = {}
    if semantics is not None:
        sem_prob = semantics.detach().cpu().float().softmax(dim=-1)      # :55
        out["sem_prob"] = sem_prob
        out["sem"] = torch.argmax(sem_prob, -1).unsqueeze(-1).to(torch.int32)   # :56, :59-60 astype(int32)
    if rgb is not None and target is not None:
        mse = img2mse(rgb, target)                                        # :83
        out["mse"] = mse.reshape(1)
        out["psnr"] = mse2psnr(mse).reshape(1)                            # :85
    return out


# ------------------------------------------------------------------------------------------ correlation losses
@dataclass
class CorrParams:
    """Attributes of CorrelationLoss / GeoCorrelationLoss after __init__ (utils/image.py:265-290, 374-402);
    shift/weight values come from args.app_corr_params / args.geo_corr_params."""
    self_shift: float = 0.18
    self_weight: float = 0.67
    neg_shift: float = 0.46
    neg_weight: float = 0.63
    zero_clamp: bool = True
    stabalize: bool = False
    pointwise: bool = True
    feature_samples: int = 11
    max_depth: float = 15.0


def _norm(t):                      # utils/image.py:300-301
    return F.normalize(t, dim=1, eps=1e-10)


def _sample(t, coords):            # utils/image.py:303-304
    return F.grid_sample(t, coords.permute(0, 2, 1, 3), padding_mode='border', align_corners=True)


def _dot_correlation(a, b):        # utils/image.py:297-298
    return torch.einsum("nchw,ncij->nhwij", a, b)


def _l1_correlation(a, b, max_depth):   # utils/image.py:404-413
    x = a.unsqueeze(-1).unsqueeze(-1)
    y = b.unsqueeze(2).unsqueeze(3)
    ret = torch.sum(torch.abs(x - y), dim=1)
    ret = torch.abs(ret)
    ret = ret + (torch.ones_like(ret) * 5e-2)
    ret = 1 / ret
    ret[ret > max_depth] = torch.Tensor([max_depth])
    return ret


def _helper(fd, c1, c2, shift, p: CorrParams, code_corr=_dot_correlation):
    """utils/image.py:311-333 and :415-438 after `fd` has been formed.  `code_corr` is the class's own
    `tensor_correlation`: the dot product for CorrelationLoss, and -- because GeoCorrelationLoss overrides the
    method -- the clamped inverse L1 distance for the geometric loss's code term as well (:427)."""
    with torch.no_grad():
        if p.pointwise:
            old_mean = fd.mean()
            fd -= fd.mean([3, 4], keepdim=True)
            fd = fd - fd.mean() + old_mean
    cd = code_corr(_norm(c1), _norm(c2))
    min_val = 0.0 if p.zero_clamp else -9999.0
    if p.stabalize:
        loss = - cd.clamp(min_val, .8) * (fd - shift)
    else:
        loss = - cd.clamp(min_val) * (fd - shift)
    return loss, cd


def neg_index(sim_matrix: torch.Tensor) -> torch.Tensor:
    """utils/image.py:354 -- index of the least similar patch per column."""
    return torch.min(sim_matrix, dim=0)[1]


def correlation_loss(orig_feats, orig_code, neg_indx, coords1, coords2, p: CorrParams):
    """CorrelationLoss.forward (utils/image.py:335-370) with the random draws and the negative index injected."""
    feats = _sample(orig_feats, coords1)
    code = _sample(orig_code, coords1)
    neg_feats = _sample(orig_feats[neg_indx], coords2)
    neg_code = _sample(orig_code[neg_indx], coords2)
    with torch.no_grad():
        fd_neg = _dot_correlation(_norm(feats), _norm(neg_feats))
        fd_self = _dot_correlation(_norm(feats), _norm(feats))
    neg_loss, _ = _helper(fd_neg, code, neg_code, p.neg_shift, p)
    self_loss, _ = _helper(fd_self, code, code, p.self_shift, p)
    return p.neg_weight * neg_loss.mean() + p.self_weight * self_loss.mean()


def depth_filter_(depth: torch.Tensor, max_depth: float) -> torch.Tensor:
    """utils/image.py:455 -- IN PLACE, as the reference (it mutates the caller's ret_dict['depth'])."""
    depth[depth > max_depth] = depth[depth < max_depth].max()
    return depth


def geo_correlation_loss(depth, orig_code, ray_o, ray_d, neg_indx, p: CorrParams):
    """GeoCorrelationLoss.forward (utils/image.py:448-487): depth [B,1,P,P], code [B,C,P,P], rays [B,3,P,P]."""
    depth = depth_filter_(depth, p.max_depth)
    B, _, ph, pw = depth.shape
    xyz = (ray_o + ray_d * depth).reshape(B, -1, ph * pw).view(B, 3, ph, pw)   # depth2pts :440-446
    neg_xyz = xyz[neg_indx]
    neg_code = orig_code[neg_indx]
    with torch.no_grad():
        fd_neg = _l1_correlation(xyz, neg_xyz, p.max_depth)
    l1 = lambda a, b: _l1_correlation(a, b, p.max_depth)  # noqa: E731
    neg_loss, _ = _helper(fd_neg, orig_code, neg_code, p.neg_shift, p, l1)
    del fd_neg
    with torch.no_grad():
        fd_self = _l1_correlation(xyz, xyz, p.max_depth)
    self_loss, _ = _helper(fd_self, orig_code, orig_code, p.self_shift, p, l1)
    return p.neg_weight * neg_loss.mean() + p.self_weight * self_loss.mean()


# ------------------------------------------------------------------------------------------ contrastive loss on class tokens
def nerf_contrastive(embeddings: torch.Tensor) -> torch.Tensor:
    """NeRFContrastive.forward, min_max_contrast=True (utils/image.py:200-217; call site engines/trainer.py:168-170 on the
    DINO class tokens `cls_` [B,384]): cosine-similarity matrix, its off-diagonal entries, loss = -log(max / (max + min)).
    (`temperature` is registered but unused by the reference; `min_max_contrast=False` raises there.)"""
    sim = F.cosine_similarity(embeddings.unsqueeze(1), embeddings.unsqueeze(0), dim=2)     # :202
    mask = torch.eye(embeddings.shape[0], dtype=torch.bool)                                # :205
    sim = sim[~mask]                                                                       # :206
    mn = sim[torch.argmin(sim)]                                                            # :207
    mx = sim[torch.argmax(sim)]                                                            # :208
    return -torch.log(mx / (mx + mn))                                                      # :209
