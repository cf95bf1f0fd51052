"""Image-quality statistics of one render against another (BASELINE's metric is "rays/sec ...; PSNR vs ref").

The reference reports PSNR only (engines/eval.py:83-86: mse2psnr of the image MSE).  A mean hides a tail: on a trained field a
reduced-precision render agrees with the exact one to 1e-4 on 99 % of the rays and moves a few silhouette rays by tenths, which 56 dB
does not say.  `tail_stats` therefore returns, next to the PSNR, the percentiles and the COUNTS of rays over fixed thresholds for
|d rgb| (max over the three channels) and |d depth| / depth -- what bench.py prints and tests/test_gpu_trained.py asserts.
"""
from __future__ import annotations

import math
from typing import Dict, Optional

import torch


def _quantiles(t: torch.Tensor, qs) -> list:
    t = t.reshape(-1).double()
    if t.numel() == 0:
        return [float("nan")] * len(qs)
    srt = torch.sort(t).values                       # exact order statistics (torch.quantile refuses > 16 M elements)
    n = srt.numel()
    return [float(srt[min(n - 1, max(0, int(math.ceil(q * n)) - 1))]) for q in qs]


def tail_stats(rgb: torch.Tensor, ref_rgb: torch.Tensor, depth: Optional[torch.Tensor] = None,
               ref_depth: Optional[torch.Tensor] = None, labels: Optional[torch.Tensor] = None,
               ref_labels: Optional[torch.Tensor] = None) -> Dict[str, object]:
    """rgb / ref_rgb [R,3] (any float dtype, same device); depth / ref_depth [R] or [R,1]; labels [R] (argmax of the semantic logits)."""
    rgb, ref_rgb = rgb.reshape(-1, 3).double(), ref_rgb.reshape(-1, 3).double()
    R = rgb.shape[0]
    e = (rgb - ref_rgb).abs().amax(-1)
    mse = float(((rgb - ref_rgb) ** 2).mean())
    p50, p99, p999, p9999 = _quantiles(e, (0.5, 0.99, 0.999, 0.9999))
    out: Dict[str, object] = {
        "rays": int(R), "psnr_db": round(-10.0 * math.log10(max(mse, 1e-30)), 2),
        "abs_rgb": {"p50": p50, "p99": p99, "p99.9": p999, "p99.99": p9999, "max": float(e.max()),
                    "n_gt_0.01": int((e > 0.01).sum()), "n_gt_0.02": int((e > 0.02).sum()), "n_gt_0.05": int((e > 0.05).sum())},
        "share_of_rays_within_0.02": round(1.0 - float((e > 0.02).sum()) / max(R, 1), 6),
    }
    if depth is not None and ref_depth is not None:
        dp, rf = depth.reshape(-1).double(), ref_depth.reshape(-1).double()
        rd = (dp - rf).abs() / rf.abs().clamp_min(1e-30)
        q50, q99, q999 = _quantiles(rd, (0.5, 0.99, 0.999))
        out["rel_depth"] = {"p50": q50, "p99": q99, "p99.9": q999, "max": float(rd.max()),
                            "n_gt_0.01": int((rd > 0.01).sum()), "n_gt_0.1": int((rd > 0.1).sum())}
    if labels is not None and ref_labels is not None:
        out["label_agreement"] = round(float((labels.reshape(-1) == ref_labels.reshape(-1)).double().mean()), 6)
    return out


def compact(stats: Dict[str, object], brief: bool = False) -> Dict[str, object]:
    """The one-row form bench.py's driver line carries (brief: PSNR, max, the two counts and the share within 0.02 only)."""
    a = stats["abs_rgb"]
    row = {"rays": stats["rays"], "psnr_db": stats["psnr_db"], "p99.9": round(a["p99.9"], 5), "max": round(a["max"], 4),
           "n_gt_0.01": a["n_gt_0.01"], "n_gt_0.05": a["n_gt_0.05"], "within_0.02": stats["share_of_rays_within_0.02"]}
    if brief:
        del row["rays"], row["p99.9"]
        return row
    if "rel_depth" in stats:
        row["depth_n_gt_0.01"] = stats["rel_depth"]["n_gt_0.01"]
        row["depth_max_rel"] = round(stats["rel_depth"]["max"], 4)
    if "label_agreement" in stats:
        row["labels"] = stats["label_agreement"]
    return row
