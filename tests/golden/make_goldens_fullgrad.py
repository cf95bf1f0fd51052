#!/usr/bin/env python3
"""Golden gradients for the full backward (every parameter trainable, as configs/*_full.txt train the reference
unless --fix_backbone is given).  Run in the BUILD CONTAINER only (needs /root/reference):

    python tests/golden/make_goldens_fullgrad.py

Runs the REAL reference NeRFNet (imported by make_goldens.py's recipe) in eval mode on 12 rays, back-propagates a
random linear functional of all rendered maps, and stores inputs + every parameter's gradient in
tests/golden/full_grads.npz.  Data only.
"""
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import make_goldens as mg  # noqa: E402  (sets up the reference import; does not run its main)
from oracle import torch_port as tp  # noqa: E402


def main():
    torch.set_grad_enabled(True)
    out = {}
    rays = tp.synthetic_rays(12, seed=21)
    out["rays"] = mg.np32(rays)
    gg = torch.Generator().manual_seed(99)
    KEYS = ("rgb", "semantics", "depth", "acc", "weights", "raw")
    for name, peaky, white, n_imp in (("semcoord", True, False, 128), ("sem", False, True, 128), ("nosem", True, False, 0)):
        tag = f"{name}_{'peaky' if peaky else 'default'}{'_white' if white else ''}{'_coarse' if n_imp == 0 else ''}"
        net, pc, sd = mg.build_ref(name, n_importance=n_imp, white_bkgd=white, peaky=peaky)
        net.eval()
        ret = net(rays, (tp.NEAR, tp.FAR), radii=None)
        loss = 0.0
        for k in list(ret.keys()):
            if k.rstrip("0") not in KEYS:
                continue
            G = torch.randn(ret[k].shape, generator=gg) * (0.05 if k.startswith("raw") else 1.0)
            out[f"{tag}_G_{k}"] = mg.np32(G)
            loss = loss + (ret[k] * G).sum()
        loss.backward()
        seen = set()
        for n_, p_ in net.named_parameters():
            if id(p_) in seen:
                continue
            seen.add(id(p_))
            gr = p_.grad if p_.grad is not None else torch.zeros_like(p_)
            if gr.numel() > 8192:   # big matrices: 24 rows and 24 columns (fixed stride) instead of everything
                out[f"{tag}_gradrows_{n_}"] = mg.np32(gr[::max(1, gr.shape[0] // 24)])
                out[f"{tag}_gradcols_{n_}"] = mg.np32(gr[:, ::max(1, gr.shape[1] // 24)])
            else:
                out[f"{tag}_grad_{n_}"] = mg.np32(gr)
        out[f"{tag}_loss"] = mg.np32(loss.detach().reshape(1))
        print(tag, float(loss), len(seen), "parameters")
    path = os.path.join(HERE, "full_grads.npz")
    np.savez_compressed(path, **out)
    print(f"wrote {path}: {os.path.getsize(path) / 1e6:.2f} MB")


if __name__ == "__main__":
    main()
