#!/usr/bin/env python3
"""Golden GRADIENTS for architectures other than the shipped one (training a generic-architecture net: csrc/mlp_generic.hip's
backward program), from the REAL reference's autograd (build container only; /root/reference mounted read-only):

    python tests/golden/make_goldens_generic_grads.py

For every case of make_goldens_generic.py (same constructor arguments, seed, spiky density head and rays) the reference's own
`NeRFNet` renders in eval mode WITH autograd, a random linear functional of all rendered maps (coarse and fine) is
back-propagated, and the upstream gradients G, the loss and every parameter's gradient are written to
tests/golden/generic_grads.npz (matrices over 8192 elements: 24 rows + 24 columns at a fixed stride).  Only data is written.
"""
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import make_goldens_generic as mgg  # noqa: E402  (sets up the reference import; does not run its main)
from oracle import torch_port as tp  # noqa: E402

KEYS = ("rgb", "semantics", "depth", "acc", "weights", "raw")


def main():
    out = {}
    gg = torch.Generator().manual_seed(4242)
    fwd = dict(np.load(os.path.join(HERE, "generic.npz")))
    for name, (ref_kw, port_kw) in mgg.CASES.items():
        seed = int(fwd[f"{name}__seed"][0])        # the case's seed as make_goldens_generic.py drew it
        cfg = tp.PortConfig(**port_kw)
        sd = mgg.generic_state(cfg, seed)
        torch.set_grad_enabled(False)
        torch.manual_seed(seed)
        model = mgg.NeRFNet(**ref_kw).eval()
        model.load_state_dict(sd)
        rays = tp.synthetic_rays(24, seed=seed)
        assert np.array_equal(rays.numpy(), fwd[f"{name}__rays"])
        torch.set_grad_enabled(True)
        ret = model(rays, (tp.NEAR, tp.FAR))
        for k in ret:                              # the same render generic.npz recorded, now with an autograd graph
            assert np.array_equal(ret[k].detach().numpy(), fwd[f"{name}__out__{k}"]), (name, k)
        loss = 0.0
        for k in list(ret.keys()):
            if k.rstrip("0") not in KEYS or ret[k].numel() == 0:
                continue
            G = torch.randn(ret[k].shape, generator=gg) * (0.05 if k.startswith("raw") else 1.0)
            out[f"{name}__G__{k}"] = G.numpy()
            loss = loss + (ret[k] * G).sum()
        loss.backward()
        seen, n = set(), 0
        for n_, p_ in model.named_parameters():
            if id(p_) in seen:
                continue
            seen.add(id(p_))
            gr = p_.grad if p_.grad is not None else torch.zeros_like(p_)
            if gr.numel() > 8192:
                out[f"{name}__gradrows__{n_}"] = gr[::max(1, gr.shape[0] // 24)].numpy().copy()
                out[f"{name}__gradcols__{n_}"] = gr[:, ::max(1, gr.shape[1] // 24)].numpy().copy()
            else:
                out[f"{name}__grad__{n_}"] = gr.numpy().copy()
            n += 1
        out[f"{name}__loss"] = loss.detach().reshape(1).numpy()
        torch.set_grad_enabled(False)
        print(f"{name}: loss {float(loss):.6f}, {n} parameters")
    path = os.path.join(HERE, "generic_grads.npz")
    np.savez_compressed(path, **out)
    print(f"wrote {path}: {os.path.getsize(path) / 1e6:.2f} MB")


if __name__ == "__main__":
    main()
