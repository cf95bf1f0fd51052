#!/usr/bin/env python3
"""Golden gradients WITH RESPECT TO THE RAYS (pose refinement: the reference's autograd flows through pts = o + d z, viewdirs =
d / |d| and the renderer's dists * |d|), from the REAL reference (build container only; /root/reference mounted read-only):

    python tests/golden/make_goldens_raygrad.py

Cases: the shipped architecture (semcoord, spiky density, 64 + 128 samples, white background on and off) and two generic ones
(d6w96_m6, noview of make_goldens_generic.py).  The reference's NeRFNet renders 12 rays that require a gradient in eval mode, a
random linear functional of its rendered maps is back-propagated, and rays, the upstream gradients, the coarse weights (from which
the tests rebuild the reference's fine sample positions with the port, bit-identically) and rays.grad go to
tests/golden/ray_grads.npz.  Only data is written.
"""
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import make_goldens as mg  # noqa: E402
import make_goldens_generic as mgg  # noqa: E402
from oracle import torch_port as tp  # noqa: E402

KEYS = ("rgb", "semantics", "depth", "acc", "weights", "raw")


def run(model, rays, gg, out, tag):
    rays = rays.clone().requires_grad_(True)
    torch.set_grad_enabled(True)
    ret = model(rays, (tp.NEAR, tp.FAR))
    loss = 0.0
    for k in list(ret.keys()):
        if k.rstrip("0") not in KEYS or ret[k].numel() == 0:
            continue
        G = torch.randn(ret[k].shape, generator=gg) * (0.05 if k.startswith("raw") else 1.0)
        if k.rstrip("0") in ("depth", "disp"):
            G = G * (ret[k].detach().abs() < 1e9)       # empty rays carry depth 1e10 (no gradient): keep the loss finite-sized
        out[f"{tag}__G__{k}"] = G.numpy()
        loss = loss + (ret[k] * G).sum()
    loss.backward()
    torch.set_grad_enabled(False)
    out[f"{tag}__rays"] = rays.detach().numpy().copy()
    out[f"{tag}__g_rays"] = rays.grad.numpy().copy()
    if "weights0" in ret:
        out[f"{tag}__weights0"] = ret["weights0"].detach().numpy().copy()
    out[f"{tag}__rgb"] = ret["rgb"].detach().numpy().copy()
    print(f"{tag}: loss {float(loss):.6f}, |g_o| max {float(rays.grad[0].abs().max()):.4g}, |g_d| max {float(rays.grad[1].abs().max()):.4g}")


def main():
    out = {}
    gg = torch.Generator().manual_seed(777)
    for name, peaky, white in (("semcoord", True, False), ("sem", True, True)):
        torch.set_grad_enabled(False)
        net, pc, sd = mg.build_ref(name, n_importance=128, white_bkgd=white, peaky=peaky)
        net.eval()
        run(net, tp.synthetic_rays(12, seed=31), gg, out, f"{name}{'_white' if white else ''}")
    fwd = dict(np.load(os.path.join(HERE, "generic.npz")))
    for name in ("d6w96_m6", "noview"):
        ref_kw, port_kw = mgg.CASES[name]
        seed = int(fwd[f"{name}__seed"][0])
        torch.set_grad_enabled(False)
        torch.manual_seed(seed)
        model = mgg.NeRFNet(**ref_kw).eval()
        model.load_state_dict(mgg.generic_state(tp.PortConfig(**port_kw), seed))
        run(model, tp.synthetic_rays(12, seed=seed + 1), gg, out, name)
    path = os.path.join(HERE, "ray_grads.npz")
    np.savez_compressed(path, **out)
    print(f"wrote {path}: {os.path.getsize(path) / 1e6:.2f} MB")


if __name__ == "__main__":
    main()
