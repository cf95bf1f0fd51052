#!/usr/bin/env python3
"""The REAL reference's autograd on the TRAINED field (training side).  Run in the BUILD CONTAINER only:

    python tests/golden/make_goldens_trained_grads.py

tests/golden/trained_scene.ckpt loaded into the unmodified reference NeRFNet; 48 rays of trained.npz's batch (12 per held-out view);
eval mode (deterministic), z_fine recorded with the forward hook; a loss shaped like the reference's own training losses --
img2mse of rgb and rgb0 against the analytic colours (engines/trainer.py:113-121) plus a random linear functional of the two
semantic maps (what the correlation losses hand back: a gradient per rendered logit) -- back-propagated
  (a) into every parameter (configs/*_full.txt), and
  (b) into semantic_linear.* alone (--fix_backbone, run_nerf.py:307-318): the same loss, so (b) is a subset of (a) -- stored once.
Written: tests/golden/trained_grads.npz -- rays, z_fine, the upstream gradient G_sem*, the loss, every small gradient whole and 24
rows + 24 columns of every big one (make_goldens_fullgrad.py's convention).  Data only.
"""
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import make_goldens as mg  # noqa: E402
from make_goldens_zfine import record_z  # noqa: E402


def main():
    torch.set_grad_enabled(True)
    g = dict(np.load(os.path.join(HERE, "trained.npz")))
    sel = np.concatenate([np.arange(v * 64, v * 64 + 12) for v in range(4)])
    rays = torch.from_numpy(g["rays"][:, sel])
    gt = torch.from_numpy(g["gt_rgb"][sel])
    near, far = (float(v) for v in g["near_far"])
    net = mg.NeRFNet(N_samples=64, N_importance=128, perturb=1.0, raw_noise_std=1.0, pts_chuck=1024 * 64, use_semantics=True, sem_with_coord=True)
    net.load_state_dict(torch.load(os.path.join(HERE, "trained_scene.ckpt"), map_location="cpu")["model"], strict=True)
    net.eval()
    box, h = record_z(net)
    ret = net(rays, (near, far), radii=None)
    h.remove()
    for k in ("rgb", "semantics", "depth"):
        assert np.array_equal(mg.np32(ret[k]), g[f"eval_{k}"][sel]), k            # the same render as trained.npz's
    gg = torch.Generator().manual_seed(5)
    G1, G0 = torch.randn(ret["semantics"].shape, generator=gg) * 0.05, torch.randn(ret["semantics0"].shape, generator=gg) * 0.05
    loss = ((ret["rgb"] - gt) ** 2).mean() + ((ret["rgb0"] - gt) ** 2).mean() + (ret["semantics"] * G1).sum() + (ret["semantics0"] * G0).sum()
    loss.backward()
    out = {"sel": sel, "rays": mg.np32(rays), "gt": mg.np32(gt), "z_fine": mg.np32(box["z"]), "G_semantics": mg.np32(G1), "G_semantics0": mg.np32(G0),
           "loss": mg.np32(loss.detach().reshape(1))}
    for n_, p_ in net.named_parameters():
        gr = p_.grad
        assert gr is not None and torch.isfinite(gr).all(), n_
        if gr.numel() > 8192:
            out[f"gradrows_{n_}"] = mg.np32(gr[::max(1, gr.shape[0] // 24)])
            out[f"gradcols_{n_}"] = mg.np32(gr[:, ::max(1, gr.shape[1] // 24)])
        else:
            out[f"grad_{n_}"] = mg.np32(gr)
        out[f"gradmax_{n_}"] = np.array([float(gr.abs().max())])
    path = os.path.join(HERE, "trained_grads.npz")
    np.savez_compressed(path, **out)
    print(f"loss {float(loss):.6f}; wrote {path}: {os.path.getsize(path) / 1e6:.2f} MB, {len(out)} arrays")


if __name__ == "__main__":
    main()
