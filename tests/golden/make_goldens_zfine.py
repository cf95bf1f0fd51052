#!/usr/bin/env python3
"""The REFERENCE's own fine-pass sample positions for the committed end-to-end and full-gradient cases.

Run in the BUILD CONTAINER only (needs /root/reference):

    python tests/golden/make_goldens_zfine.py

`end_to_end.npz` / `full_grads.npz` hold the reference's outputs but not the intermediate `z_vals` its importance
sampler handed to the fine network (models/nerf_net.py:113).  A forward hook on the real reference's
`importance_sampler` module records them here, for exactly the same cases and seeds (the script re-runs the reference
and asserts its outputs equal the stored goldens bit for bit, so the recorded samples belong to those outputs).
With them the GPU tests can hold the fine network + compositing to a strict 1e-4 on every key, separately from the
(documented) last-ulp index flips of a free-running hierarchical sampler.  Data only: `zfine.npz`.
"""
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import make_goldens as mg  # noqa: E402  (sets up the reference import; does not run its main)
from oracle import torch_port as tp  # noqa: E402


def record_z(net):
    box = {}
    h = net.importance_sampler.register_forward_hook(lambda m, a, out: box.__setitem__("z", out[1].detach().clone()))
    return box, h


def main():
    e2e = dict(np.load(os.path.join(HERE, "end_to_end.npz")))
    out = {}
    rays = torch.as_tensor(e2e["rays"])
    cases = [("nosem", False, False, 128), ("semcoord", False, False, 128), ("semcoord", True, False, 128), ("sem", True, True, 128)]
    for name, peaky, white, n_imp in cases:
        tag = f"{name}_{'peaky' if peaky else 'default'}{'_white' if white else ''}"
        net, pc, sd = mg.build_ref(name, n_importance=n_imp, white_bkgd=white, peaky=peaky, raw_noise_std=1.0)
        box, h = record_z(net)
        net.eval()
        with torch.no_grad():
            ret = net(rays, (tp.NEAR, tp.FAR), radii=None)
        for k in ret:
            assert np.array_equal(mg.np32(ret[k]), e2e[f"{tag}_eval_{k}"]), f"{tag} eval {k}: not the stored golden"
        out[f"{tag}_eval_z_fine"] = mg.np32(box["z"])
        net.train()
        torch.manual_seed(99)
        with torch.no_grad():
            ret = net(rays, (tp.NEAR, tp.FAR), radii=None)
        for k in ret:
            assert np.array_equal(mg.np32(ret[k]), e2e[f"{tag}_train_{k}"]), f"{tag} train {k}: not the stored golden"
        out[f"{tag}_train_z_fine"] = mg.np32(box["z"])
        h.remove()
        print(tag, "ok")

    fg = dict(np.load(os.path.join(HERE, "full_grads.npz")))
    rays = torch.as_tensor(fg["rays"])
    for name, peaky, white in (("semcoord", True, False), ("sem", False, True)):
        tag = f"{name}_{'peaky' if peaky else 'default'}{'_white' if white else ''}"
        net, pc, sd = mg.build_ref(name, n_importance=128, white_bkgd=white, peaky=peaky)
        box, h = record_z(net)
        net.eval()
        with torch.no_grad():
            ret = net(rays, (tp.NEAR, tp.FAR), radii=None)
        loss = sum(float((ret[k] * torch.as_tensor(fg[f"{tag}_G_{k}"])).sum()) for k in ret if f"{tag}_G_{k}" in fg)
        assert abs(loss - float(fg[f"{tag}_loss"][0])) <= 1e-3 * (1 + abs(loss)), (loss, fg[f"{tag}_loss"])
        out[f"fullgrad_{tag}_z_fine"] = mg.np32(box["z"])
        h.remove()
        print("fullgrad", tag, "ok")
    path = os.path.join(HERE, "zfine.npz")
    np.savez_compressed(path, **out)
    print(f"wrote {path}: {os.path.getsize(path) / 1e3:.1f} kB, {len(out)} arrays")


if __name__ == "__main__":
    main()
