#!/usr/bin/env python3
"""Golden vectors for `NeRFContrastive` (utils/image.py:192-218), the class-token contrastive loss BASELINE configs[2]
names (call site engines/trainer.py:168-170).  Run in the BUILD CONTAINER only (needs /root/reference):

    python tests/golden/make_goldens_contrastive.py

Imports the real reference `utils/image.py` (`imageio` / `lpips` stubbed in memory, unused here -- as in
make_goldens_losses.py), runs the REAL class forward + autograd backward on seeded class-token batches, asserts
oracle/losses_port.nerf_contrastive is bit-identical (loss and gradient), and writes tests/golden/contrastive.npz:
inputs, loss, d loss / d embeddings.  Only data is written.
"""
import os
import sys
import types

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
REF = os.environ.get("NERF_SOS_REFERENCE", "/root/reference")

sys.modules["imageio"] = types.ModuleType("imageio")
_lp = types.ModuleType("lpips")
_lp.LPIPS = lambda *a, **k: None
sys.modules["lpips"] = _lp
sys.path.insert(0, REF)
sys.path.insert(0, ROOT)

import utils.image as ref_image  # noqa: E402  (reference)
from oracle import losses_port as lp  # noqa: E402


def main():
    out = {}
    g = torch.Generator().manual_seed(42)
    cases = {"b2": (2, 384, 0.0), "b8": (8, 384, 0.0), "b16": (16, 384, 0.0), "b5_d7": (5, 7, 0.0),
             # class tokens of similar crops: a common component makes every similarity positive and close (the regime of
             # a real batch: patches of one scene), min + max well away from zero
             "b8_common": (8, 384, 3.0), "b64": (64, 384, 1.0), "b5_d7_common": (5, 7, 2.0)}
    # "b5_d7": uncorrelated low-dimensional tokens -> max + min < 0 -> log of a negative number: the reference returns NaN
    # (loss and gradient); the kernel must as well
    for tag, (B, D, common) in cases.items():
        e = torch.randn(B, D, generator=g) + common * torch.randn(1, D, generator=g)
        a = e.clone().requires_grad_(True)
        loss = ref_image.NeRFContrastive(device="cpu")(a)            # the real class
        loss.backward()
        b = e.clone().requires_grad_(True)
        lport = lp.nerf_contrastive(b)
        lport.backward()
        same = lambda x, y: torch.equal(torch.nan_to_num(x, nan=123.0), torch.nan_to_num(y, nan=123.0))   # noqa: E731
        assert same(loss, lport) and same(a.grad, b.grad), f"port != reference for {tag}"
        out[f"{tag}_emb"] = e.numpy()
        out[f"{tag}_loss"] = loss.detach().reshape(1).numpy()
        out[f"{tag}_grad"] = a.grad.numpy()
    np.savez_compressed(os.path.join(HERE, "contrastive.npz"), **out)
    assert np.isnan(out["b5_d7_loss"]).all() and sum(np.isnan(v).any() for k, v in out.items() if k.endswith("_loss")) == 1
    print("wrote contrastive.npz:", {k: float(v[0]) for k, v in out.items() if "loss" in k})


if __name__ == "__main__":
    main()
