#!/usr/bin/env python3
"""Golden vectors for the rows next to the render path: correlation losses and evaluation
post-processing.  Run in the BUILD CONTAINER only (needs /root/reference):

    python tests/golden/make_goldens_losses.py

Imports the real reference `utils/image.py` (with `imageio` and `lpips` stubbed -- neither is used by the functions
exercised here), feeds both the reference classes and oracle/losses_port.py the same inputs with the same random
draws injected, asserts the port is bit-identical (loss values and gradients), and writes tests/golden/losses.npz.
Only data is written: inputs, injected draws, expected outputs.
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


def np32(t):
    return t.detach().cpu().numpy()


def ref_args(app, geo, patch_stride=6):
    a = types.SimpleNamespace()
    a.rand_neg = False
    a.self_corr_w = 0
    a.use_sim_matrix = True
    a.app_corr_params = [str(x) for x in app]
    a.geo_corr_params = [str(x) for x in geo]
    a.patch_stride = patch_stride
    return a


class InjectRand:
    """Replace torch.rand by a queue of prepared tensors (the reference draws coords1 then coords2)."""

    def __init__(self, *tensors):
        self.q = list(tensors)

    def __enter__(self):
        self._rand = torch.rand
        torch.rand = lambda *a, **k: self.q.pop(0)
        return self

    def __exit__(self, *exc):
        torch.rand = self._rand


def main():
    out = {}
    g = torch.Generator().manual_seed(777)
    APP, GEO = (0.18, 1, 0.46, 1), (0.5, 1, 3, 1)       # scripts/train_fortress_node0.sh
    args = ref_args(APP, GEO)

    # ---------------------------------------------------------------- evaluation post-processing
    sem = torch.randn(9, 13, 2, generator=g) * 3
    sem[0, 0] = torch.tensor([0.25, 0.25])               # tie -> argmax returns the first index
    sem[0, 1] = torch.tensor([-40.0, 45.0])              # saturated softmax
    rgb = torch.rand(9, 13, 3, generator=g)
    tgt = torch.rand(9, 13, 3, generator=g)
    sem_prob = sem.detach().cpu().float().softmax(dim=-1)                       # engines/eval.py:55
    sem_pred = torch.argmax(sem_prob, -1).unsqueeze(-1).numpy().astype(np.int32)  # :56,:60
    mse = ref_image.img2mse(rgb, tgt)
    psnr = ref_image.mse2psnr(mse)
    mine = lp.eval_postprocess(sem, rgb, tgt)
    assert torch.equal(mine["sem_prob"], sem_prob) and np.array_equal(mine["sem"].numpy(), sem_pred)
    assert torch.equal(mine["mse"], mse.reshape(1)) and torch.equal(mine["psnr"], psnr.reshape(1))
    out.update(post_sem=np32(sem), post_rgb=np32(rgb), post_tgt=np32(tgt), post_prob=np32(sem_prob), post_pred=sem_pred,
               post_mse=np32(mse.reshape(1)), post_psnr=np32(psnr.reshape(1)))
    sem5 = torch.randn(257, 5, generator=g)              # wider head (sem_dim is a constructor argument)
    out.update(post5_sem=np32(sem5), post5_prob=np32(sem5.softmax(-1)),
               post5_pred=torch.argmax(sem5.softmax(-1), -1).unsqueeze(-1).numpy().astype(np.int32))

    # ---------------------------------------------------------------- CorrelationLoss (appearance)
    for tag, B, Cf, Hf, P in (("app_small", 3, 16, 5, 12), ("app_full", 8, 384, 14, 64)):
        feats = torch.randn(B, Cf, Hf, Hf, generator=g)
        code = (torch.randn(B, 2, P, P, generator=g) * 2).requires_grad_(True)
        sim = torch.rand(B, B, generator=g)
        c1 = torch.rand(B, 11, 11, 2, generator=g)
        c2 = torch.rand(B, 11, 11, 2, generator=g)
        if tag == "app_small":
            c1[0, 0, 0] = torch.tensor([0.0, 1.0])       # exactly on the border after *2-1
            code.data[1, :, 3, 4] = 0.0                  # zero vector through F.normalize(eps=1e-10)
        mod = ref_image.CorrelationLoss(args)
        with InjectRand(c1.clone(), c2.clone()):
            loss = mod(feats, code, sim)
        loss.backward()
        gref = code.grad.clone()
        code2 = code.detach().clone().requires_grad_(True)
        p = lp.CorrParams(*APP)
        mine = lp.correlation_loss(feats, code2, lp.neg_index(sim), c1 * 2 - 1, c2 * 2 - 1, p)
        mine.backward()
        assert torch.equal(mine, loss), (tag, float(mine), float(loss))
        # grid_sample's CPU backward accumulates across threads in a run-dependent order at this size
        gerr = float((code2.grad - gref).abs().max() / gref.abs().max())
        assert gerr < 1e-6, (tag, gerr)
        out.update({f"{tag}_feats": np32(feats), f"{tag}_code": np32(code), f"{tag}_sim": np32(sim), f"{tag}_rand1": np32(c1),
                    f"{tag}_rand2": np32(c2), f"{tag}_loss": np32(loss.reshape(1)), f"{tag}_grad": np32(gref)})
        print(tag, float(loss.detach()), float(gref.abs().max()), 'grad port-vs-ref rel err', gerr)

    # ---------------------------------------------------------------- GeoCorrelationLoss
    for tag, B, P in (("geo_small", 3, 16), ("geo_full", 8, 64)):
        depth = 2.0 + 9.0 * torch.rand(B, 1, P, P, generator=g)
        depth[0, 0, :2, :3] = 1e10                        # empty rays: depth = 1e10 (models/renderer.py:72)
        depth[B - 1, 0, 5, 5] = 15.0                      # exactly max_depth: untouched by the filter
        code = (torch.randn(B, 2, P, P, generator=g) * 2).requires_grad_(True)
        ray_o = torch.randn(B, 3, 1, 1, generator=g).expand(B, 3, P, P).contiguous() * 0.3
        ray_d = torch.randn(B, 3, P, P, generator=g) * 0.2
        ray_d[:, 2] -= 1.0
        sim = torch.rand(B, B, generator=g)
        mod = ref_image.GeoCorrelationLoss(args)
        d_ref = depth.clone()
        loss = mod(d_ref, code, [ray_o, ray_d, None], sim)
        loss.backward()
        gref = code.grad.clone()
        code2 = code.detach().clone().requires_grad_(True)
        p = lp.CorrParams(*GEO)
        d_mine = depth.clone()
        mine = lp.geo_correlation_loss(d_mine, code2, ray_o, ray_d, lp.neg_index(sim), p)
        mine.backward()
        assert torch.equal(mine, loss), (tag, float(mine), float(loss))
        gerr = float((code2.grad - gref).abs().max() / gref.abs().max())
        assert gerr < 1e-6 and torch.equal(d_mine, d_ref), (tag, gerr)
        out.update({f"{tag}_depth": np32(depth), f"{tag}_code": np32(code), f"{tag}_ray_o": np32(ray_o[:, :, 0, 0]),
                    f"{tag}_ray_d": np32(ray_d), f"{tag}_sim": np32(sim), f"{tag}_loss": np32(loss.reshape(1)),
                    f"{tag}_grad": np32(gref), f"{tag}_depth_after": np32(d_ref)})
        print(tag, float(loss.detach()), float(gref.abs().max()), 'grad port-vs-ref rel err', gerr)

    path = os.path.join(HERE, "losses.npz")
    np.savez_compressed(path, **out)
    print(f"wrote {path}: {os.path.getsize(path) / 1e6:.2f} MB")


if __name__ == "__main__":
    main()
