"""CPU: the torch restatement of the loss / post-processing rows (oracle/losses_port.py) against the goldens captured
from the real reference classes.  The O(P^4) full-size geometric case is left to the GPU test (the port needs
several GB and ~20 s for it); the small case exercises the same code."""
import os

import numpy as np
import pytest
import torch

from oracle import losses_port as lp

GOLD = np.load(os.path.join(os.path.dirname(__file__), "golden", "losses.npz"))
APP, GEO = (0.18, 1, 0.46, 1), (0.5, 1, 3, 1)
t = lambda k: torch.from_numpy(GOLD[k])  # noqa: E731


def test_eval_postprocess_port():
    out = lp.eval_postprocess(t("post_sem"), t("post_rgb"), t("post_tgt"))
    assert np.array_equal(out["sem"].numpy(), GOLD["post_pred"]) and out["sem"].dtype == torch.int32
    assert np.array_equal(out["sem_prob"].numpy(), GOLD["post_prob"])
    assert np.array_equal(out["mse"].numpy(), GOLD["post_mse"]) and np.array_equal(out["psnr"].numpy(), GOLD["post_psnr"])
    assert out["sem"][0, 0, 0] == 0   # tie -> first index


@pytest.mark.parametrize("tag", ["app_small", "app_full"])
def test_correlation_loss_port(tag):
    code = t(f"{tag}_code").clone().requires_grad_(True)
    loss = lp.correlation_loss(t(f"{tag}_feats"), code, lp.neg_index(t(f"{tag}_sim")), t(f"{tag}_rand1") * 2 - 1,
                               t(f"{tag}_rand2") * 2 - 1, lp.CorrParams(*APP))
    loss.backward()
    assert abs(loss.item() - GOLD[f"{tag}_loss"][0]) <= 1e-6 * abs(GOLD[f"{tag}_loss"][0])
    g = GOLD[f"{tag}_grad"]
    assert np.abs(code.grad.numpy() - g).max() <= 1e-6 * np.abs(g).max()


def test_geo_correlation_loss_port():
    tag = "geo_small"
    depth = t(f"{tag}_depth").clone()
    B, _, P, _ = depth.shape
    code = t(f"{tag}_code").clone().requires_grad_(True)
    ray_o = t(f"{tag}_ray_o")[:, :, None, None].expand(B, 3, P, P)
    loss = lp.geo_correlation_loss(depth, code, ray_o, t(f"{tag}_ray_d"), lp.neg_index(t(f"{tag}_sim")), lp.CorrParams(*GEO))
    loss.backward()
    assert abs(loss.item() - GOLD[f"{tag}_loss"][0]) <= 1e-6 * abs(GOLD[f"{tag}_loss"][0])
    assert np.array_equal(depth.numpy(), GOLD[f"{tag}_depth_after"])
    g = GOLD[f"{tag}_grad"]
    assert np.abs(code.grad.numpy() - g).max() <= 1e-6 * np.abs(g).max()


CON = np.load(os.path.join(os.path.dirname(__file__), "golden", "contrastive.npz"))
CON_CASES = sorted(k[:-4] for k in CON.files if k.endswith("_emb"))


@pytest.mark.parametrize("tag", CON_CASES)
def test_contrastive_loss_port(tag):
    """oracle/losses_port.nerf_contrastive vs the real NeRFContrastive (utils/image.py:192-218): bit-identical on CPU
    (goldens from tests/golden/make_goldens_contrastive.py); the b5_d7 case is the reference's own NaN (max + min < 0)."""
    e = torch.from_numpy(CON[f"{tag}_emb"]).clone().requires_grad_(True)
    loss = lp.nerf_contrastive(e)
    loss.backward()
    assert np.array_equal(loss.detach().reshape(1).numpy(), CON[f"{tag}_loss"], equal_nan=True)
    assert np.array_equal(e.grad.numpy(), CON[f"{tag}_grad"], equal_nan=True)
