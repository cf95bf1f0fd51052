"""CPU: the trained-field fixture.  tests/golden/trained_scene.ckpt is a reference-format checkpoint of the shipped
architecture trained on the procedural scene (scripts/make_trained_scene.py, on an MI355X, this package's own training path);
tests/golden/trained.npz holds what the REAL reference renders from it (make_goldens_trained.py: 256 rays of the held-out views, eval
and train mode, its z_fine, its draws).  Here: the fixture is self-consistent, the checkpoint loads strict=True into the product's
module, the torch port reproduces the reference bit for bit on the trained weights too, and the C oracle meets its bars."""
import hashlib
import os

import numpy as np
import pytest
import torch

import nerf_sos_amd
from oracle import c_oracle as co
from oracle import torch_port as tp
from helpers import close, state_sha

HERE = os.path.dirname(os.path.abspath(__file__))
CKPT = os.path.join(HERE, "golden", "trained_scene.ckpt")
KW = dict(use_semantics=True, sem_with_coord=True)


@pytest.fixture(scope="module")
def trained(golden):
    g = golden("trained")
    assert hashlib.sha256(open(CKPT, "rb").read()).digest() == bytes(g["ckpt_sha256"]), "trained.npz was generated from another checkpoint"
    ck = torch.load(CKPT, map_location="cpu")
    assert set(ck) == {"global_step", "model", "optimizer"}                       # engines/trainer.py:216-222
    assert state_sha(ck["model"]) == bytes(g["state_sha256"]).hex()
    return g, ck["model"]


def test_checkpoint_loads_strict_and_is_a_trained_field(trained):
    g, sd = trained
    net = nerf_sos_amd.NeRFNet(N_samples=64, N_importance=128, **KW)
    net.load_state_dict(sd, strict=True)                                           # run_nerf.py:349-353
    # a trained field, not an initialisation: opaque surfaces (acc = 1), colours that match the analytic scene, a sharp density
    assert float(g["eval_acc"].min()) > 0.999
    psnr = -10 * np.log10(np.mean((g["eval_rgb"] - g["gt_rgb"]) ** 2))
    assert psnr > 35.0, psnr
    sig0, sig = np.maximum(g["eval_raw0"][..., 3], 0), np.maximum(g["eval_raw"][..., 3], 0)
    assert sig.max() > 100 and (sig0 == 0).mean() > 0.25, (sig.max(), (sig0 == 0).mean())   # solid surfaces behind empty space
    init = tp.init_state_dict(tp.PortConfig(n_importance=128, **KW), seed=0)
    assert float((sd["nerf_fine.mlp.pts_linears.3.weight"] - init["nerf_fine.mlp.pts_linears.3.weight"]).abs().max()) > 0.05
    assert not torch.equal(sd["nerf_fine.mlp.semantic_linear.2.weight"], init["nerf_fine.mlp.semantic_linear.2.weight"])


@pytest.mark.parametrize("mode", ["eval", "train"])
def test_port_equals_reference_on_trained_weights(trained, mode):
    g, sd = trained
    cfg = tp.PortConfig(n_importance=128, **KW)
    rays = torch.from_numpy(g["rays"])
    near, far = (float(v) for v in g["near_far"])
    with torch.no_grad():
        if mode == "eval":
            out = tp.render(sd, cfg, rays, (near, far))
        else:
            dr = [torch.from_numpy(g[f"train_draw{i}"]) for i in range(4)]
            out = tp.render(sd, cfg, rays, (near, far), raw_noise_std=1.0, draws_per_chunk=[tp.Draws(*dr)])
    for k, v in out.items():
        assert np.array_equal(v.numpy(), g[f"{mode}_{k}"]), f"port != reference on the trained field: {mode} {k}"


@pytest.mark.parametrize("mode", ["eval", "train"])
def test_c_oracle_on_trained_weights(trained, mode):
    """The C restatement against the reference's render of the trained field: coarse pass strictly within 1e-4; fine maps with at most
    one ray of 256 outside (a last-ulp cdf difference may flip a bisect index); per-sample fine tensors in bulk."""
    g, sd = trained
    rays = g["rays"]
    near, far = (float(v) for v in g["near_far"])
    kw = dict(n_importance=128, **KW)
    if mode == "train":
        dr = [g[f"train_draw{i}"] for i in range(4)]
        kw.update(raw_noise_std=1.0, t_rand=dr[0], noise0=dr[1], u=dr[2], noise1=dr[3])
    out = co.render(sd, rays[0], rays[1], near, far, **kw)
    for k in ("rgb0", "depth0", "acc0", "disp0", "semantics0", "weights0", "raw0"):
        close(out[k].reshape(g[f"{mode}_{k}"].shape), g[f"{mode}_{k}"], what=f"trained {mode} {k}")
    for k in ("rgb", "depth", "acc", "semantics"):
        want = g[f"{mode}_{k}"]
        bad = (np.abs(out[k].reshape(want.shape).astype(np.float64) - want) > 1e-4 * (1 + np.abs(want))).any(-1)
        assert bad.sum() <= 1, f"trained {mode} {k}: {bad.sum()} rays outside 1e-4"
    for k in ("weights", "raw"):
        want = g[f"{mode}_{k}"]
        err = np.abs(out[k].reshape(want.shape).astype(np.float64) - want)
        assert (err > 1e-4 + 1e-4 * np.abs(want)).mean() < 5e-3
