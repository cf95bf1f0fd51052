"""GPU (-m gpu): stage-wise pins that close the parity loopholes of round 1 ("What's weak" 1-3).

  * the device positional encoder against the reference's PositionEncoder goldens, directly (models/embedder.py:34-48);
  * the fine network + compositing on the REFERENCE's own fine-pass sample positions: every output key strictly
    within 1e-4, no percentage allowances (models/nerf_net.py:107-121);
  * the fine network's gradients on the reference's sample positions: 1e-4 of scale (was 6e-2 free-running);
  * the free-running hierarchical sampler at C2 size: the measured share of rays touched by a last-ulp index flip
     is asserted, not a loose bound.
"""
import json
import os

import numpy as np
import pytest
import torch

import nerf_sos_amd
from nerf_sos_amd import ops
from oracle import torch_port as tp
from helpers import CFGS, close, ref_state, tag_of

pytestmark = pytest.mark.gpu
DEV = "cuda:0"
HERE = os.path.dirname(os.path.abspath(__file__))


def T(a):
    return torch.as_tensor(np.ascontiguousarray(a)).to(DEV)


def N(t):
    return t.detach().cpu().numpy()


# ------------------------------------------------------------------------------------------------ a4: the encoder itself
@pytest.mark.parametrize("precision", ["fp32", "fp16x3"])
def test_device_positional_encoding_vs_reference(golden, manifest, precision):
    """`posenc.npz`: 257 points in [-15, 15]^3 (arguments up to 2^9 * 15 = 7680 rad -- the regime the hand-written
    Cody-Waite reduction exists for) and 257 unit directions, encoded by the reference's PositionEncoder.  The SAVE=2
    variant of the fused kernel stores the encodings it feeds to the MLP (acts[:, ACTS_X:], acts[:, ACTS_D:]): one ray
    per point with origin = the point and z = 0, so o + d*0 is the point exactly."""
    g = golden("posenc")
    x, v = g["x"], g["v"]
    n = x.shape[0]
    net = nerf_sos_amd.NeRFNet(N_samples=64, N_importance=0, **CFGS["semcoord"]).to(DEV)
    net.load_state_dict(ref_state("semcoord", manifest, False, 0))
    d = np.tile(np.array([[0.3, -0.2, -1.0]], np.float32), (n, 1))
    z = torch.zeros((n, 1), device=DEV)
    raw, acts, _ = ops.mlp_forward_rays_save_all(net.nerf.packed_weights(precision), net.nerf.sem_mode, T(x), T(d), T(v), z, precision)
    e10 = N(acts[:, ops.ACTS_X:ops.ACTS_X + 63])
    e4 = N(acts[:, ops.ACTS_D:ops.ACTS_D + 27])
    assert np.array_equal(e10[:, :3], x) and np.array_equal(e4[:, :3], v), "include_input: the raw coordinates come first"
    tol = 2e-7 if precision == "fp32" else 1e-6          # split-fp16 keeps hi + lo (22 mantissa bits)
    err10, err4 = np.abs(e10.astype(np.float64) - g["e10"]).max(), np.abs(e4.astype(np.float64) - g["e4"]).max()
    assert err10 <= tol, f"xyz encoding: max abs err {err10:.3e} (bar {tol:.0e})"
    assert err4 <= tol, f"direction encoding: max abs err {err4:.3e} (bar {tol:.0e})"
    assert np.array_equal(N(acts[:, ops.ACTS_X + 63]), np.ones(n, np.float32))


# ------------------------------------------------------------------- fine pass on the reference's own sample positions
PINNED = [("nosem", False, False), ("semcoord", False, False), ("semcoord", True, False), ("sem", True, True)]


@pytest.mark.parametrize("name,peaky,white", PINNED)
@pytest.mark.parametrize("mode", ["eval", "train"])
@pytest.mark.parametrize("precision", ["fp32", "fp16x3"])
def test_fine_pass_on_reference_z_fine_strict(golden, manifest, name, peaky, white, mode, precision):
    """Fine MLP + compositing on the z_fine the REAL reference produced for the committed end-to-end cases
    (tests/golden/make_goldens_zfine.py): rgb, disp, acc, depth, semantics, weights and raw all within 1e-4 -- every
    element, no allowances.  Train mode uses the reference's captured sigma-noise draw."""
    g, zf = golden("end_to_end"), golden("zfine")
    tag = tag_of(name, peaky, white) + "_" + mode
    net = nerf_sos_amd.NeRFNet(N_samples=64, N_importance=128, white_bkgd=white, **CFGS[name]).to(DEV)
    net.load_state_dict(ref_state(name, manifest, peaky, 128))
    rays = T(g["rays"])
    o, d = rays[0].contiguous(), rays[1].contiguous()
    z_fine = T(zf[f"{tag}_z_fine"])
    R = d.shape[0]
    near = torch.full((R,), tp.NEAR, device=DEV)
    far = torch.full((R,), tp.FAR, device=DEV)
    _, viewdirs = ops.ray_setup(d, near, far, 64, None)
    mlp = net.nerf_fine
    if precision == "fp32":
        raw = ops.mlp_forward_rays(mlp.packed_weights(), mlp.sem_mode, o, d, viewdirs, z_fine)
    else:
        raw = ops.mlp_forward_rays_lp(mlp.packed_weights(precision), mlp.sem_mode, precision, o, d, viewdirs, z_fine)
    noise = T(g[f"{tag}_draw3"]) if mode == "train" else None
    ret = ops.composite(raw, z_fine, d, noise, 1.0 if mode == "train" else 0.0, white)
    ret["raw"] = raw
    for k, got in ret.items():
        close(N(got), g[f"{tag}_{k}"], atol=1e-4, rtol=1e-4, what=f"{tag} {k} on the reference's z_fine ({precision})")


@pytest.mark.parametrize("name,peaky,white", PINNED[1:])
@pytest.mark.parametrize("mode", ["eval", "train"])
def test_module_with_pinned_z_fine_equals_reference_everywhere(golden, manifest, monkeypatch, name, peaky, white, mode):
    """The whole module (NeRFNet.forward) with only the fine sample positions pinned to the reference's: every key of
    the output dict -- coarse and fine -- strictly within 1e-4 (z_std comes from the sampler: 1e-4 as well, it is a
    mean over 128 samples and a flipped index moves it by < 1e-5)."""
    g, zf = golden("end_to_end"), golden("zfine")
    tag = tag_of(name, peaky, white) + "_" + mode
    net = nerf_sos_amd.NeRFNet(N_samples=64, N_importance=128, perturb=1.0, raw_noise_std=1.0, white_bkgd=white, **CFGS[name]).to(DEV)
    net.load_state_dict(ref_state(name, manifest, peaky, 128))
    net.train(mode == "train")
    if mode == "train":
        from test_gpu_parity import _Draws
        dr = [torch.as_tensor(g[f"{tag}_draw{i}"]) for i in range(4)]
        monkeypatch.setattr(torch, "rand", _Draws([dr[0], dr[2]]))
        monkeypatch.setattr(torch, "randn", _Draws([dr[1], dr[3]]))
    with torch.no_grad():
        out = net(T(g["rays"]), (tp.NEAR, tp.FAR), radii=None, z_fine_override=T(zf[f"{tag}_z_fine"]))
    for k, got in out.items():
        close(N(got), g[f"{tag}_{k}"], atol=1e-4, rtol=1e-4, what=f"{tag} {k}")


# --------------------------------------------------------------- fine-network gradients on the reference's positions
@pytest.mark.parametrize("name,peaky,white", [("semcoord", True, False), ("sem", False, True)])
@pytest.mark.parametrize("precision", ["fp32", "fp16x3"])
def test_full_backward_fine_net_on_reference_z_fine(golden, manifest, name, peaky, white, precision):
    """`full_grads.npz` (the real reference's autograd, every parameter trainable) with the fine sample positions pinned
    to the reference's: the FINE network's gradients now meet the same 1e-4 of scale as the coarse network's
    (free-running they sat at a few % because a flipped bisect index moves a sample)."""
    FULL, zf = golden("full_grads"), golden("zfine")
    tag = tag_of(name, peaky, white)
    net = nerf_sos_amd.NeRFNet(N_samples=64, N_importance=128, white_bkgd=white, **CFGS[name]).to(DEV)
    net.load_state_dict(ref_state(name, manifest, peaky, 128))
    net.eval()
    net.mlp_precision = precision
    rays = T(FULL["rays"])
    ret = net(rays, (tp.NEAR, tp.FAR), radii=None, z_fine_override=T(zf[f"fullgrad_{tag}_z_fine"]))
    loss = 0.0
    for k in ret:
        gk = f"{tag}_G_{k}"
        if gk in FULL:
            loss = loss + (ret[k] * T(FULL[gk])).sum()
    want_loss = float(FULL[f"{tag}_loss"][0])
    assert abs(float(loss) - want_loss) <= 1e-4 * (1 + abs(want_loss)) * 10, (float(loss), want_loss)
    loss.backward()
    worst, n_checked = {}, 0
    for n_, p_ in net.named_parameters():
        got = p_.grad
        assert got is not None, n_
        refs = []
        if f"{tag}_grad_{n_}" in FULL:
            refs.append((got, FULL[f"{tag}_grad_{n_}"]))
        elif f"{tag}_gradrows_{n_}" in FULL:
            refs.append((got[::max(1, got.shape[0] // 24)], FULL[f"{tag}_gradrows_{n_}"]))
            refs.append((got[:, ::max(1, got.shape[1] // 24)], FULL[f"{tag}_gradcols_{n_}"]))
        for a, b in refs:
            b = torch.from_numpy(b)
            scale = float(b.abs().max()) + 1e-20
            worst[n_] = max(worst.get(n_, 0.0), float((a.detach().cpu() - b).abs().max()) / scale)
        n_checked += bool(refs)
    assert n_checked >= 48, n_checked
    bad = {k: v for k, v in worst.items() if v > 1e-4}
    assert not bad, f"gradients off by more than 1e-4 of their scale: {bad}"


# --------------------------------------------------------------------------- free-running sampler: measured flip rate
def test_free_running_flip_rate_at_c2_size(manifest):
    """BASELINE C2 size, free-running (nothing pinned), spiky density: the HIP render against the CPU port of the
    reference (bit-identical to the reference on CPU) on 4096 rays.  The coarse pass is strictly within 1e-4; in the
    fine pass a ray counts as 'flipped' when any of rgb / depth / acc / semantics leaves the 1e-4 band.  Measured in
    round 1: ~0.1 % of rays (profiles/r01/quality_report_2048rays.json); asserted here: <= 0.3 %."""
    cfg = tp.PortConfig(n_importance=128, **CFGS["semcoord"])
    sd = ref_state("semcoord", manifest, peaky=True)
    rays = tp.synthetic_rays(4096, seed=0)
    torch.set_num_threads(min(32, os.cpu_count() or 1))
    with torch.no_grad():
        ref = tp.render(sd, cfg, rays, (tp.NEAR, tp.FAR))
    net = nerf_sos_amd.NeRFNet(N_samples=64, N_importance=128, **CFGS["semcoord"]).to(DEV).eval()
    net.load_state_dict(sd)
    with torch.no_grad():
        out = net(rays.to(DEV), (tp.NEAR, tp.FAR))
    for k in ("rgb0", "depth0", "acc0", "disp0", "semantics0", "weights0", "raw0"):
        close(N(out[k]), N(ref[k]), what=f"coarse {k} at C2 size")
    flipped = np.zeros(4096, bool)
    for k in ("rgb", "depth", "acc", "semantics"):
        a, b = N(out[k]).astype(np.float64), N(ref[k]).astype(np.float64)
        flipped |= (np.abs(a - b) > 1e-4 + 1e-4 * np.abs(b)).reshape(4096, -1).any(-1)
    rate = flipped.mean()
    print(f"flip rate at C2 size: {flipped.sum()} of 4096 rays = {100 * rate:.3f} %")
    assert rate <= 3e-3, f"{100 * rate:.3f} % of rays outside 1e-4 (expected ~0.1 %: last-ulp bisect flips only)"
    mse = float(((N(out['rgb']) - N(ref['rgb'])) ** 2).mean())
    assert 10 * np.log10(1.0 / max(mse, 1e-30)) > 80.0, "PSNR of rgb vs the reference path"
