"""The point-query entry and MLP.forward under autograd (the reference's `model.nerf_fine(pts, viewdirs)` and `MLP.forward(x)` are
ordinary differentiable modules, models/nerf_mlp.py:179-215 and 67-100): on the generic fp32 kernels, for the shipped architecture and
generic ones -- values, parameter gradients and input gradients against autograd through the CPU port (bit-identical to the reference's
forward on CPU, the same ATen backward formulas)."""
import numpy as np
import pytest
import torch

import nerf_sos_amd
from oracle import torch_port as tp
from helpers import CFGS, GENERIC_CASES, generic_state, ref_state

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def _case(tag, golden, manifest):
    if tag in GENERIC_CASES:
        cfg, sd = generic_state(tag, golden)
        net = nerf_sos_amd.NeRFNet(**GENERIC_CASES[tag][0])
    else:
        cfg = tp.PortConfig(n_importance=128, **CFGS[tag])
        sd = ref_state(tag, manifest, peaky=True)
        net = nerf_sos_amd.NeRFNet(N_samples=64, N_importance=128, **CFGS[tag])
    net = net.to(DEV).eval()
    net.load_state_dict(sd)
    return cfg, sd, net


def _rel(a, b):
    return float((a - b).abs().max() / (b.abs().max() + 1e-20))


@pytest.mark.parametrize("tag", ["semcoord", "d6w96_m6", "noview", "deepsem3_geo"])
def test_point_query_autograd_vs_port(golden, manifest, tag):
    """raw = nerf_fine(pts, viewdirs) with every parameter, the points and the directions requiring grad, 77 points (ragged tiles)."""
    cfg, sd, net = _case(tag, golden, manifest)
    g = torch.Generator().manual_seed(3)
    pts = (torch.rand(7, 11, 3, generator=g) * 4 - 2).requires_grad_(True)
    dirs = torch.nn.functional.normalize(torch.randn(7, 11, 3, generator=g), dim=-1).requires_grad_(True) if cfg.use_viewdirs else None
    sdg = {k: v.clone().requires_grad_(True) for k, v in sd.items()}
    ref = tp.point_query(sdg, "nerf_fine", pts, dirs, cfg)
    up = torch.randn(ref.shape, generator=g)
    (ref * up).sum().backward()

    p_gpu = pts.detach().to(DEV).requires_grad_(True)
    d_gpu = dirs.detach().to(DEV).requires_grad_(True) if dirs is not None else None
    out = net.nerf_fine(p_gpu, viewdirs=d_gpu)
    assert out.shape == ref.shape and out.requires_grad
    assert float(((out.detach().cpu() - ref.detach()).abs() / (1 + ref.detach().abs())).max()) <= 1e-4
    with torch.no_grad():
        assert float(((net.nerf_fine(p_gpu, viewdirs=d_gpu) - out.detach()).abs() / (1 + out.detach().abs())).max()) <= 2e-5   # (tuned kernel vs generic)
    (out * up.to(DEV)).sum().backward()
    bad = {}
    for n_, p_ in net.nerf_fine.named_parameters():
        e = _rel(p_.grad.cpu(), sdg["nerf_fine." + n_].grad)
        if e > 1e-4:
            bad[n_] = e
    assert not bad, bad
    assert all(p_.grad is None for p_ in net.nerf.parameters()) or net.nerf is net.nerf_fine
    assert _rel(p_gpu.grad.cpu(), pts.grad) <= 2e-4
    if dirs is not None:
        assert _rel(d_gpu.grad.cpu(), dirs.grad) <= 2e-4


@pytest.mark.parametrize("tag", ["semcoord", "d4w128"])
def test_mlp_forward_on_encoded_inputs_vs_port(golden, manifest, tag):
    """MLP.forward(x) on PRE-ENCODED rows (the reference's own signature): values, parameter gradients and d / d x against the port's
    mlp_forward on the same encoded rows (the encodings are inputs here: nothing of the kernel's sin / cos is involved)."""
    cfg, sd, net = _case(tag, golden, manifest)
    g = torch.Generator().manual_seed(4)
    pts = torch.rand(45, 3, generator=g) * 4 - 2
    dirs = torch.nn.functional.normalize(torch.randn(45, 3, generator=g), dim=-1)
    x = torch.cat([tp.posenc(pts, cfg.multires), tp.posenc(dirs, cfg.multires_views)], -1).requires_grad_(True)
    sdg = {k: v.clone().requires_grad_(True) for k, v in sd.items()}
    ref = tp.mlp_forward(sdg, "nerf", x, cfg)
    up = torch.randn(ref.shape, generator=g)
    (ref * up).sum().backward()
    mlp = net.nerf.mlp
    xg = x.detach().to(DEV).requires_grad_(True)
    with torch.no_grad():
        plain = mlp(xg.reshape(5, 9, -1))
    out = mlp(xg.reshape(5, 9, -1))
    assert tuple(out.shape) == (5, 9, ref.shape[-1]) and torch.equal(out.detach(), plain)
    assert float(((out.detach().cpu().reshape(ref.shape) - ref.detach()).abs() / (1 + ref.detach().abs())).max()) <= 2e-5
    (out.reshape(ref.shape) * up.to(DEV)).sum().backward()
    bad = {n_: _rel(p_.grad.cpu(), sdg["nerf.mlp." + n_].grad) for n_, p_ in mlp.named_parameters()}
    assert max(bad.values()) <= 1e-4, {k: v for k, v in bad.items() if v > 1e-4}
    assert _rel(xg.grad.cpu(), x.grad) <= 1e-4
    with pytest.raises(ValueError):
        mlp(xg[:, :-1])


def test_point_query_gradients_refuse_16_bit(golden, manifest):
    cfg, sd, net = _case("semcoord", golden, manifest)
    net.mlp_precision = "bf16"
    pts = torch.rand(8, 3, device=DEV)
    with pytest.raises(NotImplementedError):
        net.nerf_fine(pts, viewdirs=pts)
    with torch.no_grad():
        assert torch.isfinite(net.nerf_fine(pts, viewdirs=pts)).all()
