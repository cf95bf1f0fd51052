"""GPU: backward kernels against torch autograd through the CPU port (oracle/torch_port.py, op-for-op the reference)."""
import numpy as np
import pytest
import torch

import nerf_sos_amd
from nerf_sos_amd import ops
from oracle import torch_port as tp

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


@pytest.mark.parametrize("S,C,white,noisy", [(64, 6, False, False), (192, 6, True, True), (64, 4, False, True), (50, 5, True, False),
                                             (300, 6, False, False)])
def test_composite_backward_vs_autograd(S, C, white, noisy):
    g = torch.Generator().manual_seed(S * 10 + C)
    R = 37
    raw = torch.randn(R, S, C, generator=g)
    raw[..., 3] = raw[..., 3] * 3 + 0.5
    raw[0, :, 3] = -1.0                      # empty ray: acc = 0, depth -> 1e10 (no gradient through depth / disp)
    raw[1, 3, 3] = 80.0                      # opaque sample
    z = torch.sort(1.2 + 13 * torch.rand(R, S, generator=g), -1)[0]
    d = torch.randn(R, 3, generator=g)
    noise = torch.randn(R, S, generator=g) * 0.7 if noisy else None
    cfg = tp.PortConfig(use_semantics=C > 4, white_bkgd=white)
    ups = dict(rgb=torch.randn(R, 3, generator=g), depth=torch.randn(R, 1, generator=g), acc=torch.randn(R, 1, generator=g),
               disp=torch.randn(R, 1, generator=g) * 0.1, weights=torch.randn(R, S, generator=g))
    if C > 4:
        ups["semantics"] = torch.randn(R, C - 4, generator=g)
    ups["disp"][0] = 0.0                     # the reference's autograd gives NaN for an empty ray with g_disp != 0
    rd = raw.double().requires_grad_(True)
    ret = tp.composite(rd, z.double(), d.double(), None if noise is None else noise.double(), cfg)
    loss = sum((ret[k] * ups[k].double()).sum() for k in ups)
    loss.backward()
    want = rd.grad.float()
    T = lambda t: None if t is None else t.to(DEV)  # noqa: E731
    got = ops.composite_backward(T(raw), T(z), T(d), T(noise), 1.0 if noisy else 0.0, white, g_rgb=T(ups["rgb"]),
                                 g_sem=T(ups.get("semantics")), g_depth=T(ups["depth"]), g_acc=T(ups["acc"]),
                                 g_disp=T(ups["disp"]), g_weights=T(ups["weights"])).cpu()
    assert torch.isfinite(got).all()
    scale = want.abs().max()
    err = (got - want).abs().max()
    assert err <= 1e-4 * scale, f"max err {err:.3e} (scale {scale:.3e})"
    # partial upstream sets (NULL pointers)
    rd2 = raw.double().requires_grad_(True)
    ret2 = tp.composite(rd2, z.double(), d.double(), None if noise is None else noise.double(), cfg)
    (ret2["rgb"] * ups["rgb"].double()).sum().backward()
    got2 = ops.composite_backward(T(raw), T(z), T(d), T(noise), 1.0 if noisy else 0.0, white, g_rgb=T(ups["rgb"])).cpu()
    assert (got2 - rd2.grad.float()).abs().max() <= 1e-4 * rd2.grad.abs().max()
