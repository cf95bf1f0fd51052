"""Round-5 mechanisms of the generic kernels, each held to an independent statement of what it must equal (GPU, through the C ABI):

  * nsos_wgrad_batch == the nsos_wgrad calls it replaces, bit for bit (same kernels, same order);
  * the ReLU bit words a SAVE launch stores behind the blocks == [saved block > 0], bit by bit, for every ReLU Linear of a net
    (four-wave and eight-wave workgroups, 32- and 16-point tiles);
  * a backward cut to a trainable subset that ALSO reaches the rays (bit 31 of the mask; pose refinement against a frozen net: no
    activation block saved, nothing written to gbuf) gives the rays the SAME gradient, bit for bit, as the all-trainable backward;
  * K-split heads: the forward of a net whose heads are split over the waves equals the port within the usual bar (covered by
    test_generic_arch.py for values) and is deterministic run to run.
"""
import ctypes as C

import numpy as np
import pytest
import torch

import nerf_sos_amd
from nerf_sos_amd import _lib, ops
from nerf_sos_amd import synthetic as syn

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def test_wgrad_batch_equals_single_calls():
    g = torch.Generator(device=DEV).manual_seed(5)
    P_, ldg, ldx = 3001, 96 + 64, 256 + 32                    # ragged point count; column blocks inside wider rows
    G = torch.randn((P_, ldg), device=DEV, generator=g)
    X = torch.randn((P_, ldx), device=DEV, generator=g)
    # (w_off, b_off, g_col, x_col, M, N, ldw): two items into one [96, 288] block (ldw 288) with a bias, one [64, 32] block without
    items = [(0, 96 * 288, 0, 0, 64, 256, 288), (64 * 288, 96 * 288 + 64, 64, 0, 32, 256, 288), (256, -1, 0, 256, 64, 32, 288),
             (64 * 288 + 256, -1, 64, 256, 32, 32, 288), (96 * 288 + 96, -1, 96, 256, 64, 32, 32)]
    arr = (_lib.WgradItem * len(items))()
    for i, (w, b, gc, xc, m, n, ldw) in enumerate(items):
        arr[i].w_off, arr[i].b_off, arr[i].g_col, arr[i].x_col, arr[i].M, arr[i].N, arr[i].ldw = w, b, gc, xc, m, n, ldw
    total = 96 * 288 + 96 + 64 * 32
    flat = torch.full((total,), float("nan"), device=DEV)
    ops.wgrad_batch(arr, len(items), G, X, flat)
    want = torch.full((total,), float("nan"), device=DEV)
    for w, b, gc, xc, m, n, ldw in items:
        dW = want[w: w + (m - 1) * ldw + n].as_strided((m, n), (ldw, 1))
        db = want[b: b + m] if b >= 0 else None
        ops.wgrad(G[:, gc: gc + m], X[:, xc: xc + n], dW, db)
    torch.cuda.synchronize()
    a, b_ = flat.cpu().numpy(), want.cpu().numpy()
    assert np.array_equal(np.isnan(a), np.isnan(b_)), "the batch wrote other elements than the single calls"
    assert np.array_equal(a[~np.isnan(a)], b_[~np.isnan(b_)])
    ref = (G[:, :96].double().T @ X[:, :256].double()).float()
    got = flat[: 96 * 288].view(96, 288)[:, :256]
    assert float((got - ref).abs().max() / ref.abs().max()) < 2e-6


ARCHS = {
    "8x256_viewdirs": dict(multires=6),                                                        # two workgroups per CU: four waves
    "deep_head_geo": dict(use_semantics=True, sem_layer=3, sem_dim=5, sem_with_geo=True),      # one workgroup per CU: eight waves
    "4x96": dict(netdepth=4, netwidth=96, netdepth_fine=4, netwidth_fine=96),        # ragged width: padded tiles
    "2x608_16pt": dict(netdepth=2, netwidth=608, netdepth_fine=2, netwidth_fine=608),  # past the 32-point tiles' LDS budget: 16-point tiles
}


@pytest.mark.parametrize("name", list(ARCHS))
def test_saved_relu_bit_words_equal_the_sign_of_the_saved_blocks(name):
    torch.manual_seed(1)
    net = nerf_sos_amd.NeRFNet(N_samples=16, N_importance=0, **ARCHS[name]).to(DEV).train()
    mlp = net.nerf
    R, S = 37, 16                                                 # 592 points: ragged last tile
    rays = syn.synthetic_rays(R, seed=2, device=DEV)
    near = torch.full((R,), syn.NEAR, device=DEV); far = torch.full((R,), syn.FAR, device=DEV)
    z, v = ops.ray_setup(rays[1].contiguous(), near, far, S, None)
    raw, acts = mlp.query_rays(rays[0].contiguous(), rays[1].contiguous(), v, z, save=True)          # every block + the bit words
    raw2, acts2 = mlp.query_rays(rays[0].contiguous(), rays[1].contiguous(), v, z, save=True)
    assert torch.equal(raw, mlp.query_rays(rays[0].contiguous(), rays[1].contiguous(), v, z)), "SAVE changed the outputs"
    ld, layout = mlp._gplan.layout()
    params = dict(mlp.mlp.named_parameters())
    pad = lambda n: (n + 31) // 32 * 32  # noqa: E731
    end = max(col + pad(out_dim) for _, col, out_dim, _ in layout)
    names = [n for n, *_ in layout]
    sem = [n for n in names if n.startswith("semantic_linear.")]
    relu = [n for n in names if n.startswith("pts_linears.") or n == "views_linears.0" or n == "geo_map_sem.0" or (n in sem and n != sem[-1])]
    n_words = sum(pad(od) // 32 for n_, _, od, _ in layout if n_ in relu)
    used = end + n_words                                          # (the row's padding to 16 bytes behind the words is never written)
    assert torch.equal(raw, raw2) and torch.equal(acts[:, :used].contiguous().view(torch.int32), acts2[:, :used].contiguous().view(torch.int32)), "not deterministic"
    words = acts[:, end:].contiguous().view(torch.int32).cpu().numpy().astype(np.int64) & 0xFFFFFFFF
    a = acts.cpu().numpy()
    w0 = 0
    for n_, col, out_dim, _ in layout:                            # program order = layout order for the dense ops
        if n_ not in relu:
            continue
        tiles = pad(out_dim) // 32
        blk = a[:, col: col + 32 * tiles]
        for t in range(tiles):
            wt = words[:, w0 + t]
            for h in range(2):
                for r in range(16):
                    f = 32 * t + (r & 3) + 8 * (r >> 2) + 4 * h
                    bit = (wt >> (r + 16 * h)) & 1
                    assert np.array_equal(bit.astype(bool), blk[:, f] > 0), (name, n_, t, h, r)
        w0 += tiles
    assert end + w0 <= ld < end + w0 + 4
    assert params[relu[0] + ".weight"].shape[0] == layout[0][2]


@pytest.mark.parametrize("name", ["8x256_viewdirs", "deep_head_geo"])
def test_ray_gradients_of_a_frozen_net_equal_the_all_trainable_backward(name):
    """The subset program with bit 31 (every gradient of the chain formed, none stored, no activation block saved) against the full
    program on the same weights: rays.grad bit for bit; and the frozen call returns no parameter gradient."""
    R = 64
    rays0 = syn.synthetic_rays(R, seed=4, device=DEV)
    gt = torch.rand(R, 3, device=DEV)
    grads = {}
    for frozen in (False, True):
        torch.manual_seed(3)
        net = nerf_sos_amd.NeRFNet(N_samples=16, N_importance=24, perturb=0.0, raw_noise_std=0.0, **ARCHS[name]).to(DEV).train()
        for p in net.parameters():
            p.requires_grad_(not frozen)
        r = rays0.clone().requires_grad_(True)
        ret = net(r, (syn.NEAR, syn.FAR), retraw=False)
        (((ret["rgb"] - gt) ** 2).mean() + ((ret["rgb0"] - gt) ** 2).mean() + ret["depth"].mean() * 1e-3).backward()
        grads[frozen] = r.grad.clone()
        if frozen:
            assert all(p.grad is None for p in net.parameters())
    assert torch.isfinite(grads[True]).all() and float(grads[True].abs().max()) > 0
    assert torch.equal(grads[True], grads[False])


def test_a_net_whose_outputs_get_no_gradient_is_not_differentiated():
    """A loss on the fine maps only: the fine samples are detached from the coarse weights (models/sampler.py:159), so the coarse net gets
    no gradient at all (None, as autograd leaves an unvisited branch) and the fine net's gradients equal those of the same loss with a
    zero-weighted coarse term added -- bit for bit (the skipped branch contributed exact zeros)."""
    rays = syn.synthetic_rays(48, seed=6, device=DEV)
    gt = torch.rand(48, 3, device=DEV)
    out = {}
    for with_coarse in (False, True):
        torch.manual_seed(2)
        net = nerf_sos_amd.NeRFNet(N_samples=16, N_importance=16, perturb=0.0, raw_noise_std=0.0, netwidth=128, netwidth_fine=128).to(DEV).train()
        ret = net(rays, (syn.NEAR, syn.FAR), retraw=False)
        loss = ((ret["rgb"] - gt) ** 2).mean()
        if with_coarse:
            loss = loss + 0.0 * ((ret["rgb0"] - gt) ** 2).mean()
        loss.backward()
        out[with_coarse] = {n: (None if p.grad is None else p.grad.clone()) for n, p in net.named_parameters()}
    assert all(v is None for n, v in out[False].items() if n.startswith("nerf.")), "the coarse net was differentiated"
    assert all(v is not None for n, v in out[False].items() if n.startswith("nerf_fine."))
    for n, v in out[False].items():
        if n.startswith("nerf_fine."):
            assert torch.equal(v, out[True][n]), n
