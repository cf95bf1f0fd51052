#!/usr/bin/env python3
"""GPU box: race detector for the hand-synchronised kernels.  Every kernel on the path is deterministic by construction
(block-ordered reductions, no atomics), so two identical training runs must end with BIT-IDENTICAL parameters; a counted
`s_waitcnt`, an LDS ring slot or a relaxed chunk barrier that is wrong only sometimes shows up here as a mismatch.
Runs N optimiser steps twice per mode and compares every parameter.   usage: soak_determinism.py [steps] [rays]"""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import nerf_sos_amd
from nerf_sos_amd import synthetic as syn

dev = torch.device("cuda:0")
STEPS = int(sys.argv[1]) if len(sys.argv) > 1 else 25
R = int(sys.argv[2]) if len(sys.argv) > 2 else 2048


GENERIC = {   # modes on the generic-architecture kernels (csrc/mlp_generic.hip): every parameter trains
    "gen6x96": dict(netdepth=6, netwidth=96, netdepth_fine=6, netwidth_fine=96, multires=6, multires_views=2, sem_layer=3),   # 32-point tiles, 3 WGs / CU
    "gen8x256": dict(),                                                # the shipped shape forced onto them: rays require grad (pose refinement)
    "gen4x512d": dict(netdepth=4, netwidth=512, netdepth_fine=4, netwidth_fine=512, sem_layer=3),    # 16-point tiles
}


def run(mode: str, precision: str):
    torch.manual_seed(0)
    net = nerf_sos_amd.NeRFNet(N_samples=64, N_importance=128, use_semantics=True, sem_with_coord=True, perturb=1.0,
                               raw_noise_std=1.0, **GENERIC.get(mode, {})).to(dev).train()
    if mode == "frozen":                                  # run_nerf.py:307-318 (--fix_backbone)
        for n_, p_ in net.named_parameters():
            p_.requires_grad = "semantic_linear" in n_
    net.mlp_precision = precision
    opt = torch.optim.Adam([p for p in net.parameters() if p.requires_grad], lr=5e-4)
    rays = syn.synthetic_rays(R if mode != "gen4x512d" else min(R, 512), seed=1, device=dev)
    if mode == "gen8x256":
        rays = rays.clone().requires_grad_(True)          # -> both nets on the generic kernels, gradients to the rays as well
        opt.add_param_group({"params": [rays], "lr": 1e-4})
    gen = torch.Generator(dev).manual_seed(7)
    Rn = rays.shape[1]
    gt, gt_sem = torch.rand(Rn, 3, device=dev, generator=gen), torch.rand(Rn, 2, device=dev, generator=gen)
    torch.manual_seed(123)                                # the render's own draws (jitter, sigma noise)
    losses = []
    for _ in range(STEPS):
        opt.zero_grad()
        ret = net(rays, (syn.NEAR, syn.FAR), retraw=False)
        loss = ((ret["semantics"] - gt_sem) ** 2).mean() + ((ret["semantics0"] - gt_sem) ** 2).mean()
        if mode == "full" or mode in GENERIC:
            loss = loss + ((ret["rgb"] - gt) ** 2).mean() + ((ret["rgb0"] - gt) ** 2).mean()
        loss.backward()
        opt.step()
        losses.append(float(loss.detach()))
    torch.cuda.synchronize()
    params = {n: p.detach().clone() for n, p in net.named_parameters()}
    if mode == "gen8x256":
        params["rays"] = rays.detach().clone()
    return params, losses


def run_sharded_step(precision: str):
    """The C4 step (two 64x64 patches, both correlation losses, the appearance loss on its side stream, fused Adam): the
    multi-stream part of the path."""
    import types
    from nerf_sos_amd import sharding
    torch.manual_seed(0)
    net = nerf_sos_amd.NeRFNet(N_samples=64, N_importance=128, use_semantics=True, sem_with_coord=True, perturb=1.0,
                               raw_noise_std=1.0, ray_chunk=1 << 20).to(dev).train()
    for n_, p_ in net.named_parameters():
        p_.requires_grad = "semantic_linear" in n_
    net.mlp_precision = precision
    net.rng, net.rng_seed = "philox", 1
    opt = torch.optim.Adam([p for p in net.parameters() if p.requires_grad], lr=5e-4, fused=True)
    a = types.SimpleNamespace(rand_neg=False, self_corr_w=0, use_sim_matrix=True, patch_stride=6,
                              app_corr_params=["0.18", "1", "0.46", "1"], geo_corr_params=["0.5", "1", "3", "1"])
    corr, geo = nerf_sos_amd.CorrelationLoss(a), nerf_sos_amd.GeoCorrelationLoss(a)
    B = 2
    rays = syn.synthetic_patches(B, 64, 6, seed=0, device=dev)
    feat = torch.randn(B, 384, 14, 14, generator=torch.Generator().manual_seed(1)).to(dev)
    cls_ = torch.randn(B, 384, generator=torch.Generator().manual_seed(2)).to(dev)
    losses = []
    for i in range(STEPS):
        opt.zero_grad(set_to_none=True)
        losses.append(sharding.sharded_patch_step(net, rays, (syn.NEAR, syn.FAR), B, feat, cls_, corr, geo, step=i, seed=0))
        opt.step()
    torch.cuda.synchronize()
    return {n: p.detach().clone() for n, p in net.named_parameters()}, [float(l) for l in losses]


out = {}
ok_all = True
for precision in ("bf16", "fp32"):
    a, la = run_sharded_step(precision)
    b, lb = run_sharded_step(precision)
    bad = [n for n in a if not torch.equal(a[n], b[n])]
    ok = not bad and la == lb and all(torch.isfinite(v).all() for v in a.values())
    ok_all &= ok
    out[f"c4_step_{precision}"] = {"steps": STEPS, "bit_identical": not bad, "losses_identical": la == lb, "loss_first": la[0], "loss_last": la[-1]}
    print(f"c4step {precision:7s}: {'OK' if ok else 'MISMATCH'}  loss {la[0]:.5f} -> {la[-1]:.5f}", flush=True)
for mode, precision in (("full", "fp32"), ("full", "fp16x3"), ("frozen", "fp32"), ("frozen", "fp16x3"), ("frozen", "bf16"), ("frozen", "fp16"),
                        ("gen6x96", "fp32"), ("gen8x256", "fp32"), ("gen4x512d", "fp32")):
    a, la = run(mode, precision)
    b, lb = run(mode, precision)
    bad = [n for n in a if not torch.equal(a[n], b[n])]
    finite = all(torch.isfinite(v).all() for v in a.values())
    ok = not bad and la == lb and finite
    ok_all &= ok
    out[f"{mode}_{precision}"] = {"steps": STEPS, "rays": R, "bit_identical": not bad, "losses_identical": la == lb, "finite": finite,
                                  "loss_first": la[0], "loss_last": la[-1], "mismatching": bad[:4]}
    print(f"{mode:6s} {precision:7s}: {'OK' if ok else 'MISMATCH'}  loss {la[0]:.5f} -> {la[-1]:.5f}", flush=True)
os.makedirs("gpurun_out", exist_ok=True)
json.dump(out, open("gpurun_out/soak_determinism.json", "w"), indent=1)
sys.exit(0 if ok_all else 1)
