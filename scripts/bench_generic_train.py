#!/usr/bin/env python3
"""GPU box: one full training step of GENERIC-architecture nets (everything the reference's constructor builds beyond the shipped
8 x 256 net): 4096 rays x (64 + 128) samples, train mode, img2mse on rgb and rgb0 (engines/trainer.py:113-121), backward through both
networks (the generic input-gradient chain + nsos_wgrad over the saved column blocks), Adam step.  Prints one JSON line per
architecture: ms per step, rays/s, the forward / backward split, bytes saved per step and the FLOP rate of the step
(forward 2 MAC + backward 4 MAC per point and weight) against the fp32-MFMA peak."""
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import nerf_sos_amd
from nerf_sos_amd import synthetic as syn

dev = torch.device("cuda:0")
R = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
ARCHS = {
    "4x128": dict(netdepth=4, netwidth=128, netdepth_fine=4, netwidth_fine=128),
    "8x256_deep_sem_head": dict(use_semantics=True, sem_layer=4, sem_with_coord=True),
    "8x256_no_viewdirs": dict(viewdirs=False),
    "8x256_sem_dim7_geo": dict(use_semantics=True, sem_dim=7, sem_with_geo=True),
    "6x96_multires6": dict(netdepth=6, netwidth=96, netdepth_fine=6, netwidth_fine=96, multires=6, multires_views=2),
    "8x512": dict(netwidth=512, netwidth_fine=512),
    "8x512_deep_head_16_point_tiles": dict(netwidth=512, netwidth_fine=512, use_semantics=True, sem_layer=3),
}
for name, kw in ARCHS.items():
    torch.manual_seed(0)
    net = nerf_sos_amd.NeRFNet(N_samples=64, N_importance=128, perturb=1.0, raw_noise_std=1.0, **kw).to(dev).train()
    assert not net.nerf.fast
    opt = torch.optim.Adam(net.parameters(), lr=5e-4)
    rays = syn.synthetic_rays(R, seed=0, device=dev)
    gt = torch.rand(R, 3, device=dev)
    macs = sum(p.numel() for n, p in net.nerf.named_parameters() if n.endswith("weight")) * 64 + \
        sum(p.numel() for n, p in net.nerf_fine.named_parameters() if n.endswith("weight")) * 192

    def step(timing=None):
        opt.zero_grad()
        if timing is not None:
            torch.cuda.synchronize(); t0 = time.perf_counter()
        ret = net(rays, (syn.NEAR, syn.FAR), retraw=False)
        loss = ((ret["rgb"] - gt) ** 2).mean() + ((ret["rgb0"] - gt) ** 2).mean()
        if timing is not None:
            torch.cuda.synchronize(); t1 = time.perf_counter()
        loss.backward()
        if timing is not None:
            torch.cuda.synchronize(); t2 = time.perf_counter()
        opt.step()
        if timing is not None:
            torch.cuda.synchronize(); timing.append((t1 - t0, t2 - t1, time.perf_counter() - t2))
        return loss

    losses = [float(step().detach()) for _ in range(3)]
    torch.cuda.synchronize()
    K = 8
    t0 = time.perf_counter()
    for _ in range(K):
        l = step()
    torch.cuda.synchronize()
    ms = (time.perf_counter() - t0) / K * 1e3
    split = []
    for _ in range(4):
        step(split)
    f, b, o = (sum(x[i] for x in split) / len(split) * 1e3 for i in range(3))
    ld = [m._gplan.layout()[0] for m in (net.nerf, net.nerf_fine)]
    saved = (ld[0] * 64 + ld[1] * 192) * R * 4 * 2          # acts + gbuf
    print(json.dumps({"arch": name, "rays": R, "ms_per_step": round(ms, 2), "rays_per_s": round(R / ms * 1e3), "forward_ms": round(f, 2),
                      "backward_ms": round(b, 2), "adam_ms": round(o, 2), "saved_GB_per_step": round(saved / 1e9, 2),
                      "step_TFLOPs": round(6 * macs * R / ms / 1e9, 1), "frac_of_fp32_mfma_peak": round(6 * macs * R / ms / 1e9 / 157.3, 3),
                      "loss_first": round(losses[0], 5), "loss_last": round(float(l.detach()), 5)}), flush=True)
    del net, opt
    torch.cuda.empty_cache()
