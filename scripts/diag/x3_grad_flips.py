#!/usr/bin/env python3
"""Is a 1e-3 gradient difference of the split-fp16 full backward against fp32 autograd ReLU-mask flips or a bug?
For a few seeds of one odd-shaped configuration: gradient error of both precisions against the port, and the number of
hidden units whose sign differs between the split-fp16 and the exact-fp32 forward."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
import nerf_sos_amd
from nerf_sos_amd import ops
from oracle import torch_port as tp
from helpers import CFGS

dev = "cuda:0"
R, S, name = 40, 72, "semcoord"
for seed in range(6):
    torch.manual_seed(5004 + seed)
    net = nerf_sos_amd.NeRFNet(N_samples=S, N_importance=0, white_bkgd=False, **CFGS[name]).to(dev).eval()
    nerf_sos_amd.synthetic.spiky_density_(net, 2.0, 0.5)
    sd = {k: v.detach().cpu().clone().requires_grad_(True) for k, v in net.state_dict().items()}
    rays = tp.synthetic_rays(R, seed=6004 + seed)
    tgt = torch.rand(R, 3, generator=torch.Generator().manual_seed(seed))
    loss_of = lambda o, t: ((o["rgb"] - t) ** 2).mean() + (o["semantics"] ** 2).mean() + 0.1 * ((o["depth"] ** 2).mean() + (o["acc"] ** 2).mean())
    ref = tp.render(sd, tp.PortConfig(n_samples=S, n_importance=0, **CFGS[name]), rays, (tp.NEAR, tp.FAR))
    loss_of(ref, tgt).backward()
    line = [f"seed {seed}"]
    for prec in ("fp32", "fp16x3"):
        net.mlp_precision = prec
        net.zero_grad(set_to_none=True)
        out = net(rays.to(dev), (tp.NEAR, tp.FAR))
        loss_of(out, tgt.to(dev)).backward()
        worst = max((float((p.grad.cpu() - sd[n].grad).abs().max()) / float(sd[n].grad.abs().max()), n) for n, p in net.named_parameters())
        line.append(f"{prec}: {worst[0]:.2e} ({worst[1].split('mlp.')[1]})")
    # sign differences of the hidden activations between the two forwards
    o, d = rays[0].to(dev).contiguous(), rays[1].to(dev).contiguous()
    v = d / d.norm(dim=-1, keepdim=True)
    z = torch.linspace(0, 1, S, device=dev)[None] * (tp.FAR - tp.NEAR) + tp.NEAR
    z = z.expand(R, S).contiguous()
    a32 = ops.mlp_forward_rays_save_all(net.nerf.packed_weights("fp32"), net.nerf.sem_mode, o, d, v, z, "fp32")[1]
    a16 = ops.mlp_forward_rays_save_all(net.nerf.packed_weights("fp16x3"), net.nerf.sem_mode, o, d, v, z, "fp16x3")[1]
    flips = int(((a32[:, :2048] > 0) != (a16[:, :2048] > 0)).sum())
    line.append(f"flips {flips} of {a32[:, :2048].numel()}; max |dh| {float((a32 - a16).abs().max()):.2e}")
    print("  ".join(line))
