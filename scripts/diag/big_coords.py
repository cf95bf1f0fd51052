#!/usr/bin/env python3
"""16-bit MLP kernels against the exact-fp32 kernel for coordinates up to 1e5 (the hardware-sine encoder's large-argument fallback;
fp16 overflows on the raw coordinate itself beyond 65504 -- its range, not the encoder)."""
import os, sys
sys.path.insert(0, os.getcwd())
import torch, nerf_sos_amd
from nerf_sos_amd import ops
dev = "cuda:0"
torch.manual_seed(0)
net = nerf_sos_amd.NeRFNet(N_samples=64, N_importance=0, use_semantics=True, sem_with_coord=True).to(dev).eval()
for scale in (1.0, 10.0, 100.0, 1000.0, 1e5):
    R = 256
    o = (torch.rand(R, 3, device=dev) * 2 - 1) * scale
    d = torch.tensor([[0.0, 0.0, -1.0]], device=dev).expand(R, 3).contiguous()
    z = torch.zeros(R, 1, device=dev)
    r32 = ops.mlp_forward_rays(net.nerf.packed_weights("fp32"), net.nerf.sem_mode, o, d, d, z)
    out = {}
    for prec in ("fp16", "bf16"):
        r = ops.mlp_forward_rays_lp(net.nerf.packed_weights(prec), net.nerf.sem_mode, prec, o, d, d, z)
        out[prec] = float((r - r32).abs().max() / r32.abs().max())
    print(f"|x| <= {scale:g}: max |lp - fp32| / max|fp32|: fp16 {out['fp16']:.2e}  bf16 {out['bf16']:.2e}  finite {bool(torch.isfinite(r).all())}")
