"""Compositing kernels on the C5 chunk shape and the C2 / C3 shapes: time per launch and HBM rate (algorithmic bytes)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from nerf_sos_amd import ops
dev = "cuda:0"
g = torch.Generator(dev).manual_seed(0)
for R, S, C in ((65536, 192, 6), (65536, 192, 4), (4096, 192, 6), (4096, 192, 4), (65536, 64, 6), (4096, 64, 6)):
    raw = torch.randn(R, S, C, device=dev, generator=g)
    z = (1.2 + 13.5 * torch.rand(R, S, device=dev, generator=g)).sort(-1).values
    d = torch.randn(R, 3, device=dev, generator=g)
    for _ in range(3):
        ops.composite(raw, z, d)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20):
        out = ops.composite(raw, z, d)
    e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) / 20 * 1e3
    nbytes = R * S * (4 * C + 4 + 4) + R * 64
    print(f"composite R={R} S={S} C={C}: {us:8.1f} us  {nbytes / us / 1e6:6.2f} TB/s  checksum {float(out['rgb'].double().sum()):.9e} {float(out['weights'].double().sum()):.9e}")
