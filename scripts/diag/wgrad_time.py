#!/usr/bin/env python3
"""Time nsos_sem_head_wgrad_x3 at the C4 per-GPU size (8192 rays x 64 / 192 samples, bf16 sem_in).
usage: wgrad_time.py [rays]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from nerf_sos_amd import ops

R = int(sys.argv[1]) if len(sys.argv) > 1 else 8192
dev = torch.device("cuda:0")
torch.manual_seed(0)
for S in (64, 192):
    P = R * S
    w = torch.rand(R, S, device=dev) / S
    g = torch.randn(R, 2, device=dev) * 1e-4
    w2 = torch.randn(2, 128, device=dev) * 0.1
    hid = torch.relu(torch.randn(P, 128, device=dev)).to(torch.bfloat16)     # as the 16-bit forward stores it
    x = torch.randn(P, 320, device=dev).to(torch.bfloat16)
    for tag, hh, xx in (("rows / rows", hid, x), ("rows / tiled sem_in", hid, ops.sem_in_tiled(x)), ("tiled / tiled", ops.sem_hid_tiled(hid), ops.sem_in_tiled(x))):
        for _ in range(3):
            ops.sem_head_wgrad(w, g, w2, hh, xx, split_fp16=True)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        n = 20
        e0.record()
        for _ in range(n):
            ops.sem_head_wgrad(w, g, w2, hh, xx, split_fp16=True)
        e1.record()
        torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / n
        gb = P * (640 + 256 + 4) / 1e9
        print(f"S={S:4d} P={P:8d}  sem_hid / sem_in {tag:20s} {ms * 1e3:8.1f} us/call  {gb / ms * 1e3:7.1f} GB/s  ({os.environ.get('NERF_SOS_HIP_LIB', 'default')})")
