#!/usr/bin/env python3
"""A plain loop of eval-mode forward steps (4096 rays) for profilers:
    PROFILE_CMD="python $PWD/scripts/run_forward.py fp16" bash scripts/profile_gpu.sh lp_fp16
usage: run_forward.py [fp32|fp16|bf16] [steps] [--semantics]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import nerf_sos_amd
from nerf_sos_amd import synthetic as syn

prec = sys.argv[1] if len(sys.argv) > 1 else "fp32"
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 12
sem = "--semantics" in sys.argv
dev = torch.device("cuda:0")
torch.manual_seed(0)
net = nerf_sos_amd.NeRFNet(N_samples=64, N_importance=128, use_semantics=sem, sem_with_coord=sem).to(dev).eval()
net.mlp_precision = prec
rays = syn.synthetic_rays(4096, seed=0, device=dev)
with torch.no_grad():
    for _ in range(steps):
        net(rays, (syn.NEAR, syn.FAR))
torch.cuda.synchronize()
