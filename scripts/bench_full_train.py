#!/usr/bin/env python3
"""GPU box: one full training step of config C2 (configs/flower_full.txt: every parameter trainable, 4096 rays x
(64+128) samples, fp32): train-mode render -> img2mse on rgb and rgb0 (engines/trainer.py:113-121) -> backward through
both networks -> Adam step (run_nerf.py:321).  Prints ms/step, rays/s and the forward/backward split."""
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
PREC = sys.argv[2] if len(sys.argv) > 2 else "fp32"      # forward kernel: "fp32" (exact) or "fp16x3" (split fp16, fp32-accurate)
torch.manual_seed(0)
net = nerf_sos_amd.NeRFNet(N_samples=64, N_importance=128, perturb=1.0, raw_noise_std=1.0).to(dev).train()
net.mlp_precision = PREC
net.exact_weight_gradients = "--exact-wgrad" in sys.argv     # fp32 mode only: weight-gradient reductions on the exact-fp32 MFMA
net.compact_activations = "--compact" in sys.argv            # fp16x3 only: the saved activations as 16-bit floats
opt = torch.optim.Adam(net.parameters(), lr=5e-4, betas=(0.9, 0.999))
rays = syn.synthetic_rays(R, seed=0, device=dev)
gt = torch.rand(R, 3, device=dev)


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
        torch.cuda.synchronize(); t3 = time.perf_counter()
        timing.append((t1 - t0, t2 - t1, t3 - t2))
    return loss


losses = [float(step().detach()) for _ in range(3)]
torch.cuda.synchronize()
K = 10
t0 = time.perf_counter()
for _ in range(K):
    l = step()
torch.cuda.synchronize()
ms = (time.perf_counter() - t0) / K * 1e3
split = []
for _ in range(5):
    step(split)
f, b, o = (sum(x[i] for x in split) / len(split) * 1e3 for i in range(3))
out = {"rays": R, "forward_kernel": PREC, "exact_weight_gradients": net.exact_weight_gradients, "ms_per_step": round(ms, 2), "rays_per_s": round(R / ms * 1e3), "forward_ms": round(f, 2), "backward_ms": round(b, 2),
       "adam_ms": round(o, 2), "loss_first": round(losses[0], 5), "loss_last": round(float(l.detach()), 5),
       "compact_activations": bool(net.compact_activations and PREC == "fp16x3"),
       "saved_activations_GB": round(R * 256 * 2656 * (2 if net.compact_activations and PREC == "fp16x3" else 4) / 1e9, 2)}
print(json.dumps(out))
os.makedirs("gpurun_out", exist_ok=True)
json.dump(out, open(f"gpurun_out/bench_full_train_{PREC}.json", "w"), indent=1)
