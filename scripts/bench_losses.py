#!/usr/bin/env python3
"""GPU box: timings of the rows next to the render path (correlation losses, evaluation post-processing) at the
shipped recipe's shapes (B=8 patches of 64x64, DINO features [8,384,14,14]; C5 image 1008x756), with the CPU
restatement (oracle/losses_port.py = the reference's op sequence) timed beside them.  Secondary numbers for
DESIGN.md; the headline bench is bench.py."""
import json
import os
import sys
import time
import types

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import nerf_sos_amd
from nerf_sos_amd import ops
from oracle import losses_port as lp

dev = torch.device("cuda:0")
a = types.SimpleNamespace(rand_neg=False, self_corr_w=0, use_sim_matrix=True, patch_stride=6,
                          app_corr_params=["0.18", "1", "0.46", "1"], geo_corr_params=["0.5", "1", "3", "1"])
B, P = 8, 64
g = torch.Generator().manual_seed(0)
depth = 2.0 + 9.0 * torch.rand(B, 1, P, P, generator=g)
code = torch.randn(B, 2, P, P, generator=g)
ray_o = torch.zeros(B, 3, P, P)
ray_d = torch.randn(B, 3, P, P, generator=g) * 0.2
ray_d[:, 2] -= 1
feat = torch.randn(B, 384, 14, 14, generator=g)
sim = torch.rand(B, B, generator=g)


def gpu_time(fn, reps=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps


out = {}
geo, app = nerf_sos_amd.GeoCorrelationLoss(a), nerf_sos_amd.CorrelationLoss(a)
D, Cd, RO, RD, F, S = (t.to(dev) for t in (depth, code, ray_o, ray_d, feat, sim))


def geo_step():
    c = Cd.clone().requires_grad_(True)
    geo(D.clone(), c, [RO, RD, None], S).backward()


def app_step():
    c = Cd.clone().requires_grad_(True)
    app(F, c, S).backward()


ms = gpu_time(geo_step)
pairs = 2 * B * (P * P) ** 2
out["geo_loss_fwd_bwd"] = {"ms": round(ms, 3), "pairs": pairs, "pair_passes": 4, "Gpair_evals_per_s": round(4 * pairs / ms / 1e6, 1),
                           "reference_materialises_MB": round(pairs / 2 * 4 / 1e6)}
out["app_loss_fwd_bwd"] = {"ms": round(gpu_time(app_step), 3)}
R = 1008 * 756
sem, rgb, tgt = torch.randn(R, 2, device=dev), torch.rand(R, 3, device=dev), torch.rand(R, 3, device=dev)
ms = gpu_time(lambda: ops.eval_postprocess(sem, rgb, tgt))
out["eval_postprocess_762048_rays"] = {"ms": round(ms, 4), "GB_per_s": round(R * (8 + 24 + 8 + 4) / ms / 1e6, 1)}

# CPU: the reference's op sequence on this box's host cores (one repetition each; the geometric one needs ~4 GB)
torch.set_num_threads(32)
t0 = time.perf_counter()
c = code.clone().requires_grad_(True)
lp.geo_correlation_loss(depth.clone(), c, ray_o, ray_d, lp.neg_index(sim), lp.CorrParams(0.5, 1, 3, 1)).backward()
out["geo_loss_fwd_bwd"]["cpu_port_s"] = round(time.perf_counter() - t0, 2)
t0 = time.perf_counter()
c = code.clone().requires_grad_(True)
lp.correlation_loss(feat, c, lp.neg_index(sim), torch.rand(B, 11, 11, 2) * 2 - 1, torch.rand(B, 11, 11, 2) * 2 - 1,
                    lp.CorrParams(0.18, 1, 0.46, 1)).backward()
out["app_loss_fwd_bwd"]["cpu_port_s"] = round(time.perf_counter() - t0, 4)
t0 = time.perf_counter()
lp.eval_postprocess(sem.cpu(), rgb.cpu(), tgt.cpu())
out["eval_postprocess_762048_rays"]["cpu_port_s"] = round(time.perf_counter() - t0, 4)
out["cpu_threads"] = 32
print(json.dumps(out, indent=1))
os.makedirs("gpurun_out", exist_ok=True)
json.dump(out, open("gpurun_out/bench_losses.json", "w"), indent=1)
