#!/usr/bin/env python3
"""GPU box: throughput and accuracy of the reduced-precision (fp16 / bf16 MFMA) render path vs the fp32 path,
on the bench workload (4096 rays x (64+192)) and a full 1008x756 image.  Secondary numbers for DESIGN.md; the
headline bench (bench.py) is fp32."""
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import nerf_sos_amd
from nerf_sos_amd import ops
from nerf_sos_amd import synthetic as syn

dev = torch.device("cuda:0")
out = {}
for sem in (False, True):
    torch.manual_seed(0)
    net = nerf_sos_amd.NeRFNet(N_samples=64, N_importance=128, use_semantics=sem, sem_with_coord=sem,
                               ray_chunk=65536).to(dev).eval()
    syn.spiky_density_(net, gain=40.0, shift=1.0)
    rays = syn.synthetic_rays(4096, seed=0, device=dev)
    ref = None
    for prec in ("fp32", "fp16x3", "fp16", "bf16"):
        net.mlp_precision = prec
        with torch.no_grad():
            for _ in range(3):
                o = net(rays, (syn.NEAR, syn.FAR))
            torch.cuda.synchronize()
            ops.KERNEL_EVENTS = []
            t0 = time.perf_counter()
            for _ in range(20):
                o = net(rays, (syn.NEAR, syn.FAR))
            torch.cuda.synchronize()
            dt = (time.perf_counter() - t0) / 20
            ev, ops.KERNEL_EVENTS = ops.KERNEL_EVENTS, None
        fine = [a.elapsed_time(b) for n, a, b in ev if n == 4096 * 192]
        fine_ms = sum(fine) / len(fine)
        mac = 634496 if sem else 593408
        tf = 2 * mac * 4096 * 192 / (fine_ms * 1e-3) / 1e12
        rec = {"rays_per_s": round(4096 / dt), "ms_per_step": round(dt * 1e3, 3), "fine_kernel_ms": round(fine_ms, 3),
               "fine_kernel_tflops": round(tf, 1)}
        # the same step replayed from a captured HIP graph (host launch path out of the way)
        gr = nerf_sos_amd.GraphedRender(net, 4096, (syn.NEAR, syn.FAR))
        for _ in range(3):
            gr(rays)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(20):
            gr(rays)
        torch.cuda.synchronize()
        dtg = (time.perf_counter() - t0) / 20
        rec["graphed_ms_per_step"] = round(dtg * 1e3, 3)
        rec["graphed_rays_per_s"] = round(4096 / dtg)
        if prec == "fp32":
            ref = {k: v.clone() for k, v in o.items()}
        else:
            mse = (o["rgb"] - ref["rgb"]).square().mean().item()
            rec["psnr_rgb_vs_fp32_db"] = round(-10 * np.log10(max(mse, 1e-30)), 1)
            rec["max_abs_rgb"] = float((o["rgb"] - ref["rgb"]).abs().max())
            issued = tf * 3 if prec == "fp16x3" else tf     # the split kernel issues three MFMAs per useful product
            rec["frac_of_lp_peak_2500TF"] = round(issued / 2500, 3)
        out[f"{'semcoord' if sem else 'nosem'}_{prec}"] = rec
        print(f"{'semcoord' if sem else 'nosem':9s} {prec}: {json.dumps(rec)}", flush=True)
os.makedirs("gpurun_out", exist_ok=True)
json.dump(out, open("gpurun_out/bench_lp.json", "w"), indent=1)
