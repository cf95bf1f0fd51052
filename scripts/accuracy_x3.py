#!/usr/bin/env python3
"""GPU box: arithmetic error of every K2 variant against an fp64 evaluation of the same network on the same points.

The question it answers: is the split-fp16 kernel (mlp_x3.hip, three 16-bit MFMAs per product) "fp32-grade"?  The
yardsticks are (a) torch's own fp32 CPU evaluation (what the reference computes) and (b) the exact-fp32 MFMA kernel.
Errors are relative to 1 + |ref|, over all outputs of 64 rays x 192 samples; encodings are evaluated in fp32 (as the
reference does) and only the network arithmetic runs in fp64.  Writes gpurun_out/accuracy_x3.json."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
from nerf_sos_amd import ops
from oracle import torch_port as tp
from helpers import CFGS, ref_state

manifest = json.load(open(os.path.join(ROOT, "tests", "golden", "manifest.json")))
dev = torch.device("cuda:0")
out = {}
R, S = 64, 192
for name, peaky in (("semcoord", False), ("semcoord", True), ("nosem", False), ("nosem", True)):
    cfg = tp.PortConfig(**CFGS[name])
    sd = ref_state(name, manifest, peaky)
    mode = ops.sem_mode_of(**CFGS[name])
    rays = tp.synthetic_rays(R, seed=3)
    torch.manual_seed(1)
    z = tp.stratified_z(torch.full((R, 1), tp.NEAR), torch.full((R, 1), tp.FAR), S, torch.rand(R, S))
    vd = rays[1] / rays[1].norm(dim=-1, keepdim=True)
    pts = tp.ray_points(rays[0], rays[1], z).reshape(-1, 3)
    e = torch.cat([tp.posenc(pts, cfg.multires), tp.posenc(vd[:, None, :].expand(R, S, 3).reshape(-1, 3), cfg.multires_views)], -1)
    ref64 = tp.mlp_forward({k: v.double() for k, v in sd.items()}, "nerf_fine", e.double(), cfg)
    scale = 1 + ref64.abs()

    def err(x):
        d = (x.double().reshape(ref64.shape) - ref64).abs() / scale
        return {"max": float(d.max()), "rms": float(d.square().mean().sqrt())}

    rec = {"torch_cpu_fp32": err(tp.mlp_forward(sd, "nerf_fine", e, cfg))}
    params = {k[len("nerf_fine") + 5:]: t.to(dev) for k, t in sd.items() if k.startswith("nerf_fine.mlp.")}
    args = [t.to(dev).contiguous() for t in (rays[0], rays[1], vd, z)]
    rec["kernel_fp32"] = err(ops.mlp_forward_rays(ops.pack_mlp(params, mode), mode, *args).cpu())
    for prec in ("fp16x3", "fp16", "bf16"):
        rec["kernel_" + prec] = err(ops.mlp_forward_rays_lp(ops.pack_mlp(params, mode, precision=prec), mode, prec, *args).cpu())
    key = f"{name}_{'peaky' if peaky else 'default'}"
    out[key] = rec
    print(key, " ".join(f"{k}: max {v['max']:.2e} rms {v['rms']:.2e} |" for k, v in rec.items()), flush=True)
os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
json.dump(out, open(os.path.join(ROOT, "gpurun_out", "accuracy_x3.json"), "w"), indent=1)
