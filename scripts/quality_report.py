#!/usr/bin/env python3
"""GPU box: render N rays with the HIP path and with the C oracle (same weights, same rays); report PSNR of
the rendered colour and the max-abs / max-rel error of every output key (SURVEY section 8d "quality")."""
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import nerf_sos_amd
from oracle import c_oracle as co
from oracle import torch_port as tp

n = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
out = {}
for cfg_name, kw, peaky in (("C2 no-sem default-init", dict(use_semantics=False), False),
                            ("C3 sem+coord peaky", dict(use_semantics=True, sem_with_coord=True), True)):
    torch.manual_seed(0)
    net = nerf_sos_amd.NeRFNet(N_samples=64, N_importance=128, **kw).cuda().eval()
    if peaky:
        net.load_state_dict(tp.make_peaky({k: v.cpu() for k, v in net.state_dict().items()}))
    rays = tp.synthetic_rays(n, seed=0).cuda()
    with torch.no_grad():
        got = net(rays, (tp.NEAR, tp.FAR))
    torch.cuda.synchronize()
    sd = {k: v.cpu() for k, v in net.state_dict().items()}
    t0 = time.time()
    ref = co.render(sd, rays[0].cpu(), rays[1].cpu(), tp.NEAR, tp.FAR, use_semantics=kw.get("use_semantics", False),
                    sem_with_coord=kw.get("sem_with_coord", False))
    dt = time.time() - t0
    rep = {"rays": n, "oracle_seconds": round(dt, 1), "oracle_threads": co.num_threads(), "keys": {}}
    for k in sorted(got):
        a = got[k].cpu().numpy().astype(np.float64).reshape(ref[k].shape)
        b = ref[k].astype(np.float64)
        fin = np.isfinite(a) & np.isfinite(b)
        err = np.abs(a - b)[fin]
        rep["keys"][k] = {"max_abs": float(err.max()), "max_rel": float((err / (np.abs(b[fin]) + 1e-3)).max()),
                          "frac_gt_1e-4": float((err > 1e-4 * (1 + np.abs(b[fin]))).mean())}
    for k in ("rgb", "rgb0"):
        mse = float(np.mean((got[k].cpu().numpy().astype(np.float64) - ref[k]) ** 2))
        rep[f"psnr_{k}_db"] = float(-10 * np.log10(max(mse, 1e-30)))
    rep["inds_mismatch_frac"] = None
    out[cfg_name] = rep
    print(cfg_name, json.dumps({k: v for k, v in rep.items() if k != "keys"}))
    for k, v in rep["keys"].items():
        print(f"   {k:12s} max_abs {v['max_abs']:.3e}  max_rel {v['max_rel']:.3e}  frac>1e-4 {v['frac_gt_1e-4']:.2e}")
os.makedirs("gpurun_out", exist_ok=True)
json.dump(out, open("gpurun_out/quality_report.json", "w"), indent=1)
