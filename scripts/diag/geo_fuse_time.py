"""GPU box: geometric correlation loss (forward + gradient) with the column gradient fused into pass 3 vs the separate fourth
pass (NSOS_GEO_SEPARATE_COLS=1), at the C3 step's stacked shape (2 patches of 64 x 64), the C4 per-GPU shape (4) and B = 8."""
import os
import sys
import types

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import nerf_sos_amd

dev = torch.device("cuda:0")
a = types.SimpleNamespace(rand_neg=False, self_corr_w=0, use_sim_matrix=True, patch_stride=6,
                          app_corr_params=["0.18", "1", "0.46", "1"], geo_corr_params=["0.5", "1", "3", "1"])
mod = nerf_sos_amd.GeoCorrelationLoss(a)
for B in (2, 4, 8):
    g = torch.Generator(dev).manual_seed(0)
    P = 64
    depth = 2.0 + 9.0 * torch.rand(B, 1, P, P, device=dev, generator=g)
    code = torch.randn(B, 2, P, P, device=dev, generator=g)
    ray_o = torch.zeros(B, 3, P, P, device=dev)
    ray_d = torch.randn(B, 3, P, P, device=dev, generator=g) * 0.2
    sim = torch.rand(B, B, device=dev, generator=g)
    for kind in ("fused", "separate"):
        os.environ["NSOS_GEO_SEPARATE_COLS"] = "1" if kind == "separate" else ""

        def fn():
            c = code.clone().requires_grad_(True)
            mod(depth.clone(), c, [ray_o, ray_d, None], sim).backward()
        for _ in range(3):
            fn()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(30):
            fn()
        e1.record()
        torch.cuda.synchronize()
        print(f"B={B} {kind:9s} {e0.elapsed_time(e1) / 30 * 1e3:8.1f} us per loss + gradient", flush=True)
