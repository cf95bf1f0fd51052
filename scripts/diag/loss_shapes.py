#!/usr/bin/env python3
"""Both correlation losses at shapes the goldens do not cover (non-square patches, other feature-map sizes, 1..5 patches,
1..4 code channels, depths beyond max_depth) against the CPU port of the reference's loss classes: value and gradient.
(One code channel is degenerate: the code is L2-normalised, so its gradient is analytically zero and the relative error printed for it
is noise over noise.)"""
import os
import sys
import types

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch
import nerf_sos_amd
from oracle import losses_port as lp

dev = "cuda:0"
a = types.SimpleNamespace(rand_neg=False, self_corr_w=0, use_sim_matrix=True, patch_stride=6,
                          app_corr_params=["0.18", "1", "0.46", "1"], geo_corr_params=["0.5", "1", "3", "1"])
pa = lp.CorrParams(self_shift=0.18, self_weight=1.0, neg_shift=0.46, neg_weight=1.0)
pg = lp.CorrParams(self_shift=0.5, self_weight=1.0, neg_shift=3.0, neg_weight=1.0)
bad = 0
for case, (B, C, H, W, Cf, Hf, Wf) in enumerate([(3, 2, 24, 40, 20, 12, 10), (1, 2, 16, 16, 8, 5, 7), (5, 3, 20, 12, 16, 14, 14), (2, 4, 32, 32, 384, 14, 14),
                                                 (2, 1, 16, 24, 12, 9, 9), (4, 2, 8, 8, 6, 3, 3)]):
    g = torch.Generator().manual_seed(case)
    depth = 2.0 + 20.0 * torch.rand(B, 1, H, W, generator=g)            # some beyond max_depth = 15
    code = torch.randn(B, C, H, W, generator=g)
    ro = torch.randn(B, 3, 1, 1, generator=g).expand(B, 3, H, W).contiguous() * 0.1
    rd = torch.randn(B, 3, H, W, generator=g) * 0.2
    rd[:, 2] -= 1
    feat = torch.randn(B, Cf, Hf, Wf, generator=g)
    sim = torch.rand(B, B, generator=g)
    neg = lp.neg_index(sim)
    # geometric
    c_ref = code.clone().requires_grad_(True)
    want = lp.geo_correlation_loss(depth.clone(), c_ref, ro, rd, neg, pg)
    want.backward()
    c_dev = code.to(dev).requires_grad_(True)
    d_dev = depth.to(dev).clone()
    got = nerf_sos_amd.GeoCorrelationLoss(a)(d_dev, c_dev, [ro.to(dev), rd.to(dev), None], sim.to(dev))
    got.backward()
    e_l = abs(float(got) - float(want)) / (1 + abs(float(want)))
    e_g = float((c_dev.grad.cpu() - c_ref.grad).abs().max() / (c_ref.grad.abs().max() + 1e-30))
    d_ok = torch.allclose(d_dev.cpu(), lp.depth_filter_(depth.clone(), 15.0))
    # appearance (same draws: the module's generator replayed on the CPU)
    S = 11
    gen = torch.Generator(dev).manual_seed(100 + case)
    mod = nerf_sos_amd.CorrelationLoss(a)
    mod.generator = gen
    c_dev2 = code.to(dev).requires_grad_(True)
    got2 = mod(feat.to(dev), c_dev2, sim.to(dev))
    got2.backward()
    gen2 = torch.Generator(dev).manual_seed(100 + case)
    c1 = torch.rand([B, S, S, 2], device=dev, generator=gen2).cpu() * 2 - 1
    c2 = torch.rand([B, S, S, 2], device=dev, generator=gen2).cpu() * 2 - 1
    c_ref2 = code.clone().requires_grad_(True)
    want2 = lp.correlation_loss(feat, c_ref2, neg, c1, c2, pa)
    want2.backward()
    e_l2 = abs(float(got2) - float(want2)) / (1 + abs(float(want2)))
    e_g2 = float((c_dev2.grad.cpu() - c_ref2.grad).abs().max() / (c_ref2.grad.abs().max() + 1e-30))
    ok = e_l < 1e-4 and e_g < 1e-4 and d_ok and e_l2 < 1e-4 and e_g2 < 1e-4
    bad += not ok
    print(f"case {case} B={B} C={C} {H}x{W} feat {Cf}x{Hf}x{Wf}: geo loss {e_l:.1e} grad {e_g:.1e} depth filter {'ok' if d_ok else 'DIFFERS'} | "
          f"app loss {e_l2:.1e} grad {e_g2:.1e} {'OK' if ok else 'MISMATCH'}", flush=True)
print("failures:", bad)
