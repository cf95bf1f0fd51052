#!/usr/bin/env python3
"""GPU box: one training step of the shipped frozen-backbone recipe (config C3; scripts/train_fortress_node0.sh):
render B patches of PxP rays (64+128 samples, semantic head with coordinates, train-mode perturb/noise) -> both
correlation losses on semantics0 and semantics (engines/trainer.py:127-166) -> backward into the semantic heads ->
Adam step.  DINO features are synthetic (the ViT is outside the path).  Prints ms/step and rays/s per precision."""
import json
import os
import sys
import time
import types

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import nerf_sos_amd
from nerf_sos_amd import synthetic as syn

dev = torch.device("cuda:0")
args = types.SimpleNamespace(rand_neg=False, self_corr_w=0, use_sim_matrix=True, patch_stride=6,
                             app_corr_params=["0.18", "1", "0.46", "1"], geo_corr_params=["0.5", "1", "3", "1"])
out = {}
CFG = [tuple(int(x) for x in a.split("x")) for a in sys.argv[1].split(",")] if len(sys.argv) > 1 else [(8, 64), (4, 32)]
PRECS = sys.argv[2].split(",") if len(sys.argv) > 2 else ["fp32", "bf16", "fp16"]
for B, P in CFG:
    for prec in PRECS:
        torch.manual_seed(0)
        net = nerf_sos_amd.NeRFNet(N_samples=64, N_importance=128, use_semantics=True, sem_with_coord=True, perturb=1.0,
                                   raw_noise_std=1.0, ray_chunk=1 << 20).to(dev)
        syn.spiky_density_(net, gain=40.0, shift=1.0)
        for n_, p_ in net.named_parameters():                      # run_nerf.py:307-318 (--fix_backbone)
            p_.requires_grad = "semantic_linear" in n_
        net.train()
        net.mlp_precision = prec
        opt = torch.optim.Adam([p for p in net.parameters() if p.requires_grad], lr=5e-4)
        corr, geo = nerf_sos_amd.CorrelationLoss(args), nerf_sos_amd.GeoCorrelationLoss(args)
        rays = syn.synthetic_rays(B * P * P, seed=0, device=dev).reshape(2, B, P, P, 3)
        feat = torch.randn(B, 384, 14, 14, device=dev)
        sim = torch.rand(B, B, device=dev)
        ro, rd = rays[0].permute(0, 3, 1, 2), rays[1].permute(0, 3, 1, 2)

        def step():
            opt.zero_grad()
            ret = net(rays, (syn.NEAR, syn.FAR), retraw=False)
            s0, s1 = ret["semantics0"].permute(0, 3, 1, 2), ret["semantics"].permute(0, 3, 1, 2)
            depth = ret["depth"].permute(0, 3, 1, 2)
            loss = corr(feat, s0, sim) + corr(feat, s1, sim)
            loss = loss + 0.01 * (geo(depth, s0, [ro, rd, None], sim) + geo(depth, s1, [ro, rd, None], sim))
            loss.backward()
            opt.step()
            return loss

        for _ in range(3):
            l0 = step()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        K = 10
        for _ in range(K):
            l1 = step()
        torch.cuda.synchronize()
        ms = (time.perf_counter() - t0) / K * 1e3
        rec = {"ms_per_step": round(ms, 3), "rays_per_s": round(B * P * P / ms * 1e3), "loss_first": round(float(l0.detach()), 5),
               "loss_last": round(float(l1.detach()), 5)}
        out[f"B{B}_P{P}_{prec}"] = rec
        print(f"B={B} P={P} {prec}: {json.dumps(rec)}", flush=True)
os.makedirs("gpurun_out", exist_ok=True)
json.dump(out, open("gpurun_out/bench_train_step.json", "w"), indent=1)
