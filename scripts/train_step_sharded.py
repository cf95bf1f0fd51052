#!/usr/bin/env python3
"""Config C4 shape: the frozen-backbone training step with the patch batch sharded over the GPUs of one node.

    python scripts/train_step_sharded.py                                    # 1 GPU
    python -m torch.distributed.run --nproc-per-node 8 --master-addr 127.0.0.1 scripts/train_step_sharded.py

B patches of PxP rays (default 16 x 32x32 = 8192 rays per GPU at 8 GPUs... here: `--patches-per-gpu`), patch b lives on
rank b mod N (sharding.local_patches).  Per step and rank:
  render own patches (train mode, 64+128 samples, semantic head with coordinates, `--precision`)
  -> ONE flat all-gather of the rendered patch tensors the losses compare across the batch (semantics0, semantics,
     depth; RCCL over xGMI, ~0.5 MiB per patch) -> splice the rank's own gradient-carrying patches back in
  -> both correlation losses on the whole batch (cheap, identical on every rank)
  -> backward through the rank's own patches -> ONE flat all-reduce of the 82 436 head gradients -> Adam.
No other collective; weights are replicated.  Prints one JSON line on rank 0 (max step time over ranks).
"""
import argparse
import json
import os
import sys
import time
import types

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import torch.distributed as dist

import nerf_sos_amd
from nerf_sos_amd import sharding
from oracle import torch_port as tp


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--patches-per-gpu", type=int, default=2)
    ap.add_argument("--patch", type=int, default=64)
    ap.add_argument("--precision", default="bf16", choices=["fp32", "fp16x3", "bf16", "fp16"])
    ap.add_argument("--steps", type=int, default=10)
    args = ap.parse_args()
    rank, world = int(os.environ.get("RANK", 0)), int(os.environ.get("WORLD_SIZE", 1))
    local = int(os.environ.get("LOCAL_RANK", 0))
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
    B, P = args.patches_per_gpu * world, args.patch
    la = types.SimpleNamespace(rand_neg=False, self_corr_w=0, use_sim_matrix=True, patch_stride=6,
                               app_corr_params=["0.18", "1", "0.46", "1"], geo_corr_params=["0.5", "1", "3", "1"])
    torch.manual_seed(0)                                          # same weights everywhere (one checkpoint in a real run)
    net = nerf_sos_amd.NeRFNet(N_samples=64, N_importance=128, use_semantics=True, sem_with_coord=True, perturb=1.0,
                               raw_noise_std=1.0, ray_chunk=1 << 20).to(dev)
    for n_, p_ in net.named_parameters():                         # run_nerf.py:307-318 (--fix_backbone)
        p_.requires_grad = "semantic_linear" in n_
    net.train()
    net.mlp_precision = args.precision
    opt = torch.optim.Adam([p for p in net.parameters() if p.requires_grad], lr=5e-4)
    corr, geo = nerf_sos_amd.CorrelationLoss(la), nerf_sos_amd.GeoCorrelationLoss(la)
    own = sharding.local_patches(B, rank, world)
    g = torch.Generator().manual_seed(1)                          # batch-wide inputs, identical on every rank
    all_rays = tp.synthetic_rays(B * P * P, seed=0).reshape(2, B, P, P, 3)
    feat = torch.randn(B, 384, 14, 14, generator=g).to(dev)      # DINO features of the whole batch (outside the path)
    sim = torch.rand(B, B, generator=g).to(dev)
    rays = all_rays[:, own].to(dev)
    ro, rd = all_rays[0].permute(0, 3, 1, 2).to(dev), all_rays[1].permute(0, 3, 1, 2).to(dev)

    def step():
        opt.zero_grad()
        ret = net(rays, (tp.NEAR, tp.FAR), retraw=False)
        local = {k: ret[k] for k in ("semantics", "semantics0", "depth")}
        if world > 1:
            full = sharding.all_gather_patches(local, B)
            full = sharding.splice_local_patches(full, local, B)
        else:
            full = local
        s0, s1 = full["semantics0"].permute(0, 3, 1, 2), full["semantics"].permute(0, 3, 1, 2)
        depth = full["depth"].detach().permute(0, 3, 1, 2).contiguous()
        torch.manual_seed(1234)                                   # the losses' sample coordinates: same draw on every rank
        loss = corr(feat, s0, sim) + corr(feat, s1, sim)
        loss = loss + 0.01 * (geo(depth, s0, [ro, rd, None], sim) + geo(depth, s1, [ro, rd, None], sim))
        loss.backward()
        sharding.all_reduce_grads(net.parameters())               # sum: every rank back-propagated its own patches only
        opt.step()
        return loss

    for _ in range(3):
        step()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        loss = step()
    torch.cuda.synchronize()
    dt = torch.tensor([time.perf_counter() - t0], device=dev, dtype=torch.float64)
    if world > 1:
        dist.all_reduce(dt, op=dist.ReduceOp.MAX)
    if rank == 0:
        ms = float(dt.item()) / args.steps * 1e3
        print(json.dumps({"n_gpus": world, "patches": B, "patch": P, "rays_per_gpu": len(own) * P * P, "precision": args.precision,
                          "ms_per_step": round(ms, 3), "rays_per_s": round(B * P * P / ms * 1e3), "loss": round(float(loss.detach()), 5)}))
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
