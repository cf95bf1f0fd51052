#!/usr/bin/env python3
"""Full-image evaluation render (BASELINE config C5 shape), single- or multi-GPU.

    python scripts/render_image.py [--H 756 --W 1008 --chunk 65536 --semantics]
    python -m torch.distributed.run --nproc-per-node 8 --master-addr 127.0.0.1 scripts/render_image.py ...

pose + intrinsics -> rays generated on each rank for its own contiguous pixel block (nsos_generate_rays; no ray
tensors cross PCIe) -> NeRFNet.forward in `chunk`-ray chunks (eval mode) -> per-rank rows, or an all-gather of the
image-sized maps.  `raw` is dropped (retraw=False): at 762 048 rays it is 4.7 GB that only a debugger reads.
Prints one JSON line on rank 0.
"""
import argparse
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import torch.distributed as dist

import nerf_sos_amd
from nerf_sos_amd import ops, sharding


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--H", type=int, default=756)
    ap.add_argument("--W", type=int, default=1008)
    ap.add_argument("--focal", type=float, default=850.0)
    ap.add_argument("--chunk", type=int, default=65536)
    ap.add_argument("--semantics", action="store_true")
    ap.add_argument("--gather", action="store_true")
    ap.add_argument("--reps", type=int, default=2)
    ap.add_argument("--precision", default="fp32", choices=["fp32", "fp16x3", "fp16", "bf16"])
    ap.add_argument("--post", action="store_true", help="also run the on-device eval post-processing (labels, PSNR)")
    args = ap.parse_args()
    rank, world = int(os.environ.get("RANK", 0)), int(os.environ.get("WORLD_SIZE", 1))
    local = int(os.environ.get("LOCAL_RANK", 0))
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    if world > 1:
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
    torch.manual_seed(0)  # same weights on every rank (a real run loads one checkpoint everywhere)
    net = nerf_sos_amd.NeRFNet(N_samples=64, N_importance=128, use_semantics=args.semantics,
                               sem_with_coord=args.semantics, ray_chunk=args.chunk).to(dev).eval()
    net.mlp_precision = args.precision
    K = [[args.focal, 0, args.W / 2], [0, args.focal, args.H / 2], [0, 0, 1]]
    c2w = [[1, 0, 0, 0], [0, 1, 0, 0], [0, 0, 1, 0]]
    n_pix = args.H * args.W
    b, e = sharding.shard_bounds(n_pix, rank, world)
    keys = ("rgb", "depth", "acc", "disp") + (("semantics",) if args.semantics else ())
    best = float("inf")
    for _ in range(args.reps):
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        rays = ops.generate_rays(args.H, args.W, K, c2w, dev, pix_range=(b, e))
        with torch.no_grad():
            out = net(rays, (1.2, 14.72), retraw=False)
        out = {k: out[k] for k in keys}
        if args.post:   # engines/eval.py:44-57,79-86 on device: only labels and two scalars would leave the GPU
            post = ops.eval_postprocess(out.get("semantics"), out["rgb"], torch.full_like(out["rgb"], 0.5))
        if args.gather and world > 1:
            rows = [sharding.shard_bounds(n_pix, r, world) for r in range(world)]
            out = {k: sharding.all_gather_rows(v, [y - x for x, y in rows]) for k, v in out.items()}
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        best = min(best, time.perf_counter() - t0)
    ok = all(torch.isfinite(out[k]).all().item() or k == "depth" for k in keys)
    if rank == 0:
        print(json.dumps({"image": f"{args.W}x{args.H}", "rays": n_pix, "n_gpus": world, "chunk": args.chunk,
                          "seconds": round(best, 4), "rays_per_s": round(n_pix / best, 1), "finite": ok,
                          "rows_on_rank0": int(out["rgb"].shape[0]), "semantics": args.semantics, "precision": args.precision,
                          "post": args.post,
                          "peak_mem_gb": round(torch.cuda.max_memory_allocated() / 2**30, 2)}))
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
