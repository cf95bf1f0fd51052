#!/usr/bin/env python3
"""rays/s of the NeRF-SOS render path on MI355X (BASELINE.json metric).

    python bench.py [--gpus N] [--steps K] [--warmup W]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

A step = one pass of the hot path (NeRFNet.forward: coarse 64 + fine 192 MLP evaluations per ray, both
compositing passes, hierarchical resampling) over one synthetic batch already resident in HBM.
Workload at any N: BASELINE.json configs[1] per GPU ("flower_full, 4096 rays x (64+128), fp32"): weak scaling,
rays sharded across ranks, no data-path collective.  Prints ONE JSON line on rank 0.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import torch  # noqa: E402

N_RAYS = 4096
N_COARSE, N_IMPORTANCE = 64, 128
MAC_PER_POINT = 593408            # matmul MACs of one MLP evaluation, no semantic head
EVALS_PER_RAY = N_COARSE + (N_COARSE + N_IMPORTANCE)   # 64 coarse + 192 fine
FLOP_PER_RAY = 2 * MAC_PER_POINT * EVALS_PER_RAY       # 303.82 MFLOP
PEAK_FP32_MFMA_TFLOPS = 157.3     # MI355X_MICROARCH.md: v_mfma_f32_32x32x2_f32, dense
PEAK_F16_MFMA_TFLOPS = 2500.0     # v_mfma_f32_32x32x16_f16, dense (the split-fp16 variant issues 3 of them per product)


def cpu_baseline(n_rays_sample: int, budget_s: float = 25.0):
    """The reference's CPU path = the pure-torch op-for-op port (bit-identical to the reference on CPU,
    tests/golden/make_goldens.py), eval-mode forward, on a bounded sample of the workload.  torch's CPU
    kernels do not scale to every logical CPU of the GPU box (measured: 256 threads are 35x SLOWER than 32 on
    the 2x64-core EPYC host), so a few thread counts are tried inside the budget and the best is reported,
    with the thread count actually used."""
    from oracle import torch_port as tp
    ncpu = os.cpu_count() or 1
    cfg = tp.PortConfig(n_samples=N_COARSE, n_importance=N_IMPORTANCE, use_semantics=False, pts_chunk=1024 * 256)
    sd = tp.init_state_dict(cfg, seed=0)
    rays = tp.synthetic_rays(n_rays_sample, seed=0)
    best, best_threads, reps, t_all = float("inf"), 0, 0, time.perf_counter()
    with torch.no_grad():
        for threads in sorted({min(ncpu, t) for t in (32, 16, 64)}, key=lambda t: abs(t - 32)):
            if reps and (time.perf_counter() - t_all) > budget_s * 0.6:
                break
            torch.set_num_threads(threads)
            tp.render(sd, cfg, rays, (tp.NEAR, tp.FAR), retraw=True)  # warm-up
            for _ in range(3):
                t0 = time.perf_counter()
                tp.render(sd, cfg, rays, (tp.NEAR, tp.FAR), retraw=True)
                dt = time.perf_counter() - t0
                reps += 1
                if dt < best:
                    best, best_threads = dt, threads
                if (time.perf_counter() - t_all) > budget_s:
                    break
    return {"value": round(n_rays_sample / best, 1), "unit": "rays/s", "cores": best_threads, "kind": "port",
            "sample": f"{n_rays_sample} of the {N_RAYS} rays of the same batch, eval-mode forward, best of {reps} runs over "
                      f"thread counts <= 64 (host has {ncpu} logical CPUs), torch {torch.__version__} CPU ops"}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--precision", choices=("fp32", "fp16x3"), default="fp32",
                    help="fp32: exact-fp32 MFMA kernel (the headline).  fp16x3: split-fp16 kernel, fp32-grade results "
                         "(DESIGN.md 4.6b); without this flag it is measured as a side note under \"variants\"")
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            raise SystemExit("launch with torch.distributed.run --nproc-per-node N for --gpus N")
    import torch.distributed as dist
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)

    import nerf_sos_amd
    from nerf_sos_amd import ops
    from oracle import torch_port as tp  # synthetic ray generator only (inputs, not arithmetic)

    torch.manual_seed(0)
    net = nerf_sos_amd.NeRFNet(N_samples=N_COARSE, N_importance=N_IMPORTANCE, use_semantics=False,
                               perturb=1.0, raw_noise_std=1.0).to(dev).eval()
    net.mlp_precision = args.precision
    rays = tp.synthetic_rays(N_RAYS, seed=rank).to(dev)     # resident in HBM before the timed region

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    with torch.no_grad():
        for _ in range(args.warmup):
            net(rays, (tp.NEAR, tp.FAR))
        barrier()
        ops.KERNEL_EVENTS = []                               # live HIP-event timing of every MLP launch
        t0 = time.perf_counter()
        for _ in range(args.steps):
            out = net(rays, (tp.NEAR, tp.FAR))
        barrier()
        dt = time.perf_counter() - t0
    events, ops.KERNEL_EVENTS = ops.KERNEL_EVENTS, None
    assert out["rgb"].shape == (N_RAYS, 3)

    variants = None
    if world == 1 and args.precision == "fp32":
        # side note, outside the timed region: the same step on the split-fp16 kernel (fp32-grade results, see
        # tests/test_gpu_parity.py::test_mlp_x3_is_fp32_grade) and how far its coarse pass is from the exact path's
        net.mlp_precision = "fp16x3"
        with torch.no_grad():
            for _ in range(2):
                alt = net(rays, (tp.NEAR, tp.FAR))
            ref_c = alt["rgb0"]                            # eval mode: no perturbation, no noise
            torch.cuda.synchronize()
            ops.KERNEL_EVENTS = []
            t1 = time.perf_counter()
            for _ in range(args.steps):
                alt = net(rays, (tp.NEAR, tp.FAR))
            torch.cuda.synchronize()
            dt_alt = time.perf_counter() - t1
            ev_alt, ops.KERNEL_EVENTS = ops.KERNEL_EVENTS, None
            net.mlp_precision = "fp32"
            exact_c = out["rgb0"]
        f_alt = [a.elapsed_time(b) for (n_pts, a, b) in ev_alt if n_pts == N_RAYS * (N_COARSE + N_IMPORTANCE)]
        f_alt_ms = sum(f_alt) / max(1, len(f_alt))
        variants = {"fp16x3": {
            "what": "same step, MLP on the 16-bit matrix pipe with split-fp16 operands (3 MFMAs per product, fp32 accumulate)",
            "value": round(N_RAYS * args.steps / dt_alt, 1), "unit": "rays/s", "ms_per_step": round(1e3 * dt_alt / args.steps, 4),
            "kernel_ms": round(f_alt_ms, 4),
            "algorithmic_tflops": round(2.0 * MAC_PER_POINT * N_RAYS * (N_COARSE + N_IMPORTANCE) / (f_alt_ms * 1e-3) / 1e12, 1),
            "issued_frac_of_f16_peak": round(3 * 2.0 * MAC_PER_POINT * N_RAYS * (N_COARSE + N_IMPORTANCE) / (f_alt_ms * 1e-3) / 1e12 / PEAK_F16_MFMA_TFLOPS, 4),
            "max_abs_rgb0_vs_exact_fp32": float((ref_c - exact_c).abs().max()),
            "parity": "same tests and bars as the exact kernel (2e-5 vs the reference goldens; error vs an fp64 evaluation equal "
                      "to fp32 arithmetic's own: profiles/r01/k_accuracy_x3.json)"}}

    t = torch.tensor([dt], device=dev, dtype=torch.float64)
    if world > 1:
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
    dt = float(t.item())

    # dominant kernel = the fine-pass fused MLP launch (786 432 points): algorithmic FLOPs / mean duration
    fine = [a.elapsed_time(b) for (n_pts, a, b) in events if n_pts == N_RAYS * (N_COARSE + N_IMPORTANCE)]
    fine_ms = sum(fine) / max(1, len(fine))
    fine_flop = 2.0 * MAC_PER_POINT * N_RAYS * (N_COARSE + N_IMPORTANCE)
    achieved = fine_flop / (fine_ms * 1e-3) / 1e12 if fine else 0.0

    traffic = None
    tj = os.path.join(ROOT, "profiles", "r01", "traffic.json")   # PMC passes of this same command (separate runs)
    if os.path.exists(tj):
        traffic = json.load(open(tj)).get("hbm_bytes_per_launch")

    if rank == 0:
        value = world * N_RAYS * args.steps / dt
        line = {
            "metric": "rays/sec (coarse+fine, 64+128 samples)", "value": round(value, 1), "unit": "rays/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(1e3 * dt / args.steps, 4), "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": "BASELINE configs[1]: LLFF flower_full shape, 4096 rays/GPU x (64 coarse + 192 fine "
                                   "MLP evaluations), eval-mode NeRFNet.forward, fp32 exact-MFMA, no semantic head, "
                                   "random-init weights (seed 0), pinhole rays 1008x756 f=850",
                       "rays_per_gpu": N_RAYS, "parallelism": f"ray-sharded x{world}, no collective in the path",
                       "flop_per_ray": FLOP_PER_RAY},
            "roofline": {"bound": "mfma", "achieved": round(achieved, 2), "peak": PEAK_FP32_MFMA_TFLOPS,
                         "unit": "TFLOP/s", "frac": round(achieved / PEAK_FP32_MFMA_TFLOPS, 4), "traffic": traffic,
                         "traffic_note": "HBM bytes per fine-pass launch from rocprofv3 FETCH_SIZE(x2)+WRITE_SIZE, profiles/r01/traffic.json",
                         "kernel": "mlp_fused_kernel<0,true> (fine pass, 786432 points)",
                         "kernel_ms": round(fine_ms, 4), "launches_timed": len(fine),
                         "whole_path_frac": round(value / world * FLOP_PER_RAY / 1e12 / PEAK_FP32_MFMA_TFLOPS, 4)},
        }
        if args.precision == "fp16x3":
            issued = 3 * achieved
            line["dtype"] = "f16x3 (split-fp16 operands, fp32 accumulate, fp32-grade results)"
            line["config"]["workload"] = line["config"]["workload"].replace("fp32 exact-MFMA", "split-fp16 MFMA (fp32-grade)")
            line["roofline"].update({"peak": PEAK_F16_MFMA_TFLOPS, "frac": round(achieved / PEAK_F16_MFMA_TFLOPS, 4),
                                     "issued_frac": round(issued / PEAK_F16_MFMA_TFLOPS, 4), "traffic": None,
                                     "kernel": "mlp_x3_kernel<0> (fine pass, 786432 points)",
                                     "whole_path_frac": round(value / world * FLOP_PER_RAY / 1e12 / PEAK_F16_MFMA_TFLOPS, 4)})
        if variants:
            line["variants"] = variants
        if world == 1 and not args.no_cpu_baseline:
            line["cpu_baseline"] = cpu_baseline(1024)
        print(json.dumps(line), flush=True)
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
