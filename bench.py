#!/usr/bin/env python3
"""rays/s of the NeRF-SOS render path on MI355X (BASELINE.json metric).

    python bench.py [--gpus N] [--steps K] [--warmup W] [--config c2|c3|c4|c5]

`--gpus N` with N > 1 launches its own N ranks (re-exec under `python -m torch.distributed.run --nproc-per-node N
--master-addr 127.0.0.1`); it is equally happy to be started by torch.distributed.run itself (RANK / LOCAL_RANK /
WORLD_SIZE in the environment).  One process per GPU over RCCL; rank 0 prints ONE JSON line.

A step = one pass of the hot path over one synthetic batch already resident in HBM:
  c2 (default, the headline): BASELINE configs[1] -- eval-mode NeRFNet.forward, 4096 rays/GPU x (64 coarse + 192 fine MLP
      evaluations), exact-fp32 MFMA, no semantic head.  Weak scaling, rays sharded across ranks, NO data-path collective.
  c3: configs[2] -- 4096 rays (one 64x64 patch), semantic head with coordinates, bf16: train-mode render + both
      correlation losses + backward of the semantic heads (the shipped --fix_backbone recipe) + Adam.
  c4: configs[3] -- 8192 rays/GPU (two 64x64 patches per GPU), the c3 step with the patch batch sharded over the GPUs:
      ONE flat RCCL all-gather of the patch tensors the batch-wide losses read, ONE flat all-reduce of the gradients.
  c5: configs[4] -- full-image eval 1008x756 in 65536-ray chunks, fp16 MFMA, rays generated on device, row blocks
      sharded over the GPUs, on-device post-processing; a step = one image.
The default run also measures c3 / c5 / c4 (and the split-fp16 kernel) briefly, OUTSIDE the timed region, and reports
them under "variants", each with its own roofline.
"""
import argparse
import hashlib
import json
import os
import socket
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

N_COARSE, N_IMPORTANCE = 64, 128
N_FINE = N_COARSE + N_IMPORTANCE
MAC_NOSEM, MAC_SEMCOORD = 593408, 634496   # matmul MACs of one MLP evaluation
EVALS_PER_RAY = N_COARSE + N_FINE           # 64 coarse + 192 fine
PEAK_FP32_MFMA_TFLOPS = 157.3               # MI355X_MICROARCH.md: v_mfma_f32_32x32x2_f32, dense
PEAK_16BIT_MFMA_TFLOPS = 2500.0             # v_mfma_f32_32x32x16_{f16,bf16}, dense
PATCH, PATCH_STRIDE = 64, 6                 # scripts/train_flower_node0.sh:4-6


def _self_launch(args):
    """`python bench.py --gpus N` (N > 1) outside torch.distributed.run: become the launcher."""
    if args.gpus <= 1 or "WORLD_SIZE" in os.environ:
        return
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}",
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    os.environ.setdefault("OMP_NUM_THREADS", "8")
    os.execv(sys.executable, cmd)


def physical_cores() -> int:
    try:
        seen, phys, core = set(), None, None
        for ln in open("/proc/cpuinfo"):
            if ln.startswith("physical id"):
                phys = ln.split(":")[1].strip()
            elif ln.startswith("core id"):
                core = ln.split(":")[1].strip()
            elif not ln.strip():
                if phys is not None and core is not None:
                    seen.add((phys, core))
                phys = core = None
        return len(seen) or (os.cpu_count() or 1)
    except OSError:
        return os.cpu_count() or 1


def cpu_baseline(gpu=None, n_rays: int = 4096, dense=None):
    """The reference's CPU path = the pure-torch op-for-op port (bit-identical to the reference on CPU,
    tests/golden/make_goldens.py), timed as SURVEY.md section 8(d) prescribes:
      primary   eval-mode `torch.no_grad()` forward over the FULL 4096-ray batch of the headline workload, fp32,
                `time.perf_counter`, 2 warm-ups then best of 5, at the thread count where torch's CPU kernels peak on this
                class of host (32; measured in round 1: 256 threads are 35x slower on the 2x64-core box) -- AND once at all
                physical cores (the survey's literal protocol), reported beside it;
      secondary train-mode forward + backward (the reference's four random tensors, MSE on rgb + rgb0, every parameter
                trainable) on a bounded 1024-ray sample, with autograd anomaly mode on -- as the reference runs, it enables it
                process-wide at models/sampler.py:2 -- and off.
    `gpu` = (state_dict, rays, outputs) of the headline GPU step: the port renders THE SAME rays with THE SAME weights, so the
    comparison of the two renders (`parity`: PSNR, max-abs per key, share of rays outside the 1e-4 band) costs nothing extra.
    `dense` = (state_dict, outputs) of one more GPU render of the same rays with a DENSE density field (the seed-0 weights with
    the sigma head scaled: x40, -1.5 -- the "peaky" field of the parity tests): the default-init field is almost empty (sigma
    < 0 nearly everywhere, so the fine maps are exactly 0 on both sides and their PSNR says nothing), the dense one puts the
    hierarchical sampler and the compositing to work.  One untimed port render.
    The port is the checker here -- it is timed and compared against, never shipped (the product path has no CPU fallback)."""
    import torch
    from oracle import torch_port as tp            # test infrastructure: the CPU baseline / checker, never the product path
    ncpu, phys = os.cpu_count() or 1, physical_cores()
    cfg = tp.PortConfig(n_samples=N_COARSE, n_importance=N_IMPORTANCE, use_semantics=False, pts_chunk=1024 * 256)
    if gpu is not None:
        sd = {k: v.detach().cpu() for k, v in gpu[0].items()}
        rays = gpu[1].detach().cpu()
        n_rays = rays.shape[1]
    else:
        sd = tp.init_state_dict(cfg, seed=0)
        rays = tp.synthetic_rays(n_rays, seed=0)
    best_threads = min(32, ncpu)

    def forward_runs(threads, warm, reps):
        torch.set_num_threads(threads)
        times, out = [], None
        with torch.no_grad():
            for k in range(warm + reps):
                t0 = time.perf_counter()
                out = tp.render(sd, cfg, rays, (tp.NEAR, tp.FAR), retraw=True)
                if k >= warm:
                    times.append(time.perf_counter() - t0)
        return times, out

    t_all = time.perf_counter()
    times, ref = forward_runs(best_threads, 2, 5)
    best = min(times)
    res = {"value": round(n_rays / best, 1), "unit": "rays/s", "cores": best_threads, "kind": "port",
           "physical_cores": phys, "logical_cpus": ncpu, "runs": len(times), "warmups": 2,
           "median_rays_per_s": round(n_rays / sorted(times)[len(times) // 2], 1),
           "sample": f"the full {n_rays}-ray batch of the headline workload (same rays and weights as the GPU step), eval-mode "
                     f"forward, fp32, best of {len(times)} runs after 2 warm-ups at {best_threads} threads (where torch's CPU "
                     f"kernels peak; host: {phys} physical cores / {ncpu} logical CPUs), torch {torch.__version__} CPU ops"}
    # the survey's literal protocol: all physical cores.  One warm-up, one timed run (it is several times slower).
    if phys != best_threads and phys <= ncpu:
        t_phys, _ = forward_runs(phys, 1, 1)
        res["all_physical_cores"] = {"value": round(n_rays / min(t_phys), 1), "unit": "rays/s", "cores": phys, "runs": 1, "warmups": 1}
    # secondary: train-mode forward + backward, anomaly mode on (as the reference runs) and off
    torch.set_num_threads(best_threads)
    n_tr = min(1024, n_rays)
    tr_rays = rays[:, :n_tr]
    params = {k: v.clone().requires_grad_(True) for k, v in sd.items()}
    g = torch.Generator().manual_seed(0)
    draws = [tp.Draws(t_rand=torch.rand((n_tr, N_COARSE), generator=g), noise0=torch.randn((n_tr, N_COARSE), generator=g),
                      u=torch.rand((n_tr, N_IMPORTANCE), generator=g), noise1=torch.randn((n_tr, N_FINE), generator=g))]
    target = torch.rand((n_tr, 3), generator=g)
    train = {}
    for tag, anomaly in (("anomaly_mode_on", True), ("anomaly_mode_off", False)):
        ts = []
        with torch.autograd.set_detect_anomaly(anomaly):
            for k in range(3):                                     # 1 warm-up + 2 timed
                for p_ in params.values():
                    p_.grad = None
                t0 = time.perf_counter()
                out = tp.render(params, cfg, tr_rays, (tp.NEAR, tp.FAR), raw_noise_std=1.0, draws_per_chunk=draws, retraw=True)
                loss = ((out["rgb"] - target) ** 2).mean() + ((out["rgb0"] - target) ** 2).mean()   # engines/trainer.py:101-109
                loss.backward()
                if k >= 1:
                    ts.append(time.perf_counter() - t0)
        train[tag] = {"value": round(n_tr / min(ts), 1), "unit": "rays/s", "runs": len(ts), "warmups": 1}
    res["train_fwd_bwd"] = dict(train, cores=best_threads, sample=f"{n_tr} rays of the same batch, train mode (perturb=1, raw_noise_std=1), "
                                "MSE on rgb + rgb0, every parameter trainable; the reference turns autograd anomaly mode on process-wide "
                                "(models/sampler.py:2)")
    # the per-GPU workload of configs[3] (8192 rays), eval forward: 1 warm-up + 2 timed
    rays8k = tp.synthetic_rays(8192, seed=1)
    torch.set_num_threads(best_threads)
    ts = []
    with torch.no_grad():
        for k in range(3):
            t0 = time.perf_counter()
            tp.render(sd, cfg, rays8k, (tp.NEAR, tp.FAR), retraw=True)
            if k >= 1:
                ts.append(time.perf_counter() - t0)
    res["eval_8192_rays"] = {"value": round(8192 / min(ts), 1), "unit": "rays/s", "cores": best_threads, "runs": len(ts), "warmups": 1,
                             "sample": "8192 rays (the per-GPU batch of BASELINE configs[3]) of the same camera, same weights, eval-mode forward"}
    res["frozen_recipe_train_step"] = cpu_frozen_recipe_step(tp, best_threads)
    res["cpu_seconds_spent"] = round(time.perf_counter() - t_all, 1)
    parity = None
    if gpu is not None:
        torch.set_num_threads(best_threads)
        parity = {"timed_workload_default_init_field": render_parity(gpu[2], ref)}
        parity["timed_workload_default_init_field"]["yardstick"] = parity_yardstick(tp, sd, cfg, rays, ref, gpu[2], gpu[3] if len(gpu) > 3 else None)
        if dense is not None:
            sd_dense = {k: v.detach().cpu() for k, v in dense[0].items()}
            with torch.no_grad():
                ref_dense = tp.render(sd_dense, cfg, rays, (tp.NEAR, tp.FAR), retraw=True)
            parity["dense_field"] = render_parity(dense[1], ref_dense)
            parity["dense_field"]["yardstick"] = parity_yardstick(tp, sd_dense, cfg, rays, ref_dense, dense[1], dense[3] if len(dense) > 3 else None)
            parity["dense_field"]["field"] = dict(dense[2] if len(dense) > 2 else {}, mean_acc_cpu=round(float(ref_dense["acc"].mean()), 4),
                                                  what="seed-0 weights, sigma head x gain + shift (bench.make_dense_field)")
            parity["psnr_db"] = parity["dense_field"]["psnr_db"]
    return res, parity


def cpu_frozen_recipe_step(tp, threads: int):
    """The training step BASELINE configs[2] names, on the CPU port: one 64x64 patch (stride 6) = 4096 rays through the sem+coord
    net in train mode (the reference's four random tensors), appearance + 0.01 x geometric correlation losses on semantics0 and
    semantics (oracle/losses_port.py), backward with only the semantic head trainable (run_nerf.py:307-318, --fix_backbone),
    Adam.  1 warm-up + 2 timed steps; autograd anomaly mode OFF (the kinder of the two settings the full-backward timing shows)."""
    import torch
    from oracle import losses_port as lp
    torch.set_num_threads(threads)
    cfg = tp.PortConfig(n_samples=N_COARSE, n_importance=N_IMPORTANCE, use_semantics=True, sem_with_coord=True, ray_chunk=1 << 20, pts_chunk=1024 * 256)
    sd = tp.init_state_dict(cfg, seed=0)
    params = {k: v.clone().requires_grad_("semantic_linear" in k) for k, v in sd.items()}
    opt = torch.optim.Adam([p_ for p_ in params.values() if p_.requires_grad], lr=5e-4)
    H, W, focal = 756, 1008, 850.0
    ar = torch.arange(PATCH, dtype=torch.float32) * PATCH_STRIDE
    jj, ii = torch.meshgrid(100.0 + ar, 200.0 + ar, indexing="ij")
    d = torch.stack([(ii - W * 0.5) / focal, -(jj - H * 0.5) / focal, -torch.ones_like(ii)], -1)[None]     # [1, P, P, 3]
    rays = torch.stack([torch.zeros_like(d), d], 0)
    n = PATCH * PATCH
    g = torch.Generator().manual_seed(0)
    feat = torch.randn(1, 384, 14, 14, generator=g)
    neg = torch.zeros(1, dtype=torch.long)                       # one patch: its own negative (argmin of a 1x1 similarity matrix)
    pa = lp.CorrParams(self_shift=0.18, self_weight=1.0, neg_shift=0.46, neg_weight=1.0)
    pg = lp.CorrParams(self_shift=0.5, self_weight=1.0, neg_shift=3.0, neg_weight=1.0)
    ts, loss = [], None
    for k in range(3):
        draws = [tp.Draws(t_rand=torch.rand((n, N_COARSE), generator=g), noise0=torch.randn((n, N_COARSE), generator=g),
                          u=torch.rand((n, N_IMPORTANCE), generator=g), noise1=torch.randn((n, N_FINE), generator=g))]
        c = [torch.rand([1, 11, 11, 2], generator=g) * 2 - 1 for _ in range(4)]
        t0 = time.perf_counter()
        opt.zero_grad(set_to_none=True)
        ret = tp.render(params, cfg, rays, (tp.NEAR, tp.FAR), raw_noise_std=1.0, draws_per_chunk=draws, retraw=False)
        s0, s1 = ret["semantics0"].permute(0, 3, 1, 2), ret["semantics"].permute(0, 3, 1, 2)
        depth = ret["depth"].detach().reshape(1, PATCH, PATCH, 1).permute(0, 3, 1, 2).contiguous()
        ro, rd = rays[0].permute(0, 3, 1, 2), rays[1].permute(0, 3, 1, 2)
        loss = lp.correlation_loss(feat, s0, neg, c[0], c[1], pa) + lp.correlation_loss(feat, s1, neg, c[2], c[3], pa)
        loss = loss + 0.01 * (lp.geo_correlation_loss(depth.clone(), s0, ro, rd, neg, pg) + lp.geo_correlation_loss(depth.clone(), s1, ro, rd, neg, pg))
        loss.backward()
        opt.step()
        if k >= 1:
            ts.append(time.perf_counter() - t0)
    return {"value": round(n / min(ts), 1), "unit": "rays/s", "ms_per_step": round(1e3 * min(ts), 1), "cores": threads, "runs": len(ts), "warmups": 1,
            "loss": round(float(loss.detach()), 6),
            "sample": "BASELINE configs[2]'s step on the port: one 64x64 patch = 4096 rays, sem+coord net, train mode, appearance + 0.01 x geometric "
                      "correlation losses, backward into the semantic head only (--fix_backbone), Adam; anomaly mode off"}


def make_dense_field(net, rays, bounds, gain: float = 40.0):
    """Turn the (almost empty) default-init density field of `net` into a dense one, in place: sigma head x `gain` and a shift
    chosen from a fixed ladder so that the FINE pass's mean opacity lands between 0.15 and 0.9 (a fixed shift is a lottery:
    -1.5 gives mean acc 0.49 for the seed-0 sem+coord net and an empty fine field for the seed-0 net without semantics).
    Returns {"gain", "shift", "mean_acc"}.  Deterministic (seeded weights, fixed ladder); outside every timed region."""
    import torch
    mlps = [net.nerf.mlp] + ([net.nerf_fine.mlp] if net.nerf_fine is not net.nerf else [])
    orig = [(m.alpha_linear.weight.detach().clone(), m.alpha_linear.bias.detach().clone()) for m in mlps]
    chosen = None
    with torch.no_grad():
        for shift in (-1.5, 0.0, 1.5, 4.0, 8.0, 16.0):
            for m, (w0, b0) in zip(mlps, orig):
                m.alpha_linear.weight.copy_(w0 * gain)
                m.alpha_linear.bias.copy_(b0 * gain + shift)
            net.invalidate_packed()
            acc = float(net(rays, bounds, retraw=False)["acc"].mean())
            chosen = {"gain": gain, "shift": shift, "mean_acc": round(acc, 4)}
            if 0.15 < acc < 0.9:
                break
    return chosen


def render_parity(got: dict, ref: dict, tol: float = 1e-4):
    """The other half of BASELINE.json's metric ("PSNR vs ref"): the headline GPU render against the CPU port's render (= the
    reference's, bit for bit on CPU) of the same rays with the same weights.  Per key: max |a-b|, and the share of rays with
    any element outside |a-b| <= tol * (1 + |b|) -- the north star's 1e-4 fp32 band.  The coarse pass is expected inside
    the band on every ray; the fine pass leaves it on ~1 % of rays: last-ulp differences of the coarse weights move
    importance samples where the cdf is flat (the reference's own fp64-vs-fp32 sensitivity is of the same size; DESIGN.md
    section 2, tests/test_gpu_pins.py)."""
    import torch
    keys = [k for k in ("rgb", "depth", "acc", "disp", "semantics", "weights", "raw", "z_std", "rgb0", "depth0", "acc0", "disp0", "semantics0",
                        "weights0", "raw0") if k in got and k in ref]
    per_key, bad_any, bad_maps = {}, None, None
    for k in keys:
        a = got[k].detach().float().cpu().reshape(ref[k].shape)
        b = ref[k].float()
        diff = (a - b).abs()
        fin = torch.isfinite(b) & torch.isfinite(a)
        # depth = 1e10 on empty rays on both sides (models/renderer.py:72): compare where finite and below that sentinel
        use = fin & (b.abs() < 1e9)
        # a NaN / Inf on the GPU side where the reference is finite is OUTSIDE the band, not excluded from it;
        # so is a finite GPU value where the reference holds the 1e10 sentinel or a non-finite value
        nonfinite = torch.isfinite(b) & ~torch.isfinite(a)
        sentinel = (b.abs() >= 1e9) & ~((a - b).abs() <= tol * (1 + b.abs()))
        out = ((diff > tol * (1 + b.abs())) & use) | nonfinite | (torch.isfinite(b) & sentinel)
        rays_out = out.reshape(out.shape[0], -1).any(-1)
        per_key[k] = {"max_abs": float(diff[use].max()) if use.any() else 0.0, "frac_rays_outside_1e-4": round(float(rays_out.float().mean()), 5)}
        if bool(nonfinite.any()):
            per_key[k]["gpu_nonfinite_where_ref_finite"] = int(nonfinite.sum())
        if not k.endswith("0") and k != "raw":
            bad_any = rays_out if bad_any is None else (bad_any | rays_out)
        if k in ("rgb", "depth", "acc", "disp", "semantics"):
            bad_maps = rays_out if bad_maps is None else (bad_maps | rays_out)

    def psnr(k):
        mse = float(((got[k].detach().float().cpu().reshape(ref[k].shape) - ref[k]) ** 2).mean())
        return round(-10.0 * __import__("math").log10(max(mse, 1e-30)), 2)

    coarse_ok = all(per_key[k]["frac_rays_outside_1e-4"] == 0.0 for k in per_key if k.endswith("0") or "rgb0" not in ref)
    return {"vs": f"oracle/torch_port.py on the host (bit-identical to the reference on CPU), same {ref['rgb'].shape[0]} rays, same weights",
            "psnr_db": {k: psnr(k) for k in ("rgb", "rgb0") if k in got and k in ref},
            "tolerance": "|gpu - ref| <= 1e-4 * (1 + |ref|)",
            "coarse_pass_all_rays_inside_1e-4": coarse_ok,
            "frac_rays_outside_1e-4_any_fine_map": round(float(bad_any.float().mean()), 5) if bad_any is not None else None,
            # the rendered maps alone (without the per-sample weights and the sampler's z_std, which sit on the importance sampler's
            # discontinuities: bin flips, and on trained fields the `denom < 1e-5` switch of models/sampler.py:117-118)
            "frac_rays_outside_1e-4_image_maps": round(float(bad_maps.float().mean()), 5) if bad_maps is not None else None,
            "per_key": per_key}


def parity_yardstick(tp, sd, cfg, rays, ref, got, inds_hip=None, tol: float = 1e-4):
    """The yardstick beside `frac_rays_outside_1e-4`: how far the REFERENCE moves against ITSELF under its own rounding.
    The hierarchical sampler is ill-conditioned where the coarse weights are small (alpha = 1 - exp(-sigma * delta) lives on a
    6e-8 grid: a last-ulp sigma moves an alpha of 1e-4 by 6e-4 of itself and an importance sample by up to 1e-2), so:
      reference_self_sensitivity  the port with its COARSE network evaluated in fp64 (rounded to fp32 once; everything else
                                  unchanged fp32) against the port itself: rays outside the same band in rgb/depth/acc (the
                                  N_self of tests/test_gpu_pins.py::test_free_running_render_at_c2_size);
      gpu_rays_outside            the same count for the GPU render (same keys);
      index_flip_rays             rays with at least one right-bisect index (searchsorted, models/sampler.py:112) different
                                  from the port's, the u = 1 end sample excluded (both indices give the same position there);
                                  the north star's "bit-exact for sample indices" holds for (cdf, u) -> index
                                  (tests/test_gpu_parity.py::test_importance); this counts the flips caused by the GPU's
                                  last-ulp differences in the cdf itself.
    Host work outside every timed region: one fp64 coarse pass + one fp32 fine pass of the port."""
    import torch
    n = rays.shape[1]
    near, far = torch.full((n, 1), tp.NEAR), torch.full((n, 1), tp.FAR)
    u = torch.linspace(0.0, 1.0, steps=cfg.n_importance).expand(n, cfg.n_importance)
    viewdirs = rays[1] / torch.norm(rays[1], dim=-1, keepdim=True)
    keys = [k for k in ("rgb", "depth", "acc", "semantics") if k in ref and k in got]

    def fine_pass(weights0):
        z = tp.stratified_z(near, far, cfg.n_samples, None)
        mids = 0.5 * (z[..., 1:] + z[..., :-1])
        zs, inds = tp.invert_cdf(mids, tp.pdf_to_cdf(weights0[..., 1:-1]), u)
        z_fine, _ = torch.sort(torch.cat([z, zs], -1), -1)
        pts = tp.ray_points(rays[0], rays[1], z_fine)
        raw = tp.point_query(sd, "nerf_fine", pts, viewdirs[..., None, :].expand(pts.shape), cfg)
        maps = tp.composite(raw, z_fine, rays[1], None, cfg)
        maps["z_std"] = torch.std(zs, dim=-1, unbiased=False)                      # models/nerf_net.py:124
        return inds, maps

    def outside(maps):
        o = torch.zeros(n, dtype=torch.bool)
        for k in keys:
            a = maps[k].detach().double().cpu().reshape(ref[k].shape)
            b = ref[k].double()
            o |= (~((a - b).abs() <= tol + tol * b.abs())).reshape(n, -1).any(-1)
        return int(o.sum())

    with torch.no_grad():
        inds_ref, chk = fine_pass(ref["weights0"])
        staged_ok = bool(torch.equal(chk["rgb"], ref["rgb"]))
        z = tp.stratified_z(near, far, cfg.n_samples, None)
        pts = tp.ray_points(rays[0], rays[1], z)
        sd64 = {k: v.double() for k, v in sd.items()}
        raw64 = tp.point_query(sd64, "nerf", pts.double(), viewdirs.double()[..., None, :].expand(pts.shape), cfg).float()
        self_maps = fine_pass(tp.composite(raw64, z, rays[1], None, cfg)["weights"])[1]
        n_self = outside(self_maps)

    def z_std_outside(t):
        a, b = t.detach().double().cpu().reshape(-1), ref["z_std"].double().reshape(-1)
        return int((~((a - b).abs() <= tol + tol * b.abs())).sum())
    n_gpu = outside(got)
    res = {"keys": keys, "band": "|a - ref| <= 1e-4 + 1e-4 * |ref|", "rays": n,
           "gpu_rays_outside": n_gpu, "reference_self_sensitivity_rays_outside": n_self,
           "reference_self_sensitivity": "oracle/torch_port.py with its coarse network in fp64 (rounded to fp32 once) vs itself in fp32",
           "max_abs_raw0_minus_fp64": {"reference_fp32": float((ref["raw0"] - raw64).abs().max()),
                                       "gpu": float((got["raw0"].detach().float().cpu().reshape(raw64.shape) - raw64).abs().max())},
           "staged_port_reproduces_port": staged_ok}
    if "z_std" in ref and "z_std" in got:
        # z_std (std of the 128 importance samples) sits on the sampler's discontinuities and is in none of the image maps: its own
        # yardstick -- the reference against its fp64-coarse self, and the GPU against the reference
        res["z_std_rays_outside"] = {"gpu": z_std_outside(got["z_std"]), "reference_self": z_std_outside(self_maps["z_std"])}
    if inds_hip is not None:
        res["index_flip_rays"] = int((inds_hip.cpu().reshape(inds_ref.shape) != inds_ref)[:, :-1].any(-1).sum())
    return res


def gpu_bisect_indices(torch, out, rays, n_coarse, n_importance, near, far):
    """The right-bisect indices the GPU's importance sampler took for the render `out` (its own coarse weights), through the
    ABI's debug output -- what `parity_yardstick` compares with the port's searchsorted."""
    from nerf_sos_amd import ops
    n = rays.shape[1]
    dev = rays.device
    zc = ops.ray_setup(rays[1].reshape(-1, 3), torch.full((n,), float(near), device=dev), torch.full((n,), float(far), device=dev), n_coarse)[0]
    return ops.importance_sample(zc, out["weights0"].reshape(n, n_coarse), n_importance, debug=True)[4].cpu()


KERNEL_SOURCES = {   # traffic.json key -> the files whose content decides the dominant kernel's memory traffic
    "c2_fp32": ("mlp_fused.hip", "mlp_common.h"),
    "c2_fp16x3": ("mlp_x316.hip", "lp16_sched.h", "x316.h", "lp_common.h", "mlp_common.h"),
    "lp16": ("mlp_lp16.hip", "lp16_sched.h", "lp_common.h", "mlp_common.h"),
}
TRAFFIC_JSON = os.path.join(ROOT, "profiles", "r06", "traffic.json")


def kernel_source_hash(key: str = "c2_fp32") -> str:
    h = hashlib.sha256()
    for f in KERNEL_SOURCES.get(key, KERNEL_SOURCES["lp16"]):
        h.update(open(os.path.join(ROOT, "nerf-sos_amd", "csrc", f), "rb").read())
    return h.hexdigest()[:16]


def add_traffic(roof, key: str):
    """roofline.traffic = HBM bytes per launch of the dominant kernel from the PMC passes (profiles/r06/traffic.json: FETCH_SIZE
    and WRITE_SIZE collected in separate rocprofv3 --pmc runs and corrected as MI355X_MICROARCH.md prescribes), reported only
    while the kernel's sources still hash to the build the passes were measured on."""
    if roof is None or not os.path.exists(TRAFFIC_JSON):
        return roof
    t = json.load(open(TRAFFIC_JSON)).get("kernels", {}).get(key)
    if not t:
        return roof
    if t.get("kernel_source_sha16") == kernel_source_hash(key):
        roof["traffic"] = t["hbm_bytes_per_launch"]
        roof["traffic_detail"] = {"fetch_size_kb": t["fetch_size_kb"], "write_size_kb": t["write_size_kb"],
                                  "algorithmic_bytes_per_launch_without_weights": t["algorithmic_bytes_per_launch_without_weights"],
                                  "ratio_to_algorithmic": t["ratio"], "scratch_bytes": t.get("scratch_bytes"),
                                  "source": "profiles/r06/traffic.json (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, separate passes; "
                                            "2 x FETCH_SIZE + WRITE_SIZE)"}
    else:
        roof["traffic_note"] = "profiles/r06/traffic.json was measured on a different build of this kernel: not reported"
    return roof


TRAINED_CKPT = os.path.join(ROOT, "tests", "golden", "trained_scene.ckpt")
PEAK_HBM_TBPS = 8.0                          # MI355X_MICROARCH.md: HBM3E ~8 TB/s


def trained_field_parity(torch, dev, n_rays: int = 4096):
    """`parity.trained_field`: the checkpoint trained on the procedural scene (tests/golden/trained_scene.ckpt,
    scripts/make_trained_scene.py) rendered on `n_rays` pixels of the four held-out views by the exact fp32 kernels and by the CPU
    port (bit-identical to the reference on CPU, asserted on this very checkpoint by tests/golden/make_goldens_trained.py): the
    same per-key table and yardstick as the default-init and dense fields, plus every other precision against the PORT's render
    (PSNR, max-abs, label agreement).  Outside every timed region; ~10 s of host work."""
    import numpy as np
    import nerf_sos_amd
    from nerf_sos_amd import io as nio, ops, synthetic as syn
    from oracle import torch_port as tp            # the checker, never the product path
    scene = syn.ProceduralScene()
    net = nerf_sos_amd.NeRFNet(N_samples=N_COARSE, N_importance=N_IMPORTANCE, use_semantics=True, sem_with_coord=True).to(dev).eval()
    nio.load_checkpoint(TRAINED_CKPT, net)
    for p_ in net.parameters():
        p_.requires_grad_(False)
    g = torch.Generator().manual_seed(0)
    per_view = n_rays // len(scene.i_test)
    K = syn.intrinsics(scene.h, scene.w, scene.focal)
    rays, pix = [], []
    for i in scene.i_test:
        sel = torch.randperm(scene.h * scene.w, generator=g)[:per_view]
        full = ops.generate_rays(scene.h, scene.w, K, scene.poses[i, :3, :4], dev).reshape(2, -1, 3)
        rays.append(full[:, sel.to(dev)])
        pix.append((i, sel))
    rays = torch.cat(rays, 1).contiguous()
    bounds = (scene.NEAR, scene.FAR)
    outs = {}
    net.validate_precision = False
    modes = {"fp32": ("fp32", None), "fp16x3": ("fp16x3", None), "bf16": ("bf16", None), "fp16": ("fp16", None),
             "fp16_coarse_fp16x3": ("fp16", "fp16x3"), "bf16_coarse_fp16x3": ("bf16", "fp16x3")}
    with torch.no_grad():
        for name, (prec, coarse) in modes.items():
            net.mlp_precision, net.coarse_precision = prec, coarse
            outs[name] = {k: v.detach().clone() for k, v in net(rays, bounds, retraw=name == "fp32").items() if name == "fp32" or k in ("rgb", "depth", "semantics")}
    net.mlp_precision, net.coarse_precision = "fp32", None
    inds = gpu_bisect_indices(torch, outs["fp32"], rays, N_COARSE, N_IMPORTANCE, *bounds)
    torch.cuda.synchronize()
    cfg = tp.PortConfig(n_samples=N_COARSE, n_importance=N_IMPORTANCE, use_semantics=True, sem_with_coord=True, pts_chunk=1024 * 256)
    sd = {k: v.detach().cpu() for k, v in net.state_dict().items()}
    rays_c = rays.cpu()
    torch.set_num_threads(min(32, os.cpu_count() or 1))
    with torch.no_grad():
        ref = tp.render(sd, cfg, rays_c, bounds, retraw=True)
    res = render_parity(outs["fp32"], ref)
    res["yardstick"] = parity_yardstick(tp, sd, cfg, rays_c, ref, outs["fp32"], inds)
    gt = np.concatenate([scene.trace(*(t.double().cpu().numpy() for t in (rays[0][j * per_view:(j + 1) * per_view], rays[1][j * per_view:(j + 1) * per_view])))[0]
                         for j in range(len(scene.i_test))])
    psnr_gt = lambda a: round(float(-10 * np.log10(np.mean((a.detach().float().cpu().numpy() - gt) ** 2))), 2)  # noqa: E731
    res["field"] = {"what": "shipped architecture trained on synthetic.ProceduralScene by this package on an MI355X (8000 all-parameter steps "
                            "+ 1500 steps of the --fix_backbone head recipe; profiles/r05/a_trained_scene_training_log.json); rays: "
                            f"{per_view} random pixels of each of the {len(scene.i_test)} held-out views",
                    "reference_psnr_vs_analytic_gt_db": psnr_gt(ref["rgb"]), "gpu_fp32_psnr_vs_analytic_gt_db": psnr_gt(outs["fp32"]["rgb"]),
                    "mean_acc": round(float(ref["acc"].mean()), 4)}
    from nerf_sos_amd import quality
    lab_ref = ref["semantics"].argmax(-1)
    q = {}
    for name in modes:
        if name == "fp32":
            continue
        o = outs[name]
        st = quality.tail_stats(o["rgb"].cpu(), ref["rgb"], o["depth"].cpu(), ref["depth"], o["semantics"].cpu().argmax(-1), lab_ref)
        st["psnr_vs_analytic_gt_db"] = psnr_gt(o["rgb"])
        q[name] = st
    res["other_precisions_vs_reference"] = q
    res["tail_note"] = ("16-bit tail = silhouette rays whose importance samples move with the COARSE pass's 16-bit error (scripts/diag/lp_outliers.py, "
                        "profiles/r06/a_lp_outliers.json); `coarse_precision='fp16x3'` removes it (the *_coarse_fp16x3 rows)")
    return res


def hbm_kernel_rooflines(torch, dev, n_rays: int = 65536):
    """HBM rooflines of the path's memory-bound kernels at the C5 chunk (65 536 rays, sem+coord: 6 channels): algorithmic bytes
    (every input read once, every output written once) / mean HIP-event time of 20 back-to-back launches, against 8 TB/s.
    north_star: "evidenced by rocprof HBM GB/s" -- profiles/r05 holds the rocprofv3 kernel trace of the same launches."""
    from nerf_sos_amd import ops, synthetic as syn
    R = n_rays
    g = torch.Generator(device=dev).manual_seed(0)
    rays = syn.image_rays(dev, (0, R))
    d = rays[1].contiguous()
    near, far = torch.full((R,), syn.NEAR, device=dev), torch.full((R,), syn.FAR, device=dev)
    raw0 = torch.randn(R, N_COARSE, 6, device=dev, generator=g)
    raw1 = torch.randn(R, N_FINE, 6, device=dev, generator=g)
    z0, _ = ops.ray_setup(d, near, far, N_COARSE, None)
    _, z1, _, _ = ops.composite_importance(raw0, z0, d, N_IMPORTANCE)
    cases = {
        "ray_setup_kernel": (lambda: ops.ray_setup(d, near, far, N_COARSE, None), R * (12 + 8 + 4 * N_COARSE + 12)),
        "composite_importance_kernel": (lambda: ops.composite_importance(raw0, z0, d, N_IMPORTANCE),
                                        R * (4 * N_COARSE * 6 + 4 * N_COARSE + 12 + 4 * N_COARSE + 32 + 4 * N_FINE + 4 * N_IMPORTANCE + 4)),
        "composite_kernel<3>": (lambda: ops.composite(raw1, z1, d), R * (4 * N_FINE * 6 + 4 * N_FINE + 12 + 4 * N_FINE + 32)),
    }
    out = {}
    for name, (fn, nbytes) in cases.items():
        for _ in range(3):
            fn()
        us = float("inf")
        for _ in range(3):        # best of three blocks: inside the default run the first block has read 2.7 x slow (allocator / clock state
            torch.cuda.synchronize()   # left by the image renders before it; stand-alone all three agree: scripts/diag/hbm_check.py)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(20):
                fn()
            e1.record()
            torch.cuda.synchronize()
            us = min(us, e0.elapsed_time(e1) / 20 * 1e3)
        out[name] = {"bound": "hbm (VALU-issue-bound in fact: profiles/r06/d_hbm_kernels_pmc_summary.txt)", "achieved": round(nbytes / us / 1e6, 3), "peak": PEAK_HBM_TBPS, "unit": "TB/s",
                     "frac": round(nbytes / us / 1e6 / PEAK_HBM_TBPS, 4), "kernel_us": round(us, 1), "algorithmic_mb": round(nbytes / 1e6, 1)}
    return out


def c5_trained_quality(torch, dev, precision: str, chunk: int = 65536, coarse=None):
    """C5's `quality` on the TRAINED field: the full 1008x756 image of held-out pose 0 (the scene's field of view) in `precision`
    (coarse pass in `coarse`, if given: NeRFNet.coarse_precision) and through the exact fp32 kernels (= the reference within 1e-4 on
    this field, parity.trained_field): PSNR AND the tail -- percentiles, counts of rays over 0.01 / 0.05 in rgb, share within 0.02,
    relative depth, labels (nerf_sos_amd.quality.tail_stats) -- and both against the analytic image.  Outside the timed region."""
    import math
    import nerf_sos_amd
    from nerf_sos_amd import io as nio, ops, quality, synthetic as syn
    scene = syn.ProceduralScene()
    net = nerf_sos_amd.NeRFNet(N_samples=N_COARSE, N_importance=N_IMPORTANCE, use_semantics=True, sem_with_coord=True, ray_chunk=chunk).to(dev).eval()
    nio.load_checkpoint(TRAINED_CKPT, net)
    for p_ in net.parameters():
        p_.requires_grad_(False)
    i = scene.i_test[0]
    H, W = syn.H, syn.W
    focal = scene.focal * W / scene.w
    rays = ops.generate_rays(H, W, syn.intrinsics(H, W, focal), scene.poses[i, :3, :4], dev).reshape(2, -1, 3)
    res = {}
    with torch.no_grad():
        for prec, cp in ((precision, coarse), ("fp32", None)):
            net.mlp_precision, net.coarse_precision = prec, cp
            ret = net(rays, (scene.NEAR, scene.FAR), retraw=False)
            res[prec] = (ret["rgb"].clone(), ops.eval_postprocess(semantics=ret["semantics"])["sem"].clone(), ret["depth"].clone())
    gt_rgb, gt_lab, _ = scene.view(i, H, W)
    gt = torch.from_numpy(gt_rgb.reshape(-1, 3)).to(dev)
    lo, hi = res[precision], res["fp32"]
    psnr = lambda a: round(-10.0 * math.log10(max(float(((a - gt) ** 2).mean()), 1e-30)), 2)   # noqa: E731
    st = quality.tail_stats(lo[0], hi[0], lo[2], hi[2], lo[1], hi[1])
    st.update({"psnr_db_rgb_vs_exact_fp32": st["psnr_db"], "max_abs_rgb": st["abs_rgb"]["max"], "max_rel_depth": st["rel_depth"]["max"],
               "psnr_vs_analytic_image_db": {precision: psnr(lo[0]), "fp32": psnr(hi[0])}, "coarse_precision": coarse,
               "what": f"full {W}x{H} image of held-out pose {i} of the TRAINED procedural scene (tests/golden/trained_scene.ckpt), {precision}"
                       + (f" with the coarse pass in {coarse}" if coarse else "") + " vs the exact-fp32 kernels"})
    return st


# ------------------------------------------------------------------------------------------------------------------
class Ctx:
    def __init__(self, args):
        import torch
        import torch.distributed as dist
        self.torch, self.dist = torch, dist
        self.rank = int(os.environ.get("RANK", "0"))
        self.local_rank = int(os.environ.get("LOCAL_RANK", "0"))
        self.world = int(os.environ.get("WORLD_SIZE", "1"))
        if self.world != args.gpus:
            raise SystemExit(f"bench.py: --gpus {args.gpus} but WORLD_SIZE={self.world}")
        one_gpu = os.environ.get("NSOS_BENCH_SHARE_GPU") == "1"       # CI only: all ranks on cuda:0 over gloo
        n_dev = torch.cuda.device_count()
        if not one_gpu and n_dev < self.world:
            raise SystemExit(f"bench.py: --gpus {self.world} but only {n_dev} GPU(s) visible")
        self.dev = torch.device("cuda", 0 if one_gpu else self.local_rank)
        torch.cuda.set_device(self.dev)
        self.backend = None
        if self.world > 1:
            os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
            self.backend = "gloo" if one_gpu else "nccl"
            # a diverged collective sequence must fail, not hang: a bounded process-group timeout (RCCL's watchdog aborts the job
            # when a collective exceeds it) and the package's own per-collective wait (gloo honours it; sharding.collective)
            import datetime
            from nerf_sos_amd import sharding
            pg_timeout = datetime.timedelta(seconds=int(os.environ.get("NSOS_PG_TIMEOUT_S", "300")))
            sharding.COLLECTIVE_TIMEOUT_S = sharding.COLLECTIVE_TIMEOUT_S or 120.0
            if self.backend == "nccl":
                dist.init_process_group("nccl", rank=self.rank, world_size=self.world, device_id=self.dev, timeout=pg_timeout)
            else:
                dist.init_process_group("gloo", rank=self.rank, world_size=self.world, timeout=pg_timeout)

    def barrier(self):
        if self.world > 1:
            self.dist.barrier()
        self.torch.cuda.synchronize()

    def ranks_seen(self) -> int:
        """Number of ranks the collective library really connected: an all-reduce of ones on the GPU."""
        t = self.torch.ones(1, device=self.dev)
        if self.world > 1:
            self.dist.all_reduce(t)
        return int(t.item())

    def gather_times(self, dt: float):
        t = self.torch.tensor([dt], device=self.dev, dtype=self.torch.float64)
        if self.world == 1:
            return [dt]
        out = [self.torch.zeros_like(t) for _ in range(self.world)]
        self.dist.all_gather(out, t)
        return [float(x.item()) for x in out]


def timed(ctx, step, warmup: int, steps: int):
    """W untimed steps, then EXACTLY K steps between barrier + synchronize on both sides.  Returns
    (max-over-ranks seconds, per-rank seconds, MLP kernel events recorded inside the timed region)."""
    from nerf_sos_amd import ops
    import gc
    for i in range(warmup):
        step(i)
    # One c3 run of 20 steps came out at 2.45 ms per step (host share 1.93 ms) and repeated twice at 1.80 (1.16): a single
    # ~13 ms stall on the host inside a 40 ms timed region.  The interpreter's cyclic collector is one thing that can do that:
    # collect now and keep it out of the K steps (nothing is skipped: the steps create no reference cycles that need it).
    gc.collect()
    gc_was = gc.isenabled()
    gc.disable()
    ctx.barrier()
    ops.KERNEL_EVENTS = []
    t0 = time.perf_counter()
    for i in range(steps):
        step(warmup + i)
    ctx.host_enqueue_s = time.perf_counter() - t0     # the host's share: launches are asynchronous, the GPU drains behind it
    ctx.torch.cuda.synchronize()
    dt_own = time.perf_counter() - t0
    ctx.barrier()
    dt = time.perf_counter() - t0
    if gc_was:
        gc.enable()
    events, ops.KERNEL_EVENTS = ops.KERNEL_EVENTS, None
    per_rank = ctx.gather_times(dt_own)
    return max([dt] + per_rank) if ctx.world > 1 else dt, per_rank, events


def kernel_roofline(events, n_points: int, mac_per_point: int, peak: float, kernel: str, extra=None):
    """Dominant kernel = the fine-pass fused MLP launch: algorithmic FLOPs / mean HIP-event duration of those launches."""
    ms = [a.elapsed_time(b) for (n_pts, a, b) in events if n_pts == n_points]
    if not ms:
        return None
    mean_ms = sum(ms) / len(ms)
    achieved = 2.0 * mac_per_point * n_points / (mean_ms * 1e-3) / 1e12
    r = {"bound": "mfma", "achieved": round(achieved, 2), "peak": peak, "unit": "TFLOP/s",
         "frac": round(achieved / peak, 4), "traffic": None, "kernel": kernel, "kernel_ms": round(mean_ms, 4),
         "launches_timed": len(ms)}
    if extra:
        r.update(extra)
    return r


# profiles/r02/c_mfma_power.txt (scripts/ubench/mfma_power.hip): what a PURE stream of these MFMAs sustains on the whole chip
# with operands of which half are zero (what a ReLU network feeds the pipe) -- the chip clocks to its power budget, so the
# nominal 2.5 PFLOP/s (2.4 GHz) is not reachable on real data.  Reported next to `frac` (which stays against the nominal peak).
MEASURED_PIPE_CEILING_TFLOPS = {"fp16": 1881.0, "bf16": 1898.0}


_LP_CLOCK = {}


def lp_kernel_clock_ghz(torch, dev, precision: str, R: int):
    """Shader clock DURING the 16-bit MLP kernel: shader cycles of workgroup 0's first wave from its first to its last
    instruction (the diagnostics entry nsos_mlp_profile_rays_lp stamps them) over the launch's HIP-event duration, sem+coord
    net, R rays x 192 samples = the shape of the launch the roofline is quoted on (short launches clock lower than long ones).
    Outside every timed region; once per process, precision and shape."""
    if (precision, R) in _LP_CLOCK:
        return _LP_CLOCK[(precision, R)]
    import ctypes as C
    import nerf_sos_amd
    from nerf_sos_amd import _lib, ops, synthetic as syn
    net = nerf_sos_amd.NeRFNet(N_samples=N_COARSE, N_importance=N_IMPORTANCE, use_semantics=True, sem_with_coord=True).to(dev).eval()
    rays = syn.synthetic_rays(R, seed=0, device=dev)
    near, far = torch.full((R,), syn.NEAR, device=dev), torch.full((R,), syn.FAR, device=dev)
    z, v = ops.ray_setup(rays[1], near, far, N_FINE, None)
    packed = net.nerf_fine.packed_weights(precision)
    raw = torch.empty(R, N_FINE, 6, device=dev)
    stamps = torch.zeros(16 * 64, dtype=torch.int64, device=dev)
    o, d = rays[0].contiguous(), rays[1].contiguous()
    P = lambda t: C.c_void_p(t.data_ptr())   # noqa: E731
    st = C.c_void_p(torch.cuda.current_stream(dev).cuda_stream)

    def launch():
        _lib.check(_lib.lib().nsos_mlp_profile_rays_lp(P(packed), 2, {"fp16": 1, "bf16": 2}[precision], P(o), P(d), P(v), P(z), R, N_FINE,
                                                      P(raw), P(stamps), st), "nsos_mlp_profile_rays_lp")
    for _ in range(3):
        launch()
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
    ev[0].record()
    for _ in range(5):
        launch()
    ev[1].record()
    torch.cuda.synchronize()
    row = stamps.view(16, 64)[0].cpu()
    cycles = int(row[63] - row[62])
    _LP_CLOCK[(precision, R)] = round(cycles / (ev[0].elapsed_time(ev[1]) / 5) / 1e6, 3)
    return _LP_CLOCK[(precision, R)]


def x3_kernel_selected() -> int:
    """2 = mlp_x316_kernel (16x16x32, the default split-fp16 forward since round 6), 1 = mlp_x3_kernel (NSOS_X3_KERNEL=1)."""
    from nerf_sos_amd import _lib
    return int(_lib.lib().nsos_mlp_x3_selected_kernel())


def lp_kernel_name() -> str:
    """The 16-bit MLP kernel the library dispatches to (NSOS_LP_KERNEL / nsos_mlp_lp_select_kernel; default: mlp_lp16_kernel)."""
    k = os.environ.get("NSOS_LP_KERNEL", "")
    if os.environ.get("NSOS_LP_WAVES", "") == "4" or k == "lp4":
        return "mlp_lp_kernel"
    return "mlp_lp8_kernel" if k == "lp8" else "mlp_lp16_kernel"


def add_power_note(roof, precision, ctx=None, rays=4096):
    if roof and precision in MEASURED_PIPE_CEILING_TFLOPS:
        c = MEASURED_PIPE_CEILING_TFLOPS[precision]
        roof["measured_pipe_ceiling"] = {"tflops": c, "frac_of_it": round(roof["achieved"] / c, 4),
                                         "what": "pure MFMA stream, two waves per SIMD, random operands with half of them zero, whole chip "
                                                 "(power-limited clock); profiles/r02/c_mfma_power.txt"}
        if ctx is not None:
            try:
                ghz = lp_kernel_clock_ghz(ctx.torch, ctx.dev, precision, rays)
                roof["shader_clock_during_kernel"] = {
                    "ghz": ghz, "nominal_ghz": 2.4, "frac_of_peak_at_that_clock": round(roof["frac"] * 2.4 / ghz, 4),
                    "what": f"shader cycles of a whole launch of this kernel / its HIP-event duration ({rays} rays x 192 samples, outside "
                            "the timed region; non-SAVE variant): the chip runs the 16-bit matrix pipe at its power limit, not at 2.4 GHz; `frac` stays "
                            "against the nominal 2.5 PFLOP/s"}
            except Exception as e:   # diagnostics only: never fail the bench line over it
                roof["shader_clock_during_kernel"] = {"error": repr(e)}
    return roof


def timed_blocks(ctx, step, warmup: int, steps: int, blocks: int):
    """`blocks` timed regions of `steps` steps each (the variants' protocol: the power-limited 16-bit paths move by a few per
    cent from block to block, one block of 2-20 steps is thin for three digits): returns the MEDIAN block's (dt, per_rank), the
    kernel events of all blocks, and the per-block ms per step."""
    runs, events = [], []
    for b in range(blocks):
        dt, per_rank, ev = timed(ctx, step, warmup if b == 0 else 0, steps)
        runs.append((dt, per_rank))
        events += ev
    order = sorted(range(blocks), key=lambda i: runs[i][0])
    dt, per_rank = runs[order[blocks // 2]]
    ms = sorted(round(1e3 * r[0] / steps, 4) for r in runs)
    return dt, per_rank, events, {"blocks": blocks, "steps_per_block": steps, "ms_per_step_min": ms[0], "ms_per_step_median": ms[blocks // 2],
                                  "ms_per_step_max": ms[-1], "reported": "median block"}


def speed_fields(ctx, rays_per_rank_step: int, steps: int, dt: float, per_rank):
    rates = [rays_per_rank_step * steps / t for t in per_rank]
    res = {"value": round(ctx.world * rays_per_rank_step * steps / dt, 1), "unit": "rays/s",
           "ms_per_step": round(1e3 * dt / steps, 4),
           "host_enqueue_ms_per_step": round(1e3 * getattr(ctx, "host_enqueue_s", 0.0) / steps, 4),   # rank 0's Python + launch time
           "per_rank_rays_per_s": {"min": round(min(rates), 1), "max": round(max(rates), 1)}}
    if ctx.world > 1:     # every rank's host share: at 16-bit rates an 8-GPU step is bound by the slowest HOST, not by xGMI
        res["host_enqueue_ms_per_step_by_rank"] = [round(1e3 * t / steps, 4) for t in ctx.gather_times(getattr(ctx, "host_enqueue_s", 0.0))]
    return res


# ------------------------------------------------------------------------------------------------------------------ c2
def run_c2(ctx, args, precision="fp32", steps=None, warmup=None, blocks=1):
    import nerf_sos_amd
    from nerf_sos_amd import synthetic as syn
    torch = ctx.torch
    n_rays = 4096
    torch.manual_seed(0)
    net = nerf_sos_amd.NeRFNet(N_samples=N_COARSE, N_importance=N_IMPORTANCE, use_semantics=False,
                               perturb=1.0, raw_noise_std=1.0).to(ctx.dev).eval()
    net.mlp_precision = precision
    rays = syn.synthetic_rays(n_rays, seed=ctx.rank, device=ctx.dev)       # resident in HBM before the timed region
    out = {}

    def step(i):
        with torch.no_grad():
            out["ret"] = net(rays, (syn.NEAR, syn.FAR))

    k = steps or args.steps
    dt, per_rank, events, blk = timed_blocks(ctx, step, args.warmup if warmup is None else warmup, k, blocks)
    assert out["ret"]["rgb"].shape == (n_rays, 3)
    res = speed_fields(ctx, n_rays, k, dt, per_rank)
    if blocks > 1:
        res["timing_blocks"] = blk
    flop_per_ray = 2 * MAC_NOSEM * EVALS_PER_RAY
    if precision == "fp32":
        roof = kernel_roofline(events, n_rays * N_FINE, MAC_NOSEM, PEAK_FP32_MFMA_TFLOPS,
                               "mlp_fused_kernel<0,true> (fine pass, 786432 points)")
        peak = PEAK_FP32_MFMA_TFLOPS
    else:
        roof = kernel_roofline(events, n_rays * N_FINE, MAC_NOSEM, PEAK_16BIT_MFMA_TFLOPS,
                               ("mlp_x316_kernel<0>" if x3_kernel_selected() == 2 else "mlp_x3_kernel<0>") + " (fine pass, 786432 points)")
        roof["issued_frac"] = round(3 * roof["frac"], 4)   # three 16-bit MFMAs per product
        peak = PEAK_16BIT_MFMA_TFLOPS
    roof["whole_path_frac"] = round(res["value"] / ctx.world * flop_per_ray / 1e12 / peak, 4)
    # the same step replayed as ONE HIP graph (nerf_sos_amd.GraphedRender: the package's opt-in API for fixed-shape eval loops; its
    # outputs are static buffers, which is why NeRFNet.forward itself -- the timed headline -- stays eager): outside the timed region
    graphed = None
    if ctx.world == 1 and blocks == 1:
        try:
            gr = nerf_sos_amd.GraphedRender(net, n_rays, (syn.NEAR, syn.FAR))
            for _ in range(3):
                gr(rays)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(k):
                gr(rays)
            torch.cuda.synchronize()
            dtg = time.perf_counter() - t0
            same = bool(torch.equal(gr(rays)["rgb"], out["ret"]["rgb"]))
            graphed = {"ms_per_step": round(1e3 * dtg / k, 4), "rays_per_s": round(n_rays * k / dtg, 1), "bit_identical_to_eager": same,
                       "whole_path_frac": round(n_rays * k / dtg * flop_per_ray / 1e12 / peak, 4)}
            del gr
        except Exception as e:   # diagnostics only
            graphed = {"error": repr(e)[:200]}
    res.update(roofline=roof, flop_per_ray=flop_per_ray, rays_per_gpu=n_rays, out=out["ret"], net=net, rays=rays, replayed_as_hip_graph=graphed)
    return res


# ------------------------------------------------------------------------------------------------------------------ c1
def run_c1(ctx, args):
    """BASELINE configs[0]: the reference's own CPU-runnable case at its shape -- 1024 rays x 64 samples, coarse-only (N_importance
    = 0), no semantic head, eval-mode forward, fp32 -- on the GPU path; main() times the CPU port at the same shape beside it."""
    import nerf_sos_amd
    from nerf_sos_amd import synthetic as syn
    torch = ctx.torch
    n_rays = 1024
    torch.manual_seed(0)
    net = nerf_sos_amd.NeRFNet(N_samples=N_COARSE, N_importance=0, use_semantics=False, perturb=1.0, raw_noise_std=1.0).to(ctx.dev).eval()
    rays = syn.synthetic_rays(n_rays, seed=ctx.rank, device=ctx.dev)
    out = {}

    def step(i):
        with torch.no_grad():
            out["ret"] = net(rays, (syn.NEAR, syn.FAR))

    dt, per_rank, events = timed(ctx, step, args.warmup, args.steps)
    assert out["ret"]["rgb"].shape == (n_rays, 3) and "rgb0" not in out["ret"]
    res = speed_fields(ctx, n_rays, args.steps, dt, per_rank)
    roof = kernel_roofline(events, n_rays * N_COARSE, MAC_NOSEM, PEAK_FP32_MFMA_TFLOPS, "mlp_fused_kernel<0,true> (the only pass: 65536 points = 512 tiles on 256 CUs)")
    if roof:
        roof["whole_path_frac"] = round(res["value"] / ctx.world * 2 * MAC_NOSEM * N_COARSE / 1e12 / PEAK_FP32_MFMA_TFLOPS, 4)
    res.update(roofline=roof, flop_per_ray=2 * MAC_NOSEM * N_COARSE, rays_per_gpu=n_rays, out=out["ret"], net=net, rays=rays)
    return res


def cpu_baseline_c1(gpu):
    """The CPU port at BASELINE configs[0]'s own shape (1024 rays x 64 samples, coarse-only): 2 warm-ups, best of 5 at the thread
    count where torch's CPU kernels peak; the GPU render of the same rays / weights compared key by key."""
    import torch
    from oracle import torch_port as tp            # test infrastructure: the CPU baseline / checker, never the product path
    ncpu, phys = os.cpu_count() or 1, physical_cores()
    cfg = tp.PortConfig(n_samples=N_COARSE, n_importance=0, use_semantics=False, pts_chunk=1024 * 256)
    sd = {k: v.detach().cpu() for k, v in gpu[0].items()}
    rays = gpu[1].detach().cpu()
    threads = min(32, ncpu)
    torch.set_num_threads(threads)
    times = []
    with torch.no_grad():
        for k in range(7):
            t0 = time.perf_counter()
            ref = tp.render(sd, cfg, rays, (tp.NEAR, tp.FAR), retraw=True)
            if k >= 2:
                times.append(time.perf_counter() - t0)
    n = rays.shape[1]
    res = {"value": round(n / min(times), 1), "unit": "rays/s", "cores": threads, "kind": "port", "physical_cores": phys, "logical_cpus": ncpu,
           "runs": len(times), "warmups": 2, "median_rays_per_s": round(n / sorted(times)[len(times) // 2], 1),
           "sample": f"the full {n}-ray batch of BASELINE configs[0] (1024 rays x 64 samples, coarse-only; same rays and weights as the GPU step), "
                     f"eval-mode forward, fp32, best of {len(times)} after 2 warm-ups at {threads} threads, torch {torch.__version__} CPU ops"}
    par = render_parity(gpu[2], ref)
    par["vs"] = "oracle/torch_port.py on the host (bit-identical to the reference on CPU), same rays, same weights"
    return res, par


# ------------------------------------------------------------------------------------------------------------- c3 / c4
def _loss_args():
    import types
    # scripts/train_fortress_node0.sh: --use_correlation --use_geoCorr, app/geo parameters of the shipped recipe
    return types.SimpleNamespace(rand_neg=False, self_corr_w=0, use_sim_matrix=True, patch_stride=PATCH_STRIDE,
                                 app_corr_params=["0.18", "1", "0.46", "1"], geo_corr_params=["0.5", "1", "3", "1"])


def run_patch_training(ctx, args, patches_per_gpu: int, precision: str, steps: int, warmup: int, blocks: int = 1):
    """c3 (1 patch, 1 GPU) and c4 (2 patches per GPU, sharded): frozen-backbone training step on 64x64 patches."""
    import nerf_sos_amd
    from nerf_sos_amd import sharding, synthetic as syn
    torch = ctx.torch
    B = patches_per_gpu * ctx.world
    own = sharding.local_patches(B, ctx.rank, ctx.world)
    torch.manual_seed(0)                                          # same weights on every rank (one checkpoint in a real run)
    net = nerf_sos_amd.NeRFNet(N_samples=N_COARSE, N_importance=N_IMPORTANCE, use_semantics=True, sem_with_coord=True,
                               perturb=1.0, raw_noise_std=1.0, ray_chunk=1 << 20).to(ctx.dev)
    for n_, p_ in net.named_parameters():                         # run_nerf.py:307-318 (--fix_backbone)
        p_.requires_grad = "semantic_linear" in n_
    net.train()
    net.mlp_precision = precision
    net.rng, net.rng_seed = "philox", 1 + ctx.rank      # train-mode jitter / noise draws: one launch per chunk (ops.render_draws)
    # one fused update launch; the package re-packs trainable nets on every call, so updates that do not bump
    # Tensor._version (this one) still reach the kernels (tests/test_gpu_backward.py::test_optimizer_updates_reach_the_kernels)
    opt = torch.optim.Adam([p for p in net.parameters() if p.requires_grad], lr=5e-4, fused=True)
    corr, geo = nerf_sos_amd.CorrelationLoss(_loss_args()), nerf_sos_amd.GeoCorrelationLoss(_loss_args())
    all_rays = syn.synthetic_patches(B, PATCH, PATCH_STRIDE, seed=0, device=ctx.dev)
    rays = all_rays[:, own].contiguous()
    # DINO features / class tokens of the rank's OWN ground-truth crops (DINO itself is outside the path): random, seeded by
    # the global patch id, so that every world size sees the same batch
    feat = torch.stack([torch.randn(384, 14, 14, generator=torch.Generator().manual_seed(1000 + b)) for b in own]).to(ctx.dev)
    # class tokens of crops of ONE scene share a large common component (similarities all positive): without it the contrastive
    # loss -log(max / (max + min)) of uncorrelated tokens is the log of a negative number (the reference returns NaN there too)
    common = 3.0 * torch.randn(384, generator=torch.Generator().manual_seed(1999))
    cls_ = torch.stack([torch.randn(384, generator=torch.Generator().manual_seed(2000 + b)) + common for b in own]).to(ctx.dev)
    # NeRFContrastive (utils/image.py:192-218, engines/trainer.py:168-170) on the gathered class tokens; needs >= 2 patches
    contrast = nerf_sos_amd.NeRFContrastive(device=ctx.dev) if B >= 2 else None
    timings, state = {}, {}

    def step(i):
        opt.zero_grad(set_to_none=True)
        state["loss"] = sharding.sharded_patch_step(net, rays, (syn.NEAR, syn.FAR), B, feat, cls_, corr, geo,
                                                    correlation_w=1.0, geo_w=0.01, step=i, seed=0,
                                                    timings=timings if state.get("timed") else None,
                                                    contrast_loss=contrast, contrast_w=0.01)
        opt.step()

    for i in range(warmup):
        step(i)
    state["timed"] = True
    sharding.reset_collective_counts()
    dt, per_rank, events, blk = timed_blocks(ctx, step, 0, steps, blocks)
    per_step = {k: round(v / (steps * blocks), 3) for k, v in sorted(sharding.reset_collective_counts().items())}
    n_rays = len(own) * PATCH * PATCH
    res = speed_fields(ctx, n_rays, steps, dt, per_rank)
    if blocks > 1:
        res["timing_blocks"] = blk
    peak = PEAK_FP32_MFMA_TFLOPS if precision == "fp32" else PEAK_16BIT_MFMA_TFLOPS
    kname = {"fp32": "mlp_fused_kernel<2,true,1>", "fp16x3": "mlp_x3_kernel<2,1>"}.get(precision, f"{lp_kernel_name()}<{precision},2,SAVE>")
    roof = add_power_note(kernel_roofline(events, n_rays * N_FINE, MAC_SEMCOORD, peak, f"{kname} (fine pass of the rank's {n_rays} rays, {n_rays * N_FINE} points)"), precision, ctx, n_rays)
    flop_per_ray = 2 * MAC_SEMCOORD * EVALS_PER_RAY
    if roof:
        roof["whole_step_frac_forward_flops_only"] = round(res["value"] / ctx.world * flop_per_ray / 1e12 / peak, 4)

    def mean_ms(pairs):
        v = [a.elapsed_time(b) for a, b in pairs]
        return round(sum(v) / max(1, len(v)), 4)

    # The same step as ONE HIP graph (single process only: graphs.GraphedPatchStep -- device-side Philox counter, loss generator
    # registered with the graph, capturable fused Adam; bit-identical to the eager step, tests/test_gpu_configs.py): one
    # hipGraphLaunch per step, the host out of the loop.  Timed over `steps` replays, outside the eager timed region.
    # N > 1 (round 5): the step's four collectives are captured too when the group runs on RCCL; a gloo group (the one-GPU CI) or a
    # capture RCCL refuses falls back to the eager step inside GraphedPatchStep -- `captured` / `fallback` say which ran.
    # In THIS bench the N > 1 capture is opt-in (NSOS_BENCH_GRAPH_N=1): RCCL under capture could not be exercised on the builder's
    # one-GPU boxes, and a rank that hung here would take the scaling record of the whole line with it.  tests/test_gpu_sharded.py::
    # test_graphed_step_two_ranks_over_rccl is the place that answers it on the first multi-GPU box.
    graph = None
    if ctx.world > 1 and os.environ.get("NSOS_BENCH_GRAPH_N", "0") != "1":
        graph = {"skipped": "N > 1: set NSOS_BENCH_GRAPH_N=1 to time the captured sharded step (GraphedPatchStep(group=...))"}
    try:
        if graph is not None:
            raise StopIteration
        opt_g = torch.optim.Adam([p for p in net.parameters() if p.requires_grad], lr=5e-4, fused=True, capturable=True)
        g = nerf_sos_amd.GraphedPatchStep(net, opt_g, rays, (syn.NEAR, syn.FAR), feat, cls_, corr, geo, contrast, correlation_w=1.0,
                                          geo_w=0.01, contrast_w=0.01, seed=0, warmup=2, n_patches=B)
        for _ in range(3):
            g()
        ctx.barrier()
        t0 = time.perf_counter()
        for _ in range(steps):
            g()
        host = time.perf_counter() - t0
        torch.cuda.synchronize()
        ctx.barrier()
        dtg = time.perf_counter() - t0
        graph = {"ms_per_step": round(1e3 * dtg / steps, 4), "host_enqueue_ms_per_step": round(1e3 * host / steps, 4),
                 "rays_per_s": round(ctx.world * n_rays * steps / dtg, 1), "loss": round(float(g.loss), 6), "captured": g.graph is not None,
                 "what": "the whole step (render, losses, backward, Adam; N > 1: and its four collectives) captured once and replayed: GraphedPatchStep"}
        if g.capture_fallback:
            graph["fallback"] = g.capture_fallback
        if ctx.world > 1:
            graph["host_enqueue_ms_per_step_by_rank"] = [round(1e3 * t / steps, 4) for t in ctx.gather_times(host)]
        sharding.reset_collective_counts()
    except StopIteration:
        pass
    except Exception as e:   # the eager numbers above stand on their own
        graph = {"error": repr(e)[:300]}
        if ctx.world > 1:
            raise                # (a rank that dropped out of the collective sequence would hang the others: fail loudly instead)
    # diagnostics, outside the timed regions: three more eager steps with every collective bracketed by HIP events and host clocks
    # (sharding.COLLECTIVE_EVENTS) and the optimizer bracketed here -- what the first real multi-GPU run needs to be read from one line
    sharding.COLLECTIVE_EVENTS = {}
    diag_t, opt_ev = {}, []
    for i in range(3):
        opt.zero_grad(set_to_none=True)
        sharding.sharded_patch_step(net, rays, (syn.NEAR, syn.FAR), B, feat, cls_, corr, geo, correlation_w=1.0, geo_w=0.01, step=10_000 + i,
                                    seed=0, timings=diag_t, contrast_loss=contrast, contrast_w=0.01)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        opt.step()
        e1.record()
        opt_ev.append((e0, e1))
    torch.cuda.synchronize()
    coll_ev, sharding.COLLECTIVE_EVENTS = sharding.COLLECTIVE_EVENTS, None
    sharding.reset_collective_counts()
    by_kind = {k: {"gpu_ms": round(sum(a.elapsed_time(b) for a, b, _ in v) / len(v), 4), "host_ms": round(1e3 * sum(h for _, _, h in v) / len(v), 4),
                   "calls": len(v) // 3} for k, v in sorted(coll_ev.items())}
    breakdown = {k: mean_ms(diag_t.get(k, [])) for k in ("render", "gather", "losses_backward", "allreduce")}
    breakdown["optimizer"] = mean_ms(opt_ev)
    st = timings.get("stats", {})
    res.update(step_breakdown_ms=dict(breakdown, what="HIP-event spans on rank 0's stream, mean of 3 eager steps outside the timed regions: train-mode "
                                                      "render of the own patches | flat patch all-gather | losses + backward (N > 1: with the two loss "
                                                      "reductions inside) | flat gradient all-reduce | fused Adam"),
               collective_ms_by_kind=by_kind)
    res.update(roofline=roof, whole_step_graph=graph, rays_per_gpu=n_rays, patches=B, loss=round(float(state["loss"]), 6), precision=precision,
               contrastive_loss=("NeRFContrastive on the batch's class tokens, weight 0.01" if contrast is not None else
                                 "not evaluated: one patch has no off-diagonal similarity (the reference's argmin fails on B = 1)"),
               collectives={"backend": ctx.backend, "all_gather_ms": mean_ms(timings.get("gather", [])),
                            "all_reduce_ms": mean_ms(timings.get("allreduce", [])),
                            "gathered_bytes_per_patch": st.get("bytes_per_patch"),
                            "gathered_keys": list(sharding.PATCH_KEYS), "all_gathers_per_step": st.get("collectives", 0 if ctx.world == 1 else None),
                            # every collective this rank issued in the timed steps, by kind (sharding.collective counts them): the
                            # flat patch gather, the row-partitioned losses' two reductions (means, sums), the gradient all-reduce
                            "calls_per_step_by_kind": per_step, "calls_per_step": round(sum(per_step.values()), 3),
                            "all_reduce_floats": sum(p.numel() for p in net.parameters() if p.requires_grad)})
    return res


# ------------------------------------------------------------------------------------------------------------------ c5
def run_c5(ctx, args, precision: str, steps: int, warmup: int, blocks: int = 1, coarse=None):
    """Full-image eval: every rank renders its contiguous block of the 762 048 rays in 65 536-ray chunks (rays generated
    on device from the pose, `raw` never materialised) and post-processes its rows on device."""
    import nerf_sos_amd
    from nerf_sos_amd import ops, sharding, synthetic as syn
    torch = ctx.torch
    chunk = 65536
    torch.manual_seed(0)
    net = nerf_sos_amd.NeRFNet(N_samples=N_COARSE, N_importance=N_IMPORTANCE, use_semantics=True, sem_with_coord=True,
                               perturb=1.0, raw_noise_std=1.0, ray_chunk=chunk).to(ctx.dev).eval()
    net.mlp_precision, net.coarse_precision = precision, coarse
    net.validate_precision = False                         # (random-init weights, far inside fp16's range; no extra render in the timed loop)
    s, e = sharding.shard_bounds(syn.H * syn.W, ctx.rank, ctx.world)
    state = {}

    def step(i):
        with torch.no_grad():
            rays = syn.image_rays(ctx.dev, (s, e))
            ret = net(rays, (syn.NEAR, syn.FAR), retraw=False)
            state["post"] = ops.eval_postprocess(semantics=ret["semantics"])
            state["rgb"] = ret["rgb"]

    dt, per_rank, events, blk = timed_blocks(ctx, step, warmup, steps, blocks)
    n_rays = e - s
    res = speed_fields(ctx, n_rays, steps, dt, per_rank)
    if blocks > 1:
        res["timing_blocks"] = blk
    # the total over ranks is the image, not world x the first rank's block
    res["value"] = round(syn.H * syn.W * steps / dt, 1)
    peak = PEAK_FP32_MFMA_TFLOPS if precision == "fp32" else PEAK_16BIT_MFMA_TFLOPS
    full_chunk = min(chunk, n_rays)
    roof = add_power_note(kernel_roofline(events, full_chunk * N_FINE, MAC_SEMCOORD, peak,
                                          f"{lp_kernel_name()}<{precision},2> (fine pass of a {full_chunk}-ray chunk, {full_chunk * N_FINE} points)"), precision, ctx, full_chunk)
    flop_per_ray = 2 * MAC_SEMCOORD * EVALS_PER_RAY
    if roof:
        roof["whole_path_frac"] = round(res["value"] / ctx.world * flop_per_ray / 1e12 / peak, 4)
    # quality of the 16-bit render ("PSNR vs ref" for this configuration): one 65 536-ray chunk through a DENSE field (the seed-0
    # weights with the sigma head x40, -1.5: the default-init field is empty and its PSNR says nothing) in this precision and
    # through the exact-fp32 kernels (= the reference within 1e-4, tests/test_gpu_parity.py); outside the timed region
    quality = None
    if precision != "fp32" and ctx.rank == 0:
        rays = syn.image_rays(ctx.dev, (s, min(s + chunk, e)))
        field = make_dense_field(net, rays, (syn.NEAR, syn.FAR))
        with torch.no_grad():
            lo = net(rays, (syn.NEAR, syn.FAR), retraw=False)
            net.mlp_precision, net.coarse_precision = "fp32", None
            hi = net(rays, (syn.NEAR, syn.FAR), retraw=False)
            net.mlp_precision, net.coarse_precision = precision, coarse
            mse = float(((lo["rgb"] - hi["rgb"]) ** 2).mean())
            agree = float((ops.eval_postprocess(semantics=lo["semantics"])["sem"] == ops.eval_postprocess(semantics=hi["semantics"])["sem"]).float().mean())
            import math
            quality = {"psnr_db_rgb_vs_exact_fp32": round(-10.0 * math.log10(max(mse, 1e-30)), 2),
                       "max_abs_rgb": float((lo["rgb"] - hi["rgb"]).abs().max()), "label_agreement": round(agree, 5),
                       "mean_acc": round(float(hi["acc"].mean()), 4), "field": field,
                       "what": f"{rays.shape[1]} rays of the image through a dense field, {precision} vs the exact-fp32 kernels"}
    res.update(roofline=roof, rays_per_gpu=n_rays, precision=precision, coarse_precision=coarse, image=f"{syn.W}x{syn.H}", chunk=chunk,
               finite=bool(torch.isfinite(state["rgb"]).all().item()), quality_dense_random_field=quality)
    if ctx.rank == 0:
        del net
        torch.cuda.empty_cache()
        if precision != "fp32":
            res["quality"] = c5_trained_quality(torch, ctx.dev, precision, chunk, coarse)
        if coarse is None:
            res["hbm_kernels"] = hbm_kernel_rooflines(torch, ctx.dev, chunk)
    return res


def compact_line(line: dict) -> dict:
    """The ONE line of the contract, under 6 KB: the headline fields in full, `roofline` and `cpu_baseline` with their required
    keys, one short summary per parity field and per variant.  Everything else (per-key tables, timing blocks, traffic detail,
    step breakdowns, the long descriptions) is in the `bench_detail` line printed just before and in ./bench_detail.json."""
    def pick(d, keys):
        return {k: d[k] for k in keys if d is not None and k in d and d[k] is not None}

    def roof(r):
        if not r:
            return None
        c = pick(r, ("bound", "achieved", "peak", "unit", "frac", "traffic", "kernel_ms"))
        c["kernel"] = str(r.get("kernel", ""))[:44]
        if "traffic_detail" in r:
            c["traffic_ratio_to_algorithmic"] = r["traffic_detail"].get("ratio_to_algorithmic")
        for k in ("whole_path_frac", "whole_step_frac_forward_flops_only", "issued_frac"):
            if k in r:
                c[k] = r[k]
        c.setdefault("traffic", None)
        return c

    out = pick(line, ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline",
                      "dtype", "data", "host_enqueue_ms_per_step", "finite", "loss"))
    out["vs_baseline"] = line.get("vs_baseline")          # (null: BASELINE.md holds no published number for this metric)
    cfg = dict(line.get("config", {}))
    cfg["workload"] = str(cfg.get("workload", ""))[:230]
    out["config"] = cfg
    out["roofline"] = roof(line.get("roofline"))
    cb = line.get("cpu_baseline")
    if cb:
        c = pick(cb, ("value", "unit", "cores", "kind", "physical_cores"))
        c["sample"] = str(cb.get("sample", ""))[:150]
        if "all_physical_cores" in cb:
            c["all_physical_cores_rays_per_s"] = cb["all_physical_cores"].get("value")
        if "frozen_recipe_train_step" in cb:
            c["frozen_recipe_train_step_rays_per_s"] = cb["frozen_recipe_train_step"].get("value")
        if cb.get("value"):
            c["gpu_over_cpu"] = round(line["value"] / cb["value"], 1)
        out["cpu_baseline"] = c
    par = line.get("parity")
    if par:
        cp = {}
        for name, f in par.items():
            if not isinstance(f, dict) or "per_key" not in f:
                continue
            y = f.get("yardstick", {})
            e = {"psnr_db_rgb": f["psnr_db"].get("rgb"), "coarse_all_inside_1e-4": f["coarse_pass_all_rays_inside_1e-4"],
                 "frac_rays_outside_1e-4_any_fine_map": f["frac_rays_outside_1e-4_any_fine_map"],
                 "frac_rays_outside_1e-4_image_maps": f.get("frac_rays_outside_1e-4_image_maps"),
                 "gpu_rays_outside": y.get("gpu_rays_outside"), "reference_self_rays_outside": y.get("reference_self_sensitivity_rays_outside"),
                 "index_flip_rays": y.get("index_flip_rays"), "rays": y.get("rays"), "z_std_rays_outside": y.get("z_std_rays_outside"),
                 "max_abs_raw0_minus_fp64": {k_: float(f"{v_:.3g}") for k_, v_ in (y.get("max_abs_raw0_minus_fp64") or {}).items()}}
            if "other_precisions_vs_reference" in f:
                from nerf_sos_amd import quality
                e["vs_reference"] = {k: quality.compact(v, brief=True) for k, v in f["other_precisions_vs_reference"].items()}
                for row in e["vs_reference"].values():
                    row.pop("within_0.02", None)
                e["vs_reference"]["rays"] = y.get("rays")
                e["reference_psnr_vs_analytic_gt_db"] = f["field"].get("reference_psnr_vs_analytic_gt_db")
            cp[name] = e
        out["parity"] = cp
    if line.get("variants"):
        cv = {}
        for name, v in line["variants"].items():
            if "value" not in v:                       # generic_paths_fp32: two sub-records
                cv[name] = {k: pick(x, ("ms_per_step", "rays_per_s", "frac_of_fp32_mfma_peak_over_6_mac_per_weight_and_point", "ms_per_step_loss_on_both_maps")) for k, x in v.items() if isinstance(x, dict)}
                continue
            e = pick(v, ("value", "ms_per_step", "host_enqueue_ms_per_step", "host_enqueue_ms_per_step_by_rank", "loss", "max_abs_rgb0_vs_exact_fp32"))
            if v.get("finite") is False:
                e["finite"] = False
            e["roofline"] = pick(roof(v.get("roofline")) or {}, ("frac", "issued_frac", "kernel_ms", "kernel", "traffic", "whole_path_frac", "whole_step_frac_forward_flops_only"))
            if "timing_blocks" in v:
                e["ms_per_step_min_max"] = [v["timing_blocks"]["ms_per_step_min"], v["timing_blocks"]["ms_per_step_max"]]
            g = v.get("whole_step_graph")
            if isinstance(g, dict):
                e["replayed"] = pick(g, ("ms_per_step", "rays_per_s", "host_enqueue_ms_per_step"))
            if isinstance(v.get("quality"), dict):
                from nerf_sos_amd import quality
                e["quality_trained_field"] = dict(quality.compact(v["quality"]), psnr_vs_analytic_image_db=v["quality"].get("psnr_vs_analytic_image_db"))
                for drop in ("p99.9", "depth_n_gt_0.01"):
                    e["quality_trained_field"].pop(drop, None)
            if isinstance(v.get("hbm_kernels"), dict):
                e["hbm_kernels"] = {k: pick(x, ("achieved", "frac", "kernel_us")) for k, x in v["hbm_kernels"].items()}
            cv[name] = e
        out["variants"] = cv
    for k in ("quality", "hbm_kernels", "collectives", "whole_step_graph", "replayed_as_hip_graph"):
        if k in line and line[k] is not None:
            if k == "quality":
                from nerf_sos_amd import quality
                out[k] = dict(quality.compact(line[k]), psnr_vs_analytic_image_db=line[k].get("psnr_vs_analytic_image_db"))
            else:
                out[k] = line[k]
    d = line.get("distributed", {})
    out["distributed"] = pick(d, ("backend", "ranks_seen_by_collective", "calls_per_step_by_kind", "step_breakdown_ms", "host_enqueue_ms_per_step_by_rank"))
    out["detail"] = "full tables: the stdout line before this one ({\"bench_detail\": ...}) and ./bench_detail.json"
    return out


def _strip(res):
    return {k: v for k, v in res.items() if k not in ("out", "net", "rays") and not (k == "replayed_as_hip_graph" and v is None)}


# ---------------------------------------------------------------------------------------------------------------- main
def run_generic_paths(ctx, rays_n: int = 4096, steps: int = 10):
    """The paths no BASELINE config names but the drop-in contract includes, timed briefly on the generic fp32 kernels (csrc/mlp_generic.hip):
    a FULL training step of a non-shipped architecture (8 x 256 without view directions: render in train mode, MSE on rgb + rgb0,
    backward through both nets, Adam) and a pose-refinement step on the shipped architecture (rays require grad, frozen net)."""
    import time
    import torch
    import nerf_sos_amd
    from nerf_sos_amd import synthetic as syn
    dev = ctx.dev
    out = {"what": "generic-architecture training (any ctor kwargs of models/nerf_mlp.py:40-64) and gradients w.r.t. the rays, 4096 rays x (64 + 128), "
                   "exact-fp32 MFMA; whole steps, wall clock between synchronisations"}
    rays = syn.synthetic_rays(rays_n, seed=0, device=dev)
    gt = torch.rand(rays_n, 3, device=dev)

    def clock(step):
        for _ in range(2):
            step()
        torch.cuda.synchronize(dev)
        t0 = time.perf_counter()
        for _ in range(steps):
            step()
        torch.cuda.synchronize(dev)
        return (time.perf_counter() - t0) / steps * 1e3

    torch.manual_seed(0)
    net = nerf_sos_amd.NeRFNet(N_samples=64, N_importance=128, perturb=1.0, raw_noise_std=1.0, viewdirs=False).to(dev).train()
    opt = torch.optim.Adam(net.parameters(), lr=5e-4)
    macs = sum(p.numel() for n, p in net.nerf.named_parameters() if n.endswith("weight")) * 64 + \
        sum(p.numel() for n, p in net.nerf_fine.named_parameters() if n.endswith("weight")) * 192

    def train_step():
        opt.zero_grad()
        ret = net(rays, (syn.NEAR, syn.FAR), retraw=False)
        (((ret["rgb"] - gt) ** 2).mean() + ((ret["rgb0"] - gt) ** 2).mean()).backward()
        opt.step()

    ms = clock(train_step)
    out["train_8x256_no_viewdirs"] = {"ms_per_step": round(ms, 3), "rays_per_s": round(rays_n / ms * 1e3),
                                      "frac_of_fp32_mfma_peak_over_6_mac_per_weight_and_point": round(6 * macs * rays_n / ms / 1e9 / 157.3, 3)}
    del net, opt
    net = nerf_sos_amd.NeRFNet(N_samples=64, N_importance=128, use_semantics=True, sem_with_coord=True).to(dev).eval()
    for p_ in net.parameters():
        p_.requires_grad_(False)

    def pose_step(both=False):
        r = rays.clone().requires_grad_(True)
        ret = net(r, (syn.NEAR, syn.FAR), retraw=False)
        loss = ((ret["rgb"] - gt) ** 2).mean()
        if both:
            loss = loss + ((ret["rgb0"] - gt) ** 2).mean()
        loss.backward()
        return r.grad

    # the loss on the fine map (this entry's definition since round 4): the coarse net gets no gradient -- its samples are detached,
    # models/sampler.py:159 -- and since round 5 is not differentiated; `ms_per_step_loss_on_both_maps`: both networks' chains run
    ms = clock(pose_step)
    ms2 = clock(lambda: pose_step(True))
    g = pose_step()
    out["pose_step_shipped_architecture"] = {"ms_per_step": round(ms, 3), "rays_per_s": round(rays_n / ms * 1e3), "ms_per_step_loss_on_both_maps": round(ms2, 3),
                                             "g_rays_finite": bool(torch.isfinite(g).all())}
    del net
    torch.cuda.empty_cache()
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--config", choices=("c1", "c2", "c3", "c4", "c5"), default="c2",
                    help="BASELINE.json configs[0..4]; c2 is the headline the metric is quoted on")
    ap.add_argument("--precision", choices=("fp32", "fp16x3", "bf16", "fp16"), default=None,
                    help="MLP arithmetic; default: fp32 for c2 (the reference's), bf16 for c3/c4, fp16 for c5 (BASELINE's dtypes)")
    ap.add_argument("--coarse-precision", choices=("fp32", "fp16x3", "bf16", "fp16"), default=None,
                    help="c5 only: NeRFNet.coarse_precision (the coarse pass's arithmetic; default: the same as --precision)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-variants", action="store_true")
    args = ap.parse_args()
    _self_launch(args)

    ctx = Ctx(args)
    torch = ctx.torch
    seen = ctx.ranks_seen()
    prec = args.precision or {"c1": "fp32", "c2": "fp32", "c3": "bf16", "c4": "bf16", "c5": "fp16"}[args.config]
    dist_info = {"backend": ctx.backend, "ranks_seen_by_collective": seen}

    line = {"metric": "rays/sec (coarse+fine, 64+128 samples)", "unit": "rays/s", "n_gpus": ctx.world, "steps": args.steps,
            "warmup": args.warmup, "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "data": "synthetic"}
    variants = {}

    if args.config == "c1":
        if prec != "fp32":
            raise SystemExit("bench.py: c1 is the reference's own fp32 CPU case: --precision fp32")
        res = run_c1(ctx, args)
        line.update({k: res[k] for k in ("value", "ms_per_step", "host_enqueue_ms_per_step", "per_rank_rays_per_s")})
        line["metric"] = "rays/sec (coarse only, 64 samples)"
        line["dtype"] = "f32"
        line["config"] = {"workload": "BASELINE configs[0]: LLFF flower shape, 1024 rays/GPU x 64 samples, coarse-only MLP (N_importance = 0), eval-mode "
                                      "NeRFNet.forward, fp32 exact-MFMA, no semantic head, random-init weights (seed 0), pinhole rays 1008x756 f=850",
                          "rays_per_gpu": res["rays_per_gpu"], "parallelism": f"ray-sharded x{ctx.world}, no collective in the path", "flop_per_ray": res["flop_per_ray"]}
        line["roofline"] = res["roofline"]
    elif args.config == "c2":
        if prec not in ("fp32", "fp16x3"):
            raise SystemExit("bench.py: c2 is the fp32 configuration: --precision fp32 (exact) or fp16x3 (split-fp16, fp32-grade)")
        res = run_c2(ctx, args, prec)
        roof = res["roofline"]
        add_traffic(roof, "c2_fp32" if prec == "fp32" else "c2_fp16x3")
        line.update({k: res[k] for k in ("value", "ms_per_step", "host_enqueue_ms_per_step", "per_rank_rays_per_s")})
        line["dtype"] = "f32" if prec == "fp32" else "f16x3 (split-fp16 operands, fp32 accumulate, fp32-grade results)"
        line["config"] = {"workload": "BASELINE configs[1]: LLFF flower_full shape, 4096 rays/GPU x (64 coarse + 192 fine MLP "
                                      "evaluations), eval-mode NeRFNet.forward, " +
                                      ("fp32 exact-MFMA" if prec == "fp32" else "split-fp16 MFMA (fp32-grade)") +
                                      ", no semantic head, random-init weights (seed 0), pinhole rays 1008x756 f=850",
                          "rays_per_gpu": res["rays_per_gpu"], "parallelism": f"ray-sharded x{ctx.world}, no collective in the path",
                          "flop_per_ray": res["flop_per_ray"]}
        line["roofline"] = roof
        line["replayed_as_hip_graph"] = res.get("replayed_as_hip_graph")
        if not args.no_variants:
            vsteps = max(3, min(args.steps, 20))      # the variants ride along briefly, outside the headline's timed region
            if ctx.world == 1 and prec == "fp32":
                exact_c = res["out"]["rgb0"]
                alt = run_c2(ctx, args, "fp16x3", steps=vsteps, warmup=3, blocks=5)
                v = _strip(alt)
                v["what"] = "the c2 step with the MLP on the 16-bit matrix pipe, split-fp16 operands (3 MFMAs per product, fp32 accumulate)"
                v["max_abs_rgb0_vs_exact_fp32"] = float((alt["out"]["rgb0"] - exact_c).abs().max())
                v["parity"] = "same tests and bars as the exact kernel (2e-5 vs the reference goldens)"
                add_traffic(v.get("roofline"), "c2_fp16x3")
                variants["c2_fp16x3"] = v
                v = _strip(run_patch_training(ctx, args, 1, "bf16", vsteps, 6, blocks=5))
                v["what"] = "BASELINE configs[2]: one 64x64 patch = 4096 rays, sem+coord head, bf16 MFMA: train-mode render + appearance & geometric correlation losses + semantic-head backward + Adam"
                add_traffic(v.get("roofline"), "c3_bf16")
                variants["c3_bf16"] = v
            if ctx.world == 1:
                variants["generic_paths_fp32"] = run_generic_paths(ctx)
            v = _strip(run_c5(ctx, args, "fp16", 2 if ctx.world == 1 else 4, 1, blocks=5))
            v["what"] = ("BASELINE configs[4]: full 1008x756 image, 65536-ray chunks, fp16 MFMA, sem+coord, rays generated on device, on-device "
                         "softmax/argmax; row blocks sharded over the GPUs (strong scaling, no collective); a step = one image")
            add_traffic(v.get("roofline"), "c5_fp16")
            variants["c5_fp16"] = v
            if ctx.world == 1:
                v = _strip(run_c5(ctx, args, "fp16", 2, 1, blocks=3, coarse="fp16x3"))
                v["what"] = ("c5 with NeRFNet.coarse_precision = 'fp16x3': the coarse pass (1/4 of the points) on the split-fp16 kernel (fp32-grade), the fine "
                             "pass on the fp16 kernel -- removes the 16-bit tail on a trained field (quality_trained_field) at this cost")
                variants["c5_fp16_coarse_fp16x3"] = v
            v = _strip(run_patch_training(ctx, args, 2, "bf16", vsteps, 6, blocks=5))
            v["what"] = ("BASELINE configs[3]: 8192 rays/GPU (2 patches of 64x64 per GPU), the c3 step sharded over the GPUs: one flat "
                         "all-gather of semantics0/semantics/depth/feat/cls_/ray_o/ray_d, one flat gradient all-reduce")
            add_traffic(v.get("roofline"), "c4_bf16")
            variants["c4_bf16"] = v
    elif args.config in ("c3", "c4"):
        if args.config == "c3" and ctx.world != 1:
            raise SystemExit("bench.py: c3 is the single-GPU configuration; the sharded one is c4")
        res = _strip(run_patch_training(ctx, args, 1 if args.config == "c3" else 2, prec, args.steps, args.warmup))
        line.update({k: res[k] for k in ("value", "ms_per_step", "host_enqueue_ms_per_step", "per_rank_rays_per_s")})
        line["dtype"] = {"fp32": "f32", "fp16x3": "f16x3"}.get(prec, prec)
        line["config"] = {"workload": ("BASELINE configs[2]: 4096 rays = one 64x64 patch" if args.config == "c3" else
                                       "BASELINE configs[3]: 8192 rays/GPU = two 64x64 patches per GPU, patch batch sharded over the GPUs "
                                       "(RCCL all-gather of the patch tensors + gradient all-reduce)") +
                                      f", (64+192) MLP evaluations/ray, sem+coord head, {prec} MLP, train mode, appearance + geometric "
                                      "correlation losses, semantic-head backward (--fix_backbone recipe), Adam; train-mode draws from the package's one-launch Philox stream",
                          "rays_per_gpu": res["rays_per_gpu"], "patches": res["patches"], "parallelism": f"patch-sharded x{ctx.world}",
                          "flop_per_ray_forward": 2 * MAC_SEMCOORD * EVALS_PER_RAY}
        line["roofline"] = add_traffic(res["roofline"], "c3_bf16" if args.config == "c3" else "c4_bf16") if prec == "bf16" else res["roofline"]
        line["collectives"] = res["collectives"]
        dist_info.update(collective_ms_by_kind=res.get("collective_ms_by_kind"), step_breakdown_ms=res.get("step_breakdown_ms"),
                         calls_per_step_by_kind=res["collectives"]["calls_per_step_by_kind"],
                         host_enqueue_ms_per_step_by_rank=res.get("host_enqueue_ms_per_step_by_rank"))
        line["loss"] = res["loss"]
        line["whole_step_graph"] = res.get("whole_step_graph")
        line["contrastive_loss"] = res.get("contrastive_loss")
    else:
        res = _strip(run_c5(ctx, args, prec, args.steps, args.warmup, coarse=args.coarse_precision))
        line.update({k: res[k] for k in ("value", "ms_per_step", "host_enqueue_ms_per_step", "per_rank_rays_per_s")})
        line["scaling"] = "strong"
        line["dtype"] = {"fp32": "f32", "fp16x3": "f16x3"}.get(prec, prec)
        line["config"] = {"workload": f"BASELINE configs[4]: full-image render {res['image']} = 762048 rays in {res['chunk']}-ray chunks, "
                                      f"eval mode, sem+coord head, {prec} MLP" + (f" (coarse pass {args.coarse_precision})" if args.coarse_precision else "") + f", rays generated on device, on-device post-processing; step = one image",
                          "rays_per_gpu": res["rays_per_gpu"], "parallelism": f"row blocks sharded x{ctx.world}, no collective",
                          "flop_per_ray": 2 * MAC_SEMCOORD * EVALS_PER_RAY}
        line["roofline"] = add_traffic(res["roofline"], "c5_fp16") if prec == "fp16" else res["roofline"]
        line["finite"] = res["finite"]
        line["quality"] = res.get("quality")

    line["distributed"] = dist_info
    if variants:
        line["variants"] = variants
    if ctx.rank == 0:
        if ctx.world == 1 and not args.no_cpu_baseline and args.config == "c1":
            net = res["net"]
            line["cpu_baseline"], line["parity"] = cpu_baseline_c1(({k: v.detach().clone() for k, v in net.state_dict().items()}, res["rays"], res["out"]))
            for k in ("net", "rays", "out"):
                res.pop(k, None)
        if ctx.world == 1 and not args.no_cpu_baseline and args.config == "c2":
            gpu = dense = None
            if prec in ("fp32", "fp16x3"):
                net = res["net"]
                gpu = ({k: v.detach().clone() for k, v in net.state_dict().items()}, res["rays"], res["out"],
                       gpu_bisect_indices(torch, res["out"], res["rays"], N_COARSE, N_IMPORTANCE, 1.2, 14.72))
                field = make_dense_field(net, res["rays"], (1.2, 14.72))     # the same rays through a dense field, outside every timed region
                with torch.no_grad():
                    out_dense = net(res["rays"], (1.2, 14.72))
                dense = ({k: v.detach().clone() for k, v in net.state_dict().items()}, out_dense, field,
                         gpu_bisect_indices(torch, out_dense, res["rays"], N_COARSE, N_IMPORTANCE, 1.2, 14.72))
                torch.cuda.synchronize()
            line["cpu_baseline"], parity = cpu_baseline(gpu, dense=dense)
            if parity is not None:
                line["parity"] = parity
                if os.path.exists(TRAINED_CKPT) and prec == "fp32":
                    del net, gpu, dense
                    torch.cuda.empty_cache()
                    parity["trained_field"] = trained_field_parity(torch, ctx.dev)
        # The driver keeps the last 8 KB of stdout: the FULL record goes out first (one line, {"bench_detail": ...}, and
        # ./bench_detail.json), then the line the contract asks for -- every headline field, every variant's value / ms / roofline,
        # the parity summaries -- kept under 6 KB so that it survives whole.
        print(json.dumps({"bench_detail": line}), flush=True)
        try:
            with open(os.path.join(os.getcwd(), "bench_detail.json"), "w") as f:
                json.dump(line, f, indent=1)
        except OSError:
            pass
        print(json.dumps(compact_line(line)), flush=True)
    if ctx.world > 1:
        ctx.dist.barrier()
        ctx.dist.destroy_process_group()


if __name__ == "__main__":
    main()
