"""GPU (-m gpu): the N > 1 path on the HIP kernels -- two processes, one per rank.

On a box with >= 2 GPUs the ranks use RCCL ("nccl"), one GPU each.  The round-end test box has ONE GPU, and RCCL refuses
two ranks on one device, so there the same code runs with both ranks on cuda:0 and "gloo" carrying the (GPU-resident)
tensors: everything but the transport is the production path -- HIP render per rank, the flat patch all-gather, the
splice, the batch-wide losses, backward through the rank's own patches, the flat gradient all-reduce, and bench.py's own
multi-rank launch.
"""
import json
import os
import subprocess
import sys
import types

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _loss_args():
    return types.SimpleNamespace(rand_neg=False, self_corr_w=0, use_sim_matrix=True, patch_stride=6,
                                 app_corr_params=["0.18", "1", "0.46", "1"], geo_corr_params=["0.5", "1", "3", "1"])


def _build(dev, arch="shipped"):
    """"shipped": the 8 x 256 net with the frozen-backbone recipe; "generic": a 4 x 64 net with a three-Linear semantic head, EVERY
    parameter trainable (the generic kernels' full backward under the same sharded step)."""
    import nerf_sos_amd
    torch.manual_seed(0)
    kw = dict(netdepth=4, netwidth=64, netdepth_fine=4, netwidth_fine=64, sem_layer=3) if arch == "generic" else {}
    net = nerf_sos_amd.NeRFNet(N_samples=64, N_importance=128, perturb=0.0, raw_noise_std=0.0, ray_chunk=1 << 20,
                               use_semantics=True, sem_with_coord=True, **kw).to(dev)
    for n_, p_ in net.named_parameters():
        p_.requires_grad = arch == "generic" or "semantic_linear" in n_
    net.train()
    return net


def _batch(B, P, dev):
    from nerf_sos_amd import synthetic as syn
    rays = syn.synthetic_patches(B, P, 6, seed=5, device=dev)
    feat = torch.stack([torch.randn(384, 14, 14, generator=torch.Generator().manual_seed(100 + b)) for b in range(B)]).to(dev)
    cls_ = torch.stack([torch.randn(384, generator=torch.Generator().manual_seed(200 + b)) for b in range(B)]).to(dev)
    return rays, feat, cls_


def _worker(rank, world, port, backend, q, arch="shipped"):
    import nerf_sos_amd
    from nerf_sos_amd import sharding, synthetic as syn
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dev = torch.device("cuda", rank if backend == "nccl" else 0)
    torch.cuda.set_device(dev)
    if backend == "nccl":
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
    else:
        dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        ok, why = True, []
        B, P = 5, 16                                         # 5 patches over 2 ranks: ragged ownership (3 + 2)
        net = _build(dev, arch)
        rays, feat, cls_ = _batch(B, P, dev)
        own = sharding.local_patches(B, rank, world)
        corr, geo = nerf_sos_amd.CorrelationLoss(_loss_args()), nerf_sos_amd.GeoCorrelationLoss(_loss_args())
        # --- single-process reference on this rank's GPU: the whole batch, no process group involved
        ref_net = _build(dev, arch)
        solo = [dist.new_group([r]) for r in range(world)][rank]      # every rank creates every group, in the same order
        ref_loss = sharding.sharded_patch_step(ref_net, rays, (syn.NEAR, syn.FAR), B, feat, cls_, corr, geo, step=4, seed=9,
                                               group=solo)
        # --- sharded step
        stats = {}
        loss = sharding.sharded_patch_step(net, rays[:, own].contiguous(), (syn.NEAR, syn.FAR), B, feat[own], cls_[own],
                                           corr, geo, step=4, seed=9, timings=stats)
        if abs(float(loss) - float(ref_loss)) > 1e-6 * (1 + abs(float(ref_loss))):
            ok = False
            why.append(f"loss {float(loss)} vs single-process {float(ref_loss)}")
        for (n_, p_), (_, r_) in zip(net.named_parameters(), ref_net.named_parameters()):
            if not p_.requires_grad:
                continue
            scale = float(r_.grad.abs().max()) + 1e-30
            err = float((p_.grad - r_.grad).abs().max()) / scale
            if err > (1e-4 if arch == "generic" else 2e-5):      # (generic: every parameter, a chain of fp32 reductions per layer)
                ok = False
                why.append(f"grad {n_}: {err:.2e} of scale")
        if stats["stats"]["collectives"] != 1:
            ok = False
            why.append(f"{stats['stats']['collectives']} all-gathers per step (one flat buffer expected)")
        want_bytes = 4 * (2 * P * P * 2 + P * P + 384 * 14 * 14 + 384 + 2 * P * P * 3)
        if stats["stats"]["bytes_per_patch"] != want_bytes:
            ok = False
            why.append(f"gathered {stats['stats']['bytes_per_patch']} B/patch, expected {want_bytes}")
        # --- eval: ray-sharded render of an image == the single-process render, bit for bit
        net.eval()
        img = syn.image_rays(dev, (1000, 1000 + 2001))       # 2001 rays: not divisible by the world size
        with torch.no_grad():
            full = net(img, (syn.NEAR, syn.FAR), retraw=False)
            got = sharding.render_image_sharded(net, img[0], img[1], (syn.NEAR, syn.FAR), gather=True,
                                                keys=("rgb", "depth", "semantics", "acc"), retraw=False)
        for k in got:
            if not torch.equal(got[k], full[k]):
                ok = False
                why.append(f"sharded render differs in {k}")
        q.put((rank, ok, "; ".join(why)))
    finally:
        dist.destroy_process_group()


def _run(backend, world=2, arch="shipped"):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29500 + (os.getpid() % 2000) + (7 if backend == "nccl" else 0) + 13 * world + (101 if arch != "shipped" else 0)
    procs = [ctx.Process(target=_worker, args=(r, world, port, backend, q, arch)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=400) for _ in procs]
    for p in procs:
        p.join(60)
    assert sorted(r[:2] for r in res) == [(r_, True) for r_ in range(world)], res


@pytest.mark.timeout(600)
def test_two_ranks_sharing_the_gpu_over_gloo():
    _run("gloo")


@pytest.mark.timeout(600)
def test_two_ranks_generic_net_full_backward_over_gloo():
    """The same sharded step with a generic-architecture net and every parameter trainable: loss and all gradients equal the
    single-process step's, the sharded render equals the single-process render bit for bit."""
    _run("gloo", 2, "generic")


@pytest.mark.timeout(900)
def test_six_ranks_sharing_the_gpu_over_gloo():
    """More ranks than patches (5 patches over 6 ranks: one rank renders nothing but takes part in every collective) and a
    2001-ray image over 6 ranks."""
    _run("gloo", 6)


@pytest.mark.timeout(600)
@pytest.mark.skipif(torch.cuda.device_count() < 2, reason="RCCL needs one GPU per rank; this box has one")
def test_two_ranks_over_rccl():
    _run("nccl")


def _bench_lines(r):
    """bench.py prints the full record first ({"bench_detail": ...}) and the contract's ONE line last, kept under 6 KB so that it
    survives the driver's 8 KB stdout tail whole.  Returns (detail, line)."""
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert r.returncode == 0 and len(lines) == 2, (r.returncode, r.stdout[-2000:], r.stderr[-3000:])
    detail, line = json.loads(lines[0])["bench_detail"], json.loads(lines[1])
    assert len(lines[1]) < 6144, len(lines[1])
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "dtype", "data", "config", "roofline"):
        assert k in line, k
    assert "vs_baseline" not in line or line["vs_baseline"] is None
    return detail, line


@pytest.mark.timeout(900)
def test_bench_launches_its_own_ranks():
    """`python bench.py --gpus 2` (the driver's command shape) must start two ranks itself and print ONE JSON line whose
    n_gpus is 2 and whose collective really saw two ranks; the C4 variant (patch all-gather + gradient all-reduce) rides
    along.  On a one-GPU box both ranks share cuda:0 over gloo (NSOS_BENCH_SHARE_GPU=1): the numbers mean nothing there,
    the code path is the point."""
    env = dict(os.environ)
    if torch.cuda.device_count() < 2:
        env["NSOS_BENCH_SHARE_GPU"] = "1"
    env.pop("WORLD_SIZE", None), env.pop("RANK", None), env.pop("LOCAL_RANK", None)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "3", "--warmup", "1"],
                       env=env, capture_output=True, text=True, timeout=800)
    j, line = _bench_lines(r)
    assert line["n_gpus"] == 2 and line["distributed"]["ranks_seen_by_collective"] == 2 and line["value"] == j["value"]
    assert line["variants"]["c4_bf16"]["value"] == j["variants"]["c4_bf16"]["value"]
    assert j["n_gpus"] == 2 and j["distributed"]["ranks_seen_by_collective"] == 2
    assert j["scaling"] == "weak" and j["config"]["rays_per_gpu"] == 4096 and j["value"] > 0
    assert j["per_rank_rays_per_s"]["min"] > 0
    c4 = j["variants"]["c4_bf16"]
    assert c4["patches"] == 4 and c4["rays_per_gpu"] == 8192
    assert c4["collectives"]["all_gathers_per_step"] == 1 and c4["collectives"]["gathered_bytes_per_patch"] > 400_000
    assert c4["collectives"]["all_gather_ms"] > 0 and c4["collectives"]["all_reduce_floats"] == 2 * 41218


@pytest.mark.timeout(1500)
@pytest.mark.parametrize("config,steps", [("c2", 2), ("c4", 2), ("c5", 1)])
def test_bench_eight_ranks_smoke(config, steps):
    """`python bench.py --gpus 8 --config ...` -- the driver's 8-GPU command shape -- for the three configurations that
    shard: c2 (4096 rays per rank, no collective), c4 (B = 16 patches, two per rank: the flat patch all-gather, the
    geometric loss's phase reductions, the gradient all-reduce) and c5 (95 256 rays per rank, ragged last 65 536-ray
    chunk).  On a one-GPU box all eight ranks share cuda:0 over gloo (NSOS_BENCH_SHARE_GPU=1): the numbers mean nothing,
    the launch, the per-rank seeds / shards and the collective sequence are the point; every collective runs under the
    package's watchdog (a diverged sequence fails instead of hanging)."""
    env = dict(os.environ)
    if torch.cuda.device_count() < 8:
        env["NSOS_BENCH_SHARE_GPU"] = "1"
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK"):
        env.pop(k, None)
    env["NSOS_COLLECTIVE_TIMEOUT_S"] = "240"
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "8", "--config", config, "--steps", str(steps),
                        "--warmup", "1", "--no-variants"], env=env, capture_output=True, text=True, timeout=1400)
    j, line = _bench_lines(r)
    assert line["n_gpus"] == 8 and line["value"] == j["value"]
    assert j["n_gpus"] == 8 and j["distributed"]["ranks_seen_by_collective"] == 8 and j["value"] > 0
    if config == "c2":
        assert j["scaling"] == "weak" and j["config"]["rays_per_gpu"] == 4096
    if config == "c4":
        c = j["collectives"]
        assert j["config"]["patches"] == 16 and j["config"]["rays_per_gpu"] == 8192
        assert c["all_gathers_per_step"] == 1 and c["gathered_bytes_per_patch"] == 482816
        # per step and rank FOUR collectives: ONE flat patch gather, the row-partitioned losses' two reductions (the means and the
        # sums of the geometric AND both appearance evaluations, one buffer each), ONE flat gradient all-reduce -- counted by
        # sharding.collective, nothing hidden
        assert c["calls_per_step_by_kind"] == {"all_gather": 1.0, "loss_means_all_reduce": 1.0, "loss_sums_all_reduce": 1.0,
                                                "grad_all_reduce": 1.0}, c
        assert c["calls_per_step"] == 4.0
        assert c["all_reduce_floats"] == 2 * 41218 and abs(j["loss"]) < 10
    if config == "c5":
        assert j["scaling"] == "strong" and j["config"]["rays_per_gpu"] == 95256 and j["finite"] is True


def test_collective_watchdog_names_the_diverged_collective():
    """A rank that skips a collective its peers issue must produce an error naming the collective, not a hang: two gloo
    ranks, rank 1 leaves out one all-reduce; rank 0's wait gives up after COLLECTIVE_TIMEOUT_S."""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29500 + (os.getpid() % 2000) + 777
    procs = [ctx.Process(target=_watchdog_worker, args=(r, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = dict(q.get(timeout=120) for _ in procs)
    for p in procs:
        p.join(30)
    assert "grad_all_reduce" in res[0] and "did not complete" in res[0], res
    assert res[1] == "skipped"


def _watchdog_worker(rank, port, q):
    import datetime
    from nerf_sos_amd import sharding
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    torch.cuda.set_device(0)
    dist.init_process_group("gloo", rank=rank, world_size=2, timeout=datetime.timedelta(seconds=60))
    sharding.COLLECTIVE_TIMEOUT_S = 5.0
    p = torch.nn.Parameter(torch.ones(4, device="cuda:0"))
    p.grad = torch.ones(4, device="cuda:0")
    try:
        sharding.all_reduce_grads([p])                     # both ranks: fine
        if rank == 0:
            try:
                sharding.all_reduce_grads([p])             # rank 1 never issues this one
                q.put((0, "no error"))
            except RuntimeError as e:
                q.put((0, str(e)))
        else:
            q.put((1, "skipped"))
            import time
            time.sleep(8)
    finally:
        try:
            dist.destroy_process_group()
        except Exception:
            pass


# ------------------------------------------------------------------------- the sharded step as a HIP graph (round 5)
def _graph_worker(rank, world, port, backend, q):
    """Every rank: GraphedPatchStep under the process group (captured with its four collectives on RCCL; automatic eager fall-back on
    gloo) for three steps, and a twin net stepped eagerly by sharding.sharded_patch_step with the same generator / Philox counter:
    same loss and same parameters after every step, on every rank."""
    import nerf_sos_amd
    from nerf_sos_amd import sharding, synthetic as syn
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dev = torch.device("cuda", rank if backend == "nccl" else 0)
    torch.cuda.set_device(dev)
    if backend == "nccl":
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
    else:
        dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        ok, why = True, []
        B, P = 4, 16
        rays, feat, cls_ = _batch(B, P, dev)
        own = sharding.local_patches(B, rank, world)
        nets, opts, steps = [], [], []
        for twin in range(2):
            net = _build(dev)
            net.perturb, net.raw_noise_std = 1.0, 1.0
            net.render_kwargs_train.update(perturb=1.0, raw_noise_std=1.0)
            net.rng, net.rng_seed = "philox", 3 + rank
            opt = torch.optim.Adam([p for p in net.parameters() if p.requires_grad], lr=5e-3, fused=True, capturable=True)
            corr, geo = nerf_sos_amd.CorrelationLoss(_loss_args()), nerf_sos_amd.GeoCorrelationLoss(_loss_args())
            g = nerf_sos_amd.GraphedPatchStep(net, opt, rays[:, own].contiguous(), (syn.NEAR, syn.FAR), feat[own], cls_[own], corr, geo,
                                              seed=21, warmup=2, capture=(twin == 0), n_patches=B)
            nets.append(net), opts.append(opt), steps.append(g)
        graphed, eager = steps
        if backend == "gloo":
            if graphed.graph is not None or "gloo" not in (graphed.capture_fallback or ""):
                ok = False
                why.append(f"a gloo group must fall back to eager: graph={graphed.graph}, reason={graphed.capture_fallback!r}")
        elif graphed.graph is None:
            ok = False
            why.append(f"RCCL capture fell back: {graphed.capture_fallback}")      # (the world-size-1 test shows this build captures RCCL work)
        else:
            eager.eager_step(); eager.eager_step()                                 # the captured twin's two warm-up steps
        for k in range(3):
            la, lb = float(graphed()), float(eager())
            if abs(la - lb) > 1e-6 * (1 + abs(lb)):
                ok = False
                why.append(f"step {k}: loss {la} (graphed) vs {lb} (eager)")
        for (n_, p_), (_, r_) in zip(nets[0].named_parameters(), nets[1].named_parameters()):
            if p_.requires_grad and not torch.equal(p_, r_):
                err = float((p_ - r_).abs().max())
                if err > 1e-7:
                    ok = False
                    why.append(f"{n_}: graphed and eager parameters differ by {err:.2e} after three steps")
        # the ranks hold the same parameters (one gradient all-reduce per step)
        flat = torch.cat([p_.detach().reshape(-1) for p_ in nets[0].parameters() if p_.requires_grad])
        mx, mn = flat.clone(), flat.clone()
        dist.all_reduce(mx, op=dist.ReduceOp.MAX)
        dist.all_reduce(mn, op=dist.ReduceOp.MIN)
        if not torch.equal(mx, mn):
            ok = False
            why.append("the ranks' parameters diverged")
        q.put((rank, ok, "; ".join(why)))
    finally:
        dist.destroy_process_group()


def _run_graph(backend, world=2):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29500 + (os.getpid() % 2000) + 211 + (7 if backend == "nccl" else 0)
    procs = [ctx.Process(target=_graph_worker, args=(r, world, port, backend, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=400) for _ in procs]
    for p in procs:
        p.join(60)
    assert sorted(r[:2] for r in res) == [(r_, True) for r_ in range(world)], res
    return res


@pytest.mark.timeout(600)
def test_graphed_step_under_a_gloo_group_falls_back_to_eager():
    _run_graph("gloo")


@pytest.mark.timeout(600)
@pytest.mark.skipif(torch.cuda.device_count() < 2, reason="RCCL needs one GPU per rank; this box has one")
def test_graphed_step_two_ranks_over_rccl():
    """The first multi-GPU box answers whether this build's RCCL accepts the capture: either way the replayed and the eager step
    must agree; the outcome (captured / fell back, with the reason) is printed."""
    print(_run_graph("nccl"))


# ---------------------------------------------------------- RCCL inside a HIP-graph capture on ONE GPU (round 6)
def _rccl_world1_worker(port, q):
    """One process, `backend="nccl"`, world size 1, sharding.FORCE_COLLECTIVES: the sharded step takes its N > 1 branches and issues
    its four collectives as real RCCL launches -- inside the capture of GraphedPatchStep.  Asserted: the capture happened (no
    fall-back), the per-step collective table is the four-collective one, five replays equal five eager steps of a twin bit for
    bit, and the forced-collective step equals the plain single-process step (the N > 1 branches compute the same numbers)."""
    import nerf_sos_amd
    from nerf_sos_amd import sharding, synthetic as syn
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(dev)
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
    try:
        ok, why = True, []
        B, P = 4, 16
        rays, feat, cls_ = _batch(B, P, dev)

        def make(capture, force):
            sharding.FORCE_COLLECTIVES = force
            net = _build(dev)
            net.perturb, net.raw_noise_std = 1.0, 1.0
            net.render_kwargs_train.update(perturb=1.0, raw_noise_std=1.0)
            net.rng, net.rng_seed = "philox", 3
            opt = torch.optim.Adam([p for p in net.parameters() if p.requires_grad], lr=5e-3, fused=True, capturable=True)
            corr, geo = nerf_sos_amd.CorrelationLoss(_loss_args()), nerf_sos_amd.GeoCorrelationLoss(_loss_args())
            g = nerf_sos_amd.GraphedPatchStep(net, opt, rays, (syn.NEAR, syn.FAR), feat, cls_, corr, geo, seed=21, warmup=2,
                                              capture=capture, n_patches=B)
            return net, g

        net_g, graphed = make(True, True)
        if graphed.graph is None or graphed.capture_fallback is not None:
            ok = False
            why.append(f"the RCCL step was not captured: graph={graphed.graph}, reason={graphed.capture_fallback!r}")
        net_e, eager = make(False, True)
        net_p, plain = make(False, False)                   # the ordinary single-process step: no collectives at all
        sharding.FORCE_COLLECTIVES = True
        eager.eager_step(); eager.eager_step()              # the captured instance's two warm-up steps
        sharding.reset_collective_counts()
        eager()
        table = sharding.reset_collective_counts()
        want = {"all_gather": 1, "loss_means_all_reduce": 1, "loss_sums_all_reduce": 1, "grad_all_reduce": 1}
        if table != want:
            ok = False
            why.append(f"collectives per eager step {table} != {want}")
        graphed()
        for k in range(1, 5):
            la, lb = float(graphed()), float(eager())
            if la != lb:
                ok = False
                why.append(f"step {k}: loss {la!r} (replayed) vs {lb!r} (eager)")
        if sharding.reset_collective_counts():
            pass                                            # (replays issue no Python-level collectives; eager ones were counted)
        for (n_, a), (_, b) in zip(net_g.named_parameters(), net_e.named_parameters()):
            if a.requires_grad and not torch.equal(a, b):
                ok = False
                why.append(f"{n_}: replayed and eager parameters differ by {float((a.detach() - b.detach()).abs().max()):.2e} after five steps")
        sharding.FORCE_COLLECTIVES = False
        for k in range(7):
            plain()
        for (n_, a), (_, b) in zip(net_p.named_parameters(), net_e.named_parameters()):
            if a.requires_grad:
                err = float((a.detach() - b.detach()).abs().max()) / (1e-12 + float(b.detach().abs().max()))
                if err > 2e-5:
                    ok = False
                    why.append(f"{n_}: forced-collective and plain single-process steps differ by {err:.2e} of scale after five steps")
        # a capture that raises is LOUD by default, and with allow_eager_fallback the step runs eagerly after the agreement collective
        sharding.FORCE_COLLECTIVES = True
        net_f = _build(dev)
        net_f.rng = "philox"
        opt_f = torch.optim.Adam([p for p in net_f.parameters() if p.requires_grad], lr=5e-3, fused=True, capturable=True)

        class Boom(nerf_sos_amd.CorrelationLoss):
            def rows_phased(self, *a, **k):
                if torch.cuda.is_current_stream_capturing():
                    raise RuntimeError("injected failure inside the capture")
                return super().rows_phased(*a, **k)
        kw = dict(seed=21, warmup=1, capture=True, n_patches=B)
        try:
            nerf_sos_amd.GraphedPatchStep(net_f, opt_f, rays, (syn.NEAR, syn.FAR), feat, cls_, Boom(_loss_args()),
                                          nerf_sos_amd.GeoCorrelationLoss(_loss_args()), **kw)
            ok = False
            why.append("a failing capture did not raise")
        except RuntimeError as e:
            if "injected failure" not in str(e) or "allow_eager_fallback" not in str(e):
                ok = False
                why.append(f"unexpected message: {e}")
        torch.cuda.synchronize()
        fb = nerf_sos_amd.GraphedPatchStep(net_f, opt_f, rays, (syn.NEAR, syn.FAR), feat, cls_, Boom(_loss_args()),
                                           nerf_sos_amd.GeoCorrelationLoss(_loss_args()), allow_eager_fallback=True, **kw)
        if fb.graph is not None or "stepping eagerly on every rank" not in (fb.capture_fallback or ""):
            ok = False
            why.append(f"fall-back state: graph={fb.graph}, reason={fb.capture_fallback!r}")
        l0 = float(fb())
        l1 = float(fb())
        # the escape hatch: capture_collectives=False keeps the multi-rank step eager without attempting a capture
        net_h = _build(dev)
        net_h.rng = "philox"
        opt_h = torch.optim.Adam([p for p in net_h.parameters() if p.requires_grad], lr=5e-3, fused=True, capturable=True)
        hatch = nerf_sos_amd.GraphedPatchStep(net_h, opt_h, rays, (syn.NEAR, syn.FAR), feat, cls_, nerf_sos_amd.CorrelationLoss(_loss_args()),
                                              nerf_sos_amd.GeoCorrelationLoss(_loss_args()), capture_collectives=False, **kw)
        if hatch.graph is not None or "capture_collectives" not in (hatch.capture_fallback or ""):
            ok = False
            why.append(f"capture_collectives=False: graph={hatch.graph}, reason={hatch.capture_fallback!r}")
        float(hatch())
        if not (l0 == l0 and l1 == l1):
            ok = False
            why.append("the eager fall-back step returned NaN")
        q.put((0, ok, "; ".join(why)))
    finally:
        dist.destroy_process_group()


@pytest.mark.timeout(600)
def test_graphed_step_rccl_world_1_captures_its_collectives():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29500 + (os.getpid() % 2000) + 331
    p = ctx.Process(target=_rccl_world1_worker, args=(port, q))
    p.start()
    try:
        res = q.get(timeout=400)
        p.join(60)
    finally:
        if p.is_alive():                                   # a hung collective must not outlive the test (exact PID: our own child)
            p.kill()
    assert res[:2] == (0, True), res
