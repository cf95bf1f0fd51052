"""CPU: the N>1 path -- ray sharding and the patch all-gather -- on world_size-2 gloo."""
import os
import sys

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from nerf_sos_amd import sharding


def test_shard_bounds_cover_and_balance():
    for n in (0, 1, 7, 762048, 95256 * 8 + 3):
        for world in (1, 2, 4, 8):
            spans = [sharding.shard_bounds(n, r, world) for r in range(world)]
            assert spans[0][0] == 0 and spans[-1][1] == n
            assert all(spans[i][1] == spans[i + 1][0] for i in range(world - 1))
            sizes = [b - a for a, b in spans]
            assert max(sizes) - min(sizes) <= 1
    assert sharding.shard_bounds(762048, 0, 8) == (0, 95256)
    assert sharding.local_patches(16, 3, 8) == [3, 11]


def _fake_render(ray_batch, bounds, **kw):
    o, d = ray_batch
    near, far = bounds
    near = near if isinstance(near, torch.Tensor) else torch.full((d.shape[0],), float(near))
    return {"rgb": d * 2.0 + o, "depth": (d.sum(-1, keepdim=True) + near[:, None]), "weights": d.repeat(1, 2)}


def _worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        torch.manual_seed(0)
        R = 1001  # not divisible by the world size
        o, d = torch.randn(R, 3), torch.randn(R, 3)
        near = torch.rand(R)
        full = _fake_render((o, d), (near, 6.0))
        got = sharding.render_image_sharded(_fake_render, o, d, (near, 6.0), gather=True)
        ok = all(torch.equal(got[k], full[k]) for k in full)
        mine = sharding.render_image_sharded(_fake_render, o, d, (near, 6.0), gather=False, keys=("rgb",))
        s, e = sharding.shard_bounds(R, rank, world)
        ok = ok and torch.equal(mine["rgb"], full["rgb"][s:e]) and list(mine) == ["rgb"]
        # patches: B=5 patches of 4x4, owner = b mod world
        B, P = 5, 4
        allp = {"semantics": torch.randn(B, P, P, 2), "depth": torch.randn(B, P, P, 1), "rgb": torch.randn(B, P, P, 3),
                "feat": torch.randn(B, 6, 3, 3), "cls_": torch.randn(B, 6)}
        own = sharding.local_patches(B, rank, world)
        local = {k: v[own].clone().requires_grad_(k == "semantics") for k, v in allp.items()}
        st = {}
        g = sharding.all_gather_patches(local, B, keys=("semantics", "depth", "semantics0", "feat", "cls_"), stats=st)
        ok = ok and set(g) == {"semantics", "depth", "feat", "cls_"} and all(torch.equal(g[k], allp[k]) for k in g)
        # one flat buffer for all keys -> ONE collective; payload per patch = the sum of the per-patch tensors
        ok = ok and st["collectives"] == 1 and st["bytes_per_patch"] == 4 * (P * P * 2 + P * P + 6 * 9 + 6)
        ok = ok and not g["semantics"].requires_grad
        # training: batch-wide loss on gathered patches, gradient through the rank's own patches, one grad all-reduce
        torch.manual_seed(1)
        head = torch.nn.Linear(3, 2)                       # stands in for the semantic head (same init on every rank)
        feats = torch.randn(B, P, P, 3)
        neg = torch.tensor([2, 0, 4, 1, 3])

        def batch_loss(sem):                               # couples every patch with its negative, like the losses
            return (sem * sem[neg].flip(1)).mean() + sem.square().mean()

        ref_head = torch.nn.Linear(3, 2)
        ref_head.load_state_dict(head.state_dict())
        batch_loss(ref_head(feats)).backward()
        mine_sem = head(feats[own])
        gathered = sharding.all_gather_patches({"semantics": mine_sem}, B, keys=("semantics",))
        spliced = sharding.splice_local_patches(gathered, {"semantics": mine_sem}, B)
        loss = batch_loss(spliced["semantics"])
        loss.backward()
        sharding.all_reduce_grads(head.parameters())
        ok = ok and torch.allclose(loss.detach(), batch_loss(ref_head(feats)).detach(), atol=1e-6)
        ok = ok and all(torch.allclose(a.grad, b.grad, atol=1e-6) for a, b in zip(head.parameters(), ref_head.parameters()))
        q.put((rank, bool(ok)))
    finally:
        dist.destroy_process_group()


@pytest.mark.timeout(180)
def test_gloo_world2_shard_and_gather():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29500 + (os.getpid() % 2000)
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=150) for _ in procs]
    for p in procs:
        p.join(30)
    assert sorted(res) == [(0, True), (1, True)]


def test_device_index_is_uploaded_once():
    a = sharding.device_index([3, 1, 2], torch.long, torch.device("cpu"))
    b = sharding.device_index((3, 1, 2), torch.long, torch.device("cpu"))
    c = sharding.device_index([3, 1, 2], torch.int32, torch.device("cpu"))
    assert a is b and a.tolist() == [3, 1, 2] and c.dtype == torch.int32 and c is not a


def test_geo_loss_both_stacks_codes_geometry_and_negatives():
    """What sharding.geo_loss_both hands to the loss module: the two codes stacked as 2B patches, the geometry repeated, the
    row list repeated with an offset of B, and a similarity matrix whose argmin per column stays inside its own half (so the
    negatives of the second half are the first half's, shifted by B); the result is twice the stacked mean."""
    B, P = 3, 4
    g = torch.Generator().manual_seed(0)
    depth, ro, rd = torch.rand(B, 1, P, P, generator=g), torch.rand(B, 3, P, P, generator=g), torch.rand(B, 3, P, P, generator=g)
    c0, c1 = torch.rand(B, 2, P, P, generator=g), torch.rand(B, 2, P, P, generator=g)
    sim = torch.rand(B, B, generator=g)
    seen = {}

    def fake(d, code, rays, sim2, rows=None, group=None):
        seen.update(d=d, code=code, rays=rays, sim=sim2, rows=rows)
        return code.sum()

    out = sharding.geo_loss_both(fake, depth, c0, c1, ro, rd, sim, rows=[0, 2])
    assert torch.equal(seen["code"], torch.cat([c0, c1], 0)) and torch.equal(seen["d"], torch.cat([depth, depth], 0))
    assert torch.equal(seen["rays"][0], torch.cat([ro, ro], 0)) and torch.equal(seen["rays"][1], torch.cat([rd, rd], 0))
    assert seen["rows"] == [0, 2, 3, 5]
    neg = torch.min(sim, dim=0)[1]
    assert torch.equal(torch.min(seen["sim"], dim=0)[1], torch.cat([neg, neg + B]))
    assert torch.equal(out, 2.0 * (c0.sum() + c1.sum()))


# ---------------------------------------------------------------- round 6: the N > 1 branches on ONE process (FORCE_COLLECTIVES)
def _forced_worker(port, q):
    """gloo, world size 1, sharding.FORCE_COLLECTIVES: the gather / splice / gradient all-reduce take their multi-rank branches and
    issue their collectives (counted), and return what the single-process short cuts return."""
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=0, world_size=1)
    try:
        ok, why = True, []
        torch.manual_seed(0)
        B = 3
        local = {"semantics": torch.randn(B, 4, 4, 2, requires_grad=True), "semantics0": torch.randn(B, 4, 4, 2), "depth": torch.randn(B, 4, 4, 1),
                 "feat": torch.randn(B, 6, 2, 2), "cls_": torch.randn(B, 6), "ray_o": torch.randn(B, 4, 4, 3), "ray_d": torch.randn(B, 4, 4, 3)}
        assert not sharding.multi_process(None)
        plain = sharding.all_gather_patches(local, B, None)
        sharding.FORCE_COLLECTIVES = True
        assert sharding.multi_process(None)
        sharding.reset_collective_counts()
        stats = {}
        forced = sharding.all_gather_patches(local, B, None, stats=stats)
        full = sharding.splice_local_patches(forced, local, B, None)
        counts = sharding.reset_collective_counts()
        if counts != {"all_gather": 1}:
            ok = False
            why.append(f"collectives {counts}")
        for k in plain:
            if not torch.equal(plain[k], forced[k]):
                ok = False
                why.append(f"gathered `{k}` differs from the local batch")
        if not (full["semantics"].requires_grad and torch.equal(full["semantics"].detach(), local["semantics"].detach())):
            ok = False
            why.append("the rank's own gradient-carrying patches were not spliced back")
        lin = torch.nn.Linear(3, 2)
        lin(torch.randn(5, 3)).sum().backward()
        want = [p.grad.clone() for p in lin.parameters()]
        sharding.all_reduce_grads(lin.parameters(), None)
        counts = sharding.reset_collective_counts()
        if counts != {"grad_all_reduce": 1} or any(not torch.equal(p.grad, w) for p, w in zip(lin.parameters(), want)):
            ok = False
            why.append(f"gradient all-reduce: {counts}")
        sharding.FORCE_COLLECTIVES = False
        q.put((ok, "; ".join(why)))
    finally:
        dist.destroy_process_group()


def test_force_collectives_runs_the_multi_rank_branches_on_one_process():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    p = ctx.Process(target=_forced_worker, args=(29500 + (os.getpid() % 2000) + 431, q))
    p.start()
    try:
        res = q.get(timeout=120)
        p.join(30)
    finally:
        if p.is_alive():
            p.kill()
    assert res[0], res


def test_tail_stats_counts_and_percentiles():
    """nerf_sos_amd.quality.tail_stats: PSNR, order statistics and counts over fixed thresholds of |d rgb| (max over channels)."""
    from nerf_sos_amd import quality
    ref = torch.zeros(10000, 3)
    got = ref.clone()
    got[:5, 1] = 0.03          # five rays over 0.02
    got[5:7, 2] = 0.2          # two over 0.05
    got[7:107, 0] = 0.005      # a hundred small ones
    depth, ref_depth = torch.full((10000,), 2.0), torch.full((10000,), 2.0)
    depth[0] = 3.0
    lab = torch.zeros(10000, dtype=torch.long)
    st = quality.tail_stats(got, ref, depth, ref_depth, lab, lab)
    a = st["abs_rgb"]
    assert (a["n_gt_0.01"], a["n_gt_0.02"], a["n_gt_0.05"]) == (7, 7, 2) and abs(a["max"] - 0.2) < 1e-7
    assert a["p50"] == 0.0 and abs(a["p99"] - 0.005) < 1e-7 and abs(a["p99.99"] - 0.2) < 1e-7
    assert st["share_of_rays_within_0.02"] == round(1 - 7 / 10000, 6) and st["label_agreement"] == 1.0
    assert st["rel_depth"]["n_gt_0.1"] == 1 and abs(st["rel_depth"]["max"] - 0.5) < 1e-7
    mse = (5 * 0.03 ** 2 + 2 * 0.2 ** 2 + 100 * 0.005 ** 2) / 30000
    assert abs(st["psnr_db"] - round(-10 * __import__("math").log10(mse), 2)) < 0.011
    row = quality.compact(st)
    assert row["n_gt_0.01"] == 7 and row["rays"] == 10000 and row["depth_n_gt_0.01"] == 1
