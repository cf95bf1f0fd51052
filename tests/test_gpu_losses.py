"""GPU parity of the rows next to the render path against goldens captured from the real
reference classes (tests/golden/make_goldens_losses.py): evaluation post-processing and the two correlation losses.
Tolerance: 1e-4 of the quantity's scale (fp32); predicted labels and the in-place depth filter are exact."""
import os
import types

import numpy as np
import pytest
import torch

import nerf_sos_amd
from nerf_sos_amd import ops

pytestmark = pytest.mark.gpu
DEV = "cuda:0"
GOLD = np.load(os.path.join(os.path.dirname(__file__), "golden", "losses.npz"))
APP, GEO = (0.18, 1, 0.46, 1), (0.5, 1, 3, 1)   # scripts/train_fortress_node0.sh


def T(a):
    return torch.from_numpy(np.asarray(a)).to(DEV)


def ref_args():
    a = types.SimpleNamespace()
    a.rand_neg, a.self_corr_w, a.use_sim_matrix, a.patch_stride = False, 0, True, 6
    a.app_corr_params = [str(x) for x in APP]
    a.geo_corr_params = [str(x) for x in GEO]
    return a


def rel(a, b):
    a, b = a.detach().float().cpu(), torch.as_tensor(np.asarray(b)).float()
    return float((a.reshape(b.shape) - b).abs().max() / b.abs().max().clamp_min(1e-30))


def test_eval_postprocess_golden():
    out = ops.eval_postprocess(T(GOLD["post_sem"]), T(GOLD["post_rgb"]), T(GOLD["post_tgt"]))
    assert out["sem"].dtype == torch.int32 and tuple(out["sem"].shape) == (9, 13, 1)
    assert np.array_equal(out["sem"].cpu().numpy(), GOLD["post_pred"])          # labels: exact (incl. the tie -> 0)
    assert np.abs(out["sem_prob"].cpu().numpy() - GOLD["post_prob"]).max() < 1e-6
    assert abs(out["mse"].item() - GOLD["post_mse"][0]) < 1e-6 * GOLD["post_mse"][0]
    assert abs(out["psnr"].item() - GOLD["post_psnr"][0]) < 1e-5
    again = ops.eval_postprocess(T(GOLD["post_sem"]), T(GOLD["post_rgb"]), T(GOLD["post_tgt"]))
    assert torch.equal(out["mse"], again["mse"])                                  # fixed-order reduction
    out5 = ops.eval_postprocess(T(GOLD["post5_sem"]))
    assert set(out5) == {"sem_prob", "sem"} and np.array_equal(out5["sem"].cpu().numpy(), GOLD["post5_pred"])
    assert np.abs(out5["sem_prob"].cpu().numpy() - GOLD["post5_prob"]).max() < 1e-6
    only = ops.eval_postprocess(rgb=T(GOLD["post_rgb"]), target=T(GOLD["post_tgt"]))
    assert set(only) == {"mse", "psnr"} and torch.equal(only["mse"], out["mse"])


def test_eval_postprocess_on_a_render():
    """The C5 flow: render, post-process on device, only labels + two scalars leave the GPU."""
    from oracle import torch_port as tp
    net = nerf_sos_amd.NeRFNet(N_samples=64, N_importance=128, use_semantics=True, sem_with_coord=True).to(DEV).eval()
    rays = tp.synthetic_rays(512, seed=1).to(DEV)
    with torch.no_grad():
        ret = net(rays, (tp.NEAR, tp.FAR), retraw=False)
    tgt = torch.rand_like(ret["rgb"])
    out = ops.eval_postprocess(ret["semantics"], ret["rgb"], tgt)
    prob = ret["semantics"].softmax(-1)
    assert torch.equal(out["sem"][..., 0].long(), prob.argmax(-1)) and (out["sem_prob"] - prob).abs().max() < 1e-6
    mse = ((ret["rgb"] - tgt) ** 2).mean()
    assert abs(out["mse"].item() - mse.item()) < 1e-6 and abs(out["psnr"].item() + 10 * np.log10(mse.item())) < 1e-4


class InjectRand:
    def __init__(self, *arrays):
        self.q = [T(a) for a in arrays]

    def __enter__(self):
        self._rand = torch.rand
        torch.rand = lambda *a, **k: self.q.pop(0)

    def __exit__(self, *exc):
        torch.rand = self._rand


@pytest.mark.parametrize("tag", ["app_small", "app_full"])
def test_correlation_loss_golden(tag):
    mod = nerf_sos_amd.CorrelationLoss(ref_args())
    assert (mod.self_shift, mod.self_weight, mod.neg_shift, mod.neg_weight) == APP
    feats, sim = T(GOLD[f"{tag}_feats"]), T(GOLD[f"{tag}_sim"])
    code = T(GOLD[f"{tag}_code"]).requires_grad_(True)
    with InjectRand(GOLD[f"{tag}_rand1"], GOLD[f"{tag}_rand2"]):
        loss = mod(feats, code, sim)
    assert loss.shape == () and loss.dtype == torch.float32
    assert abs(loss.item() - GOLD[f"{tag}_loss"][0]) < 1e-4 * (1 + abs(GOLD[f"{tag}_loss"][0])), (loss.item(), GOLD[f"{tag}_loss"][0])
    (3.0 * loss).backward()
    assert rel(code.grad / 3.0, GOLD[f"{tag}_grad"]) < 1e-4
    code2 = T(GOLD[f"{tag}_code"]).requires_grad_(True)
    with InjectRand(GOLD[f"{tag}_rand1"], GOLD[f"{tag}_rand2"]):
        loss2 = mod(feats, code2, sim)
    loss2.backward()
    assert torch.equal(loss2, loss.detach()) and torch.equal(code2.grad * 3.0, code.grad)   # deterministic
    with torch.no_grad(), InjectRand(GOLD[f"{tag}_rand1"], GOLD[f"{tag}_rand2"]):
        assert torch.equal(mod(feats, code.detach(), sim), loss.detach())                      # no-grad path


@pytest.mark.parametrize("tag", ["geo_small", "geo_full"])
def test_geo_correlation_loss_golden(tag):
    mod = nerf_sos_amd.GeoCorrelationLoss(ref_args())
    assert (mod.self_shift, mod.self_weight, mod.neg_shift, mod.neg_weight, mod.max_depth) == GEO + (15,)
    depth, sim = T(GOLD[f"{tag}_depth"]), T(GOLD[f"{tag}_sim"])
    B, _, P, _ = depth.shape
    ray_o = T(GOLD[f"{tag}_ray_o"])[:, :, None, None].expand(B, 3, P, P)
    ray_d = T(GOLD[f"{tag}_ray_d"])
    code = T(GOLD[f"{tag}_code"]).requires_grad_(True)
    loss = mod(depth, code, [ray_o, ray_d, None], sim)
    assert abs(loss.item() - GOLD[f"{tag}_loss"][0]) < 1e-4 * (1 + abs(GOLD[f"{tag}_loss"][0])), (loss.item(), GOLD[f"{tag}_loss"][0])
    assert np.array_equal(depth.cpu().numpy(), GOLD[f"{tag}_depth_after"])     # in-place depth filter, exact
    loss.backward()
    assert rel(code.grad, GOLD[f"{tag}_grad"]) < 1e-4
    # trainer layout (engines/trainer.py:151-160): channel-last tensors permuted to [B,C,P,P] views
    depth_cl = T(GOLD[f"{tag}_depth"]).permute(0, 2, 3, 1).contiguous()
    code_cl = T(GOLD[f"{tag}_code"]).permute(0, 2, 3, 1).contiguous().requires_grad_(True)
    loss2 = mod(depth_cl.permute(0, 3, 1, 2), code_cl.permute(0, 3, 1, 2), [ray_o, ray_d, None], sim)
    loss2.backward()
    assert torch.equal(loss2, loss.detach())
    assert torch.equal(code_cl.grad.permute(0, 3, 1, 2), code.grad)
    assert np.array_equal(depth_cl.permute(0, 3, 1, 2).cpu().numpy(), GOLD[f"{tag}_depth_after"])


@pytest.mark.parametrize("tag", ["geo_small", "geo_full"])
def test_geo_correlation_loss_row_partitioned_equals_golden(tag):
    """The phased, row-partitioned entry point (nsos_geo_correlation_loss_rows: what each rank of the patch-sharded step
    calls for its own row patches) with rows = every patch and nothing to reduce is the same loss: golden value and
    gradient within 1e-4; and the sum of the per-subset role sums over a 2-way split of the rows equals the whole --
    emulated here in one process by running both halves' phases and adding the reduced slots by hand, the way the
    all-reduce does across ranks (tests/test_gpu_sharded.py runs it over a real process group)."""
    import ctypes as C
    from nerf_sos_amd import _lib
    mod = nerf_sos_amd.GeoCorrelationLoss(ref_args())
    depth, sim = T(GOLD[f"{tag}_depth"]), T(GOLD[f"{tag}_sim"])
    B, _, P, _ = depth.shape
    ray_o = T(GOLD[f"{tag}_ray_o"])[:, :, None, None].expand(B, 3, P, P).contiguous()
    ray_d = T(GOLD[f"{tag}_ray_d"])
    code = T(GOLD[f"{tag}_code"]).requires_grad_(True)
    loss = mod(depth.clone(), code, [ray_o, ray_d, None], sim, rows=list(range(B)))
    want = GOLD[f"{tag}_loss"][0]
    assert abs(loss.item() - want) < 1e-4 * (1 + abs(want)), (loss.item(), want)
    loss.backward()
    assert rel(code.grad, GOLD[f"{tag}_grad"]) < 1e-4
    # two "ranks" in one process: phases interleaved, slots summed by hand
    lib = _lib.lib()
    neg = torch.min(sim, dim=0)[1].to(torch.int64).contiguous()
    Cn = code.shape[1]
    nbytes = lib.nsos_corr_workspace_bytes(1, B, P * P, 0)
    so, go, gn = C.c_int64(), C.c_int64(), C.c_int64()
    _lib.check(lib.nsos_corr_workspace_slots(B, P * P, C.byref(so), C.byref(go), C.byref(gn)), "slots")
    halves = [list(range(0, B, 2)), list(range(1, B, 2))]
    wss = [torch.zeros((nbytes + 15) // 16 * 2, device=DEV, dtype=torch.float64) for _ in halves]
    rws = [torch.tensor(h, dtype=torch.int32, device=DEV) for h in halves]
    dbs = [T(GOLD[f"{tag}_depth"]).contiguous() for _ in halves]
    cd = code.detach().contiguous()
    P_ = lambda t: C.c_void_p(t.data_ptr())  # noqa: E731
    out_loss = [torch.empty((), device=DEV) for _ in halves]
    out_grad = [torch.empty_like(cd) for _ in halves]
    prm = (mod.self_shift, mod.self_weight, mod.neg_shift, mod.neg_weight)

    def call(phase, k):
        _lib.check(lib.nsos_geo_correlation_loss_rows(phase, P_(dbs[k]), P_(cd), P_(ray_o), P_(ray_d.contiguous()), P_(neg),
                                                      P_(rws[k]) if halves[k] else None, len(halves[k]), B, Cn, P, P, *prm, 15.0, 1,
                                                      P_(out_loss[k]), P_(out_grad[k]), P_(wss[k]), wss[k].numel() * 8, None, None, None), "rows")

    assert gn.value == lib.nsos_corr_exchange_floats(B, P * P) == B * P * P * 4 + 8
    for phase in range(3):                       # phase 0 -> [means summed] -> phase 1 -> [sums summed] -> phase 2
        for k in range(2):
            call(phase, k)
        torch.cuda.synchronize()
        if phase == 0:
            sl = slice(so.value // 8, so.value // 8 + 4)
            tot = wss[0][sl] + wss[1][sl]
            wss[0][sl] = tot
            wss[1][sl] = tot
        if phase == 1:
            gl = slice(go.value // 4, go.value // 4 + gn.value)
            g = wss[0].view(torch.float32)[gl] + wss[1].view(torch.float32)[gl]
            wss[0].view(torch.float32)[gl] = g
            wss[1].view(torch.float32)[gl] = g
    for k in range(2):
        assert abs(out_loss[k].item() - want) < 1e-4 * (1 + abs(want))
        assert rel(out_grad[k], GOLD[f"{tag}_grad"]) < 1e-4
    assert torch.equal(out_grad[0], out_grad[1]) and torch.equal(out_loss[0], out_loss[1])
    # phase 3 = the three phases in one call (what a single process uses: merged finishing launches), rows = every patch, against the
    # separate phases over the same rows: the same kernels on the same partial sums (the loss sums' fp32 split is exact when nothing
    # is added) -- loss and gradient bit for bit; and with the reduced slots handed in from outside (exchange buffers)
    all_rows = torch.arange(B, dtype=torch.int32, device=DEV)
    res = {}
    for name, phases, ext in (("one call", (3,), False), ("phases", (0, 1, 2), False), ("phases, external slots", (0, 1, 2), True)):
        ws = torch.zeros((nbytes + 15) // 16 * 2, device=DEV, dtype=torch.float64)
        db = T(GOLD[f"{tag}_depth"]).contiguous()
        lo, gr = torch.empty((), device=DEV), torch.empty_like(cd)
        xm = torch.full((8,), float("nan"), device=DEV, dtype=torch.float64) if ext else None
        xs = torch.full((gn.value,), float("nan"), device=DEV) if ext else None
        for ph in phases:
            _lib.check(lib.nsos_geo_correlation_loss_rows(ph, P_(db), P_(cd), P_(ray_o), P_(ray_d.contiguous()), P_(neg), P_(all_rows), B, B, Cn,
                                                          P, P, *prm, 15.0, 1, P_(lo), P_(gr), P_(ws), ws.numel() * 8,
                                                          P_(xm) if ext else None, P_(xs) if ext else None, None), "rows")
        torch.cuda.synchronize()
        res[name] = (lo.clone(), gr.clone())
        if ext:
            assert torch.isfinite(xm).all() and torch.isfinite(xs).all()
    for name in ("phases", "phases, external slots"):
        assert torch.equal(res["one call"][0], res[name][0]) and torch.equal(res["one call"][1], res[name][1]), name
    assert torch.equal(res["one call"][0], loss.detach())


@pytest.mark.parametrize("tag", ["app_small", "app_full"])
def test_correlation_loss_row_partitioned_equals_golden(tag):
    """CorrelationLoss row-partitioned (nsos_app_correlation_loss_rows: what each rank of the patch-sharded step calls for its own
    patches): two "ranks" in one process -- both halves' phases, the two reduced buffers added by hand the way the all-reduces do
    -- give the golden loss on both, and each rank's gradient equals the golden gradient on ITS patches (zeros elsewhere);
    with one rank owning everything the result equals the single-call entry bit for bit."""
    mod = nerf_sos_amd.CorrelationLoss(ref_args())
    feats, sim = T(GOLD[f"{tag}_feats"]), T(GOLD[f"{tag}_sim"])
    code = T(GOLD[f"{tag}_code"])
    B = code.shape[0]
    want, want_grad = GOLD[f"{tag}_loss"][0], torch.from_numpy(np.asarray(GOLD[f"{tag}_grad"]))
    from nerf_sos_amd.losses import exchange_floats
    nx = exchange_floats(B, mod.feature_samples ** 2)
    assert nx == B * mod.feature_samples ** 2 * 4 + 8
    for split in ([list(range(B))], [list(range(0, B, 2)), list(range(1, B, 2))], [[], list(range(B))]):
        evals = []
        for rows in split:
            xm, xs = torch.zeros(8, device=DEV, dtype=torch.float64), torch.full((nx,), float("nan"), device=DEV)
            with InjectRand(GOLD[f"{tag}_rand1"], GOLD[f"{tag}_rand2"]):
                run, (lo, gr) = mod.rows_phased(feats, code, sim, rows, (xm, xs))
            evals.append((run, lo, gr, xm, xs, rows))
        for phase in range(3):
            for e in evals:
                e[0](phase)
            torch.cuda.synchronize()
            if phase < 2:
                bufs = [e[3 + phase] for e in evals]
                tot = torch.stack(bufs).sum(0)
                for b_ in bufs:
                    b_.copy_(tot)
        for run, lo, gr, xm, xs, rows in evals:
            assert abs(lo.item() - want) < 1e-4 * (1 + abs(want)), (split, lo.item(), want)
            own = torch.zeros(B, dtype=torch.bool)
            own[rows] = True
            if rows:
                assert rel(gr.cpu()[own], want_grad[own].numpy()) < 1e-4
            assert (not (~own).any()) or float(gr.cpu()[~own].abs().max()) == 0.0
        assert all(torch.equal(e[1], evals[0][1]) for e in evals)
        if len(split) == 1:
            with InjectRand(GOLD[f"{tag}_rand1"], GOLD[f"{tag}_rand2"]):
                l1, g1 = mod.value_and_grad(feats, code, sim)
            assert torch.equal(l1, evals[0][1]) and torch.equal(g1, evals[0][2])
    # the renderer's channel-last maps through the same entry
    code_cl = code.permute(0, 2, 3, 1).contiguous().permute(0, 3, 1, 2)
    xm, xs = torch.zeros(8, device=DEV, dtype=torch.float64), torch.zeros(nx, device=DEV)
    with InjectRand(GOLD[f"{tag}_rand1"], GOLD[f"{tag}_rand2"]):
        run, (lo, gr) = mod.rows_phased(feats, code_cl, sim, list(range(B)), (xm, xs))
    for phase in range(3):
        run(phase)
    assert torch.equal(lo, evals[0][1]) if len(split) == 1 else abs(lo.item() - want) < 1e-4 * (1 + abs(want))
    assert rel(gr, want_grad.numpy()) < 1e-4 and gr.permute(0, 2, 3, 1).is_contiguous()


def test_losses_on_rendered_patches_train_the_semantic_head():
    """The training step right after the path (engines/trainer.py:127-166, 201-203): rendered semantics -> both
    correlation losses -> backward through the frozen-backbone render -> the semantic head's gradients."""
    from oracle import torch_port as tp
    B, P = 2, 16
    torch.manual_seed(0)
    net = nerf_sos_amd.NeRFNet(N_samples=64, N_importance=128, use_semantics=True, sem_with_coord=True).to(DEV)
    for n_, p_ in net.named_parameters():
        p_.requires_grad = "semantic_linear" in n_
    net.eval()
    rays = tp.synthetic_rays(B * P * P, seed=5).to(DEV).reshape(2, B, P, P, 3)
    ret = net(rays, (tp.NEAR, tp.FAR), retraw=False)
    sem = ret["semantics"].permute(0, 3, 1, 2)
    depth = ret["depth"].permute(0, 3, 1, 2)
    feat = torch.randn(B, 384, 14, 14, device=DEV)
    sim = torch.rand(B, B, device=DEV)
    ro, rd = rays[0].permute(0, 3, 1, 2), rays[1].permute(0, 3, 1, 2)
    a = ref_args()
    loss = nerf_sos_amd.CorrelationLoss(a)(feat, sem, sim) + 0.01 * nerf_sos_amd.GeoCorrelationLoss(a)(depth, sem, [ro, rd, None], sim)
    loss.backward()
    g = net.nerf_fine.mlp.semantic_linear[2].weight.grad
    assert g is not None and torch.isfinite(g).all() and g.abs().max() > 0
    assert net.nerf_fine.mlp.pts_linears[0].weight.grad is None


@pytest.mark.parametrize("tag", ["geo_small", "geo_full"])
def test_geo_loss_of_two_codes_as_one_stacked_evaluation(tag):
    """sharding.geo_loss_both: the coarse and the fine semantic map against the same geometry as ONE evaluation over the 2B
    stacked patches (block-diagonal similarity matrix, geometry repeated) equals the two separate calls of the reference's
    training step (engines/trainer.py:147-166) in value and in both gradients."""
    from nerf_sos_amd import sharding
    mod = nerf_sos_amd.GeoCorrelationLoss(ref_args())
    depth, sim = T(GOLD[f"{tag}_depth"]), T(GOLD[f"{tag}_sim"])
    B, _, P, _ = depth.shape
    ray_o = T(GOLD[f"{tag}_ray_o"])[:, :, None, None].expand(B, 3, P, P).contiguous()
    ray_d = T(GOLD[f"{tag}_ray_d"])
    c0 = T(GOLD[f"{tag}_code"]).requires_grad_(True)
    g = torch.Generator(DEV).manual_seed(5)
    c1 = (T(GOLD[f"{tag}_code"]) + 0.3 * torch.randn(c0.shape, device=DEV, generator=g)).requires_grad_(True)
    rows = list(range(B))
    two = mod(depth.clone(), c0, [ray_o, ray_d, None], sim, rows=rows) + mod(depth.clone(), c1, [ray_o, ray_d, None], sim, rows=rows)
    two.backward()
    want0, want1 = c0.grad.clone(), c1.grad.clone()
    c0.grad = c1.grad = None
    one = sharding.geo_loss_both(mod, depth.clone(), c0, c1, ray_o, ray_d, sim, rows)
    one.backward()
    one_v, two_v = float(one.detach()), float(two.detach())
    assert abs(one_v - two_v) < 1e-6 * (1 + abs(two_v)), (one_v, two_v)
    assert rel(c0.grad, want0.cpu().numpy()) < 1e-5 and rel(c1.grad, want1.cpu().numpy()) < 1e-5
    # a subset of the rows (what one rank of the sharded step evaluates) stacks the same way
    if B >= 2:
        sub = [0]
        a = sharding.geo_loss_both(mod, depth.clone(), c0.detach(), c1.detach(), ray_o, ray_d, sim, sub)
        b = mod(depth.clone(), c0.detach(), [ray_o, ray_d, None], sim, rows=sub) + mod(depth.clone(), c1.detach(), [ray_o, ray_d, None], sim, rows=sub)
        assert abs(float(a) - float(b)) < 1e-6 * (1 + abs(float(b)))


@pytest.mark.parametrize("tag", ["geo_small", "geo_full"])
def test_geo_loss_pair_on_channel_last_tensors_equals_the_stacked_evaluation(tag):
    """GeoCorrelationLoss.forward_pair (nsos_geo_correlation_loss_pair: both codes against one geometry, stacked batch never
    materialised, the renderer's channel-last tensors read in place) is the SAME computation as sharding.geo_loss_both on
    permuted / repeated / concatenated copies: loss and both gradients bit for bit, all rows and a subset; the depth the
    caller hands in is left untouched."""
    from nerf_sos_amd import sharding
    mod = nerf_sos_amd.GeoCorrelationLoss(ref_args())
    depth, sim = T(GOLD[f"{tag}_depth"]), T(GOLD[f"{tag}_sim"])
    B, _, P, _ = depth.shape
    ray_o = T(GOLD[f"{tag}_ray_o"])[:, :, None, None].expand(B, 3, P, P).contiguous()
    ray_d = T(GOLD[f"{tag}_ray_d"])
    g = torch.Generator(DEV).manual_seed(5)
    code = T(GOLD[f"{tag}_code"])
    for rows in ([*range(B)], [0]) if B >= 2 else ([*range(B)],):
        c0 = code.clone().requires_grad_(True)
        c1 = (code + 0.3 * torch.randn(code.shape, device=DEV, generator=g)).requires_grad_(True)
        want = sharding.geo_loss_both(mod, depth.clone(), c0, c1, ray_o, ray_d, sim, rows)
        (3.0 * want).backward()
        # channel-last: what the renderer returns -- depth [B,P,P,1], semantics [B,P,P,C], rays [B,P,P,3]
        n0 = c0.detach().permute(0, 2, 3, 1).contiguous().requires_grad_(True)
        n1 = c1.detach().permute(0, 2, 3, 1).contiguous().requires_grad_(True)
        d_cl = depth.permute(0, 2, 3, 1).contiguous()
        keep = d_cl.clone()
        got = mod.forward_pair(d_cl, n0, n1, ray_o.permute(0, 2, 3, 1).contiguous(), ray_d.permute(0, 2, 3, 1).contiguous(), sim, rows=rows)
        (3.0 * got).backward()
        assert torch.equal(got.detach(), want.detach()), (float(got), float(want))
        assert torch.equal(n0.grad.permute(0, 3, 1, 2), c0.grad) and torch.equal(n1.grad.permute(0, 3, 1, 2), c1.grad)
        assert torch.equal(d_cl, keep)
    with torch.no_grad():      # no gradient wanted: same value
        assert torch.equal(mod.forward_pair(d_cl, n0, n1, ray_o.permute(0, 2, 3, 1).contiguous(), ray_d.permute(0, 2, 3, 1).contiguous(), sim, rows=rows), got.detach())


@pytest.mark.parametrize("tag", ["app_small", "app_full"])
def test_appearance_loss_reads_a_channel_last_code_in_place(tag):
    """CorrelationLoss on `semantics.permute(0,3,1,2)` -- a view of the renderer's [B,P,P,C] tensor -- takes the channel-last
    kernel path (no copy in, gradient written channel-last): loss and gradient bit-identical to the NCHW call with the same draws."""
    mod = nerf_sos_amd.CorrelationLoss(ref_args())
    feats, sim = T(GOLD[f"{tag}_feats"]), T(GOLD[f"{tag}_sim"])
    code = T(GOLD[f"{tag}_code"])
    a = code.clone().requires_grad_(True)
    mod.generator = torch.Generator(DEV).manual_seed(3)
    la = mod(feats, a, sim)
    la.backward()
    nhwc = code.permute(0, 2, 3, 1).contiguous().requires_grad_(True)
    view = nhwc.permute(0, 3, 1, 2)
    assert not view.is_contiguous()
    mod.generator = torch.Generator(DEV).manual_seed(3)
    lb = mod(feats, view, sim)
    lb.backward()
    assert torch.equal(la.detach(), lb.detach()) and torch.equal(nhwc.grad.permute(0, 3, 1, 2), a.grad)


_CON = np.load(os.path.join(os.path.dirname(__file__), "golden", "contrastive.npz"))


@pytest.mark.parametrize("tag", sorted(k[:-4] for k in _CON.files if k.endswith("_emb")))
def test_contrastive_loss_vs_reference(tag):
    """nsos_contrastive_loss / nerf_sos_amd.NeRFContrastive against the REAL class (utils/image.py:192-218; goldens from
    tests/golden/make_goldens_contrastive.py): loss and d loss / d embeddings within 1e-5 of scale (fp32 rounding of the
    cosine sums; the picked min / max pairs must be the reference's), NaN where the reference is NaN, run-to-run identical."""
    e = torch.from_numpy(_CON[f"{tag}_emb"]).to("cuda:0").requires_grad_(True)
    mod = nerf_sos_amd.NeRFContrastive(device="cuda:0")
    loss = mod(e)
    assert loss.shape == () and mod.batch_size == e.shape[0]
    (2.5 * loss).backward()
    want_l, want_g = _CON[f"{tag}_loss"][0], _CON[f"{tag}_grad"]
    if np.isnan(want_l):
        assert torch.isnan(loss).item()
        return
    assert abs(loss.item() - want_l) <= 1e-5 * (1 + abs(want_l)), (loss.item(), want_l)
    g = e.grad.cpu().numpy() / 2.5
    assert (np.abs(want_g) > 0).any(-1).sum() <= 4 and np.array_equal(np.abs(g) > 0, np.abs(want_g) > 0)   # the same (<= 4) rows carry gradient
    assert np.abs(g - want_g).max() <= 1e-5 * np.abs(want_g).max()
    e2 = e.detach().clone().requires_grad_(True)
    l2 = mod(e2)
    l2.backward()
    assert torch.equal(l2, loss) and np.array_equal(e2.grad.cpu().numpy() * 2.5, e.grad.cpu().numpy())
    with torch.no_grad():
        assert torch.equal(mod(e.detach()), loss.detach())
    with pytest.raises(NotImplementedError):
        nerf_sos_amd.NeRFContrastive(device="cuda:0", min_max_contrast=False)(e)


@pytest.mark.parametrize("shape", [(2, 64, 64, 2), (3, 9, 11, 2), (2, 20, 20, 3), (2, 16, 16, 1), (2, 13, 17, 4), (1, 64, 64, 2)])
def test_geo_column_gradient_from_the_row_pass_equals_the_separate_pass(shape, monkeypatch):
    """Pass 3 of the geometric loss also produces the gradient w.r.t. the column codes (a transposing wave reduction of the
    row-side terms, one partial per block of 64 rows, folded in fp64) instead of a fourth pass over all N^2 pairs
    (pair_cols_kernel, kept as the checker here and for the 32-row shape): same loss bit for bit, same gradient up to the
    summation order (fp32 over 64 rows then fp64, vs fp32 over 32 rows then fp64).  Ragged N, 1-4 channels, and the wide shape
    forced on patches small enough that the launcher would pick 32-row workgroups."""
    B, H, W, C = shape
    g = torch.Generator(DEV).manual_seed(11 + B + H)
    depth = 0.5 + 4.0 * torch.rand(B, 1, H, W, device=DEV, generator=g)
    depth[0, 0, 0, :3] = 50.0                                         # beyond max_depth: filtered
    code = torch.randn(B, C, H, W, device=DEV, generator=g)
    code[0, :, 1, 1] = code[0, :, 1, 2]                               # equal codes: the sign(0) branch
    ray_o = torch.randn(B, 3, device=DEV, generator=g)[:, :, None, None].expand(B, 3, H, W).contiguous()
    ray_d = torch.nn.functional.normalize(torch.randn(B, 3, H, W, device=DEV, generator=g), dim=1)
    sim = torch.rand(B, B, device=DEV, generator=g)
    mod = nerf_sos_amd.GeoCorrelationLoss(ref_args())
    monkeypatch.setenv("NSOS_GEO_FORCE_WIDE", "1")
    out = {}
    for kind in ("fused", "separate"):
        if kind == "separate":
            monkeypatch.setenv("NSOS_GEO_SEPARATE_COLS", "1")
        c = code.clone().requires_grad_(True)
        loss = mod(depth.clone(), c, [ray_o, ray_d, None], sim)
        loss.backward()
        out[kind] = (loss.detach().clone(), c.grad.clone())
    assert torch.equal(out["fused"][0], out["separate"][0])
    assert torch.isfinite(out["fused"][1]).all() and (C == 1 or out["separate"][1].abs().max() > 0)   # one channel: normalised code = +-1, no gradient
    assert (out["fused"][1] - out["separate"][1]).abs().max() <= 2e-6 * out["separate"][1].abs().max()
    # the fused pass is deterministic
    c = code.clone().requires_grad_(True)
    monkeypatch.delenv("NSOS_GEO_SEPARATE_COLS")
    mod(depth.clone(), c, [ray_o, ray_d, None], sim).backward()
    assert torch.equal(c.grad, out["fused"][1])


@pytest.mark.parametrize("B", [1, 2, 5, 16, 120])
def test_similarity_negatives_vs_torch(B):
    """nsos_similarity_negatives against the reference's two steps in torch: get_similarity_matrix (utils/image.py:186-189) and
    torch.min(sim, dim=0)[1] (:354).  Tokens with a common offset (like DINO class tokens: similarities near 1, small gaps)."""
    from nerf_sos_amd.losses import similarity_negatives
    g = torch.Generator(DEV).manual_seed(B)
    x = torch.randn(B, 384, device=DEV, generator=g) + 3.0 * torch.randn(1, 384, device=DEV, generator=g)
    want_sim = torch.nn.functional.cosine_similarity(x.unsqueeze(0), x.unsqueeze(1), dim=2)
    neg2, sim = similarity_negatives(x, copies=2, want_similarity=True)
    assert neg2.dtype == torch.int64 and neg2.shape == (2 * B,) and sim.shape == (B, B)
    assert float((sim - want_sim).abs().max()) < 2e-6
    assert torch.equal(neg2[B:], neg2[:B] + B)
    want_neg = torch.min(want_sim, dim=0)[1]
    # equal picks wherever the column's two smallest entries are further apart than the two evaluations' rounding
    srt = torch.sort(want_sim, dim=0)[0]
    clear = (srt[1] - srt[0] > 1e-5) if B > 1 else torch.ones(B, dtype=torch.bool, device=DEV)
    assert bool(clear.float().mean() > 0.9) and torch.equal(neg2[:B][clear], want_neg[clear])
    picked = sim[neg2[:B], torch.arange(B, device=DEV)]
    assert float((picked - sim.min(dim=0)[0]).abs().max()) == 0.0          # always an arg-min of its own matrix
    assert torch.equal(similarity_negatives(x), neg2[:B])                   # deterministic, copies = 1
