"""Architectures other than the shipped one (VERDICT r03 "missing 1": every ctor kwarg the reference accepts, not only its
default): `nerf_sos_amd.NeRFNet(**kwargs)` builds the reference's module tree (same state_dict names, shapes, seed-identical
initial values) and renders through the generic fp32 kernel (csrc/mlp_generic.hip) within 1e-4 of the REAL reference's outputs
(tests/golden/generic.npz, written by tests/golden/make_goldens_generic.py from /root/reference; the same script proves
oracle/torch_port.py bit-identical to the reference on every case).

CPU: the constructor contract and the port against the goldens.  GPU (-m gpu): the renders."""
import numpy as np
import pytest
import torch

import nerf_sos_amd
from oracle import torch_port as tp
from helpers import GENERIC_CASES, close, generic_state

NAMES = list(GENERIC_CASES)


@pytest.mark.parametrize("name", NAMES)
def test_constructor_builds_the_reference_module_tree(golden, name):
    """Same parameter names, shapes and -- under the same torch seed -- the same initial VALUES as the reference's NeRFNet (creation
    order is the RNG order: models/nerf_mlp.py:40-64)."""
    g = golden("generic")
    cfg, sd_ref = generic_state(name, golden)
    torch.manual_seed(int(g[f"{name}__seed"][0]))
    net = nerf_sos_amd.NeRFNet(**GENERIC_CASES[name][0])
    sd = net.state_dict()
    assert list(sd) == list(sd_ref)
    plain = tp.init_state_dict(cfg, seed=int(g[f"{name}__seed"][0]))          # before the spiky transform
    for k in sd:
        assert sd[k].shape == sd_ref[k].shape, k
        assert torch.equal(sd[k], plain[k]), f"{k}: initial values differ from the reference's"
    net.load_state_dict(sd_ref)                                                # strict
    assert net.nerf.fast is False and (net.nerf_fine is net.nerf) == (cfg.n_importance == 0)


@pytest.mark.parametrize("name", NAMES)
def test_port_reproduces_the_reference_goldens(golden, name):
    """oracle/torch_port.py with the round-4 PortConfig fields (use_viewdirs, use_embed, sem_layer, sem_with_geo) against the
    reference's recorded outputs: bit for bit on CPU (same ATen ops in the same order)."""
    g = golden("generic")
    cfg, sd = generic_state(name, golden)
    with torch.no_grad():
        out = tp.render(sd, cfg, torch.from_numpy(g[f"{name}__rays"]), (tp.NEAR, tp.FAR))
        q = tp.point_query(sd, "nerf_fine", torch.from_numpy(g[f"{name}__pts"]),
                           torch.from_numpy(g[f"{name}__dirs"]) if cfg.use_viewdirs else None, cfg)
    keys = sorted(k[len(name) + 7:] for k in g if k.startswith(name + "__out__"))
    assert sorted(out) == keys
    for k in keys:
        assert np.array_equal(out[k].numpy(), g[f"{name}__out__{k}"]), k
    assert np.array_equal(q.numpy(), g[f"{name}__query"])


def test_refusals_are_loud():
    with pytest.raises(NotImplementedError):
        nerf_sos_amd.NeRFNet(conv_embed=True)
    with pytest.raises(ValueError):                       # the reference's own forward fails for this combination
        nerf_sos_amd.NeRFNet(use_embed=False, viewdirs=True)
    net = nerf_sos_amd.NeRFNet(netdepth=4, netwidth=64, netdepth_fine=4, netwidth_fine=64)
    net.mlp_precision = "bf16"
    with pytest.raises(NotImplementedError):
        net.render_rays(torch.zeros(1, 3), torch.ones(1, 3), torch.ones(1), torch.ones(1) * 2)


# ------------------------------------------------------------------------------------------------------------- GPU
DEV = "cuda:0"


@pytest.mark.gpu
@pytest.mark.parametrize("name", NAMES)
def test_generic_render_vs_reference(golden, name):
    """Eval-mode NeRFNet.forward and the direct point query of every generic case against the real reference's outputs.  Coarse
    pass and point query: strictly within 1e-4.  Fine pass: within 1e-4 on the reference's own fine sample positions
    (z_fine_override: the sampler's last-ulp index flips are the shipped path's business, tests/test_gpu_pins.py), and free-running
    on all but a few rays."""
    g = golden("generic")
    cfg, sd = generic_state(name, golden)
    net = nerf_sos_amd.NeRFNet(**GENERIC_CASES[name][0]).to(DEV).eval()
    net.load_state_dict(sd)
    rays = torch.from_numpy(g[f"{name}__rays"]).to(DEV)
    ref = {k[len(name) + 7:]: g[k] for k in g if k.startswith(name + "__out__")}
    with torch.no_grad():
        out = net(rays, (tp.NEAR, tp.FAR))
        q = net.nerf_fine(torch.from_numpy(g[f"{name}__pts"]).to(DEV),
                          torch.from_numpy(g[f"{name}__dirs"]).to(DEV) if cfg.use_viewdirs else None)
    assert sorted(out) == sorted(ref)
    for k in ref:
        assert tuple(out[k].shape) == ref[k].shape, (k, out[k].shape, ref[k].shape)
    close(q.cpu().numpy(), g[f"{name}__query"], what=f"{name}: point query")
    fine = cfg.n_importance > 0
    for k in ref:
        if not fine or k.endswith("0"):
            close(out[k].cpu().numpy(), ref[k], what=f"{name}: {k}")
    if fine:
        # the reference's fine sample positions, rebuilt with the port from the reference's coarse weights (bit-identical on CPU)
        R = rays.shape[1]
        near, far = torch.full((R, 1), tp.NEAR), torch.full((R, 1), tp.FAR)
        z = tp.stratified_z(near, far, cfg.n_samples, None)
        z_fine, _ = tp.importance_z(z, torch.from_numpy(ref["weights0"]), cfg.n_importance, None)
        with torch.no_grad():
            pinned = net(rays, (tp.NEAR, tp.FAR), z_fine_override=z_fine.to(DEV))
        for k in ref:
            if not k.endswith("0") and k != "z_std":
                close(pinned[k].cpu().numpy(), ref[k], what=f"{name}: {k} on the reference's z_fine")
        bad = np.zeros(R, bool)
        for k in ("rgb", "depth", "acc"):
            a, b = out[k].cpu().numpy().reshape(R, -1), ref[k].reshape(R, -1)
            bad |= (np.abs(a - b) > 1e-4 * (1 + np.abs(b))).any(-1)
        assert bad.sum() <= 2, f"{name}: {bad.sum()} of {R} rays outside 1e-4 free-running"


@pytest.mark.gpu
@pytest.mark.parametrize("name", ["d6w96_m6", "d4w128"])
def test_generic_render_at_c2_size(golden, name):
    """BASELINE C2 size (4096 rays x (64 + 128)) on a generic architecture against the CPU port (bit-identical to the reference on CPU):
    the coarse pass strictly within 1e-4 on every key; with the port's fine positions handed in, every fine key strictly within 1e-4 on
    all 4096 rays (the free-running sampler's index flips are measured on the shipped path: tests/test_gpu_pins.py)."""
    import os
    cfg0, sd = generic_state(name, golden)
    kw_net = dict(GENERIC_CASES[name][0], N_samples=64, N_importance=128)
    cfg = tp.PortConfig(**dict(GENERIC_CASES[name][1], n_samples=64, n_importance=128))
    rays = tp.synthetic_rays(4096, seed=0)
    torch.set_num_threads(min(32, os.cpu_count() or 1))
    with torch.no_grad():
        ref = tp.render(sd, cfg, rays, (tp.NEAR, tp.FAR))
    near, far = torch.full((4096, 1), tp.NEAR), torch.full((4096, 1), tp.FAR)
    z = tp.stratified_z(near, far, 64, None)
    z_fine = tp.importance_z(z, ref["weights0"], 128, None)[0]
    net = nerf_sos_amd.NeRFNet(**kw_net).to(DEV).eval()
    net.load_state_dict(sd)
    with torch.no_grad():
        out = net(rays.to(DEV), (tp.NEAR, tp.FAR), z_fine_override=z_fine.to(DEV))
    assert sorted(out) == sorted(ref)
    for k in ref:
        if k != "z_std":
            close(out[k].cpu().numpy(), ref[k].numpy(), what=f"{name} at C2 size: {k}")


@pytest.mark.gpu
def test_generic_kernel_on_the_shipped_architecture_equals_the_fused_kernel(golden, manifest):
    """The generic kernel fed the SHIPPED architecture (MLP.fast forced off) against the hand-scheduled exact-fp32 kernel: both are
    fmaf chains on the same matrix instruction, in different contraction orders -- agreement to fp32 rounding, far inside 1e-4;
    ragged point counts, point mode and ray mode."""
    from helpers import CFGS, ref_state
    for name in ("nosem", "semcoord", "sem"):
        net = nerf_sos_amd.NeRFNet(N_samples=64, N_importance=128, **CFGS[name]).to(DEV).eval()
        net.load_state_dict(ref_state(name, manifest, peaky=True))
        mlp = net.nerf_fine
        for P in (1, 31, 33, 1000):
            gen = torch.Generator().manual_seed(P)
            pts = (torch.rand(P, 3, generator=gen) * 6 - 3).to(DEV)
            dirs = torch.nn.functional.normalize(torch.randn(P, 3, generator=gen), dim=-1).to(DEV)
            with torch.no_grad():
                want = mlp(pts, dirs)
                mlp.fast = False
                try:
                    got = mlp(pts, dirs)
                finally:
                    mlp.fast = True
                    mlp.invalidate_packed()
            assert got.shape == want.shape
            err = float(((got - want).abs() / (1 + want.abs())).max())
            assert err < 2e-5, (name, P, err)


def _fine_positions(cfg, ref, R):
    """The reference's fine sample positions, rebuilt with the port from the reference's coarse weights (bit-identical on CPU)."""
    near, far = torch.full((R, 1), tp.NEAR), torch.full((R, 1), tp.FAR)
    z = tp.stratified_z(near, far, cfg.n_samples, None)
    return tp.importance_z(z, torch.from_numpy(ref["weights0"]), cfg.n_importance, None)[0]


@pytest.mark.gpu
@pytest.mark.parametrize("name", NAMES)
def test_generic_training_vs_reference_autograd(golden, name):
    """Every parameter of every generic case trainable: the gradients of a random linear functional of all rendered maps against
    the REAL reference's autograd (tests/golden/generic_grads.npz, make_goldens_generic_grads.py), 1e-4 of each gradient's
    scale, with the fine positions pinned to the reference's (index flips of the free-running sampler move fine-net gradients by
    per cents on 24 rays: tests/test_gpu_pins.py).  The training variant's outputs are bit-identical to the inference kernel's."""
    g, gg = golden("generic"), golden("generic_grads")
    cfg, sd = generic_state(name, golden)
    net = nerf_sos_amd.NeRFNet(**GENERIC_CASES[name][0]).to(DEV).eval()
    net.load_state_dict(sd)
    rays = torch.from_numpy(g[f"{name}__rays"]).to(DEV)
    ref = {k[len(name) + 7:]: g[k] for k in g if k.startswith(name + "__out__")}
    kw = {}
    if cfg.n_importance > 0:
        kw["z_fine_override"] = _fine_positions(cfg, ref, rays.shape[1]).to(DEV)
    with torch.no_grad():
        want = net(rays, (tp.NEAR, tp.FAR), **kw)
    ret = net(rays, (tp.NEAR, tp.FAR), **kw)
    loss, n_terms = 0.0, 0
    for k in ret:
        assert torch.equal(ret[k].detach(), want[k]), f"{name}: the training variant changed {k}"
        gk = f"{name}__G__{k}"
        if gk in gg:
            assert ret[k].requires_grad, k
            loss = loss + (ret[k] * torch.from_numpy(gg[gk]).to(DEV)).sum()
            n_terms += 1
    assert n_terms == sum(1 for k in gg if k.startswith(name + "__G__"))
    loss.backward()
    worst, n_checked = {}, 0
    for n_, p_ in net.named_parameters():
        got = p_.grad
        assert got is not None, n_
        refs = []
        if f"{name}__grad__{n_}" in gg:
            refs.append((got, gg[f"{name}__grad__{n_}"]))
        elif f"{name}__gradrows__{n_}" in gg:
            refs.append((got[::max(1, got.shape[0] // 24)], gg[f"{name}__gradrows__{n_}"]))
            refs.append((got[:, ::max(1, got.shape[1] // 24)], gg[f"{name}__gradcols__{n_}"]))
        for a, b in refs:
            b = torch.from_numpy(b)
            assert a.shape == b.shape, (n_, a.shape, b.shape)
            scale = float(b.abs().max()) + 1e-20
            worst[n_] = max(worst.get(n_, 0.0), float((a.detach().cpu() - b).abs().max()) / scale)
        n_checked += bool(refs)
    assert n_checked == sum(1 for k in gg if k.startswith(name + "__grad")) - sum(1 for k in gg if k.startswith(name + "__gradcols__"))
    bad = {k: v for k, v in worst.items() if v > 1e-4}
    assert not bad, f"{name}: gradients off by more than 1e-4 of their scale: {bad}"


@pytest.mark.gpu
@pytest.mark.parametrize("name", ["d6w96_m6", "deepsem3_geo"])
def test_generic_training_of_parameter_subsets(golden, name):
    """The shipped recipe on a generic net (only semantic_linear.* trainable, run_nerf.py:307-318) and other subsets: the input-gradient
    chain is cut to what the trainable Linears need (nsos_mlp_generic_pack_bwd_subset: with a frozen backbone it stops at the head), the
    trainable parameters' gradients equal the all-parameters run's bit for bit, nothing else receives one; 37 rays x 16 + 37 x 32
    samples (ragged 32-point tiles) in train mode."""
    cfg, sd = generic_state(name, golden)
    rays = tp.synthetic_rays(37, seed=12).to(DEV)
    subsets = [None, ("semantic_linear",), ("alpha_linear", "pts_linears.3."), ("rgb_linear", "nerf_fine.mlp.pts_linears.0.")]
    if name == "deepsem3_geo":
        subsets += [("geo_map_sem",), ("semantic_linear.2.",)]
    full = None
    for subset in subsets:
        net = nerf_sos_amd.NeRFNet(**GENERIC_CASES[name][0]).to(DEV).train()
        net.load_state_dict(sd)
        net.rng, net.rng_seed = "philox", 7
        chosen = lambda n_: subset is None or any(t in n_ for t in subset)  # noqa: E731
        for n_, p_ in net.named_parameters():
            p_.requires_grad_(chosen(n_))
        ret = net(rays, (tp.NEAR, tp.FAR))
        (ret["semantics"].sum() + 0.5 * ret["semantics0"].square().sum() + ret["rgb"].sum() + ret["acc0"].sum()).backward()
        grads = {n_: (None if p_.grad is None else p_.grad.clone()) for n_, p_ in net.named_parameters()}
        if subset is None:
            full = grads
            continue
        n_sel = 0
        for n_ in grads:
            if chosen(n_):
                assert grads[n_] is not None and torch.equal(grads[n_], full[n_]), (subset, n_)
                n_sel += 1
            else:
                assert grads[n_] is None, (subset, n_)
        assert n_sel >= 2, subset


@pytest.mark.gpu
def test_generic_training_step_reduces_the_loss():
    """Thirty Adam steps of a 4 x 64 net without view directions on a fixed batch: the loss falls (the gradients point downhill and
    the re-packed streams follow the optimizer)."""
    torch.manual_seed(0)
    net = nerf_sos_amd.NeRFNet(netdepth=4, netwidth=64, netdepth_fine=4, netwidth_fine=64, N_samples=16, N_importance=16, viewdirs=False).to(DEV).train()
    rays = tp.synthetic_rays(64, seed=1).to(DEV)
    target = torch.rand(64, 3, device=DEV)
    opt = torch.optim.Adam(net.parameters(), lr=5e-3)
    losses = []
    for _ in range(30):
        opt.zero_grad(set_to_none=True)
        ret = net(rays, (tp.NEAR, tp.FAR))
        loss = (ret["rgb"] - target).square().mean() + (ret["rgb0"] - target).square().mean()
        loss.backward()
        opt.step()
        losses.append(float(loss.detach()))
    assert losses[-1] < 0.97 * losses[0] and all(losses[i + 5] < losses[i] for i in range(0, 25, 5)), losses


@pytest.mark.gpu
def test_generic_net_in_hip_graphs(golden):
    """A trainable generic-architecture net inside HIP graphs: the weight streams are re-packed INSIDE the graph by kernels alone
    (nsos_mlp_generic_repack: the program header stays where the first pack put it), so (1) a GraphedRender replay equals the eager render
    and follows an in-place parameter update, and (2) a whole captured training step (train-mode render with the device-side Philox
    counter, MSE, backward through both nets on the generic kernels, capturable Adam) replays to the same parameters as eager steps."""
    name = "d6w96_m6"
    cfg, sd = generic_state(name, golden)
    rays = tp.synthetic_rays(96, seed=4).to(DEV)
    net = nerf_sos_amd.NeRFNet(**GENERIC_CASES[name][0]).to(DEV).eval()
    net.load_state_dict(sd)
    gr = nerf_sos_amd.GraphedRender(net, 96, (tp.NEAR, tp.FAR))
    with torch.no_grad():
        want = net(rays, (tp.NEAR, tp.FAR))
        got = {k: v.clone() for k, v in gr(rays).items()}
        assert all(torch.equal(got[k], want[k]) for k in want)
        net.nerf_fine.mlp.rgb_linear.bias.add_(0.25)          # what an optimizer step does: in place
        want2 = net(rays, (tp.NEAR, tp.FAR))
        got2 = gr(rays)
        assert torch.equal(got2["rgb"], want2["rgb"]) and not torch.equal(want2["rgb"], want["rgb"])

    def make():
        torch.manual_seed(0)
        n = nerf_sos_amd.NeRFNet(**GENERIC_CASES[name][0]).to(DEV).train()
        n.load_state_dict(sd)
        n.rng, n.rng_seed = "philox", 3
        n.use_device_rng_counter(DEV)
        return n, torch.optim.Adam(n.parameters(), lr=1e-3, capturable=True)

    target = torch.rand(96, 3, device=DEV)

    def step(n, opt):
        opt.zero_grad(set_to_none=False)
        ret = n(rays, (tp.NEAR, tp.FAR), retraw=False)
        loss = ((ret["rgb"] - target) ** 2).mean() + ((ret["semantics0"]) ** 2).mean()
        loss.backward()
        opt.step()
        return loss

    a, opt_a = make()
    for _ in range(5):                 # = 2 eager warm-up steps + 3 replays (the capture itself executes nothing)
        step(a, opt_a)
    b, opt_b = make()
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        for _ in range(2):
            step(b, opt_b)
    torch.cuda.current_stream().wait_stream(side)
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        step(b, opt_b)
    for _ in range(3):
        g.replay()
    torch.cuda.synchronize()
    for (n_, pa), (_, pb) in zip(a.named_parameters(), b.named_parameters()):
        assert torch.equal(pa, pb), f"captured training step diverged from eager in {n_}"


def _fuzz_config(seed):
    import random
    r = random.Random(seed)
    viewdirs = r.random() < 0.75
    embed = True if viewdirs else r.random() < 0.7
    sem = viewdirs and r.random() < 0.7
    kw = dict(netdepth=r.choice([1, 2, 3, 5, 6, 7, 9, 12]), netwidth=r.choice([8, 40, 64, 100, 136, 256, 320]),
              multires=r.choice([0, 1, 3, 7, 10]), multires_views=r.choice([0, 1, 4, 5]), viewdirs=viewdirs, use_embed=embed,
              use_semantics=sem, sem_layer=r.choice([2, 2, 3, 4, 5]), sem_dim=r.choice([1, 2, 3, 8]),
              sem_with_coord=r.random() < 0.5, sem_with_geo=r.random() < 0.35, white_bkgd=r.random() < 0.5,
              N_samples=r.choice([6, 8, 11]), N_importance=r.choice([0, 8, 13]))
    kw["netdepth_fine"], kw["netwidth_fine"] = kw["netdepth"], kw["netwidth"]
    port = dict(net_depth=kw["netdepth"], net_width=kw["netwidth"], multires=kw["multires"], multires_views=kw["multires_views"],
                use_viewdirs=viewdirs, use_embed=embed, use_semantics=sem, sem_layer=kw["sem_layer"], sem_dim=kw["sem_dim"],
                sem_with_coord=kw["sem_with_coord"], sem_with_geo=kw["sem_with_geo"], white_bkgd=kw["white_bkgd"],
                n_samples=kw["N_samples"], n_importance=kw["N_importance"])
    return kw, port


@pytest.mark.gpu
@pytest.mark.parametrize("seed", range(14))
def test_generic_fuzz_forward_and_gradients_vs_port(seed):
    """Randomly drawn constructor arguments (depths 1-12, widths 8-320 incl. non-multiples of 32, 0-10 octaves, every head shape, with and
    without view directions / embedding / white background): the module tree loads the port's state dict strictly, the render equals the
    port's within 1e-4 (fine pass on the port's positions) and every parameter's gradient of a random functional equals the port's autograd
    within 1e-4 of its scale.  (The port is pinned bit for bit to the real reference on the eight committed cases; here it extrapolates.)"""
    kw, port_kw = _fuzz_config(seed)
    cfg = tp.PortConfig(**port_kw)
    sd = {k: v.clone() for k, v in tp.init_state_dict(cfg, seed=seed).items()}
    for net_ in ("nerf", "nerf_fine"):
        for k in (f"{net_}.mlp.alpha_linear", f"{net_}.mlp.output_linear"):
            if k + ".weight" in sd and (net_ == "nerf" or cfg.n_importance > 0):
                row = slice(None) if "alpha" in k else slice(3, 4)
                sd[k + ".weight"][row] *= 40.0
                sd[k + ".bias"][row] = sd[k + ".bias"][row] * 40.0 + 1.0
    if cfg.n_importance == 0:
        for k in [k for k in sd if k.startswith("nerf.")]:
            sd["nerf_fine." + k[len("nerf."):]] = sd[k]
    net = nerf_sos_amd.NeRFNet(**kw).to(DEV).eval()
    net.load_state_dict(sd)                                   # strict: same names and shapes as the port's view of the reference
    if net.nerf.fast:
        pytest.skip("drew the shipped architecture")
    rays = tp.synthetic_rays(19, seed=seed)
    ray_grads = seed % 2 == 1                                 # odd seeds: the rays require grad too (pose refinement)
    rc = rays.clone().requires_grad_(ray_grads)
    sdg = {k: v.clone().requires_grad_(True) for k, v in sd.items()}
    ref = tp.render(sdg, cfg, rc, (tp.NEAR, tp.FAR))
    gen = torch.Generator().manual_seed(seed)
    ups = {k: torch.randn(ref[k].shape, generator=gen) * (0.05 if k.startswith("raw") else 1.0)
           for k in ref if k.rstrip("0") in ("rgb", "semantics", "acc", "weights", "raw") and ref[k].numel()}
    sum((ref[k] * ups[k]).sum() for k in ups).backward()
    extra = {}
    if cfg.n_importance > 0:
        R = rays.shape[1]
        z = tp.stratified_z(torch.full((R, 1), tp.NEAR), torch.full((R, 1), tp.FAR), cfg.n_samples, None)
        extra["z_fine_override"] = tp.importance_z(z, ref["weights0"].detach(), cfg.n_importance, None)[0].to(DEV)
    rg = rays.to(DEV).requires_grad_(ray_grads)
    out = net(rg, (tp.NEAR, tp.FAR), **extra)
    assert sorted(out) == sorted(ref), (kw, sorted(out), sorted(ref))
    for k in ref:
        if k != "z_std":
            close(out[k].detach().cpu().numpy(), ref[k].detach().numpy(), what=f"fuzz {seed} {kw}: {k}")
    sum((out[k] * ups[k].to(DEV)).sum() for k in ups).backward()
    bad = {}
    for n_, p_ in net.named_parameters():
        want = sdg[n_].grad
        if want is None:
            continue
        e = float((p_.grad.cpu() - want).abs().max() / (want.abs().max() + 1e-20))
        if e > 1e-4:
            bad[n_] = e
    assert not bad, (kw, bad)
    if ray_grads:
        for i, what in enumerate(("rays_o", "rays_d")):
            e = float((rg.grad[i].cpu() - rc.grad[i]).abs().max() / (rc.grad[i].abs().max() + 1e-20))
            assert e <= 2e-4, (kw, what, e)
