"""CPU: host-side mirror of the reference interface -- parameter names, shapes, init order, state-dict
interchange, mode switch, unsupported-architecture errors."""
import pytest
import torch

import nerf_sos_amd
from helpers import CFGS, ref_state, state_sha


@pytest.mark.parametrize("name", list(CFGS))
def test_same_init_and_keys_as_reference(manifest, name):
    torch.manual_seed(0)
    net = nerf_sos_amd.NeRFNet(N_samples=64, N_importance=128, **CFGS[name])
    sd = net.state_dict()
    assert list(sd.keys()) == manifest["state_keys"][name]
    assert state_sha(sd) == manifest["state_sha256"][name], "seed-0 init must equal the reference's bit for bit"
    want = ref_state(name, manifest)
    for k in sd:
        assert sd[k].shape == want[k].shape


def test_coarse_only_aliases_fine(manifest):
    torch.manual_seed(0)
    net = nerf_sos_amd.NeRFNet(N_samples=64, N_importance=0)
    assert net.nerf_fine is net.nerf
    assert state_sha(net.state_dict()) == manifest["state_sha256"]["nosem_coarse_only"]


def test_load_reference_state_dict_strict(manifest):
    net = nerf_sos_amd.NeRFNet(N_samples=64, N_importance=128, use_semantics=True, sem_with_coord=True)
    net.load_state_dict(ref_state("semcoord", manifest, peaky=True), strict=True)
    # frozen-backbone recipe of run_nerf.py:307-318 works on the mirrored names
    n_train = 0
    for n, p in net.named_parameters():
        p.requires_grad = 'semantic_linear' in n
        n_train += p.numel() if p.requires_grad else 0
    assert n_train == 82436
    assert sum(p.numel() for p in net.parameters()) == 1274124


def test_mode_kwargs():
    net = nerf_sos_amd.NeRFNet(N_samples=64, N_importance=128, perturb=1.0, raw_noise_std=1.0)
    assert net.render_kwargs_train["perturb"] == 1.0 and net.render_kwargs_train["retraw"] is True
    assert net.render_kwargs_test["perturb"] == 0.0 and net.render_kwargs_test["raw_noise_std"] == 0.0
    assert net.chunk == 32768 and net.nerf.chunk == 65536


@pytest.mark.parametrize("kw", [dict(netwidth=128), dict(multires=6), dict(viewdirs=False), dict(use_semantics=True, sem_layer=4),
                                dict(use_semantics=True, sem_dim=5), dict(use_semantics=True, sem_with_geo=True), dict(netdepth=3, netdepth_fine=5)])
def test_other_architectures_construct_and_take_the_generic_kernel(kw):
    """Every ctor kwarg the reference accepts constructs (round 4); anything but the shipped architecture is marked for the
    generic fp32 kernel (tests/test_generic_arch.py renders them against the reference's goldens)."""
    net = nerf_sos_amd.NeRFNet(**kw)
    assert not (net.nerf.fast and net.nerf_fine.fast)
    assert nerf_sos_amd.NeRFNet().nerf.fast


def test_conv_embed_is_refused():
    with pytest.raises(NotImplementedError, match="conv_embed"):
        nerf_sos_amd.NeRFNet(conv_embed=True)


def test_shape_assert_matches_reference():
    net = nerf_sos_amd.NeRFNet()
    with torch.no_grad(), pytest.raises(AssertionError):
        net((torch.zeros(4, 3), torch.zeros(5, 3)), (1.0, 2.0))


def test_bench_cpu_baseline_leg_runs_on_a_tiny_sample():
    """bench.py's `cpu_baseline` object (the torch port timed on host cores) -- exercised here on 16 rays so that a
    broken import or signature shows up in the CPU suite, not only on the GPU box."""
    import importlib.util
    import os
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    spec = importlib.util.spec_from_file_location("bench_module", os.path.join(root, "bench.py"))
    bench = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(bench)
    threads = torch.get_num_threads()
    try:
        rec, parity = bench.cpu_baseline(None, 16)
        # the parity object, on a stand-in "GPU" render: the port's own output with 3e-5 of noise on the fine maps
        from oracle import torch_port as tp
        cfg = tp.PortConfig(n_samples=64, n_importance=128, use_semantics=False, pts_chunk=1024 * 256)
        sd, rays = tp.init_state_dict(cfg, seed=0), tp.synthetic_rays(16, seed=0)
        with torch.no_grad():
            ref = tp.render(sd, cfg, rays, (tp.NEAR, tp.FAR))
        got = {k: (v + (3e-4 if k == "rgb" else 0.0) * (torch.arange(v.numel()).reshape(v.shape) % 2)) for k, v in ref.items()}
        rec2, parity2 = bench.cpu_baseline((sd, rays, got), dense=(sd, ref))
        assert parity2["dense_field"]["frac_rays_outside_1e-4_any_fine_map"] == 0.0 and parity2["psnr_db"]["rgb"] > 200
        parity2 = parity2["timed_workload_default_init_field"]
    finally:
        torch.set_num_threads(threads)
    assert rec["unit"] == "rays/s" and rec["value"] > 0 and rec["kind"] == "port" and 1 <= rec["cores"] <= 64
    assert rec["physical_cores"] >= 1 and rec["runs"] == 5 and rec["warmups"] == 2 and parity is None      # SURVEY 8(d) protocol
    assert {"anomaly_mode_on", "anomaly_mode_off"} <= set(rec["train_fwd_bwd"]) and rec["train_fwd_bwd"]["anomaly_mode_on"]["value"] > 0
    assert parity2["coarse_pass_all_rays_inside_1e-4"] and parity2["per_key"]["depth"]["frac_rays_outside_1e-4"] == 0.0
    assert parity2["per_key"]["rgb"]["frac_rays_outside_1e-4"] == 1.0 and 60 < parity2["psnr_db"]["rgb"] < 80
    assert parity2["frac_rays_outside_1e-4_any_fine_map"] == 1.0 and parity2["psnr_db"]["rgb0"] > 200
    assert 2 * bench.MAC_NOSEM * bench.EVALS_PER_RAY == 2 * 593408 * 256


def test_bench_self_launch_command(monkeypatch):
    """`python bench.py --gpus N` outside torch.distributed.run must become the launcher: one rank per GPU on
    127.0.0.1 -- checked without starting anything."""
    import importlib.util
    import os
    import sys
    import types
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    spec = importlib.util.spec_from_file_location("bench_module2", os.path.join(root, "bench.py"))
    bench = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(bench)
    seen = {}
    monkeypatch.setattr(os, "execv", lambda exe, cmd: seen.update(exe=exe, cmd=cmd))
    monkeypatch.delenv("WORLD_SIZE", raising=False)
    monkeypatch.setattr(sys, "argv", ["bench.py", "--gpus", "8", "--steps", "5", "--warmup", "2"])
    bench._self_launch(types.SimpleNamespace(gpus=8))
    cmd = seen["cmd"]
    assert cmd[1:3] == ["-m", "torch.distributed.run"] and "--nproc-per-node=8" in cmd
    assert cmd[cmd.index("--master-addr") + 1] == "127.0.0.1"
    assert cmd[-6:] == ["--gpus", "8", "--steps", "5", "--warmup", "2"] and cmd[-7].endswith("bench.py")
    seen.clear()
    monkeypatch.setenv("WORLD_SIZE", "8")                     # already under the launcher: nothing to do
    bench._self_launch(types.SimpleNamespace(gpus=8))
    bench._self_launch(types.SimpleNamespace(gpus=1))
    assert not seen


class _FakePlan:
    """Stands in for ops.PackPlan on a machine without a GPU: counts the pack launches it would issue."""
    runs = 0

    def __init__(self, params, sem_mode):
        self.ptrs = tuple(p.data_ptr() for p in params.values())

    heads = 0

    def run(self, out, precision, heads_only=False):
        _FakePlan.runs += 1
        _FakePlan.heads += int(heads_only)
        return torch.zeros(1) if out is None else out


def test_packed_weights_policy(monkeypatch):
    """Trainable nets re-pack on every call (fused optimizers update parameters without bumping Tensor._version, which a
    version-keyed cache would miss); frozen nets pack once and again only when (data_ptr, _version) changes."""
    import copy
    from nerf_sos_amd import ops
    monkeypatch.setattr(ops, "PackPlan", _FakePlan)
    _FakePlan.runs = 0
    mlp = nerf_sos_amd.NeRFMLP()
    mlp.packed_weights(); mlp.packed_weights()
    assert _FakePlan.runs == 2                      # trainable: every call
    for p in mlp.parameters():
        p.requires_grad_(False)
    mlp.packed_weights(); mlp.packed_weights(); mlp.packed_weights()
    assert _FakePlan.runs == 3                      # frozen: once
    with torch.no_grad():
        next(mlp.parameters()).mul_(2.0)            # in-place op: _version changes
    mlp.packed_weights()
    assert _FakePlan.runs == 4
    next(mlp.parameters()).data.mul_(2.0)           # .data edit: invisible -> documented invalidate_packed()
    mlp.packed_weights()
    assert _FakePlan.runs == 4
    mlp.invalidate_packed()
    mlp.packed_weights()
    assert _FakePlan.runs == 5
    clone = copy.deepcopy(mlp)                      # the plan (raw pointers) and the streams do not travel
    assert clone._plan is None and clone._packed == {}
    clone.packed_weights()
    assert _FakePlan.runs == 6
    # the shipped recipe (only semantic_linear.* trainable), 16-bit streams: a full pack first, then the heads' chunks only --
    # until a frozen parameter changes
    _FakePlan.runs = _FakePlan.heads = 0
    sem = nerf_sos_amd.NeRFMLP(use_semantics=True, sem_with_coord=True)
    for n, p in sem.mlp.named_parameters():
        p.requires_grad_("semantic_linear" in n)
    sem.packed_weights("bf16"); sem.packed_weights("bf16"); sem.packed_weights("bf16")
    assert (_FakePlan.runs, _FakePlan.heads) == (3, 2)
    with torch.no_grad():
        sem.mlp.pts_linears[0].weight.mul_(2.0)
    sem.packed_weights("bf16"); sem.packed_weights("bf16")
    assert (_FakePlan.runs, _FakePlan.heads) == (5, 3)
    sem.packed_weights("fp32")                                   # the fp32 stream has no partial pack
    assert (_FakePlan.runs, _FakePlan.heads) == (6, 3)


def test_fused_adam_does_not_bump_versions():
    """The fact the packing policy rests on; if torch ever changes it the policy is merely conservative."""
    p = torch.nn.Parameter(torch.ones(4))
    try:
        opt = torch.optim.Adam([p], lr=0.1, fused=True)
    except RuntimeError:
        pytest.skip("no fused Adam on this device")
    p.grad = torch.ones(4)
    v = p._version
    opt.step()
    assert not torch.equal(p.detach(), torch.ones(4))
    assert p._version in (v, v + 1)


def test_cached_parameter_walk_equals_named_parameters():
    """nerf_net._named_params (slots resolved once, objects looked up per call) against nn.Module.named_parameters: same
    names, same order, same objects -- with both nets, coarse-only (nerf_fine is nerf: de-duplicated), after a deepcopy, and
    after parameters were replaced by load_state_dict(assign=True)."""
    import copy
    from nerf_sos_amd.nerf_net import _named_params
    for kw in (dict(N_importance=128, use_semantics=True, sem_with_coord=True), dict(N_importance=0), dict(N_importance=64, use_semantics=True)):
        net = nerf_sos_amd.NeRFNet(N_samples=64, **kw)
        for mod in (net, net.nerf.mlp, net.nerf_fine.mlp, copy.deepcopy(net)):
            for _ in range(2):           # second call: from the cache
                a, b = _named_params(mod), list(mod.named_parameters())
                assert [n for n, _ in a] == [n for n, _ in b] and all(x is y for (_, x), (_, y) in zip(a, b))
        sd = {k: v.clone() for k, v in net.state_dict().items()}
        net.load_state_dict(sd, assign=True)
        a, b = _named_params(net), list(net.named_parameters())
        assert all(x is y for (_, x), (_, y) in zip(a, b)) and len(a) == len(b)


def test_rays_that_require_grad_route_or_refuse():
    """The reference's autograd reaches the rays (o + d z, d / |d|, dists * |d|).  fp32: the render routes through the generic
    kernels' backward (tests/test_gpu_raygrad.py) -- on this CPU-only box it gets as far as the device check; a 16-bit precision has
    no such backward and refuses instead of returning outputs that silently carry no gradient.  (Host-side checks: no GPU needed.)"""
    net = nerf_sos_amd.NeRFNet(N_samples=64, N_importance=0)
    rays = torch.zeros(2, 4, 3, requires_grad=True)
    net.mlp_precision = "bf16"
    with pytest.raises(NotImplementedError, match="rays"):
        net(rays, (1.0, 2.0))
    net.mlp_precision = "fp32"
    with pytest.raises(RuntimeError, match="GPU tensor"):
        net(rays, (1.0, 2.0))
    with torch.no_grad(), pytest.raises(RuntimeError, match="GPU tensor"):
        net(rays, (1.0, 2.0))


def test_tile_major_sem_in_views_round_trip():
    """ops.sem_in_tiled / ops.sem_in_rows (the tile-major layout the 16-bit training kernel stores sem_in in, include/nerf_sos_hip.h
    NSOS_SEM_IN_TILED): channel 16 K + 8 kg + c of point 32 g + i sits at [g, K, 32 kg + i, c]; rows -> tiles -> rows is the
    identity for ragged point counts too (pure index arithmetic: checked on the CPU)."""
    from nerf_sos_amd import ops
    for P in (1, 31, 32, 33, 100):
        rows = torch.arange(P * 320, dtype=torch.float32).reshape(P, 320).to(torch.bfloat16)
        tiled = ops.sem_in_tiled(rows)
        assert tuple(tiled.shape) == ((P + 31) // 32, 20, 64, 8) and tiled.is_contiguous()
        assert torch.equal(ops.sem_in_rows(tiled, P), rows)
        p, ch = P - 1, 16 * 7 + 8 * 1 + 5
        assert tiled[p // 32, 7, 32 * 1 + p % 32, 5] == rows[p, ch]
        assert ops.sem_in_rows(rows, P) is rows                      # a row-major matrix passes through
