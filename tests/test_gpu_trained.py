"""GPU (-m gpu): parity and 16-bit quality on a TRAINED field (VERDICT r04 #1 / missing-1, weak-2, missing-6).

Fixture: tests/golden/trained_scene.ckpt (the shipped architecture trained on the procedural scene by this package on an MI355X)
rendered by the REAL reference -> tests/golden/trained.npz (make_goldens_trained.py).  What is held here:
  * fp32 exact and split-fp16: on the reference's own z_fine every key strictly within 1e-4, eval and train mode;
  * free-running: rays outside the 1e-4 band counted against the reference's own sensitivity N_self (stored in the fixture);
  * bf16 / fp16: PSNR, max-abs and label agreement against the REFERENCE's render, floors set from measurement;
  * fp16 range: the same function with one hidden layer's activations scaled up (ReLU layers are positively homogeneous, so
    W_k, b_k *= s and W_{k+1} /= s leaves the network's function unchanged): in range every precision stays correct; past
    65 504 the fp16 kernels' failure is REPORTED (the first render raises FloatingPointError: NeRFNet.check_numerics), never returned
    as silent numbers.
"""
import json
import os

import numpy as np
import pytest
import torch

import nerf_sos_amd
from helpers import close

pytestmark = pytest.mark.gpu
DEV = "cuda:0"
HERE = os.path.dirname(os.path.abspath(__file__))
CKPT = os.path.join(HERE, "golden", "trained_scene.ckpt")
KW = dict(use_semantics=True, sem_with_coord=True)
REPORT = {}


def T(a):
    return torch.as_tensor(np.ascontiguousarray(a)).to(DEV)


def N(t):
    return t.detach().float().cpu().numpy()


def _net(precision="fp32", sd=None):
    net = nerf_sos_amd.NeRFNet(N_samples=64, N_importance=128, perturb=1.0, raw_noise_std=1.0, **KW).to(DEV)
    nerf_sos_amd.io.load_checkpoint(CKPT, net) if sd is None else net.load_state_dict(sd)
    for p in net.parameters():
        p.requires_grad_(False)
    net.mlp_precision = precision
    return net.eval()


def _report(key, value):
    REPORT[key] = value
    os.makedirs("gpurun_out", exist_ok=True)
    with open("gpurun_out/trained_field_report.json", "w") as f:
        json.dump(REPORT, f, indent=1, sort_keys=True)


@pytest.mark.parametrize("precision", ["fp32", "fp16x3"])
@pytest.mark.parametrize("mode", ["eval", "train"])
def test_trained_field_on_reference_z_fine_strict(golden, monkeypatch, precision, mode):
    g = golden("trained")
    net = _net(precision)
    net.train(mode == "train")
    if mode == "train":
        from test_gpu_parity import _Draws
        dr = [torch.as_tensor(g[f"train_draw{i}"]) for i in range(4)]
        monkeypatch.setattr(torch, "rand", _Draws([dr[0], dr[2]]))
        monkeypatch.setattr(torch, "randn", _Draws([dr[1], dr[3]]))
    near, far = (float(v) for v in g["near_far"])
    with torch.no_grad():
        out = net(T(g["rays"]), (near, far), radii=None, z_fine_override=T(g[f"{mode}_z_fine"]))
    assert set(out) == {k[len(mode) + 1:] for k in g if k.startswith(mode + "_") and "draw" not in k and not k.endswith("z_fine")}
    worst = {}
    for k, got in out.items():
        if k == "z_std":                      # the sampler's own output, not pinned by z_fine_override: test_trained_field_free_running
            continue
        want = g[f"{mode}_{k}"]
        worst[k] = float(np.max(np.abs(N(got).reshape(want.shape).astype(np.float64) - want) / (1 + np.abs(want))))
        # Every key at 1e-4 -- except the split-fp16 kernel's per-sample `raw`: its operands carry 22 of fp32's 24 mantissa bits, and
        # a trained field's sigma head sums 256 products of size ~10 that cancel to ~1 (|sigma| reaches 220 here): measured 1.7e-4 on 2
        # of 294 912 elements (the reference's own fp32-vs-fp64 distance on raw0 is 2.6e-5).  Every rendered map is inside 1e-4.
        tol = 2.5e-4 if (precision == "fp16x3" and k in ("raw", "raw0")) else 1e-4
        close(N(got).reshape(want.shape), want, atol=tol, rtol=tol, what=f"trained field, {precision} {mode} {k} on the reference's z_fine")
    _report(f"pinned_{precision}_{mode}_max_err_over_1_plus_abs", worst)


@pytest.mark.parametrize("precision", ["fp32", "fp16x3"])
def test_trained_field_free_running(golden, precision):
    """Nothing pinned: the coarse pass strictly within 1e-4, and the rays whose fine maps leave the band counted against the
    reference's own sensitivity on this field (N_self, the fixture's `n_self`: 0 of 256 -- a trained field's coarse weights are not
    the flat, tiny ones of a random net, so the hierarchical sampler is well conditioned)."""
    g = golden("trained")
    net = _net(precision)
    near, far = (float(v) for v in g["near_far"])
    with torch.no_grad():
        out = net(T(g["rays"]), (near, far), radii=None)
    for k in ("rgb0", "depth0", "acc0", "disp0", "semantics0", "weights0", "raw0"):
        close(N(out[k]), g[f"eval_{k}"], atol=1e-4, rtol=1e-4, what=f"trained field, {precision} free-running {k}")
    R = g["rays"].shape[1]
    bad = np.zeros(R, bool)
    per_key = {}
    for k in ("rgb", "depth", "acc", "semantics", "weights", "z_std", "disp"):
        want = g[f"eval_{k}"].reshape(R, -1)
        b = (np.abs(N(out[k]).reshape(R, -1).astype(np.float64) - want) > 1e-4 * (1 + np.abs(want))).any(-1)
        per_key[k] = int(b.sum())
        bad |= b
    n_self, n_self_z = int(g["n_self"][0]), int(g["n_self_z_std_eval"][0])
    _report(f"free_running_{precision}", {"rays_outside_any_fine_map": int(bad.sum()), "per_key": per_key, "n_self_maps": n_self,
                                          "n_self_z_std": n_self_z, "rays": R})
    # image maps: the reference moves 0 of these 256 rays against itself (n_self) -- and so does the HIP path
    maps = max(per_key[k] for k in ("rgb", "depth", "acc", "semantics", "weights", "disp"))
    assert maps <= 2 * n_self + 1, (per_key, n_self)
    # z_std (std of the 128 importance samples): empty coarse bins sit right at the sampler's `denom < 1e-5 -> 1` switch
    # (models/sampler.py:117-118), where a last-ulp cdf difference moves a sample across its bin -- in empty space, invisible in every
    # map.  The reference itself moves n_self_z (7) rays by up to 1.7e-2 when its coarse net is evaluated in fp64; two independent
    # roundings (ours and the reference's) flip the UNION of two such sets: expectation 2 n_self_z, asserted with a 3-sigma Poisson
    # allowance.  Measured 13 (fp32) and 11 (fp16x3) against 14.
    assert per_key["z_std"] <= 2 * n_self_z + 3 * (2 * n_self_z) ** 0.5 + 1, (per_key, n_self_z)
    assert float(np.abs(N(out["z_std"]).reshape(-1) - g["eval_z_std"].reshape(-1)).max()) <= 3 * float(g["max_self_z_std_eval"][0]) + 1e-3


# measured (profiles/r05/b_trained_field_report.json): bf16 68.7 dB / max-abs 3.4e-3, fp16 77.4 dB / 2.1e-3, labels identical on all 256 rays
FLOORS = {"bf16": dict(psnr=62.0, max_abs=0.01, labels=0.995), "fp16": dict(psnr=70.0, max_abs=0.006, labels=0.995)}


@pytest.mark.parametrize("precision", ["bf16", "fp16"])
def test_trained_field_16_bit_quality_vs_reference(golden, precision):
    """The 16-bit MFMA paths against the REFERENCE's fp32 render of the same trained field, free-running (BASELINE's metric:
    'PSNR vs ref').  Floors from measurement (gpurun_out/trained_field_report.json, copied to profiles/r05/)."""
    g = golden("trained")
    net = _net(precision)
    near, far = (float(v) for v in g["near_far"])
    with torch.no_grad():
        out = net(T(g["rays"]), (near, far), radii=None)
    rec = {}
    for k in ("rgb", "rgb0"):
        d = N(out[k]).astype(np.float64) - g[f"eval_{k}"]
        rec[f"psnr_{k}_db"] = float(-10 * np.log10(np.mean(d ** 2) + 1e-30))
        rec[f"max_abs_{k}"] = float(np.abs(d).max())
    lab, want = N(out["semantics"]).argmax(-1), g["eval_semantics"].argmax(-1)
    margin = np.abs(g["eval_semantics"][:, 0] - g["eval_semantics"][:, 1])
    rec["label_agreement"] = float((lab == want).mean())
    rec["label_agreement_where_margin_over_0.05"] = float((lab == want)[margin > 0.05].mean())
    rec["max_abs_semantics"] = float(np.abs(N(out["semantics"]) - g["eval_semantics"]).max())
    rec["max_rel_depth"] = float(np.max(np.abs(N(out["depth"]) - g["eval_depth"]) / np.abs(g["eval_depth"])))
    rec["max_abs_acc"] = float(np.abs(N(out["acc"]) - g["eval_acc"]).max())
    rec["psnr_vs_analytic_gt_db"] = float(-10 * np.log10(np.mean((N(out["rgb"]) - g["gt_rgb"]) ** 2)))
    rec["reference_psnr_vs_analytic_gt_db"] = float(-10 * np.log10(np.mean((g["eval_rgb"] - g["gt_rgb"]) ** 2)))
    _report(f"quality_{precision}", rec)
    f = FLOORS[precision]
    assert np.isfinite(N(out["rgb"])).all()
    assert rec["psnr_rgb_db"] >= f["psnr"] and rec["max_abs_rgb"] <= f["max_abs"], rec
    assert rec["label_agreement_where_margin_over_0.05"] >= f["labels"], rec
    assert abs(rec["psnr_vs_analytic_gt_db"] - rec["reference_psnr_vs_analytic_gt_db"]) < 1.0, rec


# ------------------------------------------------------------------------------ the 16-bit TAIL (round 6)
# 256 rays cannot show a one-in-a-thousand ray.  trained_4k.npz: the REAL reference's eval render of the 4096 rays bench.py's
# `parity.trained_field` uses; trained_img64k.npz: of 65 536 pixels of the full 1008x756 image of held-out pose 0 (C5's workload).
# What the tail is (scripts/diag/lp_outliers.py, profiles/r06/a_lp_outliers.json): silhouette rays whose 128 importance samples land
# elsewhere when the COARSE weights carry 16-bit error; the fine network's own 16-bit error on given positions is 9 rays of 762 048
# over 0.01.  `coarse_precision = "fp16x3"` removes it.  Bars: measured values (profiles/r06/b_trained_field_report.json) with headroom;
# they are COUNTS and percentiles, not only a PSNR.
TAIL_BARS = {
    # (precision, coarse_precision): (4096 rays, 65 536 pixels): PSNR floor, max count of rays with |d rgb| > 0.01 / > 0.05, p99.9 ceiling.
    # measured (profiles/r06/b_trained_field_report.json): fp16 64.35 / 57.34 dB, 3 / 60 rays over 0.01, 0 / 11 over 0.05, max 0.039 / 0.24;
    # bf16 56.71 / 54.74 dB, 7 / 116, 1 / 17, max 0.10 / 0.24; fp16 with a split-fp16 coarse pass 85.58 / 84.00 dB, 0 / 0 rays, max 0.0022 /
    # 0.0077; bf16 with it 69.66 / 68.96 dB, 1 / 17 rays over 0.01, none over 0.05, max 0.010 / 0.029
    ("fp16", None): dict(psnr=(61.0, 54.0), n01=(10, 110), n05=(3, 25), p999=(0.02, 0.02)),
    ("bf16", None): dict(psnr=(53.5, 52.0), n01=(20, 200), n05=(5, 40), p999=(0.035, 0.03)),
    ("fp16", "fp16x3"): dict(psnr=(80.0, 80.0), n01=(0, 2), n05=(0, 0), p999=(0.002, 0.002)),
    ("bf16", "fp16x3"): dict(psnr=(66.0, 66.0), n01=(5, 40), n05=(1, 3), p999=(0.01, 0.01)),
}


def _tail_case(g, rays, precision, coarse, which):
    from nerf_sos_amd import quality
    net = _net(precision)
    net.coarse_precision = coarse
    net.chunk = 65536
    near, far = (float(v) for v in g["near_far"])
    with torch.no_grad():
        out = net(rays, (near, far), radii=None, retraw=False)
    assert torch.isfinite(out["rgb"]).all() and torch.isfinite(out["depth"]).all()
    st = quality.tail_stats(out["rgb"], T(g["eval_rgb"]), out["depth"], T(g["eval_depth"]),
                            out["semantics"].argmax(-1), T(g["eval_semantics"]).argmax(-1))
    st["psnr_vs_analytic_gt_db"] = float(-10 * np.log10(np.mean((N(out["rgb"]) - g["gt_rgb"]) ** 2)))
    st["reference_psnr_vs_analytic_gt_db"] = float(-10 * np.log10(np.mean((g["eval_rgb"] - g["gt_rgb"]) ** 2)))
    _report(f"tail_{which}_{precision}_coarse_{coarse or 'same'}", st)
    bar, i = TAIL_BARS[(precision, coarse)], 0 if which == "4k" else 1
    a = st["abs_rgb"]
    assert st["psnr_db"] >= bar["psnr"][i], st
    assert a["n_gt_0.01"] <= bar["n01"][i] and a["n_gt_0.05"] <= bar["n05"][i] and a["p99.9"] <= bar["p999"][i], st
    assert st["label_agreement"] >= 0.998, st
    assert abs(st["psnr_vs_analytic_gt_db"] - st["reference_psnr_vs_analytic_gt_db"]) < 0.5, st
    if coarse == "fp16x3" and precision == "fp16":
        # the fix's acceptance bar: max |d rgb| <= 0.02 on >= 99.99 % of the rays
        assert st["share_of_rays_within_0.02"] >= 0.9999, st
    return st


@pytest.mark.parametrize("precision,coarse", list(TAIL_BARS))
def test_trained_field_16_bit_tail_vs_reference_4096_rays(golden, precision, coarse):
    g = golden("trained_4k")
    _tail_case(g, T(g["rays"]), precision, coarse, "4k")


@pytest.mark.parametrize("precision,coarse", list(TAIL_BARS))
def test_trained_field_16_bit_tail_vs_reference_image_65536_pixels(golden, precision, coarse):
    """65 536 pixels of the full-size image (C5's workload on the trained field) against the REAL reference's render of them.  The rays
    come from K0 on the device and are asserted bit-identical (sha256) to the reference's get_persp_rays (utils/ray.py:12-22)."""
    import hashlib
    from nerf_sos_amd import ops, synthetic as syn
    g = golden("trained_img64k")
    scene = syn.ProceduralScene()
    H, W, focal = int(g["image_hwf"][0]), int(g["image_hwf"][1]), float(g["image_hwf"][2])
    full = ops.generate_rays(H, W, syn.intrinsics(H, W, focal), scene.poses[int(g["pose_index"][0]), :3, :4], DEV).reshape(2, -1, 3)
    rays = full[:, T(g["pixel_index"]).long()].contiguous()
    assert hashlib.sha256(N(rays).tobytes()).digest() == bytes(g["rays_sha256"]), "K0's rays differ from the reference's get_persp_rays"
    _tail_case(g, rays, precision, coarse, "img64k")


def test_trained_field_fp32_paths_vs_reference_4096_rays(golden):
    """The exact and the split-fp16 paths on the same 4096 rays, free-running, against the reference: image maps inside 1e-4 up to the
    reference's own sensitivity (n_self = 0 rays on this set), z_std counted against ITS yardstick (n_self_z_std: the reference moves
    that many rays against its own fp64-coarse variant) -- VERDICT r05 weak-3."""
    g = golden("trained_4k")
    near, far = (float(v) for v in g["near_far"])
    R = g["rays"].shape[1]
    n_self, n_self_z = int(g["n_self"][0]), int(g["n_self_z_std"][0])
    for precision in ("fp32", "fp16x3"):
        net = _net(precision)
        with torch.no_grad():
            out = net(T(g["rays"]), (near, far), radii=None, retraw=False)
        per_key = {}
        for k in ("rgb", "depth", "acc", "disp", "semantics", "z_std", "rgb0", "depth0", "acc0", "disp0", "semantics0"):
            want = g[f"eval_{k}"].reshape(R, -1)
            per_key[k] = int((np.abs(N(out[k]).reshape(R, -1).astype(np.float64) - want) > 1e-4 * (1 + np.abs(want))).any(-1).sum())
        _report(f"free_running_4k_{precision}", {"per_key": per_key, "n_self_maps": n_self, "n_self_z_std": n_self_z, "rays": R})
        assert all(per_key[k] == 0 for k in ("rgb0", "depth0", "acc0", "disp0", "semantics0")), per_key
        assert max(per_key[k] for k in ("rgb", "depth", "acc", "disp", "semantics")) <= 2 * n_self + 2, (per_key, n_self)
        assert per_key["z_std"] <= 2 * n_self_z + 3 * (2 * n_self_z) ** 0.5 + 1, (per_key, n_self_z)


def _rescaled(sd, s, layer=3):
    """W_k, b_k *= s; the h-columns of W_{k+1} /= s: the same function, layer k's activations s times larger."""
    sd = {k: v.clone() for k, v in sd.items()}
    for net in ("nerf", "nerf_fine"):
        sd[f"{net}.mlp.pts_linears.{layer}.weight"] *= s
        sd[f"{net}.mlp.pts_linears.{layer}.bias"] *= s
        sd[f"{net}.mlp.pts_linears.{layer + 1}.weight"] /= s
    return sd


@pytest.mark.parametrize("precision", ["fp32", "fp16x3", "bf16", "fp16"])
def test_fp16_range_in_range_rescale_is_harmless(golden, precision):
    """Layer 3's activations x 64 (|h| up to a few thousand: inside fp16's 65 504): every precision renders the same image as
    before the rescale, to its own accuracy -- bf16 by format, fp16 / fp16x3 because nothing left the range."""
    g = golden("trained")
    sd = torch.load(CKPT, map_location="cpu")["model"]
    near, far = (float(v) for v in g["near_far"])
    with torch.no_grad():
        base = _net(precision)(T(g["rays"]), (near, far), radii=None)
        net = _net(precision, _rescaled(sd, 64.0))
        out = net(T(g["rays"]), (near, far), radii=None)
        chk = net.check_numerics((T(g["rays"][0]), T(g["rays"][1])), (near, far))
    assert chk["finite"] and chk["psnr_vs_fp32_db"] > (50.0 if precision in ("fp32", "fp16x3") else 40.0), chk
    assert all(torch.isfinite(out[k]).all() for k in ("rgb", "depth", "semantics", "raw"))
    d = float((out["rgb"] - base["rgb"]).abs().max())
    dref = float(np.abs(N(out["rgb"]) - g["eval_rgb"]).max())
    _report(f"range_x64_{precision}", {"max_abs_rgb_vs_unscaled": d, "max_abs_rgb_vs_reference": dref})
    assert d <= {"fp32": 1e-4, "fp16x3": 1e-4, "bf16": 0.2, "fp16": 0.1}[precision], d


@pytest.mark.parametrize("precision", ["fp16", "fp16x3"])
def test_fp16_overflow_is_reported_not_returned_silently(golden, precision):
    """Layer 3's activations x 65 536: past fp16's largest finite value.  fp32 and bf16 keep rendering the same image; the fp16 kernels
    cannot -- and say so: the first eval render of a frozen net under "fp16" / "fp16x3" (and the first after its weights change)
    re-renders up to 1024 of its own rays with the exact fp32 kernels and raises FloatingPointError naming the precision when the
    images disagree (NeRFNet.check_numerics).  Never a silently wrong image: a ReLU can turn the NaNs of inf - inf back into finite
    numbers, so the outputs alone would not show it."""
    g = golden("trained")
    sd = torch.load(CKPT, map_location="cpu")["model"]
    near, far = (float(v) for v in g["near_far"])
    big = _rescaled(sd, 65536.0)
    with torch.no_grad():
        for ok in ("fp32", "bf16"):
            out = _net(ok, big)(T(g["rays"]), (near, far), radii=None)
            assert torch.isfinite(out["rgb"]).all()
            assert float(np.abs(N(out["rgb"]) - g["eval_rgb"]).max()) <= (1e-3 if ok == "fp32" else 0.2)
        net = _net(precision, big)
        with pytest.raises(FloatingPointError, match=precision):           # the first render of these weights checks itself
            net(T(g["rays"]), (near, far), radii=None)
        with pytest.raises(FloatingPointError):                            # ... and keeps refusing: a failed check is not remembered as passed
            net(T(g["rays"]), (near, far), radii=None)
        net.validate_precision = False                                     # opting out renders whatever the format gives ...
        out = net(T(g["rays"]), (near, far), radii=None)
        bad = float(np.abs(np.nan_to_num(N(out["rgb"]), nan=9.0, posinf=9.0, neginf=9.0) - g["eval_rgb"]).max())
        assert bad > 0.2, "the rescaled field was expected to break this precision"
        with pytest.raises(FloatingPointError):                            # ... and the explicit check still says so
            net.check_numerics((T(g["rays"][0]), T(g["rays"][1])), (near, far))
        # round 6: the guard also covers a guarded precision on the COARSE pass alone (bf16 fine pass: fp32's exponent range)
        mixed = _net("bf16", big)
        mixed.coarse_precision = precision
        with pytest.raises(FloatingPointError, match="coarse_precision"):
            mixed(T(g["rays"]), (near, far), radii=None)
        mixed.coarse_precision = None                                      # bf16 everywhere: nothing to guard, renders
        assert torch.isfinite(mixed(T(g["rays"]), (near, far), radii=None)["rgb"]).all()


# ---------------------------------------------------------------------------------- gradients on the trained field (training side)
def _grad_errors(net, G, names=None):
    worst = {}
    for n_, p_ in net.named_parameters():
        if names is not None and not any(s in n_ for s in names):
            continue
        got = p_.grad
        assert got is not None and torch.isfinite(got).all(), n_
        scale = float(G[f"gradmax_{n_}"][0]) + 1e-30
        if f"grad_{n_}" in G:
            pairs = [(got, G[f"grad_{n_}"])]
        else:
            pairs = [(got[::max(1, got.shape[0] // 24)], G[f"gradrows_{n_}"]), (got[:, ::max(1, got.shape[1] // 24)], G[f"gradcols_{n_}"])]
        worst[n_] = max(float((a.detach().cpu() - torch.from_numpy(b)).abs().max()) / scale for a, b in pairs)
    return worst


def _trained_loss(net, G, near, far):
    ret = net(T(G["rays"]), (near, far), radii=None, z_fine_override=T(G["z_fine"]))
    gt = T(G["gt"])
    return ((ret["rgb"] - gt) ** 2).mean() + ((ret["rgb0"] - gt) ** 2).mean() + (ret["semantics"] * T(G["G_semantics"])).sum() + \
        (ret["semantics0"] * T(G["G_semantics0"])).sum()


@pytest.mark.parametrize("precision", ["fp32", "fp16x3"])
def test_trained_field_full_backward_vs_reference_autograd(golden, precision):
    """Every parameter trainable (configs/*_full.txt) on the TRAINED weights: the reference's own loss shape (img2mse on rgb and rgb0 +
    a per-logit gradient on both semantic maps), fine positions pinned to the reference's -- every parameter's gradient within 1e-4
    of its scale against the real reference's autograd (tests/golden/make_goldens_trained_grads.py), as on the random-init goldens."""
    g, G = golden("trained"), golden("trained_grads")
    near, far = (float(v) for v in g["near_far"])
    net = _net(precision)
    for p in net.parameters():
        p.requires_grad_(True)
    loss = _trained_loss(net, G, near, far)
    assert abs(float(loss.detach()) - float(G["loss"][0])) <= 1e-4 * (1 + abs(float(G["loss"][0])))
    loss.backward()
    worst = _grad_errors(net, G)
    _report(f"trained_full_backward_{precision}_max_err_of_scale", {"max": max(worst.values()), "worst_parameter": max(worst, key=worst.get)})
    bad = {k: v for k, v in worst.items() if v > 1e-4}
    assert len(worst) == 56 and not bad, bad


# measured (profiles/r05/b_trained_field_report.json): fp32 4.1e-7, fp16x3 4.7e-7, fp16 3.7e-3 (cosine 0.999998), bf16 0.11 (cosine 0.9990:
# an 8-bit mantissa on the head's inputs, summed over 12 288 points with the upstream gradient's random signs cancelling)
HEAD_BARS = {"fp32": 1e-4, "fp16x3": 1e-4, "bf16": 0.2, "fp16": 1e-2}
HEAD_COS = {"fp32": 0.9999999, "fp16x3": 0.9999999, "bf16": 0.998, "fp16": 0.99999}


@pytest.mark.parametrize("precision", ["fp32", "fp16x3", "bf16", "fp16"])
def test_trained_field_head_recipe_gradients_vs_reference_autograd(golden, precision):
    """--fix_backbone (run_nerf.py:307-318) on the trained weights: only semantic_linear.* trains (the SAVE kernels + the head's
    weight-gradient kernel: the C3 / C4 step).  fp32 / split fp16 at 1e-4 of scale; the 16-bit MFMA paths against the reference's
    fp32 autograd at their formats' accuracy (bars from measurement) and with the gradient's DIRECTION pinned (cosine per parameter)."""
    g, G = golden("trained"), golden("trained_grads")
    near, far = (float(v) for v in g["near_far"])
    net = _net(precision)
    for n_, p_ in net.named_parameters():
        p_.requires_grad_("semantic_linear" in n_)
    loss = _trained_loss(net, G, near, far)
    loss.backward()
    worst = _grad_errors(net, G, names=("semantic_linear",))
    assert len(worst) == 8
    cos = {}
    for n_, p_ in net.named_parameters():
        if "semantic_linear" in n_ and f"grad_{n_}" in G:
            a, b = p_.grad.detach().cpu().double().reshape(-1), torch.from_numpy(G[f"grad_{n_}"]).double().reshape(-1)
            cos[n_] = float((a @ b) / (a.norm() * b.norm() + 1e-300))
    _report(f"trained_head_gradients_{precision}", {"max_err_of_scale": max(worst.values()), "min_cosine": min(cos.values())})
    assert max(worst.values()) <= HEAD_BARS[precision], worst
    assert min(cos.values()) > HEAD_COS[precision], cos
    assert all(p_.grad is None for n_, p_ in net.named_parameters() if "semantic_linear" not in n_)
