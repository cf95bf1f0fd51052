#!/usr/bin/env python3
"""Where does the 16-bit tail on the TRAINED field come from?  (VERDICT r05 #1d)

Full 1008x756 image of held-out pose 0 of tests/golden/trained_scene.ckpt (C5's workload), exact fp32 kernels vs fp16 / bf16:
  1. the error distribution per ray (|d rgb|, |d depth| / depth): percentiles, counts over 0.01 / 0.05, where on the image;
  2. attribution BY SUBSTITUTION on the worst rays, with the real kernels: fp32 coarse + 16-bit fine, 16-bit coarse + fp32 fine,
     16-bit everything but sigma taken from the fp32 fine pass (same sample positions), fp32 everything but sigma from the 16-bit pass;
  3. attribution by OPERAND CLASS with an fp64 torch emulation on the device (scripts/diag only; never the product): which rounding
     -- encodings, weights, hidden activations, the sigma head's operands -- produces the sigma error at the samples that matter;
  4. conditioning: the exact fp32 render with sigma perturbed by a relative 2^-11 (one fp16 ulp) / 2^-8 (one bf16 ulp) of noise:
     do the same rays move?

    python scripts/diag/lp_outliers.py [--prec fp16] [--worst 32] [--out gpurun_out/lp_outliers.json]
"""
import argparse
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402
import torch  # noqa: E402
import nerf_sos_amd  # noqa: E402
from nerf_sos_amd import io as nio, ops, synthetic as syn  # noqa: E402

CKPT = os.path.join(ROOT, "tests", "golden", "trained_scene.ckpt")
NC, NI = 64, 128


def load(dev):
    scene = syn.ProceduralScene()
    net = nerf_sos_amd.NeRFNet(N_samples=NC, N_importance=NI, use_semantics=True, sem_with_coord=True, ray_chunk=65536).to(dev).eval()
    nio.load_checkpoint(CKPT, net)
    for p in net.parameters():
        p.requires_grad_(False)
    net.validate_precision = False
    return scene, net


def query(net_mlp, prec, o, d, v, z):
    if prec == "fp32":
        return ops.mlp_forward_rays(net_mlp.packed_weights(), net_mlp.sem_mode, o, d, v, z)
    return ops.mlp_forward_rays_lp(net_mlp.packed_weights(prec), net_mlp.sem_mode, prec, o, d, v, z)


def staged(net, o, d, near, far, pc, pf, z_fine=None, sigma_fine=None, keep=False):
    """One render with the coarse net in precision `pc` and the fine net in `pf`; `z_fine` pins the fine sample positions,
    `sigma_fine` replaces the fine pass's sigma channel before compositing."""
    z, v = ops.ray_setup(d, near, far, NC, None)
    raw0 = query(net.nerf, pc, o, d, v, z)
    ret0, zf, _, z_std = ops.composite_importance(raw0, z, d, NI)
    if z_fine is not None:
        zf = z_fine
    raw = query(net.nerf_fine, pf, o, d, v, zf)
    if sigma_fine is not None:
        raw = raw.clone()
        raw[..., 3] = sigma_fine
    ret = ops.composite(raw, zf, d)
    out = dict(rgb=ret["rgb"], depth=ret["depth"][:, 0], acc=ret["acc"][:, 0], sem=ret["semantics"], z_fine=zf, rgb0=ret0["rgb"])
    if keep:
        out.update(raw=raw, raw0=raw0, weights=ret["weights"], weights0=ret0["weights"], z=z)
    return out


def chunked(fn, n, chunk=65536):
    outs = {}
    for i in range(0, n, chunk):
        for k, t in fn(slice(i, min(i + chunk, n))).items():
            outs.setdefault(k, []).append(t)
    return {k: torch.cat(v) for k, v in outs.items()}


def stats(a, b):
    """a, b: dicts with rgb [R,3], depth [R]."""
    e = (a["rgb"] - b["rgb"]).abs().amax(-1).double()
    rd = ((a["depth"] - b["depth"]).abs() / b["depth"].abs()).double()
    q = lambda t, p: float(torch.quantile(t[torch.randperm(t.numel(), device=t.device)[:4_000_000]] if t.numel() > 4_000_000 else t, p))  # noqa: E731
    mse = float(((a["rgb"].double() - b["rgb"].double()) ** 2).mean())
    return {"rays": int(e.numel()), "psnr_db": round(-10 * np.log10(max(mse, 1e-30)), 2),
            "rgb": {"p50": q(e, .5), "p99": q(e, .99), "p99.9": q(e, .999), "p99.99": q(e, .9999), "max": float(e.max()),
                    "n_gt_0.01": int((e > 0.01).sum()), "n_gt_0.02": int((e > 0.02).sum()), "n_gt_0.05": int((e > 0.05).sum())},
            "rel_depth": {"p50": q(rd, .5), "p99": q(rd, .99), "p99.9": q(rd, .999), "max": float(rd.max()),
                          "n_gt_0.01": int((rd > 0.01).sum()), "n_gt_0.1": int((rd > 0.1).sum())}}


# ------------------------------------------------------------------ fp64 emulation with selectable roundings (device, torch)
def posenc(x, L):
    f = 2.0 ** torch.arange(L, device=x.device, dtype=torch.float32)
    xf = x[..., None, :] * f[:, None]                                 # [..., L, 3]   (fp32, as the kernels and the reference)
    e = torch.stack([torch.sin(xf), torch.cos(xf)], -2).reshape(x.shape[:-1] + (-1,))
    return torch.cat([x, e], -1)


def emulate(sd, prefix, pts, dirs, dtype, which):
    """raw [P,6] (fp64 accumulation).  `which` = set of operand classes rounded to `dtype`:
    enc (both encodings), w (trunk / feature / view weights), act (hidden activations), bias0 (the rounded biases of lp16), heads
    (operands of alpha / rgb / semantic output heads: 16-bit weights and 16-bit inputs), feat."""
    r = lambda t: t.to(dtype).double()  # noqa: E731
    Q = lambda name, t: r(t) if name in which else t.double()  # noqa: E731
    W = lambda n: sd[f"{prefix}.mlp.{n}.weight"]  # noqa: E731
    B = lambda n: sd[f"{prefix}.mlp.{n}.bias"]  # noqa: E731
    ex32, ed32 = posenc(pts, 10), posenc(dirs, 4)
    ex, ed = Q("enc", ex32), Q("enc", ed32)
    # finer classes of the position encoding: the raw coordinates, octaves 0-4, octaves 5-9 (columns 3 + 6 k ... of posenc)
    for name, cols in (("enc_xyz", slice(0, 3)), ("enc_oct_lo", slice(3, 33)), ("enc_oct_hi", slice(33, 63))):
        if name in which:
            ex = ex.clone()
            ex[:, cols] = r(ex32[:, cols])
    if "enc_dir" in which:
        ed = r(ed32)
    h = Q("act", torch.relu(ex @ Q("w", W("pts_linears.0")).T + Q("bias0", B("pts_linears.0"))).float())
    for i in range(1, 8):
        w = Q("w", W(f"pts_linears.{i}"))
        if i == 5:
            z = h @ w[:, 63:].T + ex @ w[:, :63].T + Q("bias0", B(f"pts_linears.{i}"))
        else:
            z = h @ w.T + B(f"pts_linears.{i}").double()
        hf = torch.relu(z).float()
        h = Q("act", hf)
    h_head = r(hf) if ("heads" in which or "act" in which) else hf.double()
    sigma = h_head @ Q("heads", W("alpha_linear")).T + B("alpha_linear").double()
    w = Q("w", W("semantic_linear.0"))
    zs = h @ w[:, :256].T + ex @ w[:, 256:].T + Q("bias0", B("semantic_linear.0"))
    sem = Q("heads", torch.relu(zs).float()) @ Q("heads", W("semantic_linear.2")).T + B("semantic_linear.2").double()
    feat = Q("feat", (h @ Q("w", W("feature_linear")).T + B("feature_linear").double()).float())
    w = Q("w", W("views_linears.0"))
    zv = feat @ w[:, :256].T + ed @ w[:, 256:].T + Q("bias0", B("views_linears.0"))
    rgb = Q("heads", torch.relu(zv).float()) @ Q("heads", W("rgb_linear")).T + B("rgb_linear").double()
    return torch.cat([rgb, sigma, sem], -1).float()


ALL = ("enc", "w", "act", "bias0", "heads", "feat")


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--prec", default="fp16,bf16")
    ap.add_argument("--worst", type=int, default=32)
    ap.add_argument("--emul-rays", type=int, default=2048)
    ap.add_argument("--coarse-classes", type=int, default=1)
    ap.add_argument("--out", default=os.path.join(ROOT, "gpurun_out", "lp_outliers.json"))
    args = ap.parse_args()
    dev = torch.device("cuda:0")
    scene, net = load(dev)
    i = scene.i_test[0]
    H, W = syn.H, syn.W
    rays = ops.generate_rays(H, W, syn.intrinsics(H, W, scene.focal * W / scene.w), scene.poses[i, :3, :4], dev).reshape(2, -1, 3)
    o, d = rays[0].contiguous(), rays[1].contiguous()
    R = o.shape[0]
    near, far = torch.full((R,), scene.NEAR, device=dev), torch.full((R,), scene.FAR, device=dev)
    sd = {k: v.detach() for k, v in net.state_dict().items()}
    report = {"image": f"{W}x{H} held-out pose {i}", "rays": R}

    def run(pc, pf, idx=None, **kw):
        sel = (lambda s: s) if idx is None else (lambda s: idx[s])
        n = R if idx is None else idx.numel()

        def fn(s):
            j = sel(s)
            kw2 = {k: (v[s] if torch.is_tensor(v) else v) for k, v in kw.items()}
            return staged(net, o[j], d[j], near[j], far[j], pc, pf, **kw2)
        with torch.no_grad():
            return chunked(fn, n)

    exact = run("fp32", "fp32")
    gt_rgb, gt_lab, gt_t = scene.view(i, H, W)
    gt_t = torch.from_numpy(np.ascontiguousarray(gt_t.reshape(-1))).to(dev)
    for prec in args.prec.split(","):
        rep = report.setdefault(prec, {})
        lo = run(prec, prec)
        rep["free_running_vs_exact"] = stats(lo, exact)
        e = (lo["rgb"] - exact["rgb"]).abs().amax(-1)
        # --- where on the image: distance (pixels) of the outliers to the nearest depth discontinuity of the ANALYTIC scene
        tmap = gt_t.reshape(H, W)
        edge = torch.zeros_like(tmap, dtype=torch.bool)
        jump = 0.25
        edge[:, 1:] |= (tmap[:, 1:] - tmap[:, :-1]).abs() > jump
        edge[:, :-1] |= (tmap[:, 1:] - tmap[:, :-1]).abs() > jump
        edge[1:, :] |= (tmap[1:, :] - tmap[:-1, :]).abs() > jump
        edge[:-1, :] |= (tmap[1:, :] - tmap[:-1, :]).abs() > jump
        near_edge = torch.nn.functional.max_pool2d(edge[None, None].float(), 7, 1, 3)[0, 0].bool().reshape(-1)   # within 3 px
        bad = e > 0.01
        rep["outliers_gt_0.01_within_3px_of_an_analytic_depth_edge"] = [int((bad & near_edge).sum()), int(bad.sum())]
        rep["share_of_image_within_3px_of_an_edge"] = round(float(near_edge.float().mean()), 4)
        # also: silhouette in the exact render itself (acc-weighted depth gradient)
        dm = exact["depth"].reshape(H, W)
        g = torch.zeros_like(dm)
        g[:, 1:] = (dm[:, 1:] - dm[:, :-1]).abs()
        g[1:, :] = torch.maximum(g[1:, :], (dm[1:, :] - dm[:-1, :]).abs())
        rep["median_exact_depth_step_at_outliers_vs_image"] = [float(g.reshape(-1)[bad].median()) if bad.any() else None, float(g.median())]

        # --- substitution on ALL rays (cheap: a few renders)
        sub = {}
        a = run("fp32", prec)
        sub["fp32_coarse__lp_fine"] = stats(a, exact)
        b = run(prec, "fp32")
        sub["lp_coarse__fp32_fine"] = stats(b, exact)
        pinned = run(prec, prec, z_fine=exact["z_fine"])
        sub["lp_both__fine_positions_of_the_exact_run"] = stats(pinned, exact)
        rep["substitution_all_rays"] = sub

        # --- the worst rays in detail
        worst = torch.argsort(e, descending=True)[:args.worst]
        with torch.no_grad():
            ex_w = staged(net, o[worst], d[worst], near[worst], far[worst], "fp32", "fp32", keep=True)
            lo_w = staged(net, o[worst], d[worst], near[worst], far[worst], prec, prec, keep=True)
            # same positions (the exact run's), 16-bit fine net: the fine net's own error
            lp_fine = staged(net, o[worst], d[worst], near[worst], far[worst], "fp32", prec, keep=True)
            # 16-bit colour / semantics, exact sigma; and the reverse
            lp_rgb_exact_sigma = staged(net, o[worst], d[worst], near[worst], far[worst], "fp32", prec, sigma_fine=ex_w["raw"][..., 3])
            exact_rgb_lp_sigma = staged(net, o[worst], d[worst], near[worst], far[worst], "fp32", "fp32", sigma_fine=lp_fine["raw"][..., 3])
            lp_coarse_only = staged(net, o[worst], d[worst], near[worst], far[worst], prec, "fp32", keep=True)
        err = lambda t: (t["rgb"] - ex_w["rgb"]).abs().amax(-1)  # noqa: E731
        rows = []
        for k in range(worst.numel()):
            r = int(worst[k])
            w_ex = ex_w["weights"][k]
            j = int(w_ex.argmax())
            s32, s16 = ex_w["raw"][k, :, 3], lp_fine["raw"][k, :, 3]
            dz = torch.diff(ex_w["z_fine"][k])
            c32, c16 = ex_w["raw0"][k, :, 3], lo_w["raw0"][k, :, 3]
            rows.append({
                "ray": r, "pixel": [r % W, r // W], "near_analytic_edge": bool(near_edge[r]),
                "err_free_running": float(e[r]), "err_fp32coarse_lpfine": float(err(lp_fine)[k]), "err_lpcoarse_fp32fine": float(err(lp_coarse_only)[k]),
                "err_lp_rgb_with_exact_sigma": float(err(lp_rgb_exact_sigma)[k]), "err_exact_rgb_with_lp_sigma": float(err(exact_rgb_lp_sigma)[k]),
                "exact": {"rgb": [round(float(x), 4) for x in ex_w["rgb"][k]], "depth": float(ex_w["depth"][k]), "acc": float(ex_w["acc"][k]),
                          "max_weight": float(w_ex.max()), "argmax_sample": j, "n_weights_gt_0.01": int((w_ex > 0.01).sum())},
                "lp": {"rgb": [round(float(x), 4) for x in lo_w["rgb"][k]], "depth": float(lo_w["depth"][k]), "acc": float(lo_w["acc"][k])},
                "fine_sigma_same_positions": {"max_abs_diff": float((s32 - s16).abs().max()), "at_sample": int((s32 - s16).abs().argmax()),
                                              "sigma_range_fp32": [float(s32.min()), float(s32.max())],
                                              "max_abs_diff_of_relu_sigma_times_dist": float(((torch.relu(s32[:-1]) - torch.relu(s16[:-1])) * dz).abs().max()),
                                              "around_argmax_fp32": [round(float(x), 3) for x in s32[max(0, j - 3):j + 4]],
                                              "around_argmax_lp": [round(float(x), 3) for x in s16[max(0, j - 3):j + 4]]},
                "coarse_sigma": {"max_abs_diff": float((c32 - c16).abs().max()), "range_fp32": [float(c32.min()), float(c32.max())],
                                 "max_abs_weight0_diff": float((ex_w["weights0"][k] - lo_w["weights0"][k]).abs().max()),
                                 "weights0_top3_fp32": [round(float(x), 4) for x in torch.topk(ex_w["weights0"][k], 3).values],
                                 "weights0_top3_lp": [round(float(x), 4) for x in torch.topk(lo_w["weights0"][k], 3).values]},
            })
        rep["worst_rays"] = rows
        agg = lambda key: [round(float(np.median([r_[key] for r_ in rows])), 5), round(float(np.max([r_[key] for r_ in rows])), 5)]  # noqa: E731
        rep["worst_rays_median_max"] = {k: agg(k) for k in ("err_free_running", "err_fp32coarse_lpfine", "err_lpcoarse_fp32fine",
                                                            "err_lp_rgb_with_exact_sigma", "err_exact_rgb_with_lp_sigma")}

        # --- operand classes (fp64 emulation on the device) on the worst `emul_rays` rays, fine net at the exact positions
        n_em = min(args.emul_rays, R)
        sel = torch.argsort(e, descending=True)[:n_em]
        dt = torch.float16 if prec == "fp16" else torch.bfloat16
        with torch.no_grad():
            ex_s = staged(net, o[sel], d[sel], near[sel], far[sel], "fp32", "fp32", keep=True)
            kern = staged(net, o[sel], d[sel], near[sel], far[sel], "fp32", prec, keep=True)
            zf, vdir = ex_s["z_fine"], d[sel] / d[sel].norm(dim=-1, keepdim=True)
            pts = (o[sel][:, None, :] + d[sel][:, None, :] * zf[..., None]).reshape(-1, 3)
            dirs = vdir[:, None, :].expand(-1, zf.shape[1], -1).reshape(-1, 3)
            em = {}

            def emu(which):
                raw = torch.cat([emulate(sd, "nerf_fine", pts[s:s + 65536], dirs[s:s + 65536], dt, set(which)) for s in range(0, pts.shape[0], 65536)])
                raw = raw.reshape(n_em, zf.shape[1], 6)
                ret = ops.composite(raw.contiguous(), zf, d[sel])
                return raw, ret
            raw_none, ret_none = emu(())
            em["fp64_emulation_vs_exact_kernel_max_abs_rgb"] = float((ret_none["rgb"] - ex_s["rgb"]).abs().max())
            raw_all, ret_all = emu(ALL)
            em["all_classes_vs_kernel_max_abs_rgb"] = float((ret_all["rgb"] - kern["rgb"]).abs().max())
            em["all_classes_vs_kernel_max_abs_sigma"] = float((raw_all[..., 3] - kern["raw"][..., 3]).abs().max())
            table = {}
            for name, which in [("all", ALL)] + [(f"only_{c}", (c,)) for c in ALL] + [(f"all_but_{c}", tuple(x for x in ALL if x != c)) for c in ALL] + \
                               [("only_enc+w", ("enc", "w")), ("all_but_act_and_heads", ("enc", "w", "bias0", "feat"))]:
                raw_c, ret_c = emu(which)
                er = (ret_c["rgb"] - ret_none["rgb"]).abs().amax(-1)
                table[name] = {"max_abs_rgb": float(er.max()), "median_abs_rgb": float(er.median()), "n_gt_0.01": int((er > 0.01).sum()),
                               "max_abs_sigma": float((raw_c[..., 3] - raw_none[..., 3]).abs().max()),
                               "rms_sigma": float((raw_c[..., 3] - raw_none[..., 3]).pow(2).mean().sqrt())}
            em["classes_fine_net_at_exact_positions"] = table
            em["kernel_fine_at_exact_positions"] = {"max_abs_rgb": float((kern["rgb"] - ex_s["rgb"]).abs().amax(-1).max()),
                                                    "n_gt_0.01": int(((kern["rgb"] - ex_s["rgb"]).abs().amax(-1) > 0.01).sum()), "rays": n_em}
        rep["emulation"] = em

        # --- the COARSE net's operand classes over the whole image: emulated coarse pass -> the real importance kernel -> exact fine
        # pass; what each candidate fix (an operand class kept exact) would do to the free-running tail
        if args.coarse_classes:
            no_enc = tuple(c for c in ALL if c != "enc")
            sub_enc = ("enc_xyz", "enc_oct_lo", "enc_oct_hi", "enc_dir")
            cases = [("none (fp64 coarse net)", ()), ("all = the kernel's roundings", ALL),
                     ("all, raw xyz exact", no_enc + ("enc_oct_lo", "enc_oct_hi", "enc_dir")),
                     ("all, xyz + octaves 5-9 exact", no_enc + ("enc_oct_lo", "enc_dir")),
                     ("all, every encoding exact", no_enc), ("all, weights exact", tuple(c for c in ALL if c != "w")),
                     ("all, activations exact", tuple(c for c in ALL if c != "act")),
                     ("all, encodings + weights exact", tuple(c for c in ALL if c not in ("enc", "w"))),
                     ("all, encodings + activations exact", tuple(c for c in ALL if c not in ("enc", "act"))),
                     ("only raw xyz", ("enc_xyz",)), ("only octaves 0-4", ("enc_oct_lo",)), ("only octaves 5-9", ("enc_oct_hi",)),
                     ("only weights", ("w",)), ("only activations", ("act",))]
            dt = torch.float16 if prec == "fp16" else torch.bfloat16
            table = {}
            for name, which in cases:
                outs = []
                for s in range(0, R, 65536):
                    sl = slice(s, min(s + 65536, R))
                    with torch.no_grad():
                        z, v = ops.ray_setup(d[sl], near[sl], far[sl], NC, None)
                        pts = (o[sl][:, None, :] + d[sl][:, None, :] * z[..., None]).reshape(-1, 3)
                        dirs = v[:, None, :].expand(-1, NC, -1).reshape(-1, 3)
                        raw0 = torch.cat([emulate(sd, "nerf", pts[c:c + 524288], dirs[c:c + 524288], dt, set(which)) for c in range(0, pts.shape[0], 524288)])
                        raw0 = raw0.reshape(-1, NC, 6).contiguous()
                        ret0, zf, _, _ = ops.composite_importance(raw0, z, d[sl], NI)
                        raw = query(net.nerf_fine, "fp32", o[sl], d[sl], v, zf)
                        ret = ops.composite(raw, zf, d[sl])
                    outs.append(dict(rgb=ret["rgb"], depth=ret["depth"][:, 0]))
                got = {k: torch.cat([t[k] for t in outs]) for k in outs[0]}
                st = stats(got, exact)
                table[name] = {"psnr_db": st["psnr_db"], "n_gt_0.01": st["rgb"]["n_gt_0.01"], "n_gt_0.05": st["rgb"]["n_gt_0.05"], "max": st["rgb"]["max"],
                               "p99.99": st["rgb"]["p99.99"], "depth_n_gt_0.01": st["rel_depth"]["n_gt_0.01"]}
                print(prec, "coarse classes:", name, table[name], flush=True)
            rep["coarse_net_classes_whole_image_fine_pass_exact"] = table

    # --- conditioning of the exact render: sigma of BOTH passes perturbed by relative noise of one 16-bit ulp
    cond = {}
    g = torch.Generator(device=dev).manual_seed(1)
    for name, rel in (("2^-11 (fp16 ulp)", 2.0 ** -11), ("2^-8 (bf16 ulp)", 2.0 ** -8), ("2^-24 (fp32 ulp)", 2.0 ** -24)):
        outs = []
        for s in range(0, R, 65536):
            sl = slice(s, min(s + 65536, R))
            with torch.no_grad():
                z, v = ops.ray_setup(d[sl], near[sl], far[sl], NC, None)
                raw0 = query(net.nerf, "fp32", o[sl], d[sl], v, z)
                raw0[..., 3] *= 1 + rel * (2 * torch.rand(raw0.shape[:2], device=dev, generator=g) - 1)
                ret0, zf, _, _ = ops.composite_importance(raw0, z, d[sl], NI)
                raw = query(net.nerf_fine, "fp32", o[sl], d[sl], v, zf)
                raw[..., 3] *= 1 + rel * (2 * torch.rand(raw.shape[:2], device=dev, generator=g) - 1)
                ret = ops.composite(raw, zf, d[sl])
            outs.append(dict(rgb=ret["rgb"], depth=ret["depth"][:, 0]))
        pert = {k: torch.cat([t[k] for t in outs]) for k in outs[0]}
        cond[name] = stats(pert, exact)
    report["exact_render_with_sigma_perturbed_by_relative_noise"] = cond
    os.makedirs(os.path.dirname(args.out), exist_ok=True)
    with open(args.out, "w") as f:
        json.dump(report, f, indent=1)
    brief = {p: {"free": report[p]["free_running_vs_exact"], "sub": report[p]["substitution_all_rays"], "edge": report[p]["outliers_gt_0.01_within_3px_of_an_analytic_depth_edge"],
                 "worst": report[p]["worst_rays_median_max"], "emul": report[p]["emulation"]} for p in args.prec.split(",")}
    brief["cond"] = cond
    print(json.dumps(brief, indent=1))


if __name__ == "__main__":
    main()
