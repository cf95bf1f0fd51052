#!/usr/bin/env python3
"""Generate the committed golden fixtures from the REAL reference implementation.

Runs ONLY in the build container, where the upstream repo is mounted read-only at
/root/reference.  It (1) imports the reference's hot-path modules unmodified (two unused
third-party imports are stubbed in memory), (2) runs them on seeded inputs,
(3) asserts that ``oracle/torch_port.py`` reproduces every reference output bit for bit on
CPU, and (4) writes inputs + reference outputs as ``.npz`` data fixtures next to this file.
Nothing from /root/reference is copied: fixtures hold tensors only.

    python tests/golden/make_goldens.py
"""
import hashlib
import json
import os
import sys
import types

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
REF = os.environ.get("NERF_SOS_REFERENCE", "/root/reference")

sys.modules["imageio"] = types.ModuleType("imageio")
_tb = types.ModuleType("torch.utils.tensorboard")
_tb.SummaryWriter = object
sys.modules["torch.utils.tensorboard"] = _tb
sys.path.insert(0, REF)
sys.path.insert(0, ROOT)

from models.nerf_net import NeRFNet  # noqa: E402  (reference)
from models.embedder import PositionEncoder  # noqa: E402
from models.renderer import VolumetricRenderer  # noqa: E402
from models.sampler import ImportanceSampler, StratifiedSampler  # noqa: E402

torch.autograd.set_detect_anomaly(False)  # the reference turns it on at import (models/sampler.py:2)

from oracle import torch_port as tp  # noqa: E402

torch.set_grad_enabled(False)


def np32(t):
    return t.detach().cpu().numpy()


def same(a, b, what):
    assert a.shape == b.shape and torch.equal(a, b), f"port != reference for {what}"


def state_sha(sd):
    h = hashlib.sha256()
    for k in sorted(sd):
        h.update(np32(sd[k]).astype("<f4").tobytes())
    return h.hexdigest()


class Recorder:
    """Captures torch.rand / torch.randn results in call order while the reference runs."""

    def __enter__(self):
        self.draws = []
        self._rand, self._randn = torch.rand, torch.randn

        def rand(*a, **k):
            t = self._rand(*a, **k)
            self.draws.append(("rand", t.clone()))
            return t

        def randn(*a, **k):
            t = self._randn(*a, **k)
            self.draws.append(("randn", t.clone()))
            return t

        torch.rand, torch.randn = rand, randn
        return self

    def __exit__(self, *exc):
        torch.rand, torch.randn = self._rand, self._randn


CFGS = {
    "nosem": dict(use_semantics=False, sem_with_coord=False),
    "semcoord": dict(use_semantics=True, sem_with_coord=True),
    "sem": dict(use_semantics=True, sem_with_coord=False),
}


def build_ref(cfg_name, n_importance=128, white_bkgd=False, peaky=False, raw_noise_std=0.0):
    torch.manual_seed(0)
    net = NeRFNet(N_samples=64, N_importance=n_importance, perturb=1.0, raw_noise_std=raw_noise_std,
                  white_bkgd=white_bkgd, pts_chuck=1024 * 64, **CFGS[cfg_name])
    pc = tp.PortConfig(n_importance=n_importance, white_bkgd=white_bkgd, **CFGS[cfg_name])
    sd = tp.init_state_dict(pc, seed=0)
    ref_sd = net.state_dict()
    assert list(ref_sd.keys()) == list(sd.keys()), "state-dict key order differs"
    for k in sd:
        same(ref_sd[k], sd[k], f"init {k}")
    if peaky:
        sd = tp.make_peaky(sd)
        net.load_state_dict(sd)
    return net, pc, sd


def main():
    manifest = {"torch": torch.__version__, "state_sha256": {}}

    # ---------------------------------------------------------------- weights hash
    for name in CFGS:
        _, _, sd = build_ref(name)
        manifest["state_sha256"][name] = state_sha(sd)
        manifest.setdefault("state_keys", {})[name] = list(sd.keys())
    _, _, sd1 = build_ref("nosem", n_importance=0)
    manifest["state_sha256"]["nosem_coarse_only"] = state_sha(sd1)
    _, _, sd2 = build_ref("semcoord", n_importance=0)
    manifest["state_sha256"]["semcoord_coarse_only"] = state_sha(sd2)

    g = torch.Generator().manual_seed(1234)

    # ---------------------------------------------------------------- stage: stratified sampler
    out = {}
    for R in (1, 7, 64, 257):
        near = 1.2 + 0.3 * torch.rand(R, 1, generator=g)
        far = 14.72 - 2.0 * torch.rand(R, 1, generator=g)
        o = torch.randn(R, 3, generator=g)
        d = torch.randn(R, 3, generator=g)
        t_rand = torch.rand(R, 64, generator=g)
        s = StratifiedSampler(64, perturb=1.0)
        _rand = torch.rand
        torch.rand = lambda *a, **k: t_rand  # inject the jitter
        pts, z = s(o, d, torch.cat([near, far], -1))
        torch.rand = _rand
        pts0, z0 = s(o, d, torch.cat([near, far], -1), perturb=0.0)
        same(z, tp.stratified_z(near, far, 64, t_rand), "stratified z")
        same(z0, tp.stratified_z(near, far, 64, None), "stratified z det")
        same(pts, tp.ray_points(o, d, z), "stratified pts")
        out.update({f"R{R}_near": np32(near), f"R{R}_far": np32(far), f"R{R}_o": np32(o), f"R{R}_d": np32(d),
                    f"R{R}_t_rand": np32(t_rand), f"R{R}_z": np32(z), f"R{R}_z_det": np32(z0),
                    f"R{R}_pts": np32(pts)})
    np.savez_compressed(os.path.join(HERE, "stratified.npz"), **out)

    # ---------------------------------------------------------------- stage: positional encoding
    x = (torch.rand(257, 3, generator=g) * 30.0 - 15.0)
    x[0] = torch.tensor([0.0, -15.0, 15.0])
    v = torch.nn.functional.normalize(torch.randn(257, 3, generator=g), dim=-1)
    e10 = PositionEncoder(3, 10, 9)(x)
    e4 = PositionEncoder(3, 4, 3)(v)
    same(e10, tp.posenc(x, 10), "posenc L=10")
    same(e4, tp.posenc(v, 4), "posenc L=4")
    np.savez_compressed(os.path.join(HERE, "posenc.npz"), x=np32(x), v=np32(v), e10=np32(e10), e4=np32(e4))

    # ---------------------------------------------------------------- stage: MLP with per-layer taps
    out = {}
    pts = torch.rand(16, 3, generator=g) * 8.0 - 4.0
    dirs = torch.nn.functional.normalize(torch.randn(16, 3, generator=g), dim=-1)
    out["pts"], out["dirs"] = np32(pts), np32(dirs)
    for name in CFGS:
        for peaky in (False, True):
            net, pc, sd = build_ref(name, peaky=peaky)
            tag = f"{name}_{'peaky' if peaky else 'default'}"
            for prefix, sub in (("nerf", net.nerf), ("nerf_fine", net.nerf_fine)):
                raw = sub(pts, viewdirs=dirs)
                same(raw, tp.point_query(sd, prefix, pts, dirs, pc), f"mlp {tag} {prefix}")
                out[f"{tag}_{prefix}_raw"] = np32(raw)
            taps = {}
            enc = torch.cat([tp.posenc(pts, 10), tp.posenc(dirs, 4)], -1)
            same(net.nerf_fine.mlp(enc), tp.mlp_forward(sd, "nerf_fine", enc, pc, tap=lambda k, t: taps.__setitem__(k, t.clone())),
                 f"mlp taps {tag}")
            # the reference's own intermediate activations, captured with forward hooks
            hooks, ref_taps = [], {}
            m = net.nerf_fine.mlp
            for i, l in enumerate(m.pts_linears):
                hooks.append(l.register_forward_hook(lambda mod, inp, o, i=i: ref_taps.__setitem__(f"h{i}", torch.relu(o))))
            hooks.append(m.feature_linear.register_forward_hook(lambda mod, inp, o: ref_taps.__setitem__("feature", o.clone())))
            hooks.append(m.views_linears[0].register_forward_hook(lambda mod, inp, o: ref_taps.__setitem__("view_hidden", torch.relu(o))))
            m(enc)
            for h in hooks:
                h.remove()
            for k, t in ref_taps.items():
                same(t, taps[k], f"tap {k} {tag}")
                if name == "semcoord":
                    out[f"{tag}_tap_{k}"] = np32(t)
    np.savez_compressed(os.path.join(HERE, "mlp.npz"), **out)

    # ---------------------------------------------------------------- stage: compositing (+ edge cases)
    out = {}
    for C, sem in ((4, False), (6, True)):
        for S in (64, 192):
            for white in (False, True):
                R = 21
                raw = torch.randn(R, S, C, generator=g) * 2.0
                z = torch.sort(1.2 + 13.5 * torch.rand(R, S, generator=g), -1)[0]
                d = torch.randn(R, 3, generator=g)
                noise = torch.randn(R, S, generator=g)
                # edge cases: empty ray, opaque first sample, huge sigma, duplicate z
                raw[0, :, 3] = -1.0 - torch.rand(S, generator=g)
                raw[1, :, 3] = 0.0
                raw[2, 0, 3] = 1e9
                raw[3, :, 3] = 1e4
                z[4, 10:14] = z[4, 10]
                raw[5, :, 3] = 1e-7
                ren = VolumetricRenderer(raw_noise_std=0.0, white_bkgd=white, use_semantics=sem)
                pc = tp.PortConfig(use_semantics=sem, white_bkgd=white)
                for noisy in (False, True):
                    if noisy:
                        _randn = torch.randn
                        torch.randn = lambda *a, **k: noise
                        ret = ren(raw, z, d, raw_noise_std=0.7)
                        torch.randn = _randn
                        port = tp.composite(raw, z, d, noise * 0.7, pc)
                    else:
                        ret = ren(raw, z, d)
                        port = tp.composite(raw, z, d, None, pc)
                    tag = f"C{C}_S{S}_{'white' if white else 'black'}_{'noise' if noisy else 'clean'}"
                    for k in ret:
                        same(ret[k], port[k], f"composite {tag} {k}")
                        out[f"{tag}_{k}"] = np32(ret[k])
                tag = f"C{C}_S{S}_{'white' if white else 'black'}"
                out.update({f"{tag}_raw": np32(raw), f"{tag}_z": np32(z), f"{tag}_d": np32(d), f"{tag}_noise": np32(noise)})
    np.savez_compressed(os.path.join(HERE, "composite.npz"), **out)

    # ---------------------------------------------------------------- stage: importance sampling
    out = {}
    imp = ImportanceSampler(128, perturb=1.0)
    for R in (1, 7, 64, 257):
        near = torch.full((R, 1), 1.2)
        far = torch.full((R, 1), 14.72)
        z = tp.stratified_z(near, far, 64, torch.rand(R, 64, generator=g))
        w = torch.rand(R, 64, generator=g) ** 8  # peaky
        if R >= 7:
            w[0] = 0.0                     # all-zero weights -> uniform pdf from the 1e-5 floor
            w[1] = 1.0                     # flat
            w[2] = 0.0
            w[2, 31] = 1.0                 # a single spike
            w[3] = 0.0
            w[3, 1] = 0.5
            w[3, 62] = 0.5                 # mass at both ends of the inner range
            w[4, :] = 1e-7
        u = torch.rand(R, 128, generator=g)
        if R >= 7:
            u[5, 0] = 0.0
            u[5, 1] = 1.0 - 2 ** -24       # largest fp32 below 1
        o = torch.randn(R, 3, generator=g)
        d = torch.randn(R, 3, generator=g)
        mids = 0.5 * (z[..., 1:] + z[..., :-1])
        cdf = tp.pdf_to_cdf(w[..., 1:-1])
        for det in (False, True):
            uu = torch.linspace(0.0, 1.0, steps=128).expand(R, 128) if det else u
            _rand = torch.rand
            torch.rand = lambda *a, **k: u
            pts, z_fine, extras = imp(o, d, z, w, perturb=0.0 if det else 1.0)
            torch.rand = _rand
            s_port, inds = tp.invert_cdf(mids, cdf, uu)
            zf_port, zs_port = tp.importance_z(z, w, 128, None if det else u)
            same(extras["z_samples"], s_port, "z_samples")
            same(extras["z_samples"], zs_port, "z_samples (importance_z)")
            same(z_fine, zf_port, "z_fine")
            same(pts, tp.ray_points(o, d, z_fine), "fine pts")
            tag = f"R{R}_{'det' if det else 'rand'}"
            out.update({f"{tag}_inds": inds.numpy().astype(np.int64), f"{tag}_z_samples": np32(extras["z_samples"]),
                        f"{tag}_z_fine": np32(z_fine),
                        f"{tag}_z_std": np32(torch.std(extras["z_samples"], dim=-1, unbiased=False))})
            if R <= 64:
                out[f"{tag}_pts"] = np32(pts)
        out.update({f"R{R}_z": np32(z), f"R{R}_w": np32(w), f"R{R}_u": np32(u), f"R{R}_o": np32(o), f"R{R}_d": np32(d),
                    f"R{R}_cdf": np32(cdf)})
    np.savez_compressed(os.path.join(HERE, "importance.npz"), **out)

    # ---------------------------------------------------------------- end to end
    out = {}
    rays = tp.synthetic_rays(16, seed=7)
    out["rays"] = np32(rays)
    cases = [("nosem", False, False, 128), ("semcoord", False, False, 128), ("semcoord", True, False, 128),
             ("sem", True, True, 128), ("nosem", True, False, 0)]
    for name, peaky, white, n_imp in cases:
        tag = f"{name}_{'peaky' if peaky else 'default'}{'_white' if white else ''}{'_coarse' if n_imp == 0 else ''}"
        net, pc, sd = build_ref(name, n_importance=n_imp, white_bkgd=white, peaky=peaky, raw_noise_std=1.0)
        net.eval()
        ret = net(rays, (tp.NEAR, tp.FAR), radii=None)
        port = tp.render(sd, pc, rays, (tp.NEAR, tp.FAR))
        assert set(ret) == set(port), (sorted(ret), sorted(port))
        for k in ret:
            same(ret[k], port[k], f"e2e eval {tag} {k}")
            out[f"{tag}_eval_{k}"] = np32(ret[k])
        net.train()
        torch.manual_seed(99)
        with Recorder() as rec:
            ret = net(rays, (tp.NEAR, tp.FAR), radii=None)
        kinds = [k for k, _ in rec.draws]
        assert kinds == (["rand", "randn", "rand", "randn"] if n_imp else ["rand", "randn"]), kinds
        dr = [t for _, t in rec.draws]
        draws = tp.Draws(*dr) if n_imp else tp.Draws(dr[0], dr[1])
        port = tp.render(sd, pc, rays, (tp.NEAR, tp.FAR), raw_noise_std=1.0, draws_per_chunk=[draws])
        for k in ret:
            same(ret[k], port[k], f"e2e train {tag} {k}")
            out[f"{tag}_train_{k}"] = np32(ret[k])
        for i, t in enumerate(dr):
            out[f"{tag}_train_draw{i}"] = np32(t)
    np.savez_compressed(os.path.join(HERE, "end_to_end.npz"), **out)

    # ---------------------------------------------------------------- K0: ray generation (utils/ray.py)
    from utils.ray import get_persp_rays, get_persp_intrinsic
    out = {}
    for idx, (H_, W_, f_) in enumerate(((37, 53, 41.7), (63, 84, 70.83), (8, 5, 3.0))):
        K_ = get_persp_intrinsic(H_, W_, f_)
        c2w = torch.cat([torch.linalg.qr(torch.randn(3, 3, generator=g))[0], torch.randn(3, 1, generator=g) * 3], 1)
        out[f"case{idx}_HWf"] = np.array([H_, W_, f_], np.float64)
        out[f"case{idx}_K"] = np32(K_)
        out[f"case{idx}_c2w"] = np32(c2w)
        out[f"case{idx}_rays"] = np32(get_persp_rays(H_, W_, K_, c2w))
    np.savez_compressed(os.path.join(HERE, "rays.npz"), **out)

    # ---------------------------------------------------------------- K5: frozen-backbone gradients
    torch.set_grad_enabled(True)
    out = {}
    rays_g = tp.synthetic_rays(12, seed=11)
    out["rays"] = np32(rays_g)
    gg = torch.Generator().manual_seed(4321)
    for name, peaky, white, n_imp in (("semcoord", True, False, 128), ("sem", True, True, 128), ("semcoord", False, False, 0)):
        tag = f"{name}_{'peaky' if peaky else 'default'}{'_white' if white else ''}{'_coarse' if n_imp == 0 else ''}"
        net, pc, sd = build_ref(name, n_importance=n_imp, white_bkgd=white, peaky=peaky)
        for n_, p_ in net.named_parameters():           # run_nerf.py:307-318 (--fix_backbone)
            p_.requires_grad = 'semantic_linear' in n_
        net.eval()
        ret = net(rays_g, (tp.NEAR, tp.FAR), radii=None)
        G = torch.randn(ret["semantics"].shape, generator=gg)
        loss = (ret["semantics"] * G).sum()
        out[f"{tag}_G"] = np32(G)
        if n_imp:
            G0 = torch.randn(ret["semantics0"].shape, generator=gg)
            loss = loss + (ret["semantics0"] * G0).sum()
            out[f"{tag}_G0"] = np32(G0)
        loss.backward()
        seen = set()
        for n_, p_ in net.named_parameters():
            if p_.requires_grad and id(p_) not in seen:
                seen.add(id(p_))
                out[f"{tag}_grad_{n_}"] = np32(p_.grad)
        out[f"{tag}_semantics"] = np32(ret["semantics"])
    np.savez_compressed(os.path.join(HERE, "sem_grads.npz"), **out)
    torch.set_grad_enabled(False)

    with open(os.path.join(HERE, "manifest.json"), "w") as f:
        json.dump(manifest, f, indent=1)
    tot = sum(os.path.getsize(os.path.join(HERE, n)) for n in os.listdir(HERE) if n.endswith(".npz"))
    print(f"wrote goldens: {tot / 1e6:.2f} MB; sha(semcoord)={manifest['state_sha256']['semcoord']}")


if __name__ == "__main__":
    main()
