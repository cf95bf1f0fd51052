"""Shared test helpers: seed-0 reference weights (regenerated, hash-checked) and tolerances."""
import hashlib

import numpy as np
import torch

from oracle import torch_port as tp

CFGS = {
    "nosem": dict(use_semantics=False, sem_with_coord=False),
    "semcoord": dict(use_semantics=True, sem_with_coord=True),
    "sem": dict(use_semantics=True, sem_with_coord=False),
}

# fp32 parity bar of BASELINE.json's north_star ("within 1e-4 fp32")
ATOL = 1e-4
RTOL = 1e-4


def state_sha(sd):
    h = hashlib.sha256()
    for k in sorted(sd):
        h.update(sd[k].detach().cpu().numpy().astype("<f4").tobytes())
    return h.hexdigest()


_cache = {}


def ref_state(name, manifest, peaky=False, n_importance=128):
    """The reference's seed-0 default-init state dict for config `name`, rebuilt locally and
    verified against the sha256 the golden generator recorded from the real reference."""
    key = (name, peaky, n_importance)
    if key not in _cache:
        cfg = tp.PortConfig(n_importance=n_importance, **CFGS[name])
        sd = tp.init_state_dict(cfg, seed=0)
        want = manifest["state_sha256"][name if n_importance else name + "_coarse_only"]
        assert state_sha(sd) == want, "seed-0 init differs from the reference's (torch CPU generator changed?)"
        if peaky:
            sd = tp.make_peaky(sd)
        _cache[key] = sd
    return _cache[key]


def close(a, b, atol=ATOL, rtol=RTOL, what=""):
    a = np.asarray(a, dtype=np.float64)
    b = np.asarray(b, dtype=np.float64)
    assert a.shape == b.shape, f"{what}: shape {a.shape} vs {b.shape}"
    both_inf = np.isinf(a) & np.isinf(b) & (np.sign(a) == np.sign(b))
    err = np.where(both_inf, 0.0, np.abs(a - b))
    tol = atol + rtol * np.abs(b)
    bad = ~(err <= tol)
    assert not bad.any(), f"{what}: {bad.sum()} / {bad.size} outside tol; max err {np.nanmax(err):.3e} at {np.unravel_index(np.nanargmax(err - tol), err.shape)}"


def tag_of(name, peaky, white=False, coarse=False):
    return f"{name}_{'peaky' if peaky else 'default'}{'_white' if white else ''}{'_coarse' if coarse else ''}"


# ---- architectures other than the shipped one (tests/golden/make_goldens_generic.py: the same table, the same weights)
GENERIC_CASES = {   # name -> (NeRFNet ctor kwargs, PortConfig kwargs)
    "d4w128": (dict(netdepth=4, netwidth=128, netdepth_fine=4, netwidth_fine=128, N_samples=24, N_importance=40),
               dict(net_depth=4, net_width=128, n_samples=24, n_importance=40)),
    "d6w96_m6": (dict(netdepth=6, netwidth=96, netdepth_fine=6, netwidth_fine=96, multires=6, multires_views=2, N_samples=16,
                      N_importance=16, use_semantics=True, sem_dim=3, sem_with_coord=True),
                 dict(net_depth=6, net_width=96, multires=6, multires_views=2, n_samples=16, n_importance=16, use_semantics=True,
                      sem_dim=3, sem_with_coord=True)),
    "deepsem": (dict(N_samples=16, N_importance=24, use_semantics=True, sem_layer=4, sem_with_coord=True),
                dict(n_samples=16, n_importance=24, use_semantics=True, sem_layer=4, sem_with_coord=True)),
    "deepsem3_geo": (dict(netwidth=64, netwidth_fine=64, N_samples=16, N_importance=16, use_semantics=True, sem_layer=3, sem_dim=2,
                          sem_with_geo=True),
                     dict(net_width=64, n_samples=16, n_importance=16, use_semantics=True, sem_layer=3, sem_dim=2, sem_with_geo=True)),
    "noview": (dict(viewdirs=False, N_samples=20, N_importance=20), dict(use_viewdirs=False, n_samples=20, n_importance=20)),
    "noembed": (dict(use_embed=False, viewdirs=False, netwidth=128, netwidth_fine=128, N_samples=16, N_importance=0),
                dict(use_embed=False, use_viewdirs=False, net_width=128, n_samples=16, n_importance=0)),
    "sem7": (dict(netdepth=8, netwidth=256, N_samples=12, N_importance=12, use_semantics=True, sem_dim=7),
             dict(n_samples=12, n_importance=12, use_semantics=True, sem_dim=7)),
    # wider than the 32-point tiles' LDS budget: the kernels' 16-point tiles (csrc/mlp_generic.hip)
    "w512_deepsem": (dict(netdepth=4, netwidth=512, netdepth_fine=4, netwidth_fine=512, N_samples=8, N_importance=8, use_semantics=True,
                          sem_layer=3, sem_with_coord=True),
                     dict(net_depth=4, net_width=512, n_samples=8, n_importance=8, use_semantics=True, sem_layer=3, sem_with_coord=True)),
}


def generic_state(name, golden):
    """The weights of generic case `name` as the REAL reference initialised them (rebuilt from the port under the case's seed and
    checked against the hash the golden generator recorded from the reference's own state_dict), with the spiky density head."""
    g = golden("generic")
    cfg = tp.PortConfig(**GENERIC_CASES[name][1])
    seed = int(g[f"{name}__seed"][0])
    sd = {k: v.clone() for k, v in tp.init_state_dict(cfg, seed=seed).items()}
    for net in ("nerf", "nerf_fine"):
        for k in (f"{net}.mlp.alpha_linear", f"{net}.mlp.output_linear"):
            if k + ".weight" in sd and (net == "nerf" or cfg.n_importance > 0):
                row = slice(None) if "alpha" in k else slice(3, 4)
                sd[k + ".weight"][row] *= 40.0
                sd[k + ".bias"][row] = sd[k + ".bias"][row] * 40.0 + 1.0
    if cfg.n_importance == 0:
        for k in [k for k in sd if k.startswith("nerf.")]:
            sd["nerf_fine." + k[len("nerf."):]] = sd[k]
    assert state_sha(sd) == bytes(g[f"{name}__state_sha256"]).hex(), "the port's initialisation differs from the reference's"
    return cfg, sd
