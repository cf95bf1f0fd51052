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
