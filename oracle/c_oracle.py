"""TEST INFRASTRUCTURE -- ctypes/numpy front end of the plain-C oracle (oracle/nerf_oracle.c).

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess
from typing import Dict, Optional

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, "liboracle.so")
_lib = None

_f32p = C.POINTER(C.c_float)
_i64p = C.POINTER(C.c_int64)


class _Weights(C.Structure):
    _fields_ = [("pts_w", _f32p * 8), ("pts_b", _f32p * 8),
                ("alpha_w", _f32p), ("alpha_b", _f32p), ("feature_w", _f32p), ("feature_b", _f32p),
                ("views_w", _f32p), ("views_b", _f32p), ("rgb_w", _f32p), ("rgb_b", _f32p),
                ("sem0_w", _f32p), ("sem0_b", _f32p), ("sem2_w", _f32p), ("sem2_b", _f32p),
                ("use_semantics", C.c_int32), ("sem_with_coord", C.c_int32)]


class _Taps(C.Structure):
    _fields_ = [("h", _f32p * 8), ("feature", _f32p), ("view_hidden", _f32p), ("sem_hidden", _f32p)]


def build(force: bool = False) -> str:
    """Compile liboracle.so with gcc if it is missing (or ``force``)."""
    src = os.path.join(_HERE, "nerf_oracle.c")
    stale = (not os.path.exists(_LIB_PATH)) or os.path.getmtime(_LIB_PATH) < os.path.getmtime(src)
    if force or stale:
        subprocess.check_call(["make", "-C", _HERE, "-B", "liboracle.so"], stdout=subprocess.DEVNULL)
    return _LIB_PATH


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(_LIB_PATH):
            build()
        _lib = C.CDLL(_LIB_PATH)
        _lib.oracle_num_threads.restype = C.c_int32
    return _lib


def _p(a: Optional[np.ndarray], ty=_f32p):
    return None if a is None else a.ctypes.data_as(ty)


def _f(a) -> Optional[np.ndarray]:
    if a is None:
        return None
    if hasattr(a, "detach"):
        a = a.detach().cpu().numpy()
    return np.ascontiguousarray(a, dtype=np.float32)


def num_threads() -> int:
    return int(lib().oracle_num_threads())


def generate_rays(H: int, W: int, K, c2w):
    K, pose = _f(K), np.ascontiguousarray(_f(c2w)[:3, :4])
    o = np.empty((H, W, 3), np.float32)
    d = np.empty((H, W, 3), np.float32)
    lib().oracle_generate_rays(C.c_int32(H), C.c_int32(W), C.c_float(K[0, 0]), C.c_float(K[1, 1]), C.c_float(K[0, 2]),
                               C.c_float(K[1, 2]), _p(pose), _p(o), _p(d))
    return np.stack([o, d], 0)


def ray_setup(rays_o, rays_d, near, far, t_rand, n_samples: int):
    o, d, n, f, t = _f(rays_o), _f(rays_d), _f(near).reshape(-1), _f(far).reshape(-1), _f(t_rand)
    R = d.shape[0]
    z = np.empty((R, n_samples), np.float32)
    v = np.empty((R, 3), np.float32)
    lib().oracle_ray_setup(_p(o), _p(d), _p(n), _p(f), _p(t), C.c_int64(R), C.c_int32(n_samples), _p(z), _p(v))
    return z, v


def ray_points(rays_o, rays_d, z_vals):
    o, d, z = _f(rays_o), _f(rays_d), _f(z_vals)
    R, S = z.shape
    pts = np.empty((R, S, 3), np.float32)
    lib().oracle_ray_points(_p(o), _p(d), _p(z), C.c_int64(R), C.c_int32(S), _p(pts))
    return pts


def posenc(x, n_freqs: int):
    x = _f(x)
    out = np.empty(x.shape[:-1] + (3 + 6 * n_freqs,), np.float32)
    lib().oracle_posenc(_p(x), C.c_int64(x.size // 3), C.c_int32(n_freqs), _p(out))
    return out


class Weights:
    """Holds contiguous fp32 copies of one net's tensors and the C struct pointing at them."""

    def __init__(self, state_dict: Dict, prefix: str, use_semantics: bool, sem_with_coord: bool):
        g = lambda name: _f(state_dict[f"{prefix}.mlp.{name}"])  # noqa: E731
        self.keep = []
        w = _Weights()

        def put(name):
            a = g(name)
            self.keep.append(a)
            return _p(a)

        for i in range(8):
            w.pts_w[i] = put(f"pts_linears.{i}.weight")
            w.pts_b[i] = put(f"pts_linears.{i}.bias")
        for field, name in (("alpha", "alpha_linear"), ("feature", "feature_linear"),
                            ("views", "views_linears.0"), ("rgb", "rgb_linear")):
            setattr(w, field + "_w", put(name + ".weight"))
            setattr(w, field + "_b", put(name + ".bias"))
        if use_semantics:
            w.sem0_w, w.sem0_b = put("semantic_linear.0.weight"), put("semantic_linear.0.bias")
            w.sem2_w, w.sem2_b = put("semantic_linear.2.weight"), put("semantic_linear.2.bias")
        w.use_semantics, w.sem_with_coord = int(use_semantics), int(sem_with_coord)
        self.c = w
        self.n_ch = 6 if use_semantics else 4


def mlp(weights: Weights, pts, dirs, dirs_stride_pts: int = 1, taps: bool = False):
    pts, dirs = _f(pts).reshape(-1, 3), _f(dirs).reshape(-1, 3)
    P = pts.shape[0]
    assert dirs.shape[0] * dirs_stride_pts >= P
    raw = np.empty((P, weights.n_ch), np.float32)
    tp, tapd = None, None
    if taps:
        tapd = {f"h{i}": np.empty((P, 256), np.float32) for i in range(8)}
        tapd.update(feature=np.empty((P, 256), np.float32), view_hidden=np.empty((P, 128), np.float32),
                    sem_hidden=np.zeros((P, 128), np.float32))
        t = _Taps()
        for i in range(8):
            t.h[i] = _p(tapd[f"h{i}"])
        t.feature, t.view_hidden, t.sem_hidden = _p(tapd["feature"]), _p(tapd["view_hidden"]), _p(tapd["sem_hidden"])
        tp = C.byref(t)
    lib().oracle_mlp(C.byref(weights.c), _p(pts), _p(dirs), C.c_int64(P), C.c_int64(dirs_stride_pts), _p(raw), tp)
    return (raw, tapd) if taps else raw


def composite(raw, z_vals, rays_d, noise=None, noise_std: float = 0.0, white_bkgd: bool = False):
    raw, z, d, nz = _f(raw), _f(z_vals), _f(rays_d), _f(noise)
    R, S, Cn = raw.shape
    out = dict(weights=np.empty((R, S), np.float32), rgb=np.empty((R, 3), np.float32),
               depth=np.empty((R, 1), np.float32), acc=np.empty((R, 1), np.float32), disp=np.empty((R, 1), np.float32))
    sem = np.empty((R, Cn - 4), np.float32) if Cn > 4 else None
    lib().oracle_composite(_p(raw), _p(z), _p(d), _p(nz), C.c_float(noise_std), C.c_int64(R), C.c_int32(S),
                           C.c_int32(Cn), C.c_int32(int(white_bkgd)), _p(out["weights"]), _p(out["rgb"]), _p(sem),
                           _p(out["depth"]), _p(out["acc"]), _p(out["disp"]))
    if sem is not None:
        out["semantics"] = sem
    return out


def searchsorted_right(cdf, u):
    cdf, u = _f(cdf), _f(u)
    R = cdf.shape[0]
    inds = np.empty(u.shape, np.int64)
    lib().oracle_searchsorted_right(_p(cdf), C.c_int32(cdf.shape[1]), _p(u), C.c_int32(u.shape[1]), C.c_int64(R),
                                    _p(inds, _i64p))
    return inds


def importance(z_vals, weights, u, n_importance: int, cdf_in=None):
    z, w, u, ci = _f(z_vals), _f(weights), _f(u), _f(cdf_in)
    R, S = z.shape
    N = n_importance
    out = dict(cdf=np.empty((R, S - 1), np.float32), inds=np.empty((R, N), np.int64),
               z_samples=np.empty((R, N), np.float32), z_fine=np.empty((R, S + N), np.float32),
               z_std=np.empty((R,), np.float32))
    lib().oracle_importance(_p(z), _p(w), _p(u), _p(ci), C.c_int64(R), C.c_int32(S), C.c_int32(N), _p(out["cdf"]),
                            _p(out["inds"], _i64p), _p(out["z_samples"]), _p(out["z_fine"]), _p(out["z_std"]))
    return out


def render(state_dict, rays_o, rays_d, near, far, *, use_semantics=False, sem_with_coord=False, n_samples=64,
           n_importance=128, white_bkgd=False, raw_noise_std=0.0, t_rand=None, noise0=None, u=None, noise1=None):
    """One ray chunk of models/nerf_net.py:71-130 through the C oracle's stages.
    near / far: [R] or [R,1] arrays.  Returns the reference's output dict as numpy arrays."""
    o, d = _f(rays_o).reshape(-1, 3), _f(rays_d).reshape(-1, 3)
    R = d.shape[0]
    near = np.broadcast_to(_f(near).reshape(-1), (R,)) if np.ndim(near) else np.full((R,), near, np.float32)
    far = np.broadcast_to(_f(far).reshape(-1), (R,)) if np.ndim(far) else np.full((R,), far, np.float32)
    wc = Weights(state_dict, "nerf", use_semantics, sem_with_coord)
    z, v = ray_setup(o, d, near, far, t_rand, n_samples)
    raw = mlp(wc, ray_points(o, d, z), v, dirs_stride_pts=n_samples).reshape(R, n_samples, -1)
    ret = composite(raw, z, d, noise0 if raw_noise_std > 0 else None, raw_noise_std, white_bkgd)
    ret["raw"] = raw
    if n_importance > 0:
        ret0 = ret
        imp = importance(z, ret0["weights"], u, n_importance)
        wf = Weights(state_dict, "nerf_fine", use_semantics, sem_with_coord)
        S2 = n_samples + n_importance
        raw = mlp(wf, ray_points(o, d, imp["z_fine"]), v, dirs_stride_pts=S2).reshape(R, S2, -1)
        ret = composite(raw, imp["z_fine"], d, noise1 if raw_noise_std > 0 else None, raw_noise_std, white_bkgd)
        ret["raw"] = raw
        ret["z_std"] = imp["z_std"]
        ret["_z_fine"], ret["_inds"], ret["_z_vals"] = imp["z_fine"], imp["inds"], z
        for k, val in ret0.items():
            ret[k + "0"] = val
    return ret
