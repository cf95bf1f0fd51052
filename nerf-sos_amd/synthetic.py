"""Synthetic inputs of the BASELINE configs: there is no dataset in the build or on the
GPU box, so benchmarks, smoke() and the size-independent parity tests draw their rays from the pinhole camera the
survey fixes -- H x W = 756 x 1008, focal 850 px, identity pose, near/far of the LLFF scenes.

Rays come from the package's own on-device generator K0 (`ops.generate_rays` = `get_persp_rays`, utils/ray.py:12-22):
unnormalised directions ``d = [(i-W/2)/f, -(j-H/2)/f, -1]``, origin 0.  Only the *choice of pixels* is made here:
  * `synthetic_rays`   : a seeded random pixel subset (ray mode, `RayNeRFDataset`, data/datasets.py:118-162);
  * `synthetic_patches`: P x P pixel patches with a pixel stride (patch mode: `--patch_size 64 --patch_stride 6`,
                         scripts/train_flower_node0.sh:4-6; data/datasets.py:52-115);
  * `image_rays`       : a contiguous flat pixel range of the image (full-image eval, engines/eval.py:30-42).
"""
from __future__ import annotations

from typing import Tuple

import torch

from . import ops

H, W, FOCAL = 756, 1008, 850.0
NEAR, FAR = 1.2, 14.72          # models/sampler.py:45 comment; LLFF bound scaling data/gen_dataset.py:95-96
_POSE = [[1.0, 0.0, 0.0, 0.0], [0.0, 1.0, 0.0, 0.0], [0.0, 0.0, 1.0, 0.0]]


def intrinsics(h: int = H, w: int = W, focal: float = FOCAL):
    return [[focal, 0.0, w * 0.5], [0.0, focal, h * 0.5], [0.0, 0.0, 1.0]]


def image_rays(device, pix_range: Tuple[int, int] = None, h: int = H, w: int = W, focal: float = FOCAL) -> torch.Tensor:
    """[2, n, 3] rays of the flat pixel range [b, e) of the image (all of it when None), generated on `device`."""
    rng = (0, h * w) if pix_range is None else pix_range
    return ops.generate_rays(h, w, intrinsics(h, w, focal), _POSE, device, pix_range=rng)


def synthetic_rays(n_rays: int, seed: int = 0, device="cuda", h: int = H, w: int = W, focal: float = FOCAL) -> torch.Tensor:
    """[2, n_rays, 3]: a seeded random subset of the image's pixels (the pixel choice is drawn on the host so that
    it does not depend on the device generator)."""
    g = torch.Generator().manual_seed(seed)
    pix = torch.randperm(h * w, generator=g)[:n_rays]
    full = image_rays(device, None, h, w, focal)
    return full[:, pix.to(full.device)].contiguous()


def synthetic_patches(n_patches: int, patch: int = 64, stride: int = 6, seed: int = 0, device="cuda",
                      h: int = H, w: int = W, focal: float = FOCAL) -> torch.Tensor:
    """[2, n_patches, patch, patch, 3]: patches whose pixels are `stride` apart, top-left corners drawn uniformly
    so that the patch fits the image (what the patch dataset's random crops do, data/datasets.py:92-101)."""
    span = (patch - 1) * stride + 1
    if span > h or span > w:
        raise ValueError(f"a {patch}x{patch} patch at stride {stride} spans {span} pixels: larger than the {h}x{w} image")
    g = torch.Generator().manual_seed(seed)
    top = torch.randint(0, h - span + 1, (n_patches,), generator=g)
    left = torch.randint(0, w - span + 1, (n_patches,), generator=g)
    ar = torch.arange(patch) * stride
    rows = top[:, None] + ar[None, :]                        # [B, P]
    cols = left[:, None] + ar[None, :]
    pix = (rows[:, :, None] * w + cols[:, None, :]).reshape(-1)
    full = image_rays(device, None, h, w, focal)
    return full[:, pix.to(full.device)].reshape(2, n_patches, patch, patch, 3).contiguous()


def spiky_density_(net, gain: float = 40.0, shift: float = -1.5):
    """In place: scale the sigma heads of a (random-init) NeRFNet so that a few samples per ray carry almost all the
    weight -- the regime trained scenes are in, and the one that exercises the hierarchical sampler's search and the
    transmittance product.  Random-init weights alone give a flat density (uniform importance samples)."""
    with torch.no_grad():
        for mlp in {id(net.nerf): net.nerf, id(net.nerf_fine): net.nerf_fine}.values():
            mlp.mlp.alpha_linear.weight.mul_(gain)
            mlp.mlp.alpha_linear.bias.mul_(gain).add_(shift)
    return net


# ---------------------------------------------------------------------------------------------------------------------
# A procedural scene with analytic ground truth (round 5): what the trained-field parity / quality evidence is trained on.
# There is no dataset in the build or on the GPU box (the reference's LLFF / CO3D scenes and checkpoints are absent,
# /root/reference/.MISSING_LARGE_BLOBS), so `scripts/make_trained_scene.py` trains the shipped architecture on THIS scene
# with the package's own training path.  Host-side numpy: data generation, not part of the render path.
class ProceduralScene:
    """Three lit spheres (label 1) in front of a back wall and above a floor (label 0), seen by forward-facing pinhole cameras
    in the reference's convention (utils/ray.py:12-22: d = R [(i-W/2)/f, -(j-H/2)/f, -1], o = t).  `view(i)` ray-traces the
    exact image of pose i through the pixel rays the K0 kernel generates, so a NeRF trained on it has a well-defined optimum:
    opaque surfaces at known depths, smooth albedo, Lambertian shading."""

    SPHERES = (  # centre, radius, base colour
        ((-0.95, -0.35, -5.0), 0.85, (0.85, 0.25, 0.20)),
        ((1.05, 0.15, -6.2), 1.10, (0.20, 0.35, 0.85)),
        ((0.10, -0.95, -4.1), 0.50, (0.90, 0.80, 0.25)),
    )
    WALL_Z, FLOOR_Y = -9.0, -1.5
    LIGHT = (0.35, 0.80, 0.50)
    NEAR, FAR = 1.2, 14.72

    def __init__(self, n_views: int = 24, h: int = 120, w: int = 160, focal: float = 150.0, seed: int = 5):
        import numpy as np
        self.h, self.w, self.focal, self.n_views = int(h), int(w), float(focal), int(n_views)
        rng = np.random.default_rng(seed)
        pos = np.stack([rng.uniform(-1.3, 1.3, n_views), rng.uniform(-0.5, 0.9, n_views), rng.uniform(-0.4, 0.4, n_views)], -1)
        target = np.array([0.0, -0.3, -6.0]) + rng.normal(0, 0.15, (n_views, 3))
        poses = np.zeros((n_views, 3, 5), np.float32)                       # LLFF layout [R | t | (H, W, f)], data/gen_dataset.py:228
        for i in range(n_views):
            back = pos[i] - target[i]
            back /= np.linalg.norm(back)
            right = np.cross([0.0, 1.0, 0.0], back)
            right /= np.linalg.norm(right)
            up = np.cross(back, right)
            poses[i, :, 0], poses[i, :, 1], poses[i, :, 2], poses[i, :, 3] = right, up, back, pos[i]
            poses[i, :, 4] = (h, w, focal)
        self.poses = poses
        self.i_test = list(range(0, n_views, 6))
        self.i_train = [i for i in range(n_views) if i not in self.i_test]

    def trace(self, o, d):
        """Exact colour [n,3], label [n] (1 = a sphere) and ray parameter t [n] (in units of the UNNORMALISED d, i.e. what the
        renderer's `depth` map integrates) of rays o + t d, float64 in, float32 out."""
        import numpy as np
        o, d = np.asarray(o, np.float64), np.asarray(d, np.float64)
        n = o.shape[0]
        t_best = np.full(n, np.inf)
        col = np.zeros((n, 3))
        nrm = np.zeros((n, 3))
        lab = np.zeros(n, np.int64)
        light = np.array(self.LIGHT) / np.linalg.norm(self.LIGHT)
        # planes
        with np.errstate(divide="ignore", invalid="ignore"):
            tw = (self.WALL_Z - o[:, 2]) / d[:, 2]
            tf = (self.FLOOR_Y - o[:, 1]) / d[:, 1]
        pw = o + tw[:, None] * d
        hit = (tw > 0) & (tw < t_best)
        wall = 0.55 + 0.25 * np.stack([np.sin(0.9 * pw[:, 0] + 0.4) * np.cos(0.7 * pw[:, 1]), np.sin(0.6 * pw[:, 0] - 1.0) * np.sin(0.8 * pw[:, 1] + 0.5),
                                       np.cos(0.5 * pw[:, 0]) * np.cos(1.1 * pw[:, 1] - 0.3)], -1)
        t_best = np.where(hit, tw, t_best)
        col = np.where(hit[:, None], wall, col)
        nrm = np.where(hit[:, None], np.array([0.0, 0.0, 1.0]), nrm)
        pf = o + tf[:, None] * d
        hit = (tf > 0) & (tf < t_best) & (pf[:, 2] > self.WALL_Z)
        floor = 0.45 + 0.20 * np.stack([np.sin(1.2 * pf[:, 0]) * np.sin(1.0 * pf[:, 2]), np.cos(0.8 * pf[:, 0] + 0.3) * np.sin(0.9 * pf[:, 2] + 1.0),
                                        np.sin(0.7 * pf[:, 0] - 0.6) * np.cos(0.6 * pf[:, 2])], -1) + np.array([0.10, 0.05, -0.05])
        t_best = np.where(hit, tf, t_best)
        col = np.where(hit[:, None], floor, col)
        nrm = np.where(hit[:, None], np.array([0.0, 1.0, 0.0]), nrm)
        lab = np.where(hit, 0, lab)
        # spheres
        for c, r, base in self.SPHERES:
            c = np.array(c)
            oc = o - c
            a = (d * d).sum(-1)
            b = 2 * (oc * d).sum(-1)
            cc = (oc * oc).sum(-1) - r * r
            disc = b * b - 4 * a * cc
            ts = np.where(disc > 0, (-b - np.sqrt(np.maximum(disc, 0))) / (2 * a), np.inf)
            hit = (ts > 0) & (ts < t_best)
            p = o + np.where(np.isfinite(ts), ts, 0.0)[:, None] * d
            nn = (p - c) / r
            stripes = 0.80 + 0.20 * np.sin(5.0 * nn[:, 1] + 2.0 * nn[:, 0])
            t_best = np.where(hit, ts, t_best)
            col = np.where(hit[:, None], np.array(base) * stripes[:, None], col)
            nrm = np.where(hit[:, None], nn, nrm)
            lab = np.where(hit, 1, lab)
        shade = 0.40 + 0.60 * np.clip((nrm * light).sum(-1), 0.0, 1.0)
        rgb = np.clip(col * shade[:, None], 0.0, 1.0)
        miss = ~np.isfinite(t_best)
        rgb[miss] = 0.0
        return rgb.astype(np.float32), lab, np.where(miss, 0.0, t_best).astype(np.float32)

    def pixel_rays(self, i: int, h: int = None, w: int = None):
        """Host restatement of utils/ray.py:12-22 for pose i (float64; the device rays of K0 agree to fp32 rounding) at the
        scene's field of view rendered with h x w pixels."""
        import numpy as np
        h, w = h or self.h, w or self.w
        f = self.focal * w / self.w
        jj, ii = np.meshgrid(np.arange(h, dtype=np.float64), np.arange(w, dtype=np.float64), indexing="ij")
        dirs = np.stack([(ii - w * 0.5) / f, -(jj - h * 0.5) / f, -np.ones_like(ii)], -1).reshape(-1, 3)
        R, t = self.poses[i, :, :3].astype(np.float64), self.poses[i, :, 3].astype(np.float64)
        return np.broadcast_to(t, dirs.shape).copy(), dirs @ R.T

    def view(self, i: int, h: int = None, w: int = None):
        """(rgb [h,w,3] float32, mask [h,w,1] float32 in {0,1}, depth [h,w] float32) of pose i."""
        h, w = h or self.h, w or self.w
        o, d = self.pixel_rays(i, h, w)
        rgb, lab, t = self.trace(o, d)
        return rgb.reshape(h, w, 3), lab.reshape(h, w, 1).astype("float32"), t.reshape(h, w)

    def write_prepared(self, root_dir: str, with_rays: bool = False) -> None:
        """The directory data/gen_dataset.py:211-250 writes (`io.PreparedScene` / `data/datasets.py:20-115` read it):
        meta.json + rgbs / masks / poses per split (+ the [N,H,W,2,3] ray files if asked: DeviceScene regenerates them)."""
        import json
        import os
        import numpy as np
        os.makedirs(root_dir, exist_ok=True)
        views = [self.view(i) for i in range(self.n_views)]
        rgbs = np.stack([v[0] for v in views])
        masks = np.stack([v[1] for v in views])
        for split, idx in (("train", self.i_train), ("val", self.i_test), ("test", self.i_test)):
            np.save(os.path.join(root_dir, f"rgbs_{split}.npy"), rgbs[idx])
            np.save(os.path.join(root_dir, f"masks_{split}.npy"), masks[idx])
            np.save(os.path.join(root_dir, f"poses_{split}.npy"), self.poses[idx])
            if with_rays:
                rays = np.stack([np.stack(self.pixel_rays(i), 0).reshape(2, self.h, self.w, 3).transpose(1, 2, 0, 3) for i in idx])
                np.save(os.path.join(root_dir, f"rays_{split}.npy"), rays.astype(np.float32))
        meta = {"H": self.h, "W": self.w, "focal": self.focal, "near": self.NEAR, "far": self.FAR, "i_train": self.i_train,
                "i_val": self.i_test, "i_test": self.i_test, "ndc": False, "factor": 1, "spherify": False, "llffhold": 6,
                "half_res": False, "white_bkgd": False, "test_skip": 1}
        with open(os.path.join(root_dir, "meta.json"), "w") as f:
            json.dump(meta, f)
