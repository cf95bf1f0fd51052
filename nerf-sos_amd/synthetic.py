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
