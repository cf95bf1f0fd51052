"""Readers / writers of the reference's on-disk formats around the render path.

* checkpoints: `torch.save({'global_step', 'model', 'optimizer'})` (engines/trainer.py:216-222), re-loaded with
  `strict = not args.load_nostrict` and a tolerated optimizer mismatch (run_nerf.py:344-360);
* prepared scenes: the directory `data/gen_dataset.py:211-250` writes -- `meta.json` (H, W, focal, near, far, ...),
  `rays_<split>.npy [N,H,W,2,3]`, `rgbs_<split>.npy [N,H,W,3]`, `masks_<split>.npy [N,H,W,1]`, `poses_<split>.npy
  [N,3,5 or 3,4]` -- as `data/datasets.py:20-115` reads them.

`PreparedScene.rays_on_device` does not read the ray file at all: it regenerates a view's rays from its pose with the
K0 kernel (bit-identical to `utils/ray.py:12-22`), which removes the `[N,H,W,2,3]` tensors from disk and PCIe.
"""
from __future__ import annotations

import json
import math
import os
from typing import Optional

import numpy as np
import torch

from . import ops


def save_checkpoint(path: str, global_step: int, model: torch.nn.Module, optimizer: Optional[torch.optim.Optimizer]) -> None:
    """engines/trainer.py:216-222."""
    torch.save({'global_step': global_step, 'model': model.state_dict(),
                'optimizer': optimizer.state_dict() if optimizer is not None else {}}, path)


def load_checkpoint(path: str, model: torch.nn.Module, optimizer: Optional[torch.optim.Optimizer] = None,
                    strict: bool = True) -> int:
    """run_nerf.py:349-360: returns global_step; a failing optimizer restore is reported and ignored like there."""
    ckpt = torch.load(path, map_location='cpu')
    model.load_state_dict(ckpt['model'], strict=strict)
    if optimizer is not None:
        try:
            optimizer.load_state_dict(ckpt['optimizer'])
        except Exception as e:  # noqa: BLE001  (the reference swallows it too)
            print(f"[Error]: optimizer initialization failed! ({type(e).__name__})")
    return int(ckpt['global_step'])


class PreparedScene:
    """One split of a scene directory prepared by the reference's `gen_dataset` (data/datasets.py:20-115)."""

    def __init__(self, root_dir: str, split: str = 'train', subsample: int = 0, rgb: bool = True, use_masks: bool = True,
                 bin_thres: float = 0.3, load_rays: bool = True):
        with open(os.path.join(root_dir, 'meta.json'), 'r') as f:
            self.meta_dict = json.load(f)
        if not all(k in self.meta_dict for k in ('near', 'far')):
            raise IOError('Missing required meta data')                               # data/datasets.py:31-33
        sfx = f'_x{subsample}' if subsample != 0 else ''
        self.rays = np.load(os.path.join(root_dir, f'rays_{split}{sfx}.npy')) if load_rays else None   # [N,H,W,ro+rd,3]
        self.rgbs = np.load(os.path.join(root_dir, f'rgbs_{split}{sfx}.npy')) if rgb else None
        ppath = os.path.join(root_dir, f'poses_{split}.npy')
        self.poses = np.load(ppath) if os.path.exists(ppath) else None
        shape = (self.rays if self.rays is not None else self.rgbs).shape if (load_rays or rgb) else \
            (len(self.poses), int(self.meta_dict['H']), int(self.meta_dict['W']))
        self.image_count, self.height, self.width = int(shape[0]), int(shape[1]), int(shape[2])
        self.masks = None
        if use_masks:
            mpath = os.path.join(root_dir, f'masks_{split}.npy')
            masks = np.load(mpath) if os.path.exists(mpath) else np.ones([self.image_count, self.height, self.width, 1])
            self.masks = (masks > bin_thres).astype(np.int64) if bin_thres != -1 else masks.astype(np.float32)  # :66-69
        K = np.eye(3, dtype=np.float32)                                               # data/datasets.py:72-75
        K[0, 0] = K[1, 1] = self.meta_dict.get('focal', 0.0)
        K[0, -1] = self.meta_dict.get('W', self.width) / 2.
        K[1, -1] = self.meta_dict.get('H', self.height) / 2.
        self.K = K

    def num_images(self):
        return self.image_count

    def height_width(self):
        return self.height, self.width

    def near_far(self):
        return self.meta_dict['near'], self.meta_dict['far']

    def radii(self):
        return 2. / max(self.height, self.width) * 2 / math.sqrt(12)                  # data/datasets.py:114-115

    def view(self, i: int) -> dict:
        """One whole view as the eval loop consumes it (engines/eval.py:31-41): rays [2,H,W,3], target_s, masks."""
        out = {}
        if self.rays is not None:
            out['rays'] = torch.from_numpy(self.rays[i]).float().permute(2, 0, 1, 3)   # [H,W,2,3] -> [2,H,W,3]
        if self.rgbs is not None:
            out['target_s'] = torch.from_numpy(self.rgbs[i]).float()
        if self.masks is not None:
            out['masks'] = torch.from_numpy(self.masks[i])
        return out

    def rays_on_device(self, i: int, device, pix_range=None) -> torch.Tensor:
        """View i's rays generated on the GPU from its pose (K0) -- [2,H,W,3], or [2,n,3] for a flat pixel range (what a
        ray-sharded rank renders).  Equal bit for bit to the stored `rays_<split>.npy` of a reference-prepared scene."""
        if self.poses is None:
            raise IOError("poses_<split>.npy is missing (prepare the scene with --w_pose, data/gen_dataset.py:223-230)")
        mh, mw = self.meta_dict.get('H', self.height), self.meta_dict.get('W', self.width)
        if (int(mh), int(mw)) != (self.height, self.width):
            # `_x{subsample}` arrays: meta.json's focal / principal point describe the full-resolution camera, and the
            # reference's sub-sampled ray files are not a pure rescaling of K -- refuse rather than return wrong rays
            raise NotImplementedError(f"rays_on_device: the loaded arrays are {self.height}x{self.width} but meta.json "
                                      f"describes a {mh}x{mw} camera (subsample != 0); use the stored rays of this split")
        return ops.generate_rays(self.height, self.width, self.K, self.poses[i][:3, :4], device, pix_range=pix_range)
