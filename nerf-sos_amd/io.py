"""Readers / writers of the reference's on-disk formats around the render path.

* checkpoints: `torch.save({'global_step', 'model', 'optimizer'})` (engines/trainer.py:216-222), re-loaded with
  `strict = not args.load_nostrict` and a tolerated optimizer mismatch (run_nerf.py:344-360);
* prepared scenes: the directory `data/gen_dataset.py:211-250` writes -- `meta.json` (H, W, focal, near, far, ...),
  `rays_<split>.npy [N,H,W,2,3]`, `rgbs_<split>.npy [N,H,W,3]`, `masks_<split>.npy [N,H,W,1]`, `poses_<split>.npy
  [N,3,5 or 3,4]` -- as `data/datasets.py:20-115` reads them.

`PreparedScene.rays_on_device` does not read the ray file at all: it regenerates a view's rays from its pose with the
K0 kernel (bit-identical to `utils/ray.py:12-22`), which removes the `[N,H,W,2,3]` tensors from disk and PCIe.

`DeviceScene` is the training-side counterpart: a split's images, masks and poses resident in HBM, and the batches the
reference's dataset classes + collaters hand to `train_one_step` gathered by one kernel launch per step --
`patch_batch` = `PatchNeRFDataset.__getitem__` (random strided crops, data/datasets.py:240-254) + `PatchBatchCollater`,
`ray_batch` = `RayNeRFDataset` + `RayBatchCollater`, `view_batch` = `ViewNeRFDataset` + `ViewBatchCollater` -- with the
rays generated from the poses (no ray file).  Pinned against the real classes by tests/golden/io.npz
(tests/golden/make_goldens_io.py).
"""
from __future__ import annotations

import ctypes as C
import json
import math
import os
import random
from typing import Optional

import numpy as np
import torch

from . import _lib, ops


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

    def class_weights(self) -> torch.Tensor:
        """`weights_log(masks)` (utils/misc.py:7-14; data/datasets.py:139,205): balanced weights of the two mask classes."""
        m = torch.from_numpy(self.masks) if self.masks is not None else torch.zeros(1)
        freq = torch.Tensor([torch.sum(m == 0), torch.sum(m == 1)])
        w = 1 / torch.log1p(freq)
        return len(freq) * w / torch.sum(w)

    def to_device(self, device) -> "DeviceScene":
        return DeviceScene(self, device)


def draw_patch_origins(n_items: int, height: int, width: int, crop_size: int):
    """The crop origins of `n_items` consecutive `PatchNeRFDataset.__getitem__` calls (data/datasets.py:240-241): Python's
    global `random.randint(0, H - crop)` then `random.randint(0, W - crop)` per item, in that order -- after
    `random.seed(s)` these are the reference's own crops."""
    return [(random.randint(0, height - crop_size), random.randint(0, width - crop_size)) for _ in range(n_items)]


class DeviceScene:
    """One split of a prepared scene resident in device memory (uploaded once), batches gathered on the device."""

    def __init__(self, scene: PreparedScene, device):
        dev = torch.device(device)
        if dev.type != "cuda":
            raise RuntimeError("nerf_sos_amd: `device` must be a GPU -- this package has no CPU path")
        if scene.poses is None:
            raise IOError("DeviceScene generates rays from poses: poses_<split>.npy is missing (prepare the scene with "
                          "--w_pose, data/gen_dataset.py:223-230)")
        mh, mw = scene.meta_dict.get('H', scene.height), scene.meta_dict.get('W', scene.width)
        if (int(mh), int(mw)) != (scene.height, scene.width):
            raise NotImplementedError("DeviceScene: sub-sampled splits are not the camera meta.json describes (see rays_on_device)")
        self.scene, self.device = scene, dev
        self.height, self.width, self.image_count = scene.height, scene.width, scene.image_count
        # poses_<split>.npy as data/gen_dataset.py:228-233 wrote it: LLFF [N,3,5], blender / toydesk / tankstemple [N,4,4]
        # (unsliced); the kernels take the image stride rows*cols and read [:3,:4]
        if scene.poses.ndim != 3 or scene.poses.shape[1] not in (3, 4) or scene.poses.shape[2] < 4:
            raise ValueError(f"DeviceScene: poses must be [N,3,>=4] or [N,4,>=4], got {tuple(scene.poses.shape)}")
        self.poses = torch.from_numpy(np.ascontiguousarray(scene.poses)).float().to(dev)
        self.rgbs = torch.from_numpy(scene.rgbs).float().to(dev).contiguous() if scene.rgbs is not None else None
        self.masks = torch.from_numpy(scene.masks).to(dev).contiguous() if scene.masks is not None else None
        if self.masks is not None and self.masks.dtype not in (torch.int64, torch.float32):
            raise TypeError(f"DeviceScene: masks are int64 labels or float32, got {self.masks.dtype}")

    # ------------------------------------------------------------------------------------------
    def _source_args(self):
        K = self.scene.K
        m = self.masks
        words = 0 if m is None else int(m.shape[-1]) * (2 if m.dtype == torch.int64 else 1)
        return (self.height, self.width, float(K[0, 0]), float(K[1, 1]), float(K[0, 2]), float(K[1, 2]),
                ops._p(self.poses), int(self.poses.shape[-2]), int(self.poses.shape[-1]), self.image_count,
                ops._p(self.rgbs), 0 if self.rgbs is None else int(self.rgbs.shape[-1]), ops._p(m), words)

    def _outputs(self, lead):
        dev = self.device
        rays = torch.empty((2,) + lead + (3,), device=dev, dtype=torch.float32)
        tgt = torch.empty(lead + (self.rgbs.shape[-1],), device=dev, dtype=torch.float32) if self.rgbs is not None else None
        msk = torch.empty(lead + (self.masks.shape[-1],), device=dev, dtype=self.masks.dtype) if self.masks is not None else None
        return rays, tgt, msk

    def patch_batch(self, image_indices, crop_size: int, patch_stride: int = 1, origins=None,
                    sel_device: Optional[torch.Tensor] = None) -> dict:
        """B = len(image_indices) items of the reference's PatchNeRFDataset (crop_size = patch_size * patch_stride,
        run_nerf.py:407-408) collated by PatchBatchCollater, in ONE launch:
          rays [B,P*P,2,3] (a view of `rays_planar` [2,B,P,P,3], the layout NeRFNet / sharded_patch_step consume directly:
          the trainer's reshape + permute, engines/trainer.py:63-64, is already done), target_s [B,P*P,3],
          masks [B,P*P,1], poses [B,3,5] or [B,4,4] (= self.poses[i]), start_idx [B,2] float32 -- P = ceil(crop_size / patch_stride).
        `origins`: [(h_idx, w_idx)] per item; None draws them like the reference (`draw_patch_origins`).
        `sel_device`: int32 [B,3] device tensor of (image, h_idx, w_idx) instead (nothing crosses PCIe; graph capture)."""
        P = -(-int(crop_size) // int(patch_stride))
        with torch.cuda.device(self.device):
            if sel_device is not None:
                if sel_device.dtype != torch.int32 or not sel_device.is_cuda or sel_device.dim() != 2 or sel_device.shape[1] != 3:
                    raise TypeError("sel_device must be an int32 [B,3] GPU tensor of (image, h_idx, w_idx)")
                sel_device = sel_device.contiguous()
                B = int(sel_device.shape[0])
                host = None
            else:
                idx = [int(i) for i in image_indices]
                B = len(idx)
                if origins is None:
                    origins = draw_patch_origins(B, self.height, self.width, int(crop_size))
                if len(origins) != B:
                    raise ValueError(f"{len(origins)} origins for {B} items")
                for i, (h0, w0) in zip(idx, origins):   # data/datasets.py:240-241
                    if not (0 <= i < self.image_count and 0 <= h0 <= self.height - crop_size and 0 <= w0 <= self.width - crop_size):
                        raise IndexError(f"patch (image {i}, origin {h0},{w0}, crop {crop_size}) outside the {self.height}x{self.width} images")
                host = (C.c_int32 * (3 * B))(*[v for i, (h0, w0) in zip(idx, origins) for v in (i, int(h0), int(w0))])
            rays, tgt, msk = self._outputs((B, P * P))
            poses = torch.empty((B,) + tuple(self.poses.shape[1:]), device=self.device, dtype=torch.float32)
            start = torch.empty((B, 2), device=self.device, dtype=torch.float32)
            _lib.check(_lib.lib().nsos_patch_batch(*self._source_args(), host, ops._p(sel_device), B, P, int(patch_stride),
                                                   ops._p(rays[0]), ops._p(rays[1]), ops._p(tgt), ops._p(msk), ops._p(poses),
                                                   ops._p(start), ops._stream()), "nsos_patch_batch")
        out = {"rays": rays.permute(1, 2, 0, 3), "rays_planar": rays.reshape(2, B, P, P, 3), "poses": poses, "start_idx": start}
        if tgt is not None:
            out["target_s"] = tgt
        if msk is not None:
            out["masks"] = msk
        return out

    def pixel_batch(self, pix: torch.Tensor) -> dict:
        """Records of an explicit list of flat pixel indices (image*H + y)*W + x (int64 device tensor, any shape):
        rays [2, *shape, 3], target_s [*shape, 3], masks [*shape, 1]."""
        if pix.dtype != torch.int64 or not pix.is_cuda:
            raise TypeError("pix must be an int64 GPU tensor of flat pixel indices")
        pix = pix.contiguous()
        with torch.cuda.device(self.device):
            rays, tgt, msk = self._outputs(tuple(pix.shape))
            _lib.check(_lib.lib().nsos_pixel_batch(*self._source_args(), ops._p(pix), pix.numel(), ops._p(rays[0]), ops._p(rays[1]),
                                                   ops._p(tgt), ops._p(msk), ops._stream()), "nsos_pixel_batch")
        out = {"rays": rays}
        if tgt is not None:
            out["target_s"] = tgt
        if msk is not None:
            out["masks"] = msk
        return out

    def ray_batch(self, indices) -> dict:
        """RayNeRFDataset items `indices` of the flattened [N*H*W] training set (data/datasets.py:149-152,159-171) collated by
        RayBatchCollater (data/collater.py:7-29): rays [2,B,3], target_s [B,3], masks [B,1].  `indices`: a host sequence
        (one small upload) or an int64 device tensor (e.g. `torch.randperm(n, device=...)[:B]`: nothing crosses PCIe)."""
        if isinstance(indices, torch.Tensor) and indices.is_cuda:
            return self.pixel_batch(indices.long())
        return self.pixel_batch(torch.as_tensor(list(indices), dtype=torch.int64).to(self.device))

    def view_batch(self, i: int, n_rand: int, precrop_frac: Optional[float] = None) -> dict:
        """ViewNeRFDataset.__getitem__(i) (--no_batching; data/datasets.py:272-300) + ViewBatchCollater: `n_rand` pixels of
        view i chosen by `np.random.choice(..., replace=False)` from the whole view, or from the centre crop while
        pre-cropping (`precrop_frac`, :282-289) -- the reference's global numpy generator, same draws after the same seed."""
        H, W = self.height, self.width
        if precrop_frac is not None:
            dH, dW = int(H // 2 * precrop_frac), int(W // 2 * precrop_frac)
            ys = np.arange(H // 2 - dH, H // 2 + dH)             # torch.linspace(H//2 - dH, H//2 + dH - 1, 2*dH) -> .long()
            xs = np.arange(W // 2 - dW, W // 2 + dW)
        else:
            ys, xs = np.arange(H), np.arange(W)
        sel = np.random.choice(len(ys) * len(xs), size=[int(n_rand)], replace=False)                 # :291
        flat = (int(i) * H + ys[sel // len(xs)]) * W + xs[sel % len(xs)]
        return self.pixel_batch(torch.from_numpy(flat.astype(np.int64)).to(self.device))
