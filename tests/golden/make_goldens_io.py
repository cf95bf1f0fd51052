#!/usr/bin/env python3
"""Golden vectors for the data formats either side of the render path: prepared scenes as the
reference's dataset classes read them, the patch-crop sampler, the collaters, and a checkpoint written by the
reference.  Run in the BUILD CONTAINER only (needs /root/reference):

    python tests/golden/make_goldens_io.py

What comes from the REAL reference here (imported unmodified from /root/reference):
  * the rays of the tiny scene: `utils/ray.py::get_persp_rays` / `get_persp_intrinsic`, laid out and saved exactly as
    `data/gen_dataset.py:189-250` does (the generator itself reads images through `imageio`, which this image lacks, so
    the loaders cannot run; the layout lines are followed by hand and cited below);
  * everything recorded as an expected value: `data/datasets.py` (`BaseNeRFDataset`, `RayNeRFDataset`,
    `PatchNeRFDataset` -- incl. its random strided crop :240-254 --, `ViewNeRFDataset`, `ExhibitNeRFDataset`),
    `data/collater.py` (all four collaters), `utils/misc.py::weights_log`;
  * the checkpoint: `engines/trainer.py::save_checkpoint` (:216-222) on a reference `NeRFNet` after one Adam step of the
    shipped --fix_backbone recipe (run_nerf.py:307-321).

Stubbed IN MEMORY (none of it is arithmetic of the path; same trick as make_goldens_losses.py): the missing third-party
modules `cv2`, `imageio`, `configargparse`, `sacrebleu`, `lpips`, `torchvision`, `torch.utils.tensorboard` (imported at
module level by the files above, used only by code not exercised here; `cv2.imwrite` -- RayNeRFDataset's debug dump
`logs/rgb.png`, data/datasets.py:143-146 -- becomes a no-op), `np.long` (removed in numpy 1.24; data/datasets.py:67 uses it)
aliased to `np.int64`, and `Tensor.cuda` (data/datasets.py:76 puts K on the GPU in the dataset ctor; there is no GPU here).

Written: tests/golden/io_scene/ (the scene directory: data files), tests/golden/io.npz (expected outputs),
tests/golden/io_scene44/ + io44.npz (a scene whose poses file is [N,4,4], as the blender / toydesk / tankstemple loaders leave it),
tests/golden/io_ref.ckpt (the reference-format checkpoint: a dict of tensors).  Only data is written.
"""
import json
import os
import random
import sys
import types

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
REF = os.environ.get("NERF_SOS_REFERENCE", "/root/reference")

imwrites = []
for name in ("cv2", "imageio", "configargparse", "sacrebleu", "lpips", "torchvision"):
    sys.modules[name] = types.ModuleType(name)
sys.modules["cv2"].imwrite = lambda path, img: imwrites.append(path)
sys.modules["sacrebleu"].dataset = None
sys.modules["lpips"].LPIPS = lambda *a, **k: None
sys.modules["torchvision"].transforms = types.ModuleType("torchvision.transforms")
sys.modules["torchvision.transforms"] = sys.modules["torchvision"].transforms
tb = types.ModuleType("torch.utils.tensorboard")
tb.SummaryWriter = object
sys.modules["torch.utils.tensorboard"] = tb
np.long = np.int64
torch.Tensor.cuda = lambda self, *a, **k: self

sys.path.insert(0, REF)
from utils.ray import get_persp_rays, get_persp_intrinsic  # noqa: E402  (reference)
import data.datasets as ref_ds  # noqa: E402  (reference)
import data.collater as ref_col  # noqa: E402  (reference)
from models.nerf_net import NeRFNet as RefNeRFNet  # noqa: E402  (reference)
from engines.trainer import save_checkpoint as ref_save_checkpoint  # noqa: E402  (reference)
torch.autograd.set_detect_anomaly(False)   # models/sampler.py:2 turns it on process-wide

SCENE = os.path.join(HERE, "io_scene")
N, H, W, FOCAL = 6, 12, 16, 14.0
NEAR, FAR = 1.2, 14.72
CROP, STRIDE = 8, 2            # PatchNeRFDataset(crop_size = patch_size * patch_stride = 4 * 2, patch_stride = 2), run_nerf.py:407-408


def write_scene():
    """data/gen_dataset.py:181-250 for an LLFF-type scene (poses [N,3,5], --w_pose), by hand: the layout, file names and
    meta keys of the generator; rays from the reference's own get_persp_rays."""
    os.makedirs(SCENE, exist_ok=True)
    rng = np.random.default_rng(20260928)
    rot = np.linalg.qr(rng.standard_normal((N, 3, 3)))[0]
    trans = rng.standard_normal((N, 3, 1)) * 0.5
    hwf = np.tile(np.array([H, W, FOCAL], np.float64).reshape(1, 3, 1), (N, 1, 1))
    poses = np.concatenate([rot, trans, hwf], -1).astype(np.float32)                  # [N,3,5] as load_llff returns
    images = rng.random((N, H, W, 3), dtype=np.float32)
    masks = rng.random((N, H, W, 1), dtype=np.float32)
    K = get_persp_intrinsic(H, W, FOCAL)                                              # gen_dataset.py:184-185
    rays = torch.stack([get_persp_rays(H, W, K, torch.tensor(p)) for p in poses[:, :3, :4]], 0)   # :189
    rays = rays.permute([0, 2, 3, 1, 4]).numpy().astype(np.float32)                   # :190  [N,H,W,ro+rd,3]
    i_test = np.array([0, 3])
    i_val = i_test                                                                    # gen_dataset.py (llff): i_val = i_test
    i_train = np.array([i for i in range(N) if i not in i_test])
    for split, idx in (("train", i_train), ("val", i_val), ("test", i_test)):
        np.save(os.path.join(SCENE, f"rays_{split}.npy"), rays[idx])                  # :212-222
        np.save(os.path.join(SCENE, f"rgbs_{split}.npy"), images[idx])
        np.save(os.path.join(SCENE, f"masks_{split}.npy"), masks[idx])
        np.save(os.path.join(SCENE, f"poses_{split}.npy"), poses[idx])                # :224-232 (--w_pose)
    np.save(os.path.join(SCENE, "rays_exhibit.npy"), rays[i_train])                   # :199-202: render_poses None -> train poses
    meta = {"H": H, "W": W, "focal": float(FOCAL), "near": float(NEAR), "far": float(FAR),
            "i_train": i_train.tolist(), "i_val": i_val.tolist(), "i_test": i_test.tolist(),
            "ndc": False, "factor": 8, "spherify": False, "llffhold": 3,
            "half_res": False, "white_bkgd": False, "test_skip": 1, "dv_scene": "cube"}   # :235-245
    with open(os.path.join(SCENE, "meta.json"), "w") as f:
        json.dump(meta, f)


SCENE44 = os.path.join(HERE, "io_scene44")
N44, H44, W44 = 3, 10, 12


def write_scene44():
    """A blender / toydesk / tankstemple-type scene: `poses_<split>.npy` is [N,4,4] -- data/gen_dataset.py:228-233 saves
    `poses[i_split]` unsliced, and load_blender / load_toydesk return 4x4 camera-to-world matrices.  Train split only."""
    os.makedirs(SCENE44, exist_ok=True)
    rng = np.random.default_rng(44)
    poses = np.zeros((N44, 4, 4), np.float32)
    poses[:, :3, :3] = np.linalg.qr(rng.standard_normal((N44, 3, 3)))[0]
    poses[:, :3, 3] = rng.standard_normal((N44, 3)) * 0.5
    poses[:, 3, 3] = 1.0
    images = rng.random((N44, H44, W44, 3), dtype=np.float32)
    masks = rng.random((N44, H44, W44, 1), dtype=np.float32)
    K = get_persp_intrinsic(H44, W44, FOCAL)
    rays = torch.stack([get_persp_rays(H44, W44, K, torch.tensor(p)) for p in poses[:, :3, :4]], 0)   # gen_dataset.py:189
    rays = rays.permute([0, 2, 3, 1, 4]).numpy().astype(np.float32)
    np.save(os.path.join(SCENE44, "rays_train.npy"), rays)
    np.save(os.path.join(SCENE44, "rgbs_train.npy"), images)
    np.save(os.path.join(SCENE44, "masks_train.npy"), masks)
    np.save(os.path.join(SCENE44, "poses_train.npy"), poses)                          # :228-233: [N,4,4]
    meta = {"H": H44, "W": W44, "focal": float(FOCAL), "near": 2.0, "far": 6.0, "i_train": list(range(N44)), "i_val": [], "i_test": [],
            "ndc": False, "half_res": False, "white_bkgd": True, "test_skip": 1}
    with open(os.path.join(SCENE44, "meta.json"), "w") as f:
        json.dump(meta, f)


def goldens44():
    """What the reference's PatchNeRFDataset + PatchBatchCollater return for the 4x4-pose scene (its own file, io44.npz)."""
    write_scene44()
    out, args = {}, types.SimpleNamespace()
    ds = ref_ds.PatchNeRFDataset(SCENE44, args, split="train", cam_id=False, use_masks=True, crop_size=6, patch_stride=2,
                                 bin_thres=0.3, ret_k=True)
    assert ds.poses.shape == (N44, 4, 4)
    order = [2, 0, 1, 1, 2]
    random.seed(44)
    items = [ds[i] for i in order]
    out["order"] = np.array(order)
    col = ref_col.PatchBatchCollater()(items)
    for key, v in zip(("rays", "target_s", "masks", "poses", "start_idx"), col):
        out[f"batch_{key}"] = t2n(v)
    assert out["batch_poses"].shape == (5, 4, 4)
    rt = ref_ds.RayNeRFDataset(SCENE44, args, split="train", use_masks=True, bin_thres=0.3)
    picks = [0, 119, 120, 200, len(rt) - 1]                                           # pixels of image 0, 0, 1, 1, 2
    out["ray_picks"] = np.array(picks)
    col = ref_col.RayBatchCollater()([rt[i] for i in picks])
    for key, v in zip(("rays", "target_s", "masks"), col):
        out[f"ray_batch_{key}"] = t2n(v)
    np.savez_compressed(os.path.join(HERE, "io44.npz"), **out)
    print(f"wrote io44.npz ({len(out)} arrays), io_scene44/")


def t2n(t):
    return t.detach().cpu().numpy() if isinstance(t, torch.Tensor) else np.asarray(t)


def main():
    write_scene()
    out = {}
    args = types.SimpleNamespace()

    # ---- base reader: masks thresholding :66-69, K :72-75, poses, radii :114-115
    for tag, thres in (("bin", 0.3), ("soft", -1)):
        b = ref_ds.BaseNeRFDataset(SCENE, args, split="train", use_masks=True, bin_thres=thres, ret_k=True)
        out[f"base_{tag}_masks"] = b.masks
        assert b.masks.dtype == (np.int64 if thres != -1 else np.float32)
    out["base_K"] = t2n(b.K)
    out["base_poses"] = b.poses
    out["base_scalars"] = np.array([b.num_images(), *b.height_width(), *b.near_far(), b.radii()], np.float64)
    nok = ref_ds.BaseNeRFDataset(SCENE, args, split="train", use_masks=True, ret_k=False)
    out["base_poses_without_ret_k"] = nok.poses                                       # zeros [N,3,4], :85

    # ---- PatchNeRFDataset: the training sampler of the shipped recipe (run_nerf.py:406-409), random strided crops :240-254
    for tag, thres in (("bin", 0.3), ("soft", -1)):
        ds = ref_ds.PatchNeRFDataset(SCENE, args, split="train", cam_id=False, use_masks=True, crop_size=CROP,
                                     patch_stride=STRIDE, bin_thres=thres, ret_k=True)
        out[f"patch_{tag}_class_w"] = t2n(ds.class_w)
        order = [2, 0, 3, 1, 1, 2]                                                    # image index per drawn item
        random.seed(7)                                                                # the crop origins come from `random.randint`
        items = [ds[i] for i in order]
        out[f"patch_{tag}_order"] = np.array(order)
        for k, it in enumerate(items):
            for key, v in it.items():
                out[f"patch_{tag}_item{k}_{key}"] = t2n(v)
        col = ref_col.PatchBatchCollater()(items)
        for key, v in zip(("rays", "target_s", "masks", "poses", "start_idx"), col):
            out[f"patch_{tag}_batch_{key}"] = t2n(v)
    assert len(ds) == 4
    # test split of the same class: whole views with the ro/rd axis first (:224), cropped the same way
    dt = ref_ds.PatchNeRFDataset(SCENE, args, split="test", crop_size=CROP, patch_stride=STRIDE, bin_thres=0.3, ret_k=True)
    out["patch_test_rays"] = t2n(dt.rays)                                             # [N,2,H,W,3]

    # ---- RayNeRFDataset (ray batching): flattening :149-152, test split permuted :154
    rt = ref_ds.RayNeRFDataset(SCENE, args, split="train", use_masks=True, bin_thres=0.3)
    out["ray_train_len"] = np.array([len(rt)])
    picks = [0, 37, 191, 500, len(rt) - 1]
    items = [rt[i] for i in picks]
    out["ray_train_picks"] = np.array(picks)
    col = ref_col.RayBatchCollater()(items)
    for key, v in zip(("rays", "target_s", "masks"), col):
        out[f"ray_train_batch_{key}"] = t2n(v)
    re_ = ref_ds.RayNeRFDataset(SCENE, args, split="test", use_masks=True, bin_thres=0.3)
    it = re_[1]
    for key, v in it.items():
        out[f"ray_test_item1_{key}"] = t2n(v)                                         # engines/eval.py consumes these whole views
    out["ray_class_w"] = t2n(rt.class_w)
    assert imwrites == ["logs/rgb.png", "logs/msk.png"] * 2

    # ---- ViewNeRFDataset (--no_batching): np.random.choice of N_rand pixels of one view :279-300, with and without pre-crop
    for tag, kw in (("full", dict(precrop_iters=0)), ("precrop", dict(precrop_iters=10, precrop_frac=0.5))):
        vd = ref_ds.ViewNeRFDataset(SCENE, 32, args, split="train", **kw)
        np.random.seed(11)
        it = vd[2]
        col = ref_col.ViewBatchCollater()([it])
        out[f"view_{tag}_rays"] = t2n(col[0])
        out[f"view_{tag}_target_s"] = t2n(col[1])

    # ---- ExhibitNeRFDataset + ExhibitCollater
    ex = ref_ds.ExhibitNeRFDataset(SCENE, args)
    out["exhibit_len"] = np.array([len(ex)])
    out["exhibit_item2_rays"] = t2n(ex[2]["rays"])
    col = ref_col.ExhibitCollater(H, W)([{"rays": ex[2]["rays"].reshape(2, -1, 3)}])       # unused by the reference; one view of [2,H*W,3]
    out["exhibit_collated_rays"] = t2n(col[0])

    np.savez_compressed(os.path.join(HERE, "io.npz"), **out)

    # ---- a checkpoint as the reference writes it, after one optimizer step of the shipped recipe
    torch.manual_seed(0)
    net = RefNeRFNet(N_samples=8, N_importance=0, use_semantics=True, sem_with_coord=True, perturb=1.0, raw_noise_std=1.0)
    for p in net.nerf.mlp.named_parameters():                                        # run_nerf.py:313-318 (--fix_backbone)
        if "semantic_linear" not in p[0]:
            p[1].requires_grad = False
    opt = torch.optim.Adam(params=net.parameters(), lr=5e-4, betas=(0.9, 0.999))      # run_nerf.py:320
    rays = torch.from_numpy(np.load(os.path.join(SCENE, "rays_train.npy"))[0, ::4, ::4]).reshape(-1, 2, 3).permute(1, 0, 2)
    torch.manual_seed(1)
    ret = net(rays, (NEAR, FAR))
    ret["semantics"].square().mean().backward()
    opt.step()
    path = os.path.join(HERE, "io_ref.ckpt")
    ref_save_checkpoint(path, 150000, net, opt)                                       # engines/trainer.py:216-222
    import hashlib
    h = hashlib.sha256()                                                              # tests/helpers.py::state_sha
    sd = net.state_dict()
    for k in sorted(sd):
        h.update(sd[k].detach().cpu().numpy().astype("<f4").tobytes())
    man_path = os.path.join(HERE, "manifest.json")
    man = json.load(open(man_path))
    man["io_ref_ckpt"] = {"state_sha256": h.hexdigest(), "global_step": 150000, "n_state_keys": len(net.state_dict()),
                          "optimizer_state_entries": len(opt.state_dict()["state"]),
                          "sem0_weight_sum": float(net.nerf.mlp.semantic_linear[0].weight.double().sum())}
    json.dump(man, open(man_path, "w"), indent=1)
    print(f"wrote io.npz ({len(out)} arrays), io_scene/, io_ref.ckpt ({os.path.getsize(path)} bytes)")
    goldens44()


if __name__ == "__main__":
    main()
