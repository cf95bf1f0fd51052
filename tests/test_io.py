"""Reference on-disk formats and batch assembly around the path, pinned to files the REFERENCE wrote /
values the reference's own classes returned (tests/golden/make_goldens_io.py: `data/datasets.py`, `data/collater.py`,
`utils/ray.py`, `engines/trainer.py::save_checkpoint`, imported unmodified):

  tests/golden/io_scene/   a prepared scene in `data/gen_dataset.py:211-250`'s layout, rays from `get_persp_rays`
  tests/golden/io.npz      what BaseNeRFDataset / PatchNeRFDataset / RayNeRFDataset / ViewNeRFDataset / ExhibitNeRFDataset and
                           the collaters return for it (crop origins from `random.seed(7)`, pixels from `np.random.seed(11)`)
  tests/golden/io_ref.ckpt a checkpoint saved by the reference's `save_checkpoint` after one Adam step of its --fix_backbone recipe

CPU: the readers.  GPU (-m gpu): the device-side batch assembly (`DeviceScene`), bit for bit.
"""
import os
import random

import numpy as np
import pytest
import torch

import nerf_sos_amd
from nerf_sos_amd import io as nio
from helpers import state_sha

HERE = os.path.dirname(os.path.abspath(__file__))
SCENE = os.path.join(HERE, "golden", "io_scene")
SCENE44 = os.path.join(HERE, "golden", "io_scene44")      # poses_train.npy is [N,4,4] (blender / toydesk / tankstemple scenes)
CKPT = os.path.join(HERE, "golden", "io_ref.ckpt")
CROP, STRIDE = 8, 2


def test_prepared_scene_reader_vs_reference_dataset(golden):
    g = golden("io")
    for tag, thres in (("bin", 0.3), ("soft", -1)):
        sc = nio.PreparedScene(SCENE, split="train", bin_thres=thres)
        assert sc.masks.dtype == g[f"base_{tag}_masks"].dtype and np.array_equal(sc.masks, g[f"base_{tag}_masks"])   # data/datasets.py:66-69
        # utils/misc.py:7-14 (soft masks hold no exact 0 / 1: the reference's weights are NaN there, and so are these)
        assert np.array_equal(sc.class_weights().numpy(), g[f"patch_{tag}_class_w"], equal_nan=True)
    assert sc.K.dtype == np.float32 and np.array_equal(sc.K, g["base_K"])                                            # :72-75
    assert np.array_equal(sc.poses, g["base_poses"])
    n, h, w, near, far, radii = g["base_scalars"]
    assert (sc.num_images(), *sc.height_width(), *sc.near_far()) == (n, h, w, near, far) and sc.radii() == radii      # :114-115
    # whole views of the test split as the eval loop consumes them (engines/eval.py:31-41; data/datasets.py:154,224)
    st = nio.PreparedScene(SCENE, split="test")
    assert np.array_equal(torch.stack([st.view(i)["rays"] for i in range(st.num_images())]).numpy(), g["patch_test_rays"])
    v = st.view(1)
    for k in ("rays", "target_s", "masks"):
        assert v[k].dtype == torch.from_numpy(g[f"ray_test_item1_{k}"]).dtype and np.array_equal(v[k].numpy(), g[f"ray_test_item1_{k}"])
    ex = nio.PreparedScene(SCENE, split="exhibit", rgb=False, use_masks=False)
    assert ex.num_images() == int(g["exhibit_len"][0])
    assert np.array_equal(ex.view(2)["rays"].numpy(), g["exhibit_item2_rays"]) and np.array_equal(g["exhibit_item2_rays"], g["exhibit_collated_rays"])


def test_prepared_scene_with_4x4_poses_reads_like_the_reference(golden):
    g = golden("io44")
    sc = nio.PreparedScene(SCENE44, split="train", bin_thres=0.3)
    assert sc.poses.shape == (3, 4, 4)                                 # data/gen_dataset.py:228-233 saves poses[i_split] unsliced
    assert np.array_equal(sc.poses[g["order"]], g["batch_poses"])


def test_missing_meta_keys_raise(tmp_path):
    import json
    import shutil
    shutil.copytree(SCENE, tmp_path / "s")
    json.dump({"H": 12, "W": 16}, open(tmp_path / "s" / "meta.json", "w"))
    with pytest.raises(IOError):                                       # data/datasets.py:31-33
        nio.PreparedScene(str(tmp_path / "s"), split="test")


def test_crop_origins_are_the_reference_draws(golden):
    g = golden("io")
    random.seed(7)                                                     # make_goldens_io.py seeds `random` the same way
    got = nio.draw_patch_origins(6, 12, 16, CROP)                      # data/datasets.py:240-241
    assert np.array_equal(np.array(got, np.float32), g["patch_bin_batch_start_idx"])
    for k in range(6):
        assert np.array_equal(np.array(got[k], np.float32), g[f"patch_bin_item{k}_start_idx"])


def test_reference_checkpoint_loads(manifest):
    """A `.ckpt` written by the reference's save_checkpoint (engines/trainer.py:216-222) on the reference's NeRFNet."""
    man = manifest["io_ref_ckpt"]
    raw = torch.load(CKPT, map_location="cpu")
    assert set(raw) == {"global_step", "model", "optimizer"} and len(raw["model"]) == man["n_state_keys"]
    net = nerf_sos_amd.NeRFNet(N_samples=8, N_importance=0, use_semantics=True, sem_with_coord=True)
    assert list(net.state_dict()) == list(raw["model"])                # same keys in the same order
    for p in net.nerf.mlp.named_parameters():                          # run_nerf.py:313-318
        if "semantic_linear" not in p[0]:
            p[1].requires_grad = False
    opt = torch.optim.Adam(params=net.parameters(), lr=5e-4, betas=(0.9, 0.999))
    assert nio.load_checkpoint(CKPT, net, opt, strict=True) == man["global_step"]
    assert state_sha(net.state_dict()) == man["state_sha256"]
    assert abs(float(net.nerf.mlp.semantic_linear[0].weight.detach().double().sum()) - man["sem0_weight_sum"]) < 1e-12
    st = opt.state_dict()["state"]
    assert len(st) == man["optimizer_state_entries"] and all(int(v["step"]) == 1 for v in st.values())   # the one Adam step
    # our writer produces the same structure; round trip
    import tempfile
    with tempfile.TemporaryDirectory() as d:
        nio.save_checkpoint(os.path.join(d, "x.ckpt"), 7, net, opt)
        again = torch.load(os.path.join(d, "x.ckpt"), map_location="cpu")
    assert set(again) == set(raw) and list(again["model"]) == list(raw["model"])
    assert all(torch.equal(again["model"][k], raw["model"][k]) for k in raw["model"])
    assert again["optimizer"]["param_groups"] == raw["optimizer"]["param_groups"]
    # --load_nostrict: a backbone-only checkpoint into a model with semantic heads (scripts/train_*_node0.sh)
    b = nerf_sos_amd.NeRFNet(N_samples=8, N_importance=0, use_semantics=False)
    with pytest.raises(RuntimeError):
        nio.load_checkpoint(CKPT, b, strict=True)
    assert nio.load_checkpoint(CKPT, b, torch.optim.SGD(b.parameters(), lr=0.1), strict=False) == man["global_step"]
    assert torch.equal(b.nerf.mlp.pts_linears[3].weight, raw["model"]["nerf.mlp.pts_linears.3.weight"])


# ------------------------------------------------------------------------------------------------------------- GPU
def _eq(t, ref):
    t = t.cpu()
    return t.dtype == torch.from_numpy(ref).dtype and tuple(t.shape) == ref.shape and np.array_equal(t.numpy(), ref)


@pytest.mark.gpu
def test_rays_on_device_equal_the_reference_ray_files():
    for split in ("train", "test"):
        sc = nio.PreparedScene(SCENE, split=split)
        lean = nio.PreparedScene(SCENE, split=split, load_rays=False, rgb=False, use_masks=False)
        for i in range(sc.num_images()):
            got = lean.rays_on_device(i, "cuda:0").cpu().numpy()
            assert np.array_equal(got, sc.rays[i].transpose(2, 0, 1, 3))             # get_persp_rays, bit for bit
        part = lean.rays_on_device(1, "cuda:0", pix_range=(50, 170)).cpu().numpy()
        assert np.array_equal(part, sc.rays[1].transpose(2, 0, 1, 3).reshape(2, -1, 3)[:, 50:170])


@pytest.mark.gpu
@pytest.mark.parametrize("tag,thres", [("bin", 0.3), ("soft", -1)])
def test_device_patch_sampler_equals_the_reference_dataset_and_collater(golden, tag, thres):
    """PatchNeRFDataset.__getitem__ x 6 + PatchBatchCollater (data/datasets.py:240-254, data/collater.py:31-61) in one launch,
    rays from the poses: every tensor of the collated batch bit for bit, crop origins from the same `random` draws."""
    g = golden("io")
    ds = nio.PreparedScene(SCENE, split="train", bin_thres=thres, load_rays=False).to_device("cuda:0")
    order = g[f"patch_{tag}_order"].tolist()
    random.seed(7)
    b = ds.patch_batch(order, CROP, STRIDE)
    for key in ("rays", "target_s", "masks", "poses", "start_idx"):
        assert _eq(b[key], g[f"patch_{tag}_batch_{key}"]), key
    # the planar layout is the trainer's own reshape + permute of the collated rays (engines/trainer.py:63-64)
    want = torch.from_numpy(g[f"patch_{tag}_batch_rays"]).reshape(-1, 2, 3).permute(1, 0, 2)
    assert torch.equal(b["rays_planar"].reshape(2, -1, 3).cpu(), want)
    # explicit origins, and descriptors that already live on the device
    origins = [tuple(int(v) for v in g[f"patch_{tag}_item{k}_start_idx"]) for k in range(6)]
    b2 = ds.patch_batch(order, CROP, STRIDE, origins=origins)
    sel = torch.tensor([[i, h, w] for i, (h, w) in zip(order, origins)], dtype=torch.int32, device="cuda:0")
    b3 = ds.patch_batch(None, CROP, STRIDE, sel_device=sel)
    for key in ("rays", "target_s", "masks", "poses", "start_idx"):
        assert torch.equal(b2[key], b[key]) and torch.equal(b3[key], b[key]), key
    with pytest.raises(IndexError):
        ds.patch_batch([0], CROP, STRIDE, origins=[(5, 0)])                          # 5 > H - crop = 4
    with pytest.raises(IndexError):
        ds.patch_batch([4], CROP, STRIDE, origins=[(0, 0)])


@pytest.mark.gpu
def test_device_batches_of_a_scene_with_4x4_poses_equal_the_reference(golden):
    """A `poses_<split>.npy` of [N,4,4] (data/gen_dataset.py:228-233 for blender / toydesk / tankstemple): the image stride is 16
    floats, rays come from [:3,:4], and the collated `poses` is the reference's [B,4,4] (the stride was 12)."""
    g = golden("io44")
    sc = nio.PreparedScene(SCENE44, split="train", bin_thres=0.3)
    for i in range(sc.num_images()):
        assert np.array_equal(sc.rays_on_device(i, "cuda:0").cpu().numpy(), sc.rays[i].transpose(2, 0, 1, 3))
    ds = nio.PreparedScene(SCENE44, split="train", bin_thres=0.3, load_rays=False).to_device("cuda:0")
    random.seed(44)
    b = ds.patch_batch(g["order"].tolist(), 6, 2)
    for key in ("rays", "target_s", "masks", "poses", "start_idx"):
        assert _eq(b[key], g[f"batch_{key}"]), key
    r = ds.ray_batch(g["ray_picks"].tolist())
    for key in ("rays", "target_s", "masks"):
        assert _eq(r[key], g[f"ray_batch_{key}"]), key
    bad = nio.PreparedScene(SCENE44, split="train", load_rays=False)
    bad.poses = bad.poses[:, :2]
    with pytest.raises(ValueError):
        bad.to_device("cuda:0")


@pytest.mark.gpu
def test_device_ray_and_view_batches_equal_the_reference(golden):
    g = golden("io")
    ds = nio.PreparedScene(SCENE, split="train", bin_thres=0.3, load_rays=False).to_device("cuda:0")
    assert ds.image_count * ds.height * ds.width == int(g["ray_train_len"][0])
    picks = g["ray_train_picks"].tolist()
    for idx in (picks, torch.tensor(picks, device="cuda:0")):                       # RayNeRFDataset + RayBatchCollater
        b = ds.ray_batch(idx)
        for key in ("rays", "target_s", "masks"):
            assert _eq(b[key], g[f"ray_train_batch_{key}"]), key
    for tag, frac in (("full", None), ("precrop", 0.5)):                             # ViewNeRFDataset + ViewBatchCollater
        np.random.seed(11)
        b = ds.view_batch(2, 32, precrop_frac=frac)
        assert _eq(b["rays"], g[f"view_{tag}_rays"]) and _eq(b["target_s"], g[f"view_{tag}_target_s"]), tag


@pytest.mark.gpu
def test_device_patch_sampler_at_the_shipped_size(tmp_path):
    """The shipped recipe's shape (64x64 patches at stride 6 of 756x1008 views, scripts/train_flower_node0.sh:4-6) and a batch
    larger than one launch's by-value descriptor block (64): against numpy slicing of full ray images (C oracle's
    get_persp_rays, itself pinned to the reference by rays.npz) -- the strided-crop indexing of data/datasets.py:245-247."""
    import json
    from oracle import c_oracle as co
    N, H, W, focal = 3, 756, 1008, 850.0
    rng = np.random.default_rng(5)
    K = np.array([[focal, 0, W / 2], [0, focal, H / 2], [0, 0, 1]], np.float32)
    poses = np.concatenate([np.linalg.qr(rng.standard_normal((N, 3, 3)))[0], rng.standard_normal((N, 3, 1))], -1).astype(np.float32)
    rgbs = rng.random((N, H, W, 3), dtype=np.float32)
    masks = rng.random((N, H, W, 1), dtype=np.float32)
    np.save(tmp_path / "rgbs_train.npy", rgbs)
    np.save(tmp_path / "masks_train.npy", masks)
    np.save(tmp_path / "poses_train.npy", poses)
    json.dump({"H": H, "W": W, "focal": focal, "near": 1.2, "far": 14.72}, open(tmp_path / "meta.json", "w"))
    rays = np.stack([co.generate_rays(H, W, K, p) for p in poses], 0).transpose(0, 2, 3, 1, 4)     # [N,H,W,2,3]
    ds = nio.PreparedScene(str(tmp_path), split="train", load_rays=False).to_device("cuda:0")
    for B, P, stride in ((16, 64, 6), (70, 8, 3), (1, 1, 1)):
        crop = P * stride
        idx = rng.integers(0, N, B).tolist()
        random.seed(B)
        origins = nio.draw_patch_origins(B, H, W, crop)
        b = ds.patch_batch(idx, crop, stride, origins=origins)
        for k, (i, (h0, w0)) in enumerate(zip(idx, origins)):
            sl = (i, slice(h0, h0 + crop, stride), slice(w0, w0 + crop, stride))
            assert np.array_equal(b["rays"][k].cpu().numpy(), rays[sl].reshape(-1, 2, 3))
            assert np.array_equal(b["target_s"][k].cpu().numpy(), rgbs[sl].reshape(-1, 3))
            assert np.array_equal(b["masks"][k].cpu().numpy(), (masks[sl] > 0.3).astype(np.int64).reshape(-1, 1))
        assert np.array_equal(b["poses"].cpu().numpy(), poses[idx]) and np.array_equal(b["start_idx"].cpu().numpy(), np.array(origins, np.float32))
    # the sampled patch renders: what train_one_step does next (engines/trainer.py:68)
    net = nerf_sos_amd.NeRFNet(N_samples=16, N_importance=16, use_semantics=True, sem_with_coord=True).to("cuda:0").eval()
    b = ds.patch_batch([0, 2], 8 * 6, 6, origins=[(10, 20), (300, 500)])
    with torch.no_grad():
        out = net(b["rays_planar"], (1.2, 14.72), retraw=False)
    assert out["rgb"].shape == (2, 8, 8, 3) and torch.isfinite(out["rgb"]).all()
