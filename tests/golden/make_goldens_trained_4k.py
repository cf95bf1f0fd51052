#!/usr/bin/env python3
"""The REAL reference on the TRAINED field at sizes that see the 16-bit TAIL (VERDICT r05 #1a, missing-2, weak-1).

Run in the BUILD CONTAINER only (needs /root/reference):

    python tests/golden/make_goldens_trained_4k.py

`trained.npz` (make_goldens_trained.py) holds 256 rays -- enough to pin fp32 parity, too few to see a one-in-a-thousand ray.  This
script renders, with the unmodified reference `NeRFNet` (`load_state_dict(strict=True)` of tests/golden/trained_scene.ckpt, eval mode,
models/nerf_net.py:132-195):

  * `trained_4k.npz`   -- 4096 rays: 1024 seeded random pixels of each of the four held-out views at the scene's own resolution --
                          EXACTLY the set `bench.trained_field_parity` renders (same generator seed, same order), so the driver's
                          `parity.trained_field.vs_reference` and tests/test_gpu_trained.py read the same rays;
  * `trained_img64k.npz` -- 65 536 seeded random pixels of the full 1008x756 image of held-out pose 0 (C5's workload on this field,
                          `bench.c5_trained_quality`): the reference's maps + the pixel indices (the rays are regenerated on the
                          device by K0, which is bit-identical to utils/ray.py:12-22; their sha256 is stored and asserted).

Both: every image map of both passes (rgb, depth, acc, disp, semantics, z_std and the coarse ones), not `raw` / `weights` (18 MB per
4096 rays).  The port is asserted equal to the reference bit for bit on both sets.  Also stored for the 4096-ray set: the reference's
own sensitivity yardsticks (N_self for the image maps and for z_std: the port with its coarse network in fp64, rounded once).
"""
import hashlib
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import make_goldens as mg  # noqa: E402  (sets up the reference import; does not run its main)
from oracle import torch_port as tp  # noqa: E402
from utils.ray import get_persp_rays, get_persp_intrinsic  # noqa: E402  (reference)
import nerf_sos_amd  # noqa: E402,F401
from nerf_sos_amd.synthetic import ProceduralScene, H as IMG_H, W as IMG_W  # noqa: E402

CKPT = os.path.join(HERE, "trained_scene.ckpt")
MAPS = ("rgb", "depth", "acc", "disp", "semantics", "z_std", "rgb0", "depth0", "acc0", "disp0", "semantics0")


def reference_and_port(net, sd, pc, rays, bounds, what):
    with torch.no_grad():
        ret = net(rays, bounds, radii=None)
        port = tp.render(sd, pc, rays, bounds)
    assert set(ret) == set(port)
    for k in ret:
        mg.same(ret[k], port[k], f"{what} {k}")
    return ret


def main():
    torch.set_num_threads(min(32, os.cpu_count() or 1))
    ckpt = torch.load(CKPT, map_location="cpu")
    sd = ckpt["model"]
    net = mg.NeRFNet(N_samples=64, N_importance=128, perturb=1.0, raw_noise_std=1.0, pts_chuck=1024 * 256,
                     use_semantics=True, sem_with_coord=True)
    net.load_state_dict(sd, strict=True)                                   # run_nerf.py:349-353
    net.eval()
    pc = tp.PortConfig(n_importance=128, use_semantics=True, sem_with_coord=True, pts_chunk=1024 * 256)
    scene = ProceduralScene()
    bounds = (scene.NEAR, scene.FAR)
    common = {"near_far": np.array(bounds, np.float64),
              "ckpt_sha256": np.frombuffer(hashlib.sha256(open(CKPT, "rb").read()).digest(), np.uint8)}

    # ---- 4096 rays = bench.trained_field_parity's set (generator seed 0, one randperm per held-out view, first 1024)
    K = get_persp_intrinsic(scene.h, scene.w, scene.focal)
    g = torch.Generator().manual_seed(0)
    per_view = 4096 // len(scene.i_test)
    rays, pix, gt_rgb, gt_lab, gt_t = [], [], [], [], []
    for i in scene.i_test:
        sel = torch.randperm(scene.h * scene.w, generator=g)[:per_view]
        full = get_persp_rays(scene.h, scene.w, K, torch.tensor(scene.poses[i, :3, :4])).reshape(2, -1, 3)
        r = full[:, sel]
        rays.append(r)
        pix.append(torch.stack([torch.full_like(sel, i), sel], -1))
        c, lab, t = scene.trace(r[0].double().numpy(), r[1].double().numpy())
        gt_rgb.append(c), gt_lab.append(lab), gt_t.append(t)
    rays = torch.cat(rays, 1).contiguous()
    ret = reference_and_port(net, sd, pc, rays, bounds, "trained 4k")
    out = dict(common, rays=mg.np32(rays), pixels=torch.cat(pix).numpy().astype(np.int32), gt_rgb=np.concatenate(gt_rgb),
               gt_label=np.concatenate(gt_lab).astype(np.int8), gt_t=np.concatenate(gt_t))
    for k in MAPS:
        out[f"eval_{k}"] = mg.np32(ret[k])
    psnr = -10 * np.log10(np.mean((out["eval_rgb"] - out["gt_rgb"]) ** 2))
    print(f"4096 rays: reference PSNR vs analytic GT {psnr:.2f} dB, mean acc {out['eval_acc'].mean():.4f}")
    # yardsticks (reference vs its own fp64-coarse variant): image maps via bench.parity_yardstick, z_std as in make_goldens_trained.py
    sys.path.insert(0, mg.ROOT)
    import bench
    full_ret = {k: v for k, v in ret.items()}
    y = bench.parity_yardstick(tp, sd, pc, rays, full_ret, full_ret)
    assert y["staged_port_reproduces_port"] and y["gpu_rays_outside"] == 0
    out["n_self"] = np.array([y["reference_self_sensitivity_rays_outside"]])
    viewdirs = rays[1] / torch.norm(rays[1], dim=-1, keepdim=True)
    nearv, farv = torch.full((rays.shape[1], 1), bounds[0]), torch.full((rays.shape[1], 1), bounds[1])
    sd64 = {k: v.double() for k, v in sd.items()}
    with torch.no_grad():
        z = tp.stratified_z(nearv, farv, 64, None)
        pts = tp.ray_points(rays[0], rays[1], z)
        raw64 = tp.point_query(sd64, "nerf", pts.double(), viewdirs.double()[..., None, :].expand(pts.shape), pc).float()
        w64 = tp.composite(raw64, z, rays[1], None, pc)["weights"]
        zstd64 = torch.std(tp.importance_z(z, w64, 128, None)[1], dim=-1, unbiased=False).numpy()
    want = out["eval_z_std"].reshape(-1)
    out["n_self_z_std"] = np.array([int((np.abs(zstd64 - want) > 1e-4 * (1 + np.abs(want))).sum())])
    out["max_self_z_std"] = np.array([float(np.abs(zstd64 - want).max())])
    print(f"N_self (4096 rays): image maps {int(out['n_self'][0])}, z_std {int(out['n_self_z_std'][0])} (max |dz_std| {float(out['max_self_z_std'][0]):.3e})")
    path = os.path.join(HERE, "trained_4k.npz")
    np.savez_compressed(path, **out)
    print(f"wrote {path}: {os.path.getsize(path) / 1e3:.1f} kB, {len(out)} arrays")

    # ---- 65 536 pixels of the full-size image of held-out pose 0 (bench.c5_trained_quality's image)
    i = scene.i_test[0]
    focal = scene.focal * IMG_W / scene.w
    K = get_persp_intrinsic(IMG_H, IMG_W, focal)
    full = get_persp_rays(IMG_H, IMG_W, K, torch.tensor(scene.poses[i, :3, :4])).reshape(2, -1, 3).contiguous()
    sel = torch.sort(torch.randperm(IMG_H * IMG_W, generator=torch.Generator().manual_seed(64))[:65536]).values
    rays = full[:, sel].contiguous()
    ret = reference_and_port(net, sd, pc, rays, bounds, "trained image 64k")
    c, lab, t = scene.trace(rays[0].double().numpy(), rays[1].double().numpy())
    out = dict(common, pixel_index=sel.numpy().astype(np.int32), image_hwf=np.array([IMG_H, IMG_W, focal], np.float64), pose_index=np.array([i]),
               rays_sha256=np.frombuffer(hashlib.sha256(mg.np32(rays).tobytes()).digest(), np.uint8),
               full_image_rays_sha256=np.frombuffer(hashlib.sha256(mg.np32(full).tobytes()).digest(), np.uint8),
               gt_rgb=c, gt_label=lab.astype(np.int8), gt_t=t)
    for k in ("rgb", "depth", "acc", "semantics", "rgb0"):
        out[f"eval_{k}"] = mg.np32(ret[k])
    psnr = -10 * np.log10(np.mean((out["eval_rgb"] - out["gt_rgb"]) ** 2))
    print(f"65536 pixels of the {IMG_W}x{IMG_H} image of pose {i}: reference PSNR vs analytic GT {psnr:.2f} dB, mean acc {out['eval_acc'].mean():.4f}")
    path = os.path.join(HERE, "trained_img64k.npz")
    np.savez_compressed(path, **out)
    print(f"wrote {path}: {os.path.getsize(path) / 1e3:.1f} kB, {len(out)} arrays")


if __name__ == "__main__":
    main()
