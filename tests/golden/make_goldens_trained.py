#!/usr/bin/env python3
"""The REAL reference on a TRAINED field (every earlier fixture used seed-0 default-init weights).

Run in the BUILD CONTAINER only (needs /root/reference):

    python tests/golden/make_goldens_trained.py

`tests/golden/trained_scene.ckpt` is a reference-format checkpoint (engines/trainer.py:216-222: a dict of tensors) of the shipped
architecture trained on `nerf_sos_amd.synthetic.ProceduralScene` by `scripts/make_trained_scene.py` on an MI355X with this
package's own training path (8000 all-parameter steps + 1500 steps of the --fix_backbone head recipe; log:
profiles/r05/a_trained_scene_training_log.json; 38-40 dB on the held-out views).  This script loads it into the unmodified
reference `NeRFNet` (`load_state_dict(strict=True)`, run_nerf.py:349-353), renders 256 rays of the four held-out views
(64 seeded random pixels each; rays from the reference's own get_persp_rays) in eval mode and in train mode (captured draws),
records the importance sampler's z_fine with a forward hook, asserts that oracle/torch_port.py reproduces every output bit for
bit, and writes tests/golden/trained.npz -- data only: rays, the reference's outputs, its draws, the analytic ground truth of
those pixels and the checkpoint's sha256.

Also written: the reference's own sensitivity yardstick for this field (N_self: the port with its coarse network evaluated in
fp64 and rounded once) so that the free-running GPU test has its bar without /root/reference.
"""
import hashlib
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import make_goldens as mg  # noqa: E402  (sets up the reference import; does not run its main)
from make_goldens_zfine import record_z  # noqa: E402
from oracle import torch_port as tp  # noqa: E402
from utils.ray import get_persp_rays, get_persp_intrinsic  # noqa: E402  (reference)
import nerf_sos_amd  # noqa: E402,F401
from nerf_sos_amd.synthetic import ProceduralScene  # noqa: E402

CKPT = os.path.join(HERE, "trained_scene.ckpt")
PER_VIEW = 64


def main():
    ckpt = torch.load(CKPT, map_location="cpu")
    sd = ckpt["model"]
    net = mg.NeRFNet(N_samples=64, N_importance=128, perturb=1.0, raw_noise_std=1.0, pts_chuck=1024 * 64,
                     use_semantics=True, sem_with_coord=True)
    net.load_state_dict(sd, strict=True)                                   # run_nerf.py:349-353
    pc = tp.PortConfig(n_importance=128, use_semantics=True, sem_with_coord=True)
    scene = ProceduralScene()
    K = get_persp_intrinsic(scene.h, scene.w, scene.focal)
    g = torch.Generator().manual_seed(20260929)
    rays, pix, gt_rgb, gt_lab, gt_t = [], [], [], [], []
    for i in scene.i_test:
        full = get_persp_rays(scene.h, scene.w, K, torch.tensor(scene.poses[i, :3, :4]))      # [2,H,W,3]
        sel = torch.randperm(scene.h * scene.w, generator=g)[:PER_VIEW]
        r = full.reshape(2, -1, 3)[:, sel]
        rays.append(r)
        pix.append(torch.stack([torch.full_like(sel, i), sel], -1))
        c, lab, t = scene.trace(r[0].double().numpy(), r[1].double().numpy())
        gt_rgb.append(c), gt_lab.append(lab), gt_t.append(t)
    rays = torch.cat(rays, 1).contiguous()
    near, far = scene.NEAR, scene.FAR
    out = {"rays": mg.np32(rays), "pixels": torch.cat(pix).numpy(), "near_far": np.array([near, far], np.float64),
           "gt_rgb": np.concatenate(gt_rgb), "gt_label": np.concatenate(gt_lab), "gt_t": np.concatenate(gt_t),
           "ckpt_sha256": np.frombuffer(hashlib.sha256(open(CKPT, "rb").read()).digest(), np.uint8),
           "state_sha256": np.frombuffer(bytes.fromhex(mg.state_sha(sd)), np.uint8), "global_step": np.array([ckpt["global_step"]])}
    box, h = record_z(net)
    net.eval()
    with torch.no_grad():
        ret = net(rays, (near, far), radii=None)
        port = tp.render(sd, pc, rays, (near, far))
    assert set(ret) == set(port)
    for k in ret:
        mg.same(ret[k], port[k], f"trained eval {k}")
        out[f"eval_{k}"] = mg.np32(ret[k])
    out["eval_z_fine"] = mg.np32(box["z"])
    net.train()
    torch.manual_seed(99)
    with torch.no_grad(), mg.Recorder() as rec:
        ret_t = net(rays, (near, far), radii=None)
    dr = [t for _, t in rec.draws]
    assert [k for k, _ in rec.draws] == ["rand", "randn", "rand", "randn"]
    with torch.no_grad():
        port = tp.render(sd, pc, rays, (near, far), raw_noise_std=1.0, draws_per_chunk=[tp.Draws(*dr)])
    for k in ret_t:
        mg.same(ret_t[k], port[k], f"trained train {k}")
        out[f"train_{k}"] = mg.np32(ret_t[k])
    out["train_z_fine"] = mg.np32(box["z"])
    for i, t in enumerate(dr):
        out[f"train_draw{i}"] = mg.np32(t)
    h.remove()

    # what the reference's render says about the scene (information for the tests' messages, not a bar)
    psnr = -10 * np.log10(np.mean((out["eval_rgb"] - out["gt_rgb"]) ** 2))
    lab = out["eval_semantics"].argmax(-1)
    hit = out["gt_t"] > 0
    depth_err = np.abs(out["eval_depth"][hit, 0] - out["gt_t"][hit])
    out["info"] = np.array([psnr, max((lab == out["gt_label"]).mean(), (lab != out["gt_label"]).mean()), np.median(depth_err),
                            out["eval_acc"].mean(), out["eval_weights"].max(-1).mean()], np.float64)
    print(f"reference render of the trained field: PSNR vs analytic GT {psnr:.2f} dB, argmax-label agreement {out['info'][1]:.3f}, "
          f"median |depth - t| {out['info'][2]:.4f}, mean acc {out['info'][3]:.4f}, mean max-weight {out['info'][4]:.3f}")

    # the yardstick: the port with an fp64 coarse network (rounded once) vs itself -- rays leaving the 1e-4 band of a fine map
    # (bench.parity_yardstick: the same function the bench line's `parity.*.yardstick` comes from)
    sys.path.insert(0, mg.ROOT)
    import bench
    ref = {k[5:]: torch.from_numpy(v) for k, v in out.items() if k.startswith("eval_")}
    y = bench.parity_yardstick(tp, sd, pc, rays, ref, ref)
    assert y["staged_port_reproduces_port"] and y["gpu_rays_outside"] == 0
    out["n_self"] = np.array([y["reference_self_sensitivity_rays_outside"]])
    out["raw0_fp32_minus_fp64_max_abs"] = np.array([y["max_abs_raw0_minus_fp64"]["reference_fp32"]])
    # ... and for z_std (the std of the 128 importance samples, models/nerf_net.py:124), which bench.parity_yardstick does not cover: in a
    # trained field many coarse bins are EMPTY, their pdf sits right at the sampler's `denom < 1e-5 -> 1` switch (models/sampler.py:
    # 117-118: (0 + 1e-5) / sum = 0.9994e-5), and a last-ulp change of the cdf flips a sample between the bin's lower edge and anywhere
    # inside it -- in empty space, where the image does not depend on it, but z_std does.
    viewdirs = rays[1] / torch.norm(rays[1], dim=-1, keepdim=True)
    nearv, farv = torch.full((rays.shape[1], 1), near), torch.full((rays.shape[1], 1), far)
    sd64 = {k: v.double() for k, v in sd.items()}
    for mode, t_rand, noise0, u in (("eval", None, None, None), ("train", dr[0], dr[1], dr[2])):
        with torch.no_grad():
            z = tp.stratified_z(nearv, farv, 64, t_rand)
            pts = tp.ray_points(rays[0], rays[1], z)
            raw64 = tp.point_query(sd64, "nerf", pts.double(), viewdirs.double()[..., None, :].expand(pts.shape), pc).float()
            w64 = tp.composite(raw64, z, rays[1], noise0 if mode == "train" else None, pc)["weights"]
            zs64 = tp.importance_z(z, w64, 128, u)[1]
            zstd64 = torch.std(zs64, dim=-1, unbiased=False).numpy()
        want = out[f"{mode}_z_std"].reshape(-1)
        out[f"n_self_z_std_{mode}"] = np.array([int((np.abs(zstd64 - want) > 1e-4 * (1 + np.abs(want))).sum())])
        out[f"max_self_z_std_{mode}"] = np.array([float(np.abs(zstd64 - want).max())])
        print(f"N_self for z_std ({mode}): {int(out[f'n_self_z_std_{mode}'][0])} rays, max |dz_std| {float(out[f'max_self_z_std_{mode}'][0]):.3e}")
    print("N_self (reference vs its own fp64-coarse variant), rays outside 1e-4:", int(out["n_self"][0]), "of", rays.shape[1],
          "; max |raw0_fp32 - raw0_fp64| =", float(out["raw0_fp32_minus_fp64_max_abs"][0]))
    path = os.path.join(HERE, "trained.npz")
    np.savez_compressed(path, **out)
    print(f"wrote {path}: {os.path.getsize(path) / 1e3:.1f} kB, {len(out)} arrays")


if __name__ == "__main__":
    main()
