#!/usr/bin/env python3
"""GPU box: train the shipped architecture on the procedural scene (nerf_sos_amd.synthetic.ProceduralScene) with the
package's OWN training path and write a reference-format checkpoint -- the trained field every "matches the reference"
number of round 5 is re-measured on (until now every fixture used seed-0 default-init weights).

    gpurun -- python scripts/make_trained_scene.py [--steps 8000] [--head-steps 1500]
    -> gpurun_out/trained_scene.ckpt  (copy to tests/golden/trained_scene.ckpt, then tests/golden/make_goldens_trained.py
       runs the REAL reference on it in the build container)

Two stages, the reference's own recipe order:
  A. every parameter trainable, ray batches of 4096 random pixels over the training views, img2mse on rgb and rgb0,
     Adam 5e-4 decayed exponentially (run_nerf.py:321-333, engines/trainer.py:113-121), perturb 1, raw_noise_std 1
     (configs/flower_full.txt);
  B. `--fix_backbone` (run_nerf.py:307-318): only semantic_linear.* trains; 8 strided 32x32 patches per step, appearance +
     geometric correlation losses on semantics0 / semantics (engines/trainer.py:127-166, scripts/train_flower_node0.sh:30-39)
     against SYNTHETIC DINO features (the ViT is outside the path and absent): per 14x14 cell a fixed random embedding of the
     cell's foreground coverage and mean colour plus noise -- what a self-supervised ViT provides in spirit (cells of one
     object look alike).
The scene directory is written in the reference's prepared-scene layout (io.PreparedScene reads it back: the product's reader
is what feeds the training)."""
import argparse
import json
import math
import os
import sys
import time
import types

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import torch.nn.functional as F
import nerf_sos_amd
from nerf_sos_amd import io as nio, sharding, synthetic as syn

ap = argparse.ArgumentParser()
ap.add_argument("--steps", type=int, default=8000)
ap.add_argument("--head-steps", type=int, default=1500)
ap.add_argument("--rays", type=int, default=4096)
ap.add_argument("--precision", default="fp32")
ap.add_argument("--out", default="gpurun_out/trained_scene.ckpt")
ap.add_argument("--scene-dir", default="/tmp/nsos_procedural_scene")
a = ap.parse_args()
dev = torch.device("cuda:0")
torch.manual_seed(0)
np.random.seed(0)

scene = syn.ProceduralScene()
scene.write_prepared(a.scene_dir)
train = nio.PreparedScene(a.scene_dir, "train", load_rays=False, bin_thres=0.3).to_device(dev)
test = nio.PreparedScene(a.scene_dir, "test", load_rays=False, bin_thres=0.3)
near, far = train.scene.near_far()
n_pix = train.image_count * train.height * train.width
log = {"scene": {"views": scene.n_views, "H": scene.h, "W": scene.w, "focal": scene.focal, "near": near, "far": far,
                 "train_pixels": n_pix}, "stage_a": [], "stage_b": []}

net = nerf_sos_amd.NeRFNet(N_samples=64, N_importance=128, use_semantics=True, sem_with_coord=True, perturb=1.0,
                           raw_noise_std=1.0).to(dev).train()
net.mlp_precision = a.precision


def psnr_of_view(i_test: int, precision: str = "fp32"):
    net.eval()
    net.mlp_precision = precision
    rays = test.rays_on_device(i_test, dev).reshape(2, -1, 3)
    with torch.no_grad():
        ret = net(rays, (near, far), retraw=False)
    gt = torch.from_numpy(test.rgbs[i_test]).to(dev).reshape(-1, 3)
    mse = float(((ret["rgb"] - gt) ** 2).mean())
    lab = ret["semantics"].argmax(-1).reshape(-1).cpu().numpy()
    m = test.masks[i_test].reshape(-1)
    agree = float(max((lab == m).mean(), (lab != m).mean()))          # the two clusters carry no fixed names
    net.train()
    net.mlp_precision = a.precision
    return -10 * math.log10(mse), agree, ret


# ---------------------------------------------------------------------------------------------- stage A: all parameters
opt = torch.optim.Adam(net.parameters(), lr=5e-4, betas=(0.9, 0.999))
decay_rate, decay_steps = 0.1, max(a.steps, 1)
t0 = time.perf_counter()
for step in range(1, a.steps + 1):
    pix = torch.randint(0, n_pix, (a.rays,), device=dev)
    b = train.ray_batch(pix)
    opt.zero_grad(set_to_none=True)
    ret = net(b["rays"], (near, far), retraw=False)
    loss = ((ret["rgb"] - b["target_s"]) ** 2).mean() + ((ret["rgb0"] - b["target_s"]) ** 2).mean()
    loss.backward()
    opt.step()
    for g in opt.param_groups:                                         # run_nerf.py:331-333: exponential decay
        g["lr"] = 5e-4 * decay_rate ** (step / decay_steps)
    if step % 500 == 0 or step == 1:
        p, _, _ = psnr_of_view(0)
        rec = {"step": step, "loss": round(float(loss.detach()), 6), "test_psnr_view0": round(p, 2), "s": round(time.perf_counter() - t0, 1)}
        log["stage_a"].append(rec)
        print(json.dumps(rec), flush=True)

# ------------------------------------------------------------------------------- stage B: the frozen-backbone head recipe
for n_, p_ in net.named_parameters():
    p_.requires_grad = "semantic_linear" in n_
head = [p for p in net.parameters() if p.requires_grad]
opt_b = torch.optim.Adam(head, lr=5e-4)
largs = types.SimpleNamespace(rand_neg=False, self_corr_w=0, use_sim_matrix=True, patch_stride=3,
                              app_corr_params=["0.18", "1", "0.46", "1"], geo_corr_params=["0.5", "1", "3", "1"])
corr, geo = nerf_sos_amd.CorrelationLoss(largs), nerf_sos_amd.GeoCorrelationLoss(largs)
B, P, STRIDE = 8, 32, 3
g_emb = torch.Generator(device="cpu").manual_seed(11)
E = torch.randn(2, 384, generator=g_emb).to(dev)                      # foreground / background embeddings
Cc = (0.5 * torch.randn(3, 384, generator=g_emb)).to(dev)             # colour projection
rgbs_dev, masks_dev = train.rgbs, train.masks.float()


def dino_like(img_idx, origins):
    """[B,384,14,14] features and [B,384] class tokens of the crops (synthetic stand-in for the ViT, see the module docstring)."""
    feats = []
    for i, (h0, w0) in zip(img_idx, origins):
        crop_m = masks_dev[i, h0:h0 + P * STRIDE, w0:w0 + P * STRIDE, 0][None, None]
        crop_c = rgbs_dev[i, h0:h0 + P * STRIDE, w0:w0 + P * STRIDE].permute(2, 0, 1)[None]
        cov = F.adaptive_avg_pool2d(crop_m, 14)[0, 0]                   # [14,14] foreground coverage
        col = F.adaptive_avg_pool2d(crop_c, 14)[0]                      # [3,14,14]
        f = cov[None] * E[1][:, None, None] + (1 - cov[None]) * E[0][:, None, None] + torch.einsum("ck,chw->khw", Cc, col - 0.5)
        feats.append(f + 0.1 * torch.randn_like(f))
    feat = torch.stack(feats)
    return feat, feat.mean((2, 3))


t0 = time.perf_counter()
for step in range(1, a.head_steps + 1):
    idx = [int(v) for v in np.random.randint(0, train.image_count, B)]
    origins = nio.draw_patch_origins(B, train.height, train.width, P * STRIDE)
    pb = train.patch_batch(idx, P * STRIDE, STRIDE, origins=origins)
    feat, cls_ = dino_like(idx, origins)
    opt_b.zero_grad(set_to_none=True)
    loss = sharding.sharded_patch_step(net, pb["rays_planar"], (near, far), B, feat, cls_, corr_loss=corr, geo_loss=geo,
                                       correlation_w=1.0, geo_w=0.01, step=step, seed=0)
    opt_b.step()
    if step % 250 == 0 or step == 1:
        p, agree, _ = psnr_of_view(0)
        rec = {"step": step, "loss": round(float(loss), 6), "label_agreement_view0": round(agree, 4), "s": round(time.perf_counter() - t0, 1)}
        log["stage_b"].append(rec)
        print(json.dumps(rec), flush=True)

# ------------------------------------------------------------------------------------------------------ report + save
final = {}
for prec in ("fp32", "fp16x3", "bf16", "fp16"):
    ps, ag = zip(*[psnr_of_view(i, prec)[:2] for i in range(test.image_count)])
    final[prec] = {"test_psnr_vs_analytic_gt": [round(x, 2) for x in ps], "label_agreement_vs_gt_mask": [round(x, 4) for x in ag]}
log["final"] = final
sd = net.state_dict()
log["sigma_stats"] = {"alpha_w_absmax": float(sd["nerf_fine.mlp.alpha_linear.weight"].abs().max()),
                      "params": int(sum(v.numel() for v in sd.values()))}
print(json.dumps(final, indent=1))
os.makedirs(os.path.dirname(os.path.abspath(a.out)), exist_ok=True)
for p_ in net.parameters():
    p_.requires_grad = True
nio.save_checkpoint(a.out, a.steps + a.head_steps, net, None)          # engines/trainer.py:216-222 (optimizer state left out: size)
with open(os.path.splitext(a.out)[0] + "_log.json", "w") as f:
    json.dump(log, f, indent=1)
print("wrote", a.out, os.path.getsize(a.out), "bytes")
