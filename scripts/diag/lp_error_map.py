"""Where does the fp16 render differ from the fp32 one over the full image?  (diagnostic)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
import nerf_sos_amd
from nerf_sos_amd import synthetic as syn
dev = "cuda:0"
torch.manual_seed(0)
net = nerf_sos_amd.NeRFNet(N_samples=64, N_importance=128, use_semantics=True, sem_with_coord=True, ray_chunk=65536).to(dev).eval()
rays = syn.image_rays(dev)
for prec in ("fp16", "bf16", "fp16x3"):
    with torch.no_grad():
        net.mlp_precision = "fp32"; ref = net(rays, (syn.NEAR, syn.FAR), retraw=False)
        net.mlp_precision = prec; a = net(rays, (syn.NEAR, syn.FAR), retraw=False)
    for k in ("rgb0", "rgb", "acc0", "depth0"):
        e = (a[k] - ref[k]).abs().max(-1).values.cpu().numpy()
        q = np.quantile(e, [0.5, 0.9, 0.99, 0.999, 1.0])
        img = e.reshape(syn.H, syn.W)
        rowmax = img.max(1); colmax = img.max(0)
        print(prec, k, "quantiles", " ".join(f"{x:.2e}" for x in q), "mse", float((e**2).mean()),
              "| worst row", int(rowmax.argmax()), "col", int(colmax.argmax()),
              "| mean err by image quarter rows", " ".join(f"{img[i*189:(i+1)*189].mean():.2e}" for i in range(4)))
# chunk dependence of the error: first 4096 rays alone vs inside the big chunk
with torch.no_grad():
    net.mlp_precision = "fp16"
    sub = net(rays[:, :4096].contiguous(), (syn.NEAR, syn.FAR), retraw=False)
    print("first 4096 rays alone == in-chunk:", torch.equal(sub["rgb0"], a["rgb0"][:4096]) if prec == "fp16" else "n/a")
