import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
import nerf_sos_amd
from nerf_sos_amd import ops, synthetic as syn
dev = "cuda:0"
torch.manual_seed(0)
net = nerf_sos_amd.NeRFNet(N_samples=64, N_importance=128, use_semantics=True, sem_with_coord=True, ray_chunk=65536).to(dev).eval()
rays = syn.image_rays(dev)
with torch.no_grad():
    net.mlp_precision = "fp32"; ref = net(rays, (syn.NEAR, syn.FAR), retraw=True)
    net.mlp_precision = "fp16"; a = net(rays, (syn.NEAR, syn.FAR), retraw=True); b = net(rays, (syn.NEAR, syn.FAR), retraw=True)
print("deterministic:", {k: bool(torch.equal(a[k], b[k])) for k in ("rgb0", "raw0", "raw", "rgb")})
e = (a["rgb0"] - ref["rgb0"]).abs().amax(-1)
bad = (e > 0.01).nonzero().flatten()
print("bad rays (rgb0):", bad.numel(), bad[:30].tolist())
eraw = (a["raw0"] - ref["raw0"]).abs().amax(-1)   # [N,64]
print("raw0 err on bad rays: max per ray", eraw[bad[:10]].amax(-1).tolist())
r = int(bad[0])
print("ray", r, "raw0 fp16 sigma", a["raw0"][r, :, 3].tolist()[:16], "\n fp32 sigma", ref["raw0"][r, :, 3].tolist()[:16])
print(" weights0 fp16", a["weights0"][r, :8].tolist(), "fp32", ref["weights0"][r, :8].tolist())
print(" acc0", float(a["acc0"][r]), float(ref["acc0"][r]), "depth0", float(a["depth0"][r]), float(ref["depth0"][r]))
