"""Which rays does the 16-bit kernel get badly wrong over the full image?  (diagnostic)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
import nerf_sos_amd
from nerf_sos_amd import ops, synthetic as syn
dev = "cuda:0"
torch.manual_seed(0)
net = nerf_sos_amd.NeRFNet(N_samples=64, N_importance=128, use_semantics=True, sem_with_coord=True, ray_chunk=65536).to(dev).eval()
rays = syn.image_rays(dev)
N = rays.shape[1]
for prec in ("fp16", "bf16"):
    pk = net.nerf.packed_weights(prec)
    nbad = 0
    for c0 in range(0, N, 65536):
        o, d = rays[0][c0:c0 + 65536].contiguous(), rays[1][c0:c0 + 65536].contiguous()
        R = o.shape[0]
        near = torch.full((R,), syn.NEAR, device=dev); far = torch.full((R,), syn.FAR, device=dev)
        z, v = ops.ray_setup(d, near, far, 64, None)
        ref = ops.mlp_forward_rays(net.nerf.packed_weights(), net.nerf.sem_mode, o, d, v, z)
        a = ops.mlp_forward_rays_lp(pk, net.nerf.sem_mode, prec, o, d, v, z)
        err = (a - ref).abs().amax(-1)
        bad = ((err > 0.05) | ~torch.isfinite(a).all(-1)).nonzero()
        if bad.shape[0] and nbad < 3:
            nbad += 1
            pts = (bad[:, 0] * 64 + bad[:, 1]).cpu().numpy()
            print(prec, "chunk at ray", c0, ":", bad.shape[0], "bad points; nan:", int(torch.isnan(a).any(-1).sum()), "inf:", int(torch.isinf(a).any(-1).sum()))
            print("  pos in tile(256):", np.unique(pts % 256)[:64], " tiles:", np.unique(pts // 256)[:20], "n", len(np.unique(pts // 256)))
            r, s = int(bad[0, 0]), int(bad[0, 1])
            print("  example: ray", c0 + r, "sample", s, "pt", (o[r] + d[r] * z[r, s]).tolist(), "viewdir", v[r].tolist())
            print("    got", a[r, s].tolist(), "\n    ref", ref[r, s].tolist())
            sel = torch.unique(bad[:, 0])[:32]
            a2 = ops.mlp_forward_rays_lp(pk, net.nerf.sem_mode, prec, o[sel].contiguous(), d[sel].contiguous(), v[sel].contiguous(), z[sel].contiguous())
            e2 = (a2 - ref[sel])
            print("    the same rays in a launch of their own: max err", float(e2.abs().amax()), "nan", int(torch.isnan(a2).sum()))
            pts_r = (o[sel][:, None] + d[sel][:, None] * z[sel][..., None]).reshape(-1, 3)
            badmask = ((a2 - ref[sel]).abs().amax(-1) > 0.05) | torch.isnan(a2).any(-1)
            bp = pts_r[badmask.reshape(-1)]
            print("    bad point coords min", bp.min(0).values.tolist(), "max", bp.max(0).values.tolist())
            print("    some bad points", bp[:6].tolist())
