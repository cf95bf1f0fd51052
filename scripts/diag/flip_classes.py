"""Classify the rays whose free-running fine pass leaves the 1e-4 band vs the reference port (diagnostic)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))), "tests"))
import json
import numpy as np, torch
import nerf_sos_amd
from nerf_sos_amd import ops
from oracle import torch_port as tp
from helpers import CFGS, ref_state
dev = "cuda:0"
manifest = json.load(open("tests/golden/manifest.json"))
torch.set_num_threads(32)
for peaky in (False, True):
    cfg = tp.PortConfig(n_importance=128, **CFGS["semcoord"])
    sd = ref_state("semcoord", manifest, peaky=peaky)
    rays = tp.synthetic_rays(4096, seed=0)
    with torch.no_grad():
        ref = tp.render(sd, cfg, rays, (tp.NEAR, tp.FAR))
        near, far = torch.full((4096, 1), tp.NEAR), torch.full((4096, 1), tp.FAR)
        z = tp.stratified_z(near, far, 64, None)
        mids = 0.5 * (z[..., 1:] + z[..., :-1])
        cdf_ref = tp.pdf_to_cdf(ref["weights0"][..., 1:-1])
        u = torch.linspace(0.0, 1.0, steps=128).expand(4096, 128)
        zs_ref, inds_ref = tp.invert_cdf(mids, cdf_ref, u)
        z_ref, _ = torch.sort(torch.cat([z, zs_ref], -1), -1)
    net = nerf_sos_amd.NeRFNet(N_samples=64, N_importance=128, **CFGS["semcoord"]).to(dev).eval()
    net.load_state_dict(sd)
    with torch.no_grad():
        out = net(rays.to(dev), (tp.NEAR, tp.FAR))
        zc = ops.ray_setup(rays[1].to(dev), near.reshape(-1).to(dev), far.reshape(-1).to(dev), 64)[0]
        z_hip, zs_hip, _, cdf_hip, inds_hip = ops.importance_sample(zc, out["weights0"], 128, debug=True)
    outside = np.zeros(4096, bool)
    for k in ("rgb", "depth", "acc", "semantics"):
        a, b = out[k].cpu().numpy().astype(np.float64), ref[k].numpy().astype(np.float64)
        outside |= (np.abs(a - b) > 1e-4 + 1e-4 * np.abs(b)).reshape(4096, -1).any(-1)
    flip = (inds_hip.cpu() != inds_ref)
    flip_ray = flip.any(-1).numpy()
    flip_not_last = flip[:, :-1].any(-1).numpy()
    dz = (z_hip.cpu() - z_ref).abs().amax(-1).numpy()
    dzs = (zs_hip.cpu() - zs_ref).abs()
    sig_last_hip, sig_last_ref = out["raw"][:, -1, 3].cpu().numpy(), ref["raw"][:, -1, 3].numpy()
    sign_last = (sig_last_hip > 0) != (sig_last_ref > 0)
    dw0 = (out["weights0"].cpu() - ref["weights0"]).abs().amax().item()
    dcdf = (cdf_hip.cpu() - cdf_ref).abs().amax().item()
    print(f"--- {'spiky' if peaky else 'default-init'}: outside {outside.sum()}  | rays with any index flip {flip_ray.sum()} (excluding the u=1 sample: {flip_not_last.sum()})"
          f" | last-sample sigma sign differs {sign_last.sum()} | max|dw0| {dw0:.2e} max|dcdf| {dcdf:.2e}")
    print("    outside & index flip (not last):", (outside & flip_not_last).sum(), " outside & only-last flip:", (outside & flip_ray & ~flip_not_last).sum(),
          " outside & no flip:", (outside & ~flip_ray).sum(), " outside & sign_last:", (outside & sign_last).sum())
    print("    dz quantiles over rays:", np.quantile(dz, [0.5, 0.9, 0.99, 1.0]), " dz among outside-noflip:", np.sort(dz[outside & ~flip_ray])[-5:] if (outside & ~flip_ray).any() else None)
    print("    flip but inside:", (flip_not_last & ~outside).sum())
    nf = outside & ~flip_ray
    if nf.any():
        r = int(np.nonzero(nf)[0][0])
        print("    example no-flip ray", r, "rgb hip", out["rgb"][r].tolist(), "ref", ref["rgb"][r].tolist(), "acc", float(out["acc"][r]), float(ref["acc"][r]),
              "max dzs", float(dzs[r].max()), "sig_last", sig_last_hip[r], sig_last_ref[r])
