"""Pose-refinement step on the shipped architecture: 4096 rays x (64 + 128) that require a gradient (frozen network), forward + backward
through the generic kernels, against the same render under no_grad on the tuned kernel."""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import nerf_sos_amd
from nerf_sos_amd import synthetic as syn

dev = torch.device("cuda:0")
R = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
torch.manual_seed(0)
net = nerf_sos_amd.NeRFNet(N_samples=64, N_importance=128, use_semantics=True, sem_with_coord=True).to(dev).eval()
for p in net.parameters():
    p.requires_grad_(False)
rays = syn.synthetic_rays(R, seed=0, device=dev)
gt = torch.rand(R, 3, device=dev)


def step(grad):
    r = rays.clone().requires_grad_(grad)
    ret = net(r, (syn.NEAR, syn.FAR), retraw=False)
    if grad:
        ((ret["rgb"] - gt) ** 2).mean().backward()
        return r.grad
    return ret["rgb"]


out = {"rays": R}
for grad in (False, True):
    with torch.set_grad_enabled(grad):
        for _ in range(3):
            step(grad)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(8):
            g = step(grad)
        torch.cuda.synchronize()
    out["pose_step_ms_generic_kernels" if grad else "render_ms_tuned_kernel"] = round((time.perf_counter() - t0) / 8 * 1e3, 2)
out["g_rays_finite"] = bool(torch.isfinite(g).all())
print(json.dumps(out))
