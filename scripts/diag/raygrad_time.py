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
    r = rays.clone().requires_grad_(bool(grad))
    ret = net(r, (syn.NEAR, syn.FAR), retraw=False)
    if grad:
        loss = ((ret["rgb"] - gt) ** 2).mean()
        if grad == 2:           # the trainer's loss (engines/trainer.py:113-121): both maps -> both networks are differentiated
            loss = loss + ((ret["rgb0"] - gt) ** 2).mean()
        loss.backward()
        return r.grad
    return ret["rgb"]


out = {"rays": R}
# grad 1: a loss on the fine map only -- the coarse net gets no gradient (its samples are detached) and is not differentiated
for grad, key in ((0, "render_ms_tuned_kernel"), (2, "pose_step_ms_generic_kernels"), (1, "pose_step_ms_loss_on_the_fine_map_only")):
    with torch.set_grad_enabled(bool(grad)):
        for _ in range(3):
            step(grad)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(8):
            g = step(grad)
        torch.cuda.synchronize()
    out[key] = round((time.perf_counter() - t0) / 8 * 1e3, 2)
out["g_rays_finite"] = bool(torch.isfinite(g).all())
print(json.dumps(out))
