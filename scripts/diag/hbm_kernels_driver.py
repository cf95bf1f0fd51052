"""One process that launches the path's HBM / latency-bound kernels at the C5 chunk (65 536 rays, 6 channels) a few times each, for
the PMC passes of scripts/profile_hbm_kernels.sh: ray_setup_kernel, composite_importance_kernel in eval mode
(deterministic u) and in train mode (random u + sigma noise), composite_kernel<3> (the fine pass: 192 samples).  Launch order and
counts are fixed (REPS of each, in this order) so the aggregator can tell eval from train launches of the same kernel by order."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from nerf_sos_amd import ops, synthetic as syn
dev = "cuda:0"
REPS = 6
R = int(os.environ.get("NSOS_HBM_RAYS", "65536"))
g = torch.Generator(dev).manual_seed(0)
rays = syn.image_rays(dev, (0, R))
d = rays[1].contiguous()
near, far = torch.full((R,), syn.NEAR, device=dev), torch.full((R,), syn.FAR, device=dev)
raw0 = torch.randn(R, 64, 6, device=dev, generator=g)
raw0[..., 3] = raw0[..., 3] * 3 - 1
raw1 = torch.randn(R, 192, 6, device=dev, generator=g)
u = torch.rand(R, 128, device=dev, generator=g)
noise = torch.randn(R, 64, device=dev, generator=g)
z0, _ = ops.ray_setup(d, near, far, 64, None)
_, z1, _, _ = ops.composite_importance(raw0, z0, d, 128)
torch.cuda.synchronize()
for _ in range(REPS):
    ops.ray_setup(d, near, far, 64, None)
for _ in range(REPS):
    ops.composite_importance(raw0, z0, d, 128)                                  # eval
for _ in range(REPS):
    ops.composite_importance(raw0, z0, d, 128, noise, 1.0, False, u)            # train
for _ in range(REPS):
    ops.composite(raw1, z1, d)
torch.cuda.synchronize()
print("hbm kernels driver done", R)
