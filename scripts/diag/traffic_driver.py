"""One process that launches the fine-pass MLP kernel of every precision path a few times, for the PMC traffic passes
(scripts/profile_traffic.sh): exact fp32 (c2), split-fp16 (c2_fp16x3), 16-bit non-SAVE fp16 at a 65536-ray chunk (c5), 16-bit
SAVE bf16 at 4096 and 8192 rays (c3, c4).  Launch order and counts are fixed so the aggregator can tell the launches apart by
kernel name, grid and order."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import nerf_sos_amd
from nerf_sos_amd import ops, synthetic as syn
dev = "cuda:0"
REPS = 6


def setup(R, sem):
    torch.manual_seed(0)
    kw = dict(use_semantics=True, sem_with_coord=True) if sem else dict(use_semantics=False)
    net = nerf_sos_amd.NeRFNet(N_samples=64, N_importance=128, **kw).to(dev).eval()
    rays = syn.synthetic_rays(R, seed=0, device=dev)
    near = torch.full((R,), syn.NEAR, device=dev); far = torch.full((R,), syn.FAR, device=dev)
    z, v = ops.ray_setup(rays[1].contiguous(), near, far, 192, None)
    return net, rays[0].contiguous(), rays[1].contiguous(), v, z


net, o, d, v, z = setup(4096, False)
for _ in range(REPS):
    ops.mlp_forward_rays(net.nerf_fine.packed_weights(), 0, o, d, v, z)                       # mlp_fused_kernel<0,true,0>
pk = net.nerf_fine.packed_weights("fp16x3")
for _ in range(REPS):
    ops.mlp_forward_rays_lp(pk, 0, "fp16x3", o, d, v, z)                                       # mlp_x316_kernel<0,false> (the default split-fp16 forward; NSOS_X3_KERNEL=1: mlp_x3_kernel<0,0>)
net, o, d, v, z = setup(65536, True)
pk = net.nerf_fine.packed_weights("fp16")
for _ in range(REPS):
    ops.mlp_forward_rays_lp(pk, 2, "fp16", o, d, v, z)                                         # mlp_lp16_kernel<F16,2,false>, c5 chunk
for R in (4096, 8192):
    net, o, d, v, z = setup(R, True)
    pk = net.nerf_fine.packed_weights("bf16")
    for _ in range(REPS):
        ops.mlp_forward_rays_save(pk, 2, o, d, v, z, "bf16", compact=True)                     # mlp_lp16_kernel<BF16,2,true>, c3 / c4
    for _ in range(REPS):
        ops.mlp_forward_rays_lp(pk, 2, "bf16", o, d, v, z)                                     # mlp_lp16_kernel<BF16,2,false>
torch.cuda.synchronize()
print("traffic driver done")
