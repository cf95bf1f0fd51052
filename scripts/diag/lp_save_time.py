import os, sys
sys.path.insert(0, os.getcwd())
import torch, nerf_sos_amd
from nerf_sos_amd import ops, synthetic as syn
dev = torch.device("cuda:0")
torch.manual_seed(0)
net = nerf_sos_amd.NeRFNet(N_samples=64, N_importance=128, use_semantics=True, sem_with_coord=True).to(dev)
rays = syn.synthetic_rays(8192, seed=0, device=dev)
o, d = rays[0].contiguous(), rays[1].contiguous()
v = d / d.norm(dim=-1, keepdim=True)
z = torch.linspace(1.2, 14.0, 192, device=dev).expand(8192, 192).contiguous()
pk = net.nerf_fine.packed_weights("bf16")
def run(save):
    if save:
        return ops.mlp_forward_rays_save(pk, net.nerf_fine.sem_mode, o, d, v, z, "bf16", compact=True)
    return ops.mlp_forward_rays_lp(pk, net.nerf_fine.sem_mode, "bf16", o, d, v, z)
for save in (False, True):
    for _ in range(3): run(save)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20): run(save)
    e1.record(); torch.cuda.synchronize()
    print("save" if save else "plain", round(e0.elapsed_time(e1) / 20, 4), "ms", os.environ.get("NERF_SOS_HIP_LIB", "default")[-14:])
