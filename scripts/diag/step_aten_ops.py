"""The aten ops of one C3 training step that launch a kernel, with the line of this package that issued them (torch.profiler,
with_stack): what is left of the element-wise glue around the HIP launches."""
import os, sys, types, collections
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from torch.profiler import profile, ProfilerActivity
import nerf_sos_amd
from nerf_sos_amd import sharding, synthetic as syn
DEV = "cuda:0"
args = types.SimpleNamespace(rand_neg=False, self_corr_w=0, use_sim_matrix=True, patch_stride=6, app_corr_params=["0.18", "1", "0.46", "1"], geo_corr_params=["0.5", "1", "3", "1"])
B = int(sys.argv[1]) if len(sys.argv) > 1 else 1
torch.manual_seed(0)
net = nerf_sos_amd.NeRFNet(N_samples=64, N_importance=128, use_semantics=True, sem_with_coord=True, perturb=1.0, raw_noise_std=1.0, ray_chunk=1 << 20).to(DEV)
for n_, p_ in net.named_parameters():
    p_.requires_grad = "semantic_linear" in n_
net.train(); net.mlp_precision = "bf16"; net.rng = "philox"
opt = torch.optim.Adam([p for p in net.parameters() if p.requires_grad], lr=5e-4, fused=True)
rays = syn.synthetic_patches(B, 64, 6, seed=0, device=DEV)
feat = torch.randn(B, 384, 14, 14, device=DEV); cls_ = torch.randn(B, 384, device=DEV) + 3 * torch.randn(1, 384, device=DEV)
corr, geo = nerf_sos_amd.CorrelationLoss(args), nerf_sos_amd.GeoCorrelationLoss(args)
con = nerf_sos_amd.NeRFContrastive(device=DEV) if B >= 2 else None


def step(i):
    opt.zero_grad(set_to_none=True)
    sharding.sharded_patch_step(net, rays, (syn.NEAR, syn.FAR), B, feat, cls_, corr, geo, step=i, seed=0, contrast_loss=con, contrast_w=0.01)
    opt.step()


for i in range(5):
    step(i)
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=True, record_shapes=True) as prof:
    step(9)
    torch.cuda.synchronize()
ev = prof.events()
launchers = collections.Counter()
for e in ev:
    if e.device_type == torch.autograd.DeviceType.CPU and e.name.startswith("aten::") and e.kernels:
        # innermost aten op that owns a kernel: skip parents whose child also owns it
        if any(c.kernels for c in e.cpu_children if c.name.startswith("aten::")):
            continue
        chain, q = [], e.cpu_parent
        while q is not None and len(chain) < 6:
            chain.append(q.name[:48])
            q = q.cpu_parent
        where = " <- ".join(chain) or "(top level)"
        where += "   shapes " + str(e.input_shapes)[:80]
        launchers[(e.name, where)] += len(e.kernels)
for (name, where), n in sorted(launchers.items(), key=lambda kv: (-kv[1], kv[0])):
    print(f"{n:3d} {name:28s} {where}")
