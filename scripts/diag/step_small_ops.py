"""Which Python lines launch the small torch kernels (fill / copy / mul / add ...) of the C3 training step: torch.profiler with stacks."""
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
import traceback
sites = collections.Counter()


def wrap(mod, name):
    orig = getattr(mod, name)

    def f(*a, **k):
        st = traceback.extract_stack()[:-1]
        mine = [f"{os.path.basename(x.filename)}:{x.lineno} {x.name}" for x in st if "nerf-sos_amd" in x.filename or "torch/optim" in x.filename or "torch/autograd" in x.filename]
        sites[(f"{getattr(mod, '__name__', 'Tensor')}.{name}", " <- ".join(mine[-3:]))] += 1
        return orig(*a, **k)
    setattr(mod, name, f)


for m, names in ((torch, ("zeros", "zeros_like", "full", "full_like", "ones_like", "cat", "rand", "min", "empty_like", "where")),
                 (torch.Tensor, ("zero_", "fill_", "clone", "contiguous", "copy_", "repeat", "__mul__", "__rmul__", "__add__", "__radd__", "float", "to"))):
    for n in names:
        wrap(m, n)
step(9)
torch.cuda.synchronize()
for (name, where), n in sorted(sites.items(), key=lambda kv: (-kv[1], kv[0])):
    print(f"{n:3d}  {name:22s} {where}")
