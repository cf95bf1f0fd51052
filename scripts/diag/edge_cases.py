#!/usr/bin/env python3
"""Interface edge cases of NeRFNet probed once (chunked frozen-backbone training, retpts/retraw in training, non-contiguous /
float64 / tuple rays, CPU-tensor bounds, autocast, rays that require grad): prints what happens."""
import os, sys, traceback
sys.path.insert(0, os.getcwd())
import torch, nerf_sos_amd
from nerf_sos_amd import synthetic as syn
dev = "cuda:0"
torch.manual_seed(1)
def mk(**kw):
    n = syn.spiky_density_(nerf_sos_amd.NeRFNet(N_samples=32, N_importance=32, use_semantics=True, sem_with_coord=True, **kw).to(dev).eval(), 2.0, 0.5)
    return n
rays = syn.synthetic_rays(300, seed=2, device=dev)
def head_grads(net, r):
    for n, p in net.named_parameters(): p.requires_grad_("semantic_linear" in n)
    net.zero_grad(set_to_none=True)
    out = net(r, (syn.NEAR, syn.FAR))
    ((out["semantics"] ** 2).mean() + (out["semantics0"] ** 2).mean()).backward()
    return {n: p.grad.clone() for n, p in net.named_parameters() if p.grad is not None}, out
# 1. chunked frozen-backbone training
a = mk(ray_chunk=1 << 20); b = mk(ray_chunk=100); b.load_state_dict(a.state_dict())
ga, oa = head_grads(a, rays); gb, ob = head_grads(b, rays)
print("1 chunked frozen grads max rel diff:", max(float((ga[k]-gb[k]).abs().max()/(ga[k].abs().max()+1e-30)) for k in ga), "outputs equal:", all(torch.equal(oa[k], ob[k]) for k in oa))
# 2. retpts / retraw in training
try:
    out = a(rays, (syn.NEAR, syn.FAR), retpts=True, retraw=True)
    print("2 retpts keys:", sorted(k for k in out if "pts" in k or "raw" in k), out["pts"].shape, out["raw"].requires_grad, out["semantics"].requires_grad)
except Exception: traceback.print_exc()
# 3. non-contiguous rays
r2 = rays.permute(1, 0, 2).contiguous().permute(1, 0, 2)
with torch.no_grad():
    x, y = a(rays, (syn.NEAR, syn.FAR)), a(r2, (syn.NEAR, syn.FAR))
print("3 non-contiguous equal:", all(torch.equal(x[k], y[k]) for k in x))
# 4. rays requiring grad
try:
    r3 = rays.clone().requires_grad_(True)
    out = a(r3, (syn.NEAR, syn.FAR))
    print("4 rays.requires_grad: ran; rgb.requires_grad =", out["rgb"].requires_grad)
except Exception as e:
    print("4 rays.requires_grad ->", type(e).__name__, str(e)[:160])
# 5. tuple / list ray_batch, float64 rays
with torch.no_grad():
    y = a((rays[0].double(), rays[1].double()), (syn.NEAR, syn.FAR))
print("5 float64 tuple rays equal:", all(torch.equal(x[k], y[k]) for k in x))
# 6. near/far tensors on CPU
try:
    with torch.no_grad():
        y = a(rays, (torch.full((300, 1), syn.NEAR), torch.full((300, 1), syn.FAR)))
    print("6 cpu bounds equal:", all(torch.equal(x[k], y[k]) for k in x))
except Exception as e:
    print("6 cpu bounds ->", type(e).__name__, str(e)[:160])
# 7. eval under autocast
try:
    with torch.no_grad(), torch.autocast("cuda", dtype=torch.bfloat16):
        y = a(rays, (syn.NEAR, syn.FAR))
    print("7 autocast equal:", all(torch.equal(x[k], y[k]) for k in x), y["rgb"].dtype)
except Exception as e:
    print("7 autocast ->", type(e).__name__, str(e)[:160])
