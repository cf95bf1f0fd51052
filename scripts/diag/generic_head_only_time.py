import sys, time, torch
sys.path.insert(0, '/root/repo')
import nerf_sos_amd
from nerf_sos_amd import synthetic as syn
dev = 'cuda:0'
for head_only in (False, True):
    torch.manual_seed(0)
    net = nerf_sos_amd.NeRFNet(N_samples=64, N_importance=128, perturb=1.0, raw_noise_std=1.0, use_semantics=True, sem_layer=4, sem_with_coord=True).to(dev).train()
    if head_only:
        for n, p in net.named_parameters(): p.requires_grad_("semantic_linear" in n)
    opt = torch.optim.Adam([p for p in net.parameters() if p.requires_grad], lr=5e-4)
    rays = syn.synthetic_rays(4096, seed=0, device=dev); gt = torch.rand(4096, 2, device=dev)
    def step():
        opt.zero_grad(); ret = net(rays, (syn.NEAR, syn.FAR), retraw=False)
        (((ret["semantics"] - gt) ** 2).mean() + ((ret["semantics0"] - gt) ** 2).mean()).backward(); opt.step()
    for _ in range(3): step()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(5): step()
    torch.cuda.synchronize(); print("8x256 four-Linear head,", "head only" if head_only else "all parameters", round((time.perf_counter() - t0) / 5 * 1e3, 2), "ms per 4096-ray step")
