"""Whole-step graph vs eager: ms per step, C3 (1 patch) and C4-per-GPU (2 patches), bf16; capture with / without the
appearance loss on a side stream."""
import sys, os, time, types
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import nerf_sos_amd
from nerf_sos_amd import synthetic as syn
DEV = "cuda:0"
args = types.SimpleNamespace(rand_neg=False, self_corr_w=0, use_sim_matrix=True, patch_stride=6, app_corr_params=["0.18", "1", "0.46", "1"], geo_corr_params=["0.5", "1", "3", "1"])
# optional: graph_step_time.py <B> <capture 0|1> <overlap 0|1>  -> that one mode only (for a kernel trace)
MODES = [(int(sys.argv[1]), bool(int(sys.argv[2])), bool(int(sys.argv[3])))] if len(sys.argv) > 3 else [(B, c, o) for B in (1, 2) for c, o in ((False, True), (True, False), (True, True))]
for B, capture, overlap in MODES:
    if True:
        torch.manual_seed(0)
        net = nerf_sos_amd.NeRFNet(N_samples=64, N_importance=128, use_semantics=True, sem_with_coord=True, perturb=1.0, raw_noise_std=1.0, ray_chunk=1 << 20).to(DEV)
        for n_, p_ in net.named_parameters():
            p_.requires_grad = "semantic_linear" in n_
        net.train(); net.mlp_precision = "bf16"; net.rng = "philox"
        opt = torch.optim.Adam([p for p in net.parameters() if p.requires_grad], lr=5e-4, fused=True, capturable=True)
        rays = syn.synthetic_patches(B, 64, 6, seed=0, device=DEV)
        feat = torch.randn(B, 384, 14, 14, device=DEV); cls_ = torch.randn(B, 384, device=DEV) + 3 * torch.randn(1, 384, device=DEV)
        corr, geo = nerf_sos_amd.CorrelationLoss(args), nerf_sos_amd.GeoCorrelationLoss(args)
        con = nerf_sos_amd.NeRFContrastive(device=DEV) if B >= 2 else None
        try:
            g = nerf_sos_amd.GraphedPatchStep(net, opt, rays, (syn.NEAR, syn.FAR), feat, cls_, corr, geo, con, contrast_w=0.01, overlap_losses=overlap, warmup=5, capture=capture)
        except Exception as ex:
            print(f"B={B} capture={capture} overlap={overlap}: FAILED {type(ex).__name__}: {str(ex)[:300]}")
            continue
        for _ in range(5):
            g()
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(50):
            g()
        host = time.perf_counter() - t0
        torch.cuda.synchronize(); dt = time.perf_counter() - t0
        print(f"B={B} capture={capture} overlap={overlap}: {dt / 50 * 1e3:.3f} ms/step (host enqueue {host / 50 * 1e3:.3f} ms), loss {float(g.loss):.6f}")
