"""A/B of the two split-fp16 forward kernels (nsos_mlp_x3_select_kernel: 1 = mlp_x3_kernel, 32x32x16; 2 = mlp_x316_kernel, 16x16x32):
HIP-event time of the fine pass at the C2 batch (4096 rays x 192, no semantics / sem+coord), of a C5 chunk's coarse pass (65 536 rays
x 64, sem+coord) and fine pass, issued fraction of the 16-bit pipe (3 MFMAs per product against 2.5 PFLOP/s), and the agreement of
the two kernels with each other and with the exact fp32 kernel on the same points."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import nerf_sos_amd
from nerf_sos_amd import ops, synthetic as syn, _lib
dev = "cuda:0"
MAC = {0: 593408, 2: 634496}


def clock(fn, n=20):
    for _ in range(3):
        fn()
    best = 1e9
    for _ in range(3):
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(n):
            fn()
        e1.record(); torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1) / n)
    return best


for R, S, sem in ((4096, 192, 0), (4096, 192, 2), (4096, 64, 0), (65536, 64, 2), (65536, 192, 2)):
    torch.manual_seed(0)
    kw = dict(use_semantics=True, sem_with_coord=True) if sem else dict(use_semantics=False)
    net = nerf_sos_amd.NeRFNet(N_samples=64, N_importance=128, **kw).to(dev).eval()
    rays = syn.synthetic_rays(R, seed=0, device=dev)
    o, d = rays[0].contiguous(), rays[1].contiguous()
    near, far = torch.full((R,), syn.NEAR, device=dev), torch.full((R,), syn.FAR, device=dev)
    z, v = ops.ray_setup(d, near, far, S, None)
    pk = net.nerf_fine.packed_weights("fp16x3")
    exact = ops.mlp_forward_rays(net.nerf_fine.packed_weights(), net.nerf_fine.sem_mode, o, d, v, z)
    outs = {}
    for k in (1, 2):
        _lib.check(_lib.lib().nsos_mlp_x3_select_kernel(k), "select")
        outs[k] = ops.mlp_forward_rays_lp(pk, net.nerf_fine.sem_mode, "fp16x3", o, d, v, z).clone()
        ms = clock(lambda: ops.mlp_forward_rays_lp(pk, net.nerf_fine.sem_mode, "fp16x3", o, d, v, z), 20 if R * S < 4e6 else 5)
        tf = 2 * MAC[sem] * R * S / ms / 1e9
        print(f"R={R:6d} S={S:3d} sem={sem} kernel {k}: {ms:8.4f} ms  useful {tf / 2500:.4f} issued {3 * tf / 2500:.4f} of 2.5 PF | "
              f"max |raw - exact| {float((outs[k] - exact).abs().max()):.3e} finite {bool(torch.isfinite(outs[k]).all())}", flush=True)
    print(f"          max |kernel 2 - kernel 1| {float((outs[2] - outs[1]).abs().max()):.3e}   (max |exact| {float(exact.abs().max()):.3f})", flush=True)
