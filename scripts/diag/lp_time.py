"""Interleaved A/B/C timing of the three 16-bit kernels (1 = lp4, 2 = lp8, 3 = lp16) on the C3/C5 fine-pass shape (diagnostic)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import nerf_sos_amd
from nerf_sos_amd import _lib, ops, synthetic as syn
dev = "cuda:0"
R = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 5
KERNELS = tuple(int(c) for c in sys.argv[3]) if len(sys.argv) > 3 else (1, 2, 3)      # e.g. "3": mlp_lp16_kernel only (A/B libraries)
for sem, kw in ((0, dict(use_semantics=False)), (2, dict(use_semantics=True, sem_with_coord=True))):
    torch.manual_seed(0)
    net = nerf_sos_amd.NeRFNet(N_samples=64, N_importance=128, **kw).to(dev).eval()
    syn.spiky_density_(net, 8.0, 0.5)
    rays = syn.synthetic_rays(R, seed=0, device=dev)
    near = torch.full((R,), syn.NEAR, device=dev); far = torch.full((R,), syn.FAR, device=dev)
    z, v = ops.ray_setup(rays[1].contiguous(), near, far, 192, None)
    o, d = rays[0].contiguous(), rays[1].contiguous()
    mac = {0: 593408, 2: 634496}[sem]
    for prec in ("fp16", "bf16"):
        pk = net.nerf_fine.packed_weights(prec)
        best = {1: 1e9, 2: 1e9, 3: 1e9}
        for rep in range(reps):
            for wps in KERNELS:
                _lib.check(_lib.lib().nsos_mlp_lp_select_kernel(wps), "select")
                for _ in range(5):
                    ops.mlp_forward_rays_lp(pk, sem, prec, o, d, v, z)
                ev = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
                ev[0].record()
                for _ in range(40):
                    ops.mlp_forward_rays_lp(pk, sem, prec, o, d, v, z)
                ev[1].record(); torch.cuda.synchronize()
                best[wps] = min(best[wps], ev[0].elapsed_time(ev[1]) / 40)
        tf = lambda ms: 2 * mac * R * 192 / (ms * 1e-3) / 1e12
        print(f"sem {sem} {prec} R={R}: " + "   ".join(f"{ {1: 'lp4', 2: 'lp8', 3: 'lp16'}[k]} {best[k]:.4f} ms ({tf(best[k]):.0f} TF, {tf(best[k])/25.166:.1f} %)"
                                                         for k in KERNELS), flush=True)
_lib.check(_lib.lib().nsos_mlp_lp_select_kernel(3), "select")
