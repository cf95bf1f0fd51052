"""lp8 (two waves per SIMD) against lp4 (round-1 kernel): bitwise equality and launch time (diagnostic / A-B)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import nerf_sos_amd
from nerf_sos_amd import _lib, ops, synthetic as syn
dev = "cuda:0"
R = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
for sem, kw in ((0, dict(use_semantics=False)), (2, dict(use_semantics=True, sem_with_coord=True)), (1, dict(use_semantics=True))):
    torch.manual_seed(0)
    net = nerf_sos_amd.NeRFNet(N_samples=64, N_importance=128, **kw).to(dev).eval()
    syn.spiky_density_(net, 8.0, 0.5)
    rays = syn.synthetic_rays(R, seed=0, device=dev)
    near = torch.full((R,), syn.NEAR, device=dev); far = torch.full((R,), syn.FAR, device=dev)
    z, v = ops.ray_setup(rays[1].contiguous(), near, far, 192, None)
    o, d = rays[0].contiguous(), rays[1].contiguous()
    for prec in ("fp16", "bf16"):
        pk = net.nerf_fine.packed_weights(prec)
        res = {}
        for wps in (1, 2):
            _lib.check(_lib.lib().nsos_mlp_lp_select_kernel(wps), "select")
            for _ in range(3):
                a = ops.mlp_forward_rays_lp(pk, sem, prec, o, d, v, z)
            torch.cuda.synchronize()
            ev = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
            ev[0].record()
            for _ in range(10):
                a = ops.mlp_forward_rays_lp(pk, sem, prec, o, d, v, z)
            ev[1].record(); torch.cuda.synchronize()
            b = ops.mlp_forward_rays_lp(pk, sem, prec, o, d, v, z)
            res[wps] = (a, ev[0].elapsed_time(ev[1]) / 10, bool(torch.equal(a, b)))
        mac = {0: 593408, 1: 626432, 2: 634496}[sem]
        tf = lambda ms: 2 * mac * R * 192 / (ms * 1e-3) / 1e12
        same = torch.equal(res[1][0], res[2][0])
        md = float((res[1][0] - res[2][0]).abs().max())
        print(f"sem {sem} {prec}: lp4 {res[1][1]:.4f} ms ({tf(res[1][1]):.0f} TF)  lp8 {res[2][1]:.4f} ms ({tf(res[2][1]):.0f} TF)  "
              f"bit-identical {same} (max diff {md:.3e})  deterministic lp4 {res[1][2]} lp8 {res[2][2]}  finite {bool(torch.isfinite(res[2][0]).all())}")
    # SAVE variants
    if sem:
        for prec in ("bf16",):
            pk = net.nerf_fine.packed_weights(prec)
            outs = {}
            for wps in (1, 2):
                _lib.check(_lib.lib().nsos_mlp_lp_select_kernel(wps), "select")
                for compact in (False, True):
                    outs[(wps, compact)] = ops.mlp_forward_rays_save(pk, sem, o[:300], d[:300], v[:300], z[:300].contiguous(), prec, compact=compact)
            for compact in (False, True):
                rows = lambda t: ops.sem_in_rows(t, 300 * z.shape[1]) if t.dim() == 4 else t      # lp8's compact sem_in is tile-major
                eq = [bool(torch.equal(rows(x), rows(y))) for x, y in zip(outs[(1, compact)], outs[(2, compact)])]
                print(f"   SAVE sem {sem} {prec} compact={compact}: raw / sem_in / sem_hid identical: {eq}")
_lib.check(_lib.lib().nsos_mlp_lp_select_kernel(2), "select")
