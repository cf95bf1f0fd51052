"""Throughput of the generic-architecture kernel (mlp_generic_kernel) against the fp32-MFMA peak, for a few architectures, next to the
tuned exact kernel on the shipped one (diagnostic).  MACs are counted from the module's Linear layers (what the reference computes)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import nerf_sos_amd
from nerf_sos_amd import ops, synthetic as syn
dev = "cuda:0"
R, S = 4096, 192
PEAK = 157.3
rays = syn.synthetic_rays(R, seed=0, device=dev)
near = torch.full((R,), syn.NEAR, device=dev); far = torch.full((R,), syn.FAR, device=dev)
z, v = ops.ray_setup(rays[1].contiguous(), near, far, S, None)
o, d = rays[0].contiguous(), rays[1].contiguous()
CASES = [("shipped 8x256 (tuned exact kernel)", dict()),
         ("8x256, multires 6 / 4", dict(multires=6)),
         ("8x256, no view directions", dict(viewdirs=False)),
         ("4x128, skips [2]", dict(netdepth=4, netwidth=128, netdepth_fine=4, netwidth_fine=128, skips=[2]) if False else dict(netdepth=4, netwidth=128, netdepth_fine=4, netwidth_fine=128)),
         ("8x256 sem_layer 3, sem_dim 5", dict(use_semantics=True, sem_layer=3, sem_dim=5)),
         ("8x512", dict(netwidth=512, netwidth_fine=512)),
         ("16x256", dict(netdepth=16, netdepth_fine=16))]
for name, kw in CASES:
    torch.manual_seed(0)
    try:
        net = nerf_sos_amd.NeRFNet(N_samples=64, N_importance=128, **kw).to(dev).eval()
    except Exception as e:
        print(f"{name:40s} construction failed: {e!r}"[:200]); continue
    mlp = net.nerf_fine
    mac = sum(m.in_features * m.out_features for m in mlp.mlp.modules() if isinstance(m, torch.nn.Linear))
    run = (lambda: ops.mlp_forward_rays(mlp.packed_weights(), mlp.sem_mode, o, d, v, z)) if mlp.fast else (lambda: mlp.query_rays(o, d, v, z))
    try:
        for _ in range(3):
            run()
    except NotImplementedError as e:
        print(f"{name:40s} refused: {str(e)[:150]}", flush=True)
        continue
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
    ev[0].record()
    for _ in range(5):
        run()
    ev[1].record(); torch.cuda.synchronize()
    ms = ev[0].elapsed_time(ev[1]) / 5
    tf = 2 * mac * R * S / (ms * 1e-3) / 1e12
    print(f"{name:40s} {'tuned' if mlp.fast else 'generic':8s} {mac:9d} MAC/point  {ms:8.3f} ms  {tf:7.1f} TFLOP/s = {tf / PEAK:.3f} of the fp32-MFMA peak", flush=True)
