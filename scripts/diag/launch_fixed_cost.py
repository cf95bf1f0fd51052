"""Fixed cost per launch of mlp_fused_kernel: time the kernel for 1, 2, 4, 8, 16, 24 tiles per workgroup (256 persistent workgroups,
128-point tiles) and fit t = a + b * tiles.  a is what a launch pays once (dispatch, prologue, weight-stream ramp, tail skew)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import torch
import nerf_sos_amd
from nerf_sos_amd import ops, synthetic as syn

dev = torch.device("cuda:0")
net = nerf_sos_amd.NeRFNet(N_samples=64, N_importance=128).to(dev).eval()
packed = net.nerf_fine.packed_weights()
S = 64
res = []
for tiles in (1, 2, 3, 4, 8, 16, 24, 48):
    R = 256 * tiles * 128 // S
    rays = syn.synthetic_rays(R, seed=0, device=dev)
    near, far = torch.full((R,), syn.NEAR, device=dev), torch.full((R,), syn.FAR, device=dev)
    z, v = ops.ray_setup(rays[1], near, far, S, None)
    o, d = rays[0].contiguous(), rays[1].contiguous()
    for _ in range(5):
        ops.mlp_forward_rays(packed, 0, o, d, v, z)
    n = 40
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(n)]
    for a, b in ev:
        a.record(); ops.mlp_forward_rays(packed, 0, o, d, v, z); b.record()
    torch.cuda.synchronize()
    t = np.array([a.elapsed_time(b) for a, b in ev]) * 1e3
    # back to back, no events in between: the launch-to-launch period
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n):
        ops.mlp_forward_rays(packed, 0, o, d, v, z)
    b.record(); torch.cuda.synchronize()
    per = a.elapsed_time(b) * 1e3 / n
    res.append((tiles, np.median(t), t.min(), per))
    print(f"tiles/WG {tiles:3d}  rays {R:6d}  event-bracketed median {np.median(t):8.1f} us  min {t.min():8.1f}  back-to-back period {per:8.1f} us  "
          f"per tile {per / tiles:7.1f} us  TFLOP/s {2 * 593408 * R * S / per / 1e6:6.1f}")
x = np.array([r[0] for r in res], float); y = np.array([r[3] for r in res], float)
b, a = np.polyfit(x, y, 1)
print(f"fit of the back-to-back period: {a:.1f} us + {b:.1f} us per tile  (ideal tile at 157.3 TFLOP/s: {2 * 593408 * 128 * 256 / 157.3e6:.1f} us)")
