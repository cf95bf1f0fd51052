#!/usr/bin/env python3
"""GPU box: per-phase cycle attribution of the fused MLP kernel (nsos_mlp_profile_rays stamps)."""
import ctypes as C
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import nerf_sos_amd
from nerf_sos_amd import _lib, ops
from nerf_sos_amd import synthetic as syn

sem = int(sys.argv[1]) if len(sys.argv) > 1 else 0
dev = torch.device("cuda:0")
torch.manual_seed(0)
net = nerf_sos_amd.NeRFNet(N_samples=64, N_importance=128, use_semantics=sem > 0, sem_with_coord=sem == 2).to(dev).eval()
rays = syn.synthetic_rays(4096, seed=0, device=dev)
near = torch.full((4096,), syn.NEAR, device=dev)
far = torch.full((4096,), syn.FAR, device=dev)
z, v = ops.ray_setup(rays[1], near, far, 192, None)
packed = net.nerf_fine.packed_weights()
raw = torch.empty(4096, 192, 6 if sem else 4, device=dev)
stamps = torch.zeros(16 * 64, dtype=torch.int64, device=dev)
P = lambda t: C.c_void_p(t.data_ptr())
for _ in range(3):
    _lib.check(_lib.lib().nsos_mlp_profile_rays(P(packed), sem, P(rays[0].contiguous()), P(rays[1].contiguous()), P(v), P(z),
                                               4096, 192, P(raw), P(stamps), None), "profile")
torch.cuda.synchronize()
st = stamps.cpu().view(16, 64).numpy()
names = ["tile start", "inputs+encoding", "L0 (enc 2ch)"]
ideal = {"L0 (enc 2ch)": 34 * 256 + 8192}
H = 34 * 256 + 7 * 8192  # a 256->256 layer: 8 chunks, the first with the bias k-step
for l in range(1, 9):
    nm = f"L{l}" + (" (+enc 2ch)" if l == 5 else "") + (" feature" if l == 8 else "")
    names.append(nm)
    ideal[nm] = H + (2 * 8192 if l == 5 else 0) + ({0: 0, 1: 33 * 256 + 3 * 8192, 2: 33 * 256 + 4 * 8192}[sem] if l == 8 else 0)
names += ["views+rgb", "store"]
ideal["views+rgb"] = 33 * 256 + 3 * 8192 + 4096
n = len(names)
print(f"{'phase':18s}" + "".join(f" w{w:<8d}" for w in range(4)) + "   ideal_mfma")
tot = [0] * 4
for k in range(1, n):
    d = [int(st[w, k] - st[w, k - 1]) for w in range(4)]
    for w in range(4):
        tot[w] += d[w]
    print(f"{names[k]:18s}" + "".join(f" {x:<9d}" for x in d) + f"   {ideal.get(names[k], 0)}")
print(f"{'total':18s}" + "".join(f" {x:<9d}" for x in tot) + f"   {sum(ideal.values())}")
print("wave start skew (cycles):", [int(st[w, 0] - st[0, 0]) for w in range(4)])
