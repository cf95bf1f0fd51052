#!/usr/bin/env python3
"""GPU box: per-phase cycle attribution of the split-fp16 MLP kernel (nsos_mlp_profile_rays_x3 stamps: shader-clock
cycles of one 128-point tile in steady state, so independent of the clock the chip happens to run at).
usage: phase_profile_x3.py [sem_mode 0|1|2]"""
import ctypes as C
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import nerf_sos_amd
from nerf_sos_amd import _lib, ops
from nerf_sos_amd import synthetic as syn

sem = int(sys.argv[1]) if len(sys.argv) > 1 else 0
prec = "fp16x3"
dev = torch.device("cuda:0")
torch.manual_seed(0)
net = nerf_sos_amd.NeRFNet(N_samples=64, N_importance=128, use_semantics=sem > 0, sem_with_coord=sem == 2).to(dev).eval()
R = 4096
rays = syn.synthetic_rays(R, seed=0, device=dev)
near = torch.full((R,), syn.NEAR, device=dev)
far = torch.full((R,), syn.FAR, device=dev)
z, v = ops.ray_setup(rays[1], near, far, 192, None)
packed = net.nerf_fine.packed_weights(prec)
raw = torch.empty(R, 192, 6 if sem else 4, device=dev)
stamps = torch.zeros(16 * 64, dtype=torch.int64, device=dev)
P = lambda t: C.c_void_p(t.data_ptr())

for _ in range(3):
    _lib.check(_lib.lib().nsos_mlp_profile_rays_x3(P(packed), sem, P(rays[0].contiguous()), P(rays[1].contiguous()),
                                                  P(v), P(z), R, 192, P(raw), P(stamps), None), "profile")
torch.cuda.synchronize()
st = stamps.cpu().view(16, 64).numpy()
M = 32  # cycles of one 32x32x16 MFMA
names, ideal = ["tile start", "inputs + xyz enc", "L0 mfma", "L0 act"], {"L0 mfma": 32 * 3 * M}   # an item = 3 MFMAs (a bias item 2)
for l in range(1, 9):
    names += [f"L{l} mfma", f"L{l} act"]
    ideal[f"L{l} mfma"] = (8 * 2 + 128 * 3 + (32 * 3 if l == 5 else 0)) * M
    if l == 7:
        names.append("sigma+sem heads")
        ideal["sigma+sem heads"] = {0: 0, 1: 4 * 2 + 64 * 3, 2: 4 * 2 + 64 * 3 + 16 * 3}[sem] * M
names += ["view mfma", "dir enc", "dir mfma", "rgb head + store"]
ideal["view mfma"] = (4 * 2 + 64 * 3) * M
ideal["dir mfma"] = 8 * 3 * M
print(f"{'phase':18s}" + "".join(f" w{w:<8d}" for w in range(4)) + "   ideal_mfma")
tot = [0] * 4
for k in range(1, len(names)):
    d = [int(st[w, k] - st[w, k - 1]) for w in range(4)]
    for w in range(4):
        tot[w] += d[w]
    print(f"{names[k]:18s}" + "".join(f" {x:<9d}" for x in d) + f"   {ideal.get(names[k], 0)}")
print(f"{'total':18s}" + "".join(f" {x:<9d}" for x in tot) + f"   {sum(ideal.values())}")
