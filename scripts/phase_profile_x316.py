#!/usr/bin/env python3
"""GPU box: per-phase cycle attribution of the split-fp16 16x16x32 kernel (mlp_x316_kernel; nsos_mlp_profile_rays_x3 with
nsos_mlp_x3_select_kernel(2)).  usage: phase_profile_x316.py [sem_mode 0|1|2]"""
import ctypes as C
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import nerf_sos_amd
from nerf_sos_amd import _lib, ops
from nerf_sos_amd import synthetic as syn

sem = int(sys.argv[1]) if len(sys.argv) > 1 else 0
NW = 8
dev = torch.device("cuda:0")
torch.manual_seed(0)
net = nerf_sos_amd.NeRFNet(N_samples=64, N_importance=128, use_semantics=sem > 0, sem_with_coord=sem == 2).to(dev).eval()
R = 4096
rays = syn.synthetic_rays(R, seed=0, device=dev)
near = torch.full((R,), syn.NEAR, device=dev)
far = torch.full((R,), syn.FAR, device=dev)
z, v = ops.ray_setup(rays[1], near, far, 192, None)
packed = net.nerf_fine.packed_weights("fp16x3")
raw = torch.empty(R, 192, 6 if sem else 4, device=dev)
stamps = torch.zeros(16 * 64, dtype=torch.int64, device=dev)   # 2 blocks x 8 waves
_lib.check(_lib.lib().nsos_mlp_x3_select_kernel(2), 'select')
P = lambda t: C.c_void_p(t.data_ptr())
o_, d_ = rays[0].contiguous(), rays[1].contiguous()
call = lambda: _lib.check(_lib.lib().nsos_mlp_profile_rays_x3(P(packed), sem, P(o_), P(d_), P(v), P(z), R, 192, P(raw), P(stamps), None), "profile")
for _ in range(3):
    call()
torch.cuda.synchronize()
ev = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
ev[0].record()
for _ in range(10):
    call()
ev[1].record()
torch.cuda.synchronize()
launch_ms = ev[0].elapsed_time(ev[1]) / 10
st = stamps.cpu().view(16, 64).numpy()
M = 16  # cycles of one 16x16x32 MFMA; three per (tile, slice) item
names, ideal = ["tile start", "inputs + xyz enc", "L0 mfma", "L0 act"], {"L0 mfma": 32 * 3 * M}
for l in range(1, 9):
    names += [f"L{l} mfma", f"L{l} act"]
    ideal[f"L{l} mfma"] = (160 if l == 5 else 128) * 3 * M
    if l == 7:
        names.append("sigma+sem heads")
        ideal["sigma+sem heads"] = {0: 8, 1: 64 + 8 + 4, 2: 80 + 8 + 4}[sem] * 3 * M
names += ["dir enc", "view mfma", "rgb mfma", "stores"]
ideal["view mfma"] = 72 * 3 * M
ideal["rgb mfma"] = 4 * 3 * M
print(f"# mlp_x316_kernel: 8 waves x 16 points (two per SIMD) on v_mfma_f32_16x16x32, split fp16, sem_mode {sem}; cycles per phase of one 128-point tile, per wave; ideal = this wave's MFMA cycles")
print(f"{'phase':18s}" + "".join(f" w{w:<8d}" for w in range(NW)) + "   ideal_mfma")
tot = [0] * NW
for k in range(1, len(names)):
    d = [int(st[w, k] - st[w, k - 1]) for w in range(NW)]
    for w in range(NW):
        tot[w] += d[w]
    print(f"{names[k]:18s}" + "".join(f" {x:<9d}" for x in d) + f"   {ideal.get(names[k], 0)}")
print(f"{'total':18s}" + "".join(f" {x:<9d}" for x in tot) + f"   {sum(ideal.values())}")
print(f"matrix-pipe time of the tile per SIMD (both waves' MFMAs): {sum(ideal.values()) * 2}; wall per tile (wave 0): {tot[0]}  -> pipe busy {sum(ideal.values()) * 2 / tot[0]:.3f}")
whole = int(st[0, 63] - st[0, 62])
tiles = R * 192 / 128 / 256
print(f"whole kernel, block 0 wave 0: {whole} shader cycles for {tiles:.0f} tiles = {whole / tiles:.0f} per tile; launch {launch_ms:.4f} ms -> effective shader clock {whole / launch_ms / 1e6:.3f} GHz")
