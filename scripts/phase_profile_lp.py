#!/usr/bin/env python3
"""GPU box: per-phase cycle attribution of the reduced-precision MLP kernel (nsos_mlp_profile_rays_lp stamps).
usage: phase_profile_lp.py [sem_mode 0|1|2] [fp16|bf16] [kernel 3|2|1] [--save]     (3 = mlp_lp16_kernel, the default; 2 = mlp_lp8_kernel; 1 = mlp_lp_kernel)"""
import ctypes as C
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import nerf_sos_amd
from nerf_sos_amd import _lib, ops
from nerf_sos_amd import synthetic as syn

sem = int(sys.argv[1]) if len(sys.argv) > 1 else 0
prec = sys.argv[2] if len(sys.argv) > 2 else "fp16"
wps = int(sys.argv[3]) if len(sys.argv) > 3 and sys.argv[3].isdigit() else 3      # 3: mlp_lp16_kernel, 2: mlp_lp8_kernel (both 8 waves x 32 points), 1: mlp_lp_kernel (4 x 64)
NW, COLS = (8, 1) if wps >= 2 else (4, 2)
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
stamps = torch.zeros(16 * 64, dtype=torch.int64, device=dev)   # 16 rows: 4 blocks x 4 waves, or 2 blocks x 8 waves
_lib.check(_lib.lib().nsos_mlp_lp_select_kernel(wps), 'select')
P = lambda t: C.c_void_p(t.data_ptr())
dt = {"fp16": 1, "bf16": 2}[prec]
SAVE = "--save" in sys.argv          # the training (SAVE) variant: stamps through nsos_mlp_lp_set_stamp_buffer
if SAVE:
    assert sem > 0
    _lib.check(_lib.lib().nsos_mlp_lp_set_stamp_buffer(P(stamps)), "stamps")
    o_, d_ = rays[0].contiguous(), rays[1].contiguous()
    for _ in range(3):
        keep = ops.mlp_forward_rays_save(packed, sem, o_, d_, v, z, prec, compact=True)
    torch.cuda.synchronize()
    _lib.check(_lib.lib().nsos_mlp_lp_set_stamp_buffer(None), "stamps")
for _ in range(0 if SAVE else 3):
    _lib.check(_lib.lib().nsos_mlp_profile_rays_lp(P(packed), sem, dt, P(rays[0].contiguous()), P(rays[1].contiguous()),
                                                  P(v), P(z), R, 192, P(raw), P(stamps), None), "profile")
torch.cuda.synchronize()
ev = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
ev[0].record()
for _ in range(0 if SAVE else 10):
    _lib.check(_lib.lib().nsos_mlp_profile_rays_lp(P(packed), sem, dt, P(rays[0].contiguous()), P(rays[1].contiguous()),
                                                  P(v), P(z), R, 192, P(raw), P(stamps), None), "profile")
ev[1].record()
torch.cuda.synchronize()
launch_ms = ev[0].elapsed_time(ev[1]) / 10 if not SAVE else float('nan')
st = stamps.cpu().view(16, 64).numpy()
M = 32  # cycles of one 32x32x16 MFMA
if wps == 3:
    # mlp_lp16_kernel: 16-cycle MFMAs (16x16x32), two per A operand; no bias MFMAs; heads on the matrix pipe
    M = 16
    names, ideal = ["tile start", "inputs + xyz enc", "L0 mfma", "L0 act"], {"L0 mfma": 64 * M}
    for l in range(1, 9):
        names += [f"L{l} mfma", f"L{l} act"]
        ideal[f"L{l} mfma"] = (320 if l == 5 else 256) * M
        if l == 7:
            if os.environ.get("NSOS_LP16_HEAD_STAMPS") and sem:      # a -DNSOS_LP16_HEAD_STAMPS build: the heads' chunks one by one
                names += ["sem0 h chunk 0", "sem0 h chunk 1"]
                ideal["sem0 h chunk 0"] = ideal["sem0 h chunk 1"] = 64 * M
            names.append("sigma+sem heads")
            ideal["sigma+sem heads"] = {0: 16, 1: 128 + 16 + 16 + 8, 2: 160 + 16 + 8}[sem] * M - (128 * M if os.environ.get("NSOS_LP16_HEAD_STAMPS") and sem else 0)
    names += ["dir enc", "view mfma", "rgb mfma", "stores"]
    ideal["view mfma"] = 144 * M
    ideal["rgb mfma"] = 8 * M      # (operands resident in LDS: no chunk of its own)
    title = "mlp_lp16_kernel: 8 waves x 32 points (two per SIMD) on v_mfma_f32_16x16x32"
else:
    names, ideal = ["tile start", "inputs + xyz enc", "L0 mfma", "L0 act"], {"L0 mfma": 32 * COLS * M}
    for l in range(1, 9):
        names += [f"L{l} mfma", f"L{l} act"]
        ideal[f"L{l} mfma"] = (136 + (32 if l == 5 else 0)) * COLS * M
        if l == 7:
            names.append("sigma+sem heads")
            ideal["sigma+sem heads"] = {0: 0, 1: 68, 2: 84}[sem] * COLS * M
    names += ["view mfma", "dir enc", "dir mfma", "rgb head + store"]
    ideal["view mfma"] = 68 * COLS * M
    ideal["dir mfma"] = 8 * COLS * M
    title = 'mlp_lp8_kernel: 8 waves x 32 points (two per SIMD: waves w and w+4 share one matrix pipe)' if wps == 2 else 'mlp_lp_kernel: 4 waves x 64 points'
print(f"# {title}, sem_mode {sem}, {prec}{', training (SAVE) variant' if SAVE else ''}; cycles per phase of one 256-point tile, per wave; ideal = this wave's MFMA cycles")
print(f"{'phase':18s}" + "".join(f" w{w:<8d}" for w in range(NW)) + "   ideal_mfma")
tot = [0] * NW
for k in range(1, len(names)):
    d = [int(st[w, k] - st[w, k - 1]) for w in range(NW)]
    for w in range(NW):
        tot[w] += d[w]
    print(f"{names[k]:18s}" + "".join(f" {x:<9d}" for x in d) + f"   {ideal.get(names[k], 0)}")
print(f"{'total':18s}" + "".join(f" {x:<9d}" for x in tot) + f"   {sum(ideal.values())}")
print(f"matrix-pipe time of the tile per SIMD (both waves' MFMAs): {sum(ideal.values()) * (2 if wps >= 2 else 1)}; "
      f"wall per tile (wave 0): {tot[0]}  -> pipe busy {sum(ideal.values()) * (2 if wps >= 2 else 1) / tot[0]:.3f}")
if wps >= 2:
    whole = int(st[0, 63] - st[0, 62])
    print(f"whole kernel, block 0 wave 0: {whole} shader cycles for {R * 192 // 256 // 256} tiles = {whole / (R * 192 / 256 / 256):.0f} per tile; launch {launch_ms:.4f} ms "
          f"-> effective shader clock {whole / launch_ms / 1e6:.3f} GHz")
