"""Phase cycles of sem_head_wgrad16_kernel from a diagnostic build (scripts/diag/build_variant.sh wg_prof sem_wgrad16.hip
-DNSOS_WG_PROF -> exp/lib_wg_prof.so): s_memtime stamps around the phases of a step, workgroup 1, every wave; mean shader-clock
ticks per step and phase.  (A stamp is taken when the scalar unit reaches it: time spent waiting for an earlier instruction's
operands shows up in the phase that waits, not the one that issued.)"""
import ctypes, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
os.environ["NERF_SOS_HIP_LIB"] = os.path.join(ROOT, "exp", "lib_wg_prof.so")
import torch
from nerf_sos_amd import ops
dev = torch.device("cuda:0")
R, S = 4096, 192
P = R * S
w = torch.rand(R, S, device=dev) / S
g = torch.randn(R, 2, device=dev) * 1e-4
w2 = torch.randn(2, 128, device=dev) * 0.1
hid = torch.relu(torch.randn(P, 128, device=dev)).to(torch.bfloat16)
x = torch.randn(P, 320, device=dev).to(torch.bfloat16)
for _ in range(3):
    ops.sem_head_wgrad(w, g, w2, hid, x, split_fp16=True)
torch.cuda.synchronize()
lib = ctypes.CDLL(os.environ["NERF_SOS_HIP_LIB"])
buf = (ctypes.c_ulonglong * 64)()
assert lib.nsos_wg_prof_read(buf) == 0
names_g = ["steps", "issue reads", "wait fetch", "stage", "fetch", "mfma issue", "wait reads", "barrier"]
names_x = ["steps", "issue reads", "mfma issue", "wait fetch", "stage", "fetch", "wait reads", "barrier"]
for wv in range(8):
    n = buf[wv * 8]
    names = names_g if wv < 4 else names_x
    print(f"wave {wv} ({'g' if wv < 4 else 'x'}): {n} steps; ticks per step: " + ", ".join(f"{names[k]} {buf[wv * 8 + k] / max(n, 1):.0f}" for k in range(1, 8))
          + f"; total {sum(buf[wv * 8 + k] for k in range(1, 8)) / max(n, 1):.0f}")
