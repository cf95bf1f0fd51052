"""Per-op phase cycles of mlp_generic_bwd_kernel (the last generic launch of a training step's backward: the coarse net's chain) from
the diagnostic build ab/lib_gen_prof.so (see gen_prof.py).  Slot 1 = loading d loss / d raw, slots 2.. = the reversed program's ops
(head steps, transposed dense ops), by wave: work ticks and barrier-wait ticks per tile."""
import ctypes, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
os.environ["NERF_SOS_HIP_LIB"] = os.path.join(ROOT, "ab", "lib_gen_prof.so")
import torch
import nerf_sos_amd
from nerf_sos_amd import synthetic as syn
dev = torch.device("cuda:0")
R = 4096
CASES = {"8x256nov": dict(viewdirs=False), "8x256m6": dict(multires=6), "4x128": dict(netdepth=4, netwidth=128, netdepth_fine=4, netwidth_fine=128),
         "6x96": dict(netdepth=6, netwidth=96, netdepth_fine=6, netwidth_fine=96, multires=6, multires_views=2)}
lib = ctypes.CDLL(os.environ["NERF_SOS_HIP_LIB"])
for name in (sys.argv[1:] or ["8x256m6"]):
    torch.manual_seed(0)
    net = nerf_sos_amd.NeRFNet(N_samples=64, N_importance=128, perturb=1.0, raw_noise_std=1.0, **CASES[name]).to(dev).train()
    rays = syn.synthetic_rays(R, seed=0, device=dev)
    gt = torch.rand(R, 3, device=dev)
    for _ in range(2):
        net.zero_grad()
        ret = net(rays, (syn.NEAR, syn.FAR), retraw=False)
        (((ret["rgb"] - gt) ** 2).mean() + ((ret["rgb0"] - gt) ** 2).mean()).backward()
    torch.cuda.synchronize()
    buf = (ctypes.c_ulonglong * 512)()
    assert lib.nsos_gen_prof_read(buf) == 0
    g = lambda w, k, s: buf[(w * 2 + k) * 64 + s]
    tiles = g(0, 0, 0)
    print(f"== {name} backward chain (coarse net): {tiles} tiles by workgroup 1")
    tot_w = [0] * 4; tot_b = [0] * 4
    for s in range(1, 63):
        if not any(g(w, k, s) for w in range(4) for k in range(2)):
            continue
        lab = "load g" if s == 1 else ("epilog" if s == 62 else f"op {s - 2:2d}")
        ws = [g(w, 0, s) / tiles for w in range(4)]; bs = [g(w, 1, s) / tiles for w in range(4)]
        for w in range(4):
            tot_w[w] += ws[w]; tot_b[w] += bs[w]
        print(f"  {lab:7s} work " + " ".join(f"{x:8.0f}" for x in ws) + "   barrier " + " ".join(f"{x:7.0f}" for x in bs) + f"   op total {max(ws[w] + bs[w] for w in range(4)):8.0f}")
    print("  total   work " + " ".join(f"{x:8.0f}" for x in tot_w) + "   barrier " + " ".join(f"{x:7.0f}" for x in tot_b) + f"   tile {tot_w[0] + tot_b[0]:8.0f}")
