"""Per-op phase cycles of mlp_generic_kernel from a diagnostic build (scripts/diag/build_variant.sh gen_prof mlp_generic.hip
-DNSOS_GEN_PROF -> ab/lib_gen_prof.so): s_memtime stamps before and after every op's barrier, workgroup 1, all four waves.
Prints, per op: mean ticks per tile of work (from the previous barrier's release to this op's barrier) and of barrier wait, by wave,
next to the op's MFMA issue floor for the busiest wave (tiles of the wave x groups x 4 k-steps x 64 cycles)."""
import ctypes, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
os.environ["NERF_SOS_HIP_LIB"] = os.path.join(ROOT, "ab", "lib_gen_prof.so")
import torch
import nerf_sos_amd
from nerf_sos_amd import ops, synthetic as syn
dev = "cuda:0"
R, S = 4096, 192
rays = syn.synthetic_rays(R, seed=0, device=dev)
near = torch.full((R,), syn.NEAR, device=dev); far = torch.full((R,), syn.FAR, device=dev)
z, v = ops.ray_setup(rays[1].contiguous(), near, far, S, None)
o, d = rays[0].contiguous(), rays[1].contiguous()
CASES = {"8x256m6": dict(multires=6), "16x256": dict(netdepth=16, netdepth_fine=16), "8x256nov": dict(viewdirs=False),
         "4x128": dict(netdepth=4, netwidth=128, netdepth_fine=4, netwidth_fine=128),
         "deepsem": dict(use_semantics=True, sem_layer=3, sem_dim=5)}
lib = ctypes.CDLL(os.environ["NERF_SOS_HIP_LIB"])
for name in (sys.argv[1:] or ["8x256m6", "16x256"]):
    torch.manual_seed(0)
    net = nerf_sos_amd.NeRFNet(N_samples=64, N_importance=128, **CASES[name]).to(dev).eval()
    mlp = net.nerf_fine
    lins = [(n, m.in_features, m.out_features) for n, m in mlp.mlp.named_modules() if isinstance(m, torch.nn.Linear)]
    for _ in range(3):
        mlp.query_rays(o, d, v, z)
    torch.cuda.synchronize()
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
    ev[0].record(); mlp.query_rays(o, d, v, z); ev[1].record(); torch.cuda.synchronize()
    buf = (ctypes.c_ulonglong * 512)()
    assert lib.nsos_gen_prof_read(buf) == 0
    g = lambda w, k, s: buf[(w * 2 + k) * 64 + s]
    tiles = g(0, 0, 0)
    print(f"== {name}: {ev[0].elapsed_time(ev[1]):.3f} ms (instrumented), {tiles} tiles by workgroup 1; Linear layers in module order: {lins}")
    tot_w = [0] * 4; tot_b = [0] * 4
    for s in list(range(1, 62)) + [62]:
        if not any(g(w, k, s) for w in range(4) for k in range(2)):
            continue
        lab = "encode" if s == 1 else ("output" if s == 62 else f"op {s - 2:2d}")
        ws = [g(w, 0, s) / tiles for w in range(4)]; bs = [g(w, 1, s) / tiles for w in range(4)]
        for w in range(4):
            tot_w[w] += ws[w]; tot_b[w] += bs[w]
        print(f"  {lab:7s} work " + " ".join(f"{x:8.0f}" for x in ws) + "   barrier " + " ".join(f"{x:7.0f}" for x in bs) + f"   op total {max(ws[w] + bs[w] for w in range(4)):8.0f}")
    print("  total   work " + " ".join(f"{x:8.0f}" for x in tot_w) + "   barrier " + " ".join(f"{x:7.0f}" for x in tot_b) + f"   tile {tot_w[0] + tot_b[0]:8.0f}")
