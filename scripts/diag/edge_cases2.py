#!/usr/bin/env python3
"""More interface edge cases probed once: inference_mode, half / bf16 / float64 parameters (must raise), pickling and deepcopy of
a used module, state_dict round trip, cpu -> gpu moves, empty and one-ray batches, a side stream."""
import os, sys, traceback, copy, io
sys.path.insert(0, os.getcwd())
import torch, nerf_sos_amd
from nerf_sos_amd import synthetic as syn
dev = "cuda:0"
torch.manual_seed(1)
net = syn.spiky_density_(nerf_sos_amd.NeRFNet(N_samples=32, N_importance=32, use_semantics=True, sem_with_coord=True).to(dev).eval(), 2.0, 0.5)
rays = syn.synthetic_rays(100, seed=2, device=dev)
with torch.no_grad():
    ref = net(rays, (syn.NEAR, syn.FAR))
def same(y): return all(torch.equal(ref[k], y[k]) for k in ref)
# 1 inference_mode
try:
    with torch.inference_mode():
        print("1 inference_mode equal:", same(net(rays, (syn.NEAR, syn.FAR))))
except Exception as e: print("1 inference_mode ->", type(e).__name__, str(e)[:200])
# 2 half / bf16 params
for dt in (torch.float16, torch.bfloat16, torch.float64):
    try:
        n2 = copy.deepcopy(net).to(dt)
        with torch.no_grad(): n2(rays, (syn.NEAR, syn.FAR))
        print("2", dt, "ran (unexpected)")
    except Exception as e: print("2", dt, "->", type(e).__name__, str(e)[:120])
# 3 torch.save / load of the whole module, and of a trained module with cached plan
buf = io.BytesIO(); torch.save(net, buf); buf.seek(0)
n3 = torch.load(buf, weights_only=False)
with torch.no_grad(): print("3 pickled module equal:", same(n3(rays, (syn.NEAR, syn.FAR))))
# 4 deepcopy after use
n4 = copy.deepcopy(net)
with torch.no_grad(): print("4 deepcopy equal:", same(n4(rays, (syn.NEAR, syn.FAR))))
# 5 state_dict roundtrip into a fresh module, strict
n5 = nerf_sos_amd.NeRFNet(N_samples=32, N_importance=32, use_semantics=True, sem_with_coord=True).to(dev).eval()
n5.load_state_dict(net.state_dict(), strict=True)
with torch.no_grad(): print("5 state_dict equal:", same(n5(rays, (syn.NEAR, syn.FAR))))
# 6 module moved to cpu and back
n6 = copy.deepcopy(net).cpu().to(dev)
with torch.no_grad(): print("6 cpu->gpu equal:", same(n6(rays, (syn.NEAR, syn.FAR))))
# 7 zero rays, one ray
with torch.no_grad():
    z = net(rays[:, :0], (syn.NEAR, syn.FAR)); o = net(rays[:, :1], (syn.NEAR, syn.FAR))
print("7 empty ->", z, "(the reference returns {} too: its chunk loop never runs); one ray equal:", torch.equal(o["rgb"], ref["rgb"][:1]))
# 8 second CUDA stream
s = torch.cuda.Stream()
s.wait_stream(torch.cuda.current_stream())
with torch.cuda.stream(s), torch.no_grad():
    y = net(rays, (syn.NEAR, syn.FAR))
s.synchronize()
print("8 side stream equal:", same(y))
