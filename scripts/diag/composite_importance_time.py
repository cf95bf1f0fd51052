"""Where composite_importance_kernel's time goes: the fused launch against its two halves (coarse compositing, importance sampling) as
stand-alone launches, eval (deterministic u, no sort) and train mode (random u: in-wave bitonic sort; sigma noise), at the C2 / C3 batch
(4096 rays: one wave per ray, all waves resident at once = the kernel's per-wave LATENCY) and at the C5 chunk (65 536 rays: throughput)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from nerf_sos_amd import ops, synthetic as syn
dev = "cuda:0"
g = torch.Generator(dev).manual_seed(0)


def clock(fn, n=30):
    for _ in range(5):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


for R in (4096, 65536):
    rays = syn.image_rays(dev, (0, R))
    d = rays[1].contiguous()
    near, far = torch.full((R,), syn.NEAR, device=dev), torch.full((R,), syn.FAR, device=dev)
    raw = torch.randn(R, 64, 6, device=dev, generator=g)
    raw[..., 3] = raw[..., 3] * 3 - 1
    z, _ = ops.ray_setup(d, near, far, 64, None)
    u = torch.rand(R, 128, device=dev, generator=g)
    noise = torch.randn(R, 64, device=dev, generator=g)
    w = ops.composite(raw, z, d)["weights"]
    rows = [("fused, eval", lambda: ops.composite_importance(raw, z, d, 128)),
            ("fused, train (u, noise)", lambda: ops.composite_importance(raw, z, d, 128, noise, 1.0, False, u)),
            ("composite<1> alone", lambda: ops.composite(raw, z, d)),
            ("importance alone, eval", lambda: ops.importance_sample(z, w, 128, None)),
            ("importance alone, train", lambda: ops.importance_sample(z, w, 128, u))]
    for name, fn in rows:
        print(f"R={R:6d}  {name:28s} {clock(fn):8.1f} us", flush=True)
