import os, sys, time, torch
sys.path.insert(0, os.getcwd())
from oracle import torch_port as tp
cfg = tp.PortConfig(n_samples=64, n_importance=128, use_semantics=False, pts_chunk=1024*256)
sd = tp.init_state_dict(cfg, seed=0)
os.system("lscpu | egrep 'Model name|Socket|Core|Thread|^CPU\\(s\\)'")
for nt in (256, 128, 64, 32, 16):
    torch.set_num_threads(nt)
    for n in (512, 2048):
        rays = tp.synthetic_rays(n, seed=0)
        with torch.no_grad():
            t0=time.perf_counter(); tp.render(sd, cfg, rays, (tp.NEAR, tp.FAR)); tw=time.perf_counter()-t0
            best=1e9
            for _ in range(3):
                t0=time.perf_counter(); tp.render(sd, cfg, rays, (tp.NEAR, tp.FAR)); best=min(best,time.perf_counter()-t0)
        print(f"threads {nt:4d} rays {n:5d}: warm {tw:.2f}s best {best:.3f}s -> {n/best:.0f} rays/s", flush=True)
