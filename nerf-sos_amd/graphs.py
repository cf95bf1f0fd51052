"""hipGraph capture of the eval-mode render (launch-bound at small ray batches).

One 4096-ray step is five kernel launches plus a handful of allocator calls; at 16-bit rates (~1 ms of GPU work) the
host-side launch path is a visible fraction of the step.  `GraphedRender` captures `NeRFNet.forward` for one fixed ray
count in eval mode (no random draws: perturb = 0, raw_noise_std = 0, deterministic importance samples) into a HIP graph
and replays it; inputs are copied into static buffers, outputs are returned as views of static buffers (valid until the
next call).  The C ABI only enqueues on the current stream and never synchronises, which is what makes it capturable.
"""
from __future__ import annotations

from typing import Dict, Tuple

import torch


class GraphedRender:
    def __init__(self, net, n_rays: int, near_far: Tuple[float, float], warmup: int = 3, **render_kwargs):
        if net.training:
            raise ValueError("GraphedRender captures the deterministic eval-mode path: call net.eval() first")
        # Trainable nets re-pack their weight streams on every call (NeRFMLP.packed_weights), so the pack launches are part
        # of the graph and a replay reads the parameters' CURRENT values; a frozen net's streams are packed once, outside.
        self._packs_in_graph = all(any(p.requires_grad for p in m.parameters()) for m in (net.nerf, net.nerf_fine))
        self.net, self.n_rays, self.near_far, self.kw = net, int(n_rays), near_far, render_kwargs
        dev = next(net.parameters()).device
        self._rays = torch.zeros((2, self.n_rays, 3), device=dev, dtype=torch.float32)
        self._rays[1, :, 2] = -1.0
        # The graph bakes in the ADDRESSES of every tensor the kernels read.  Scalar bounds would come from NeRFNet's
        # process-wide fill cache, which may evict (and so free) them later: the graph owns its bound tensors instead.
        near, far = near_far
        self._bounds = tuple(b.to(device=dev, dtype=torch.float32).reshape(-1).contiguous().clone() if isinstance(b, torch.Tensor)
                             else torch.full((self.n_rays,), float(b), device=dev, dtype=torch.float32) for b in (near, far))
        near_far = self._bounds
        side = torch.cuda.Stream(device=dev)
        side.wait_stream(torch.cuda.current_stream(dev))
        with torch.cuda.stream(side), torch.no_grad():   # warm-up on a side stream: packs weights, sets kernel attributes
            for _ in range(warmup):
                net(self._rays, near_far, **render_kwargs)
        torch.cuda.current_stream(dev).wait_stream(side)
        self._keys = None
        self.graph = torch.cuda.CUDAGraph()
        with torch.no_grad(), torch.cuda.graph(self.graph):
            self._out = net(self._rays, near_far, **render_kwargs)
        self._versions = self._param_versions()

    def _param_versions(self):
        if self._packs_in_graph:
            return tuple(p.data_ptr() for p in self.net.parameters())
        return tuple((p.data_ptr(), p._version) for p in self.net.parameters())

    def __call__(self, ray_batch) -> Dict[str, torch.Tensor]:
        rays_o, rays_d = ray_batch
        if rays_o.numel() != self.n_rays * 3 or rays_d.numel() != self.n_rays * 3:
            raise ValueError(f"captured for {self.n_rays} rays")
        if self._param_versions() != self._versions:
            raise RuntimeError("parameters moved (or, for a frozen net, changed) since capture -- the graph holds their "
                               "addresses and a frozen net's packed weight stream: re-capture.  `.data` edits of a frozen "
                               "net are invisible here: call net.invalidate_packed() and re-capture")
        self._rays[0].copy_(rays_o.reshape(self.n_rays, 3))
        self._rays[1].copy_(rays_d.reshape(self.n_rays, 3))
        self.graph.replay()
        return self._out
