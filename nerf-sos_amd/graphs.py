"""hipGraph capture of the eval-mode render (launch-bound at small ray batches).

One 4096-ray step is five kernel launches plus a handful of allocator calls; at 16-bit rates (~1 ms of GPU work) the
host-side launch path is a visible fraction of the step.  `GraphedRender` captures `NeRFNet.forward` for one fixed ray
count in eval mode (no random draws: perturb = 0, raw_noise_std = 0, deterministic importance samples) into a HIP graph
and replays it; inputs are copied into static buffers, outputs are returned as views of static buffers (valid until the
next call).  The C ABI only enqueues on the current stream and never synchronises, which is what makes it capturable.
"""
from __future__ import annotations

import os
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


class GraphedPatchStep:
    """The WHOLE single-process patch-training step as one HIP graph: train-mode render of the patch batch (SAVE variant of
    the MLP kernels) -> correlation (+ contrastive) losses -> backward of the trainable parameters -> optimizer update
    (`sharding.sharded_patch_step` + `optimizer.step()`; engines/trainer.py:32-213 is the reference's step).

    Why: at 16-bit rates the step is ~1-2 ms of GPU work behind ~1.8 ms of Python and ~136 launches -- host-bound.  A replay is
    one `hipGraphLaunch`.  What a capture needs, and where it comes from:
      * train-mode render draws: the package's Philox stream with its call counter in device memory
        (`NeRFNet.use_device_rng_counter`, nsos_render_draws_counted) -- a by-value counter would be baked into the graph;
      * the losses' draws (`torch.rand` of the sample coordinates): ONE persistent `torch.Generator`, registered with the
        graph (`CUDAGraph.register_generator_state`): torch keeps its Philox offset in device memory and advances it per replay;
      * the optimizer: `capturable=True` (step count on the device);  parameters re-packed inside the graph (trainable nets
        re-pack on every call), so a replay renders with the weights the previous replay's update produced;
      * static inputs: `rays`, `feat`, `cls_tokens` are the graph's own buffers -- `load()` copies a new batch into them.
    `eager_step()` runs the very same function without the graph (same generator, same counter): replay k and eager step k
    produce the same bits (tests/test_gpu_sharded.py).

    N > 1 (round 5): under a process group `rays` / `feat` / `cls_tokens` are this rank's patches (patch b -> rank
    b mod N) and `n_patches` the batch's size; the step's four collectives (sharding.sharded_patch_step) are captured with the
    kernels when the group runs on RCCL ("nccl": its work is enqueued on a stream and joins the capture like any kernel) -- the
    eager step spends 1.25 of its 2.8 ms on host enqueue (profiles/r04/d_graph_step_time.txt), which is what an 8-GPU step would
    otherwise be bound by.  Every rank must construct the object (the warm-up steps and the capture issue collectives).
    What happens when the graph cannot be had (round 6):
      * a group whose collectives are host-driven (gloo) cannot be captured -- a property of the backend, the same on every rank:
        the step runs eagerly (`self.graph is None`, the reason in `self.capture_fallback`), the caller's loop is the same;
      * a capture that RAISES (RCCL or torch refusing mid-capture) is LOUD by default: the exception propagates with the reason.
        `allow_eager_fallback=True` opts into continuing eagerly -- then the ranks AGREE on it first (an all-reduce MIN of a
        "captured" flag: no rank replays a graph while another steps eagerly, which would desynchronise their collective
        sequences), the loss generator is rebuilt from its pre-capture state and the gradients are reset.
      * `capture_collectives=False` (or NSOS_GRAPH_COLLECTIVES=0 in the environment) keeps a multi-rank step eager on purpose: more
        than one RCCL rank has never run under capture on the builder's hardware.
    The path is executed on one GPU by tests/test_gpu_sharded.py::test_graphed_step_rccl_world_1_captures_its_collectives (a
    `backend="nccl"`, world-size-1 group with sharding.FORCE_COLLECTIVES: real RCCL launches inside the capture).
    """

    def __init__(self, net, optimizer, rays: torch.Tensor, bounds: Tuple[float, float], feat: torch.Tensor, cls_tokens: torch.Tensor,
                 corr_loss=None, geo_loss=None, contrast_loss=None, correlation_w: float = 1.0, geo_w: float = 0.01,
                 contrast_w: float = 0.0, seed: int = 0, overlap_losses: bool = True, warmup: int = 3, capture: bool = True,
                 group=None, n_patches: int = None, allow_eager_fallback: bool = False, capture_collectives: bool = True):
        import torch.distributed as dist
        from . import sharding
        self.group, self.capture_fallback = group, None
        multi = sharding.multi_process(group)            # collectives in the step: N > 1, or one rank under FORCE_COLLECTIVES
        if multi:
            if n_patches is None:
                raise ValueError("GraphedPatchStep under a process group: pass n_patches (the size of the whole patch batch)")
            if capture and dist.get_backend(group) != "nccl":
                capture = False
                self.capture_fallback = (f"process group backend {dist.get_backend(group)!r}: its collectives are driven by the host "
                                         "and cannot be captured in a HIP graph; stepping eagerly")
            elif capture and not (capture_collectives and os.environ.get("NSOS_GRAPH_COLLECTIVES", "1") != "0"):
                # the escape hatch for a first run on a new multi-GPU system: RCCL-under-capture has executed with ONE
                # rank only (tests/test_gpu_sharded.py); `capture_collectives=False` / NSOS_GRAPH_COLLECTIVES=0 keeps the sharded
                # step eager on every rank alike (an argument / the environment: the same everywhere by construction)
                capture = False
                self.capture_fallback = "capture_collectives=False / NSOS_GRAPH_COLLECTIVES=0: the sharded step is not captured; stepping eagerly"
        if not net.training:
            raise ValueError("GraphedPatchStep captures the train-mode step: call net.train() first")
        if net.rng != "philox":
            raise ValueError("GraphedPatchStep needs net.rng = 'philox' (torch's global generator cannot serve the render's "
                             "draws from inside a graph without changing their values)")
        for grp in optimizer.param_groups:
            if not grp.get("capturable", False):
                raise ValueError("GraphedPatchStep: construct the optimizer with capturable=True (its step counter must live on the device)")
        dev = rays.device
        self.net, self.opt, self._sharding = net, optimizer, sharding
        self.n_patches = int(rays.shape[1]) if n_patches is None else int(n_patches)
        self.rays, self.feat, self.cls = rays.clone(), feat.clone(), cls_tokens.clone()
        near, far = bounds
        n_rays = self.rays[0].numel() // 3
        self.bounds = tuple(torch.full((n_rays,), float(b), device=dev, dtype=torch.float32) for b in (near, far))
        self.losses = dict(corr_loss=corr_loss, geo_loss=geo_loss, contrast_loss=contrast_loss, correlation_w=correlation_w,
                           geo_w=geo_w, contrast_w=contrast_w, overlap_losses=overlap_losses)
        self.generator = torch.Generator(device=dev)
        self.generator.manual_seed(int(seed))
        if net.rng_counter is None:
            net.use_device_rng_counter(dev)
        self.loss = torch.zeros((), device=dev)
        self.graph = None
        self.steps = 0
        if capture:
            side = torch.cuda.Stream(device=dev)
            side.wait_stream(torch.cuda.current_stream(dev))
            with torch.cuda.stream(side):                # warm-up off the capture stream: kernel attributes, index uploads, allocator
                for _ in range(warmup):
                    self.eager_step()
            torch.cuda.current_stream(dev).wait_stream(side)
            self.opt.zero_grad(set_to_none=True)          # the captured backward allocates the gradients in the graph's pool
            graph = torch.cuda.CUDAGraph()
            gen_state = self.generator.get_state()        # (to rebuild the generator if the capture is abandoned)
            graph.register_generator_state(self.generator)
            if not multi:
                with torch.cuda.graph(graph):
                    self._step()
                self.graph = graph
            else:
                # RCCL work inside a capture: ProcessGroupNCCL enqueues the collective on its own stream, which the capture follows
                # through the wait (executed on one GPU by the world-size-1 test named above).
                try:
                    with torch.cuda.graph(graph):
                        self._step()
                    self.graph = graph
                except Exception as e:   # noqa: BLE001  (torch raises RuntimeError / DistBackendError depending on where it fails)
                    self.graph = None
                    self.capture_fallback = f"capture of the sharded step failed ({type(e).__name__}: {str(e)[:300]})"
                    if not allow_eager_fallback:
                        raise RuntimeError("nerf_sos_amd.GraphedPatchStep: " + self.capture_fallback + " -- pass allow_eager_fallback=True to "
                                           "step eagerly instead (every rank then agrees on it first), or capture=False") from e
                if allow_eager_fallback:
                    # one agreement collective: every rank learns whether EVERY rank holds a graph; a rank-dependent failure
                    # (allocator, stream state) must not leave one rank replaying while another enqueues eagerly
                    flag = torch.tensor([1 if self.graph is not None else 0], device=dev, dtype=torch.int32)
                    dist.all_reduce(flag, op=dist.ReduceOp.MIN, group=group)
                    if int(flag.item()) == 0:
                        if self.graph is not None:
                            self.capture_fallback = "another rank's capture of the sharded step failed; stepping eagerly on every rank"
                        else:
                            self.capture_fallback += "; stepping eagerly on every rank"
                        self.graph = None
                if self.graph is None:
                    torch.cuda.synchronize(dev)
                    self.generator = torch.Generator(device=dev)      # the abandoned capture left the old one registered with a dead graph
                    self.generator.set_state(gen_state)
                    self.opt.zero_grad(set_to_none=True)

    def _step(self):
        self.opt.zero_grad(set_to_none=True)
        self._sharding.sharded_patch_step(self.net, self.rays, self.bounds, self.n_patches, self.feat, self.cls,
                                          generator=self.generator, group=self.group, loss_out=self.loss, **self.losses)
        self.opt.step()

    def eager_step(self) -> torch.Tensor:
        """One step without the graph (the reference for the bit-identity test, and the warm-up)."""
        self._step()
        self.steps += 1
        return self.loss

    def load(self, rays: torch.Tensor, feat: torch.Tensor, cls_tokens: torch.Tensor) -> None:
        """Copy the next batch into the graph's static input buffers (device-to-device, on the current stream)."""
        self.rays.copy_(rays)
        self.feat.copy_(feat)
        self.cls.copy_(cls_tokens)

    def __call__(self) -> torch.Tensor:
        """One training step on the loaded batch; returns the loss (a static 0-dim tensor, valid until the next call)."""
        if self.graph is None:
            return self.eager_step()
        self.graph.replay()
        self.steps += 1
        return self.loss
