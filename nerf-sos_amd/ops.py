"""Tensor-level wrappers over the C ABI (include/nerf_sos_hip.h): one function per HIP fusion group.

PyTorch is used for device memory, streams and shapes only; all arithmetic happens in the HIP kernels.
Every function requires contiguous fp32 CUDA(ROCm) tensors and launches on the current stream.
"""
from __future__ import annotations

import ctypes as C
import functools
import operator
from typing import Dict, Optional, Tuple

import torch

from . import _lib

SEM_NONE, SEM_PLAIN, SEM_COORD = 0, 1, 2

# Optional live timing of the fused-MLP launches (bench.py): when set to a list, every launch appends
# (n_points, start_event, end_event), recorded on the stream the kernel is launched on.
KERNEL_EVENTS = None


def _ev_begin():
    if KERNEL_EVENTS is None:
        return None
    ev = (torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True))
    ev[0].record()
    return ev


def _ev_end(ev, n_points: int):
    if ev is not None:
        ev[1].record()
        KERNEL_EVENTS.append((n_points, ev[0], ev[1]))


def sem_mode_of(use_semantics: bool, sem_with_coord: bool) -> int:
    return SEM_NONE if not use_semantics else (SEM_COORD if sem_with_coord else SEM_PLAIN)


def _dev(t: torch.Tensor, name: str) -> torch.Tensor:
    if not isinstance(t, torch.Tensor) or not t.is_cuda:
        raise RuntimeError(f"nerf_sos_amd: `{name}` must be a GPU tensor -- this package has no CPU path "
                           f"(got {'device ' + str(t.device) if isinstance(t, torch.Tensor) else type(t)})")
    if t.dtype != torch.float32:
        raise TypeError(f"nerf_sos_amd: `{name}` must be float32, got {t.dtype}")
    return t.contiguous()


def _p(t: Optional[torch.Tensor]):
    return None if t is None else C.c_void_p(t.data_ptr())


_raw_stream = getattr(torch._C, "_cuda_getCurrentRawStream", None)


def _stream():
    """The current device's current stream as a hipStream_t.  (torch.cuda.current_stream() builds a Stream object through four
    Python layers: ~10 us per call, ~15 calls per training step.)"""
    if _raw_stream is not None:
        return C.c_void_p(_raw_stream(torch.cuda.current_device()))
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


# ------------------------------------------------------------------------------------------ K0
def generate_rays(H: int, W: int, K, c2w, device, pix_range: Optional[Tuple[int, int]] = None) -> torch.Tensor:
    """get_persp_rays (utils/ray.py:12-22) on device.  K: 3x3 intrinsics, c2w: [3,4] (or [4,4]) pose -- host
    tensors / arrays.  Returns [2, H, W, 3] like the reference, or [2, n, 3] for the flat pixel range given."""
    Kf = torch.as_tensor(K, dtype=torch.float32).cpu()
    pose = torch.as_tensor(c2w, dtype=torch.float32).cpu()[:3, :4].contiguous()
    b, e = (0, H * W) if pix_range is None else pix_range
    n = e - b
    dev = torch.device(device)
    if dev.type != "cuda":
        raise RuntimeError("nerf_sos_amd: `device` must be a GPU -- this package has no CPU path")
    with torch.cuda.device(dev):
        ro = torch.empty((n, 3), device=dev, dtype=torch.float32)
        rd = torch.empty((n, 3), device=dev, dtype=torch.float32)
        arr = (C.c_float * 12)(*pose.reshape(-1).tolist())
        _lib.check(_lib.lib().nsos_generate_rays(H, W, float(Kf[0, 0]), float(Kf[1, 1]), float(Kf[0, 2]), float(Kf[1, 2]),
                                                 arr, b, e, _p(ro), _p(rd), _stream()), "nsos_generate_rays")
    rays = torch.stack([ro, rd], 0)
    return rays.reshape(2, H, W, 3) if pix_range is None else rays


# ------------------------------------------------------------------------------------------ K1
def ray_setup(rays_d: torch.Tensor, near: torch.Tensor, far: torch.Tensor, n_samples: int,
              t_rand: Optional[torch.Tensor] = None) -> Tuple[torch.Tensor, torch.Tensor]:
    """z_vals [R,S] and viewdirs [R,3]  (models/sampler.py:46-68, models/nerf_net.py:163-166)."""
    rays_d = _dev(rays_d, "rays_d")
    R = rays_d.shape[0]
    near = _dev(near, "near").reshape(-1)
    far = _dev(far, "far").reshape(-1)
    if near.numel() != R or far.numel() != R:
        raise ValueError(f"near/far must have one value per ray ({R}), got {near.numel()}/{far.numel()}")
    if t_rand is not None:
        t_rand = _dev(t_rand, "t_rand")
        if tuple(t_rand.shape) != (R, n_samples):
            raise ValueError(f"t_rand must be [{R},{n_samples}]")
    z = torch.empty((R, n_samples), device=rays_d.device, dtype=torch.float32)
    v = torch.empty((R, 3), device=rays_d.device, dtype=torch.float32)
    _lib.check(_lib.lib().nsos_ray_setup(_p(rays_d), _p(near), _p(far), _p(t_rand), R, n_samples, _p(z), _p(v),
                                         _stream()), "nsos_ray_setup")
    return z, v


def ray_points(rays_o: torch.Tensor, rays_d: torch.Tensor, z_vals: torch.Tensor) -> torch.Tensor:
    """pts = o + d*z  [R,S,3]  (models/sampler.py:70,166)."""
    rays_o, rays_d, z_vals = _dev(rays_o, "rays_o"), _dev(rays_d, "rays_d"), _dev(z_vals, "z_vals")
    R, S = z_vals.shape
    pts = torch.empty((R, S, 3), device=z_vals.device, dtype=torch.float32)
    _lib.check(_lib.lib().nsos_ray_points(_p(rays_o), _p(rays_d), _p(z_vals), R, S, _p(pts), _stream()),
               "nsos_ray_points")
    return pts


# ------------------------------------------------------------------------------------------ K2
_MLP_FIELDS = (("alpha", "alpha_linear"), ("feature", "feature_linear"), ("views", "views_linears.0"),
               ("rgb", "rgb_linear"))


DTYPES = {"fp32": 0, "fp16": 1, "bf16": 2, "fp16x3": 3,   # fp16x3: split-fp16 operands, fp32-accurate results (K2-X3)
          "fp16x3_bwd": 4}                                  # the transposed stream of the fused input-gradient kernel (K7-X3)


def packed_bytes(sem_mode: int, precision: str = "fp32") -> int:
    if precision == "fp32":
        return int(_lib.lib().nsos_mlp_packed_bytes(sem_mode))
    if precision == "fp16x3":
        return int(_lib.lib().nsos_mlp_packed_bytes_x3(sem_mode))
    if precision == "fp16x3_bwd":
        return int(_lib.lib().nsos_mlp_bwd_packed_bytes_x3(sem_mode))
    return int(_lib.lib().nsos_mlp_packed_bytes_lp(sem_mode))


class PackPlan:
    """The argument block of one net's pack launches: the validated parameter pointers (MlpTensors) plus the tensors that
    own them.  Building it walks ~26 parameters; a training loop re-packs every step (NeRFMLP.packed_weights), so the
    plan is built once per set of parameter storages and a re-pack is one C call."""

    def __init__(self, params: Dict[str, torch.Tensor], sem_mode: int):
        keep = []

        def g(name):
            t = _dev(params[name].detach(), name)
            keep.append(t)
            return t.data_ptr()

        T = _lib.MlpTensors()
        for i in range(8):
            T.pts_w[i] = g(f"pts_linears.{i}.weight")
            T.pts_b[i] = g(f"pts_linears.{i}.bias")
        for field, name in _MLP_FIELDS:
            setattr(T, field + "_w", g(name + ".weight"))
            setattr(T, field + "_b", g(name + ".bias"))
        if sem_mode != SEM_NONE:
            T.sem0_w, T.sem0_b = g("semantic_linear.0.weight"), g("semantic_linear.0.bias")
            T.sem2_w, T.sem2_b = g("semantic_linear.2.weight"), g("semantic_linear.2.bias")
        shapes = {"pts_linears.0.weight": (256, 63), "pts_linears.5.weight": (256, 319),
                  "views_linears.0.weight": (128, 283), "rgb_linear.weight": (3, 128), "alpha_linear.weight": (1, 256)}
        if sem_mode != SEM_NONE:
            shapes["semantic_linear.0.weight"] = (128, 319 if sem_mode == SEM_COORD else 256)
            shapes["semantic_linear.2.weight"] = (2, 128)
        for k, shp in shapes.items():
            if tuple(params[k].shape) != shp:
                raise ValueError(f"{k}: expected shape {shp}, got {tuple(params[k].shape)}")
        self.tensors, self.keep, self.sem_mode, self.device = T, keep, sem_mode, keep[0].device
        self.ptrs = tuple(t.data_ptr() for t in keep)
        self.nbytes = {}

    def run(self, out: Optional[torch.Tensor] = None, precision: str = "fp32", heads_only: bool = False) -> torch.Tensor:
        """Pack into `out` (allocated if None / too small).  heads_only (fp16 / bf16 with a semantic head): `out` already holds a
        full pack of the same trunk -- re-pack only what depends on semantic_linear.* (nsos_mlp_pack_lp_heads)."""
        if precision not in DTYPES:
            raise ValueError(f"precision must be one of {list(DTYPES)}, got {precision!r}")
        nbytes = self.nbytes.get(precision)
        if nbytes is None:
            nbytes = self.nbytes[precision] = packed_bytes(self.sem_mode, precision)
        if out is None or out.numel() * 4 < nbytes or out.device != self.device:
            out = torch.empty(nbytes // 4, device=self.device, dtype=torch.float32)
        T, L = C.byref(self.tensors), _lib.lib()
        if precision == "fp32":
            _lib.check(L.nsos_mlp_pack(T, self.sem_mode, _p(out), nbytes, _stream()), "nsos_mlp_pack")
        elif precision == "fp16x3":
            _lib.check(L.nsos_mlp_pack_x3(T, self.sem_mode, _p(out), nbytes, _stream()), "nsos_mlp_pack_x3")
        elif precision == "fp16x3_bwd":
            _lib.check(L.nsos_mlp_bwd_pack_x3(T, self.sem_mode, _p(out), nbytes, _stream()), "nsos_mlp_bwd_pack_x3")
        elif heads_only and self.sem_mode != SEM_NONE:
            _lib.check(L.nsos_mlp_pack_lp_heads(T, self.sem_mode, DTYPES[precision], _p(out), nbytes, _stream()), "nsos_mlp_pack_lp_heads")
            _LP_PARTIAL[_storage_key(out)] = lp_selected_kernel()       # only this kernel's stream holds the new heads (check_lp_stream)
        else:
            _lib.check(L.nsos_mlp_pack_lp(T, self.sem_mode, DTYPES[precision], _p(out), nbytes, _stream()), "nsos_mlp_pack_lp")
            _LP_PARTIAL.pop(_storage_key(out), None)
        return out


# Which of a 16-bit weight buffer's three streams holds current semantic-head weights after a heads-only re-pack, keyed by the
# buffer's STORAGE (device, address): every view, reshape or slice of the buffer shares it, so the tag survives them (a
# Python attribute on the tensor object did not).  A clone is a new storage with no entry -- a clone of a partially re-packed buffer
# must be re-packed in full by its owner; the C ABI's own callers hold the same obligation (include/nerf_sos_hip.h, nsos_mlp_pack_lp_heads).
_LP_PARTIAL: Dict[tuple, int] = {}


def _storage_key(t: torch.Tensor) -> tuple:
    return (str(t.device), t.untyped_storage().data_ptr())


def check_lp_stream(packed: torch.Tensor, kernel: int) -> None:
    """A 16-bit weight buffer that had a heads-only re-pack (PackPlan.run(heads_only=True)) holds current semantic-head weights in
    ONE of its three streams.  Called with the kernel a launch is about to run on (1 = mlp_lp_kernel, the fall-back for fp32 sem_in
    saves and >= 2^31 points; 2 = lp8; 3 = lp16): raises instead of rendering with stale heads."""
    cur = _LP_PARTIAL.get(_storage_key(packed))
    if cur is not None and cur != kernel:
        raise RuntimeError(f"nerf_sos_amd: this packed 16-bit weight buffer was last updated by a heads-only re-pack for kernel {cur}; the "
                           f"launch needs kernel {kernel}'s stream, whose semantic-head weights are stale -- do a full pack first "
                           "(PackPlan.run(..., heads_only=False), or NeRFMLP.invalidate_packed())")


class GenericPlan:
    """One NeRFMLP.mlp of ANY architecture the reference's constructors accept, described for the generic kernel
    (nsos_generic_mlp, csrc/mlp_generic.hip): the Linear modules in the reference's forward order with their device pointers.
    `mlp` is the nerf_net.MLP parameter container; `multires` / `multires_views` are octave counts or None (use_embed=False)."""

    def __init__(self, mlp, multires, multires_views):
        keep = []

        def lin(dst, module):
            w, b = _dev(module.weight.detach(), "weight"), _dev(module.bias.detach(), "bias")
            keep.extend((w, b))
            dst.weight, dst.bias = w.data_ptr(), b.data_ptr()
            dst.out_dim, dst.in_dim = int(w.shape[0]), int(w.shape[1])

        G = _lib.GenericMlp()
        names = {}             # position of a Linear in nsos_generic_mlp -> its module's name inside `mlp` (nsos_mlp_generic_save_layout)
        G.depth, G.width = int(mlp.D), int(mlp.W)
        if G.depth > _lib.GENERIC_MAX_DEPTH:
            raise NotImplementedError(f"nerf_sos_amd: netdepth {G.depth} > {_lib.GENERIC_MAX_DEPTH}")
        G.skip_mask = functools.reduce(operator.or_, (1 << int(i) for i in set(mlp.skips) if 0 <= int(i) < G.depth), 0)   # `i in self.skips`: repeats count once
        G.xyz_freqs = -1 if multires is None else int(multires)
        G.dir_freqs = -1 if multires_views is None else int(multires_views)
        G.use_viewdirs, G.use_semantics = int(bool(mlp.use_viewdirs)), int(bool(mlp.use_semantics))
        G.sem_with_coord = int(bool(mlp.sem_with_coord))
        for i, m in enumerate(mlp.pts_linears):
            lin(G.pts[i], m)
            names[i] = f"pts_linears.{i}"
        if mlp.use_viewdirs:
            lin(G.alpha, mlp.alpha_linear), lin(G.feature, mlp.feature_linear), lin(G.views, mlp.views_linears[0]), lin(G.rgb, mlp.rgb_linear)
            names.update({16: "alpha_linear", 17: "feature_linear", 18: "views_linears.0", 19: "rgb_linear"})
        else:
            lin(G.output, mlp.output_linear)
            names[20] = "output_linear"
        G.sem_layers = G.sem_dim = G.sem_with_geo = 0
        if mlp.use_semantics:
            named = [(n, m) for n, m in mlp.semantic_linear.named_modules() if isinstance(m, torch.nn.Linear)]   # Sequential order = forward order
            linears = [m for _, m in named]
            if len(linears) > _lib.GENERIC_MAX_SEM:
                raise NotImplementedError(f"nerf_sos_amd: a semantic head of {len(linears)} Linear layers (> {_lib.GENERIC_MAX_SEM})")
            for k, m in enumerate(linears):
                lin(G.sem[k], m)
                names[21 + k] = "semantic_linear." + named[k][0]
            G.sem_layers, G.sem_dim = len(linears), int(linears[-1].weight.shape[0])
            if G.sem_dim > 8 and mlp.use_viewdirs:
                raise NotImplementedError(f"nerf_sos_amd: sem_dim {G.sem_dim} > 8 (the generic kernels' output tile and the compositing "
                                          "backward hold at most 8 semantic channels)")
            if getattr(mlp, "geo_map_sem", None) is not None:
                G.sem_with_geo = 1
                lin(G.geo[0], mlp.geo_map_sem[0]), lin(G.geo[1], mlp.geo_map_sem[2])
                names.update({29: "geo_map_sem.0", 30: "geo_map_sem.2"})
        self.lin_names = names
        self._layout = None
        self._headers, self._keep_bufs = {}, {}     # which buffers hold this plan's program header already (run / run_bwd)
        self.desc, self.keep, self.device = G, keep, keep[0].device
        self.ptrs = tuple(t.data_ptr() for t in keep)
        self.nbytes = int(_lib.lib().nsos_mlp_generic_packed_bytes(C.byref(G)))
        self.out_channels = int(_lib.lib().nsos_mlp_generic_out_channels(C.byref(G)))
        if self.nbytes == 0 or self.out_channels == 0:
            raise NotImplementedError(
                "nerf_sos_amd: this architecture is outside the generic kernel's limits (include/nerf_sos_hip.h: depth <= 16, "
                "activation buffers of ceil(W/32)*32 rows within 160 KiB of LDS -- W <= 256 with the deep semantic head --, "
                "sem_dim <= 8, a skip on the last layer is the reference's own shape error)")

    def run(self, out: Optional[torch.Tensor] = None) -> torch.Tensor:
        """Pack into `out` (or a new buffer).  A buffer this plan filled before keeps its program header: only the weights are
        re-packed (kernels only: capturable in a HIP graph, and no 33 KB staging copy per training step)."""
        if out is None or out.numel() * 4 < self.nbytes or out.device != self.device:
            out = torch.empty((self.nbytes + 3) // 4, device=self.device, dtype=torch.float32)
        if self._headers.get("fwd") == out.data_ptr():
            _lib.check(_lib.lib().nsos_mlp_generic_repack(C.byref(self.desc), _p(out), self.nbytes, _stream()), "nsos_mlp_generic_repack")
        else:
            _lib.check(_lib.lib().nsos_mlp_generic_pack(C.byref(self.desc), _p(out), self.nbytes, _stream()), "nsos_mlp_generic_pack")
            self._headers["fwd"] = out.data_ptr()
            self._keep_bufs["fwd"] = out            # (the pointer stays this buffer's while the plan lives)
        return out

    def trainable_mask(self, mlp) -> Optional[int]:
        """Bit p set = the Linear at position p of the description has a parameter that requires grad; None: all of them do."""
        params = dict(mlp.named_parameters())
        mask = sum(1 << p for p, n in self.lin_names.items() if params[n + ".weight"].requires_grad or params[n + ".bias"].requires_grad)
        return None if mask == sum(1 << p for p in self.lin_names) else mask

    def run_bwd(self, out: Optional[torch.Tensor] = None, input_grads: bool = False, trainable: Optional[int] = None) -> torch.Tensor:
        """The transposed weight streams + the reversed program of the input-gradient chain (nsos_mlp_generic_pack_bwd);
        input_grads: the chain also reaches the positional encodings (gradients w.r.t. the rays); trainable (a trainable_mask()):
        the chain only as far as a trainable Linear needs it, and only trainable Linears' gradients are written to gbuf
        (nsos_mlp_generic_pack_bwd_subset; with input_grads every gradient is formed and a frozen net's chain stores nothing)."""
        nbytes = int(_lib.lib().nsos_mlp_generic_bwd_packed_bytes(C.byref(self.desc), int(input_grads)))
        if trainable is not None:
            if input_grads:      # bit 31: the subset's chain continued to the encodings (pose refinement against a partly / wholly frozen net)
                trainable = int(trainable) | INPUT_GRADS_BIT
            if nbytes == 0:
                raise NotImplementedError("nerf_sos_amd: this architecture is outside the generic backward kernel's limits")
            if out is None or out.numel() * 4 < nbytes or out.device != self.device:
                out = torch.empty((nbytes + 3) // 4, device=self.device, dtype=torch.float32)
            key = ("sub", int(trainable))
            fresh = self._headers.get(key) != out.data_ptr()
            _lib.check(_lib.lib().nsos_mlp_generic_pack_bwd_subset(C.byref(self.desc), _p(out), nbytes, int(trainable), int(fresh), _stream()),
                       "nsos_mlp_generic_pack_bwd_subset")
            if fresh:
                for k in [k for k, v in self._headers.items() if v == out.data_ptr()]:
                    del self._headers[k]        # the buffer held another program
                self._headers[key] = out.data_ptr()
                self._keep_bufs[key] = out
            return out
        if nbytes == 0:
            raise NotImplementedError("nerf_sos_amd: this architecture is outside the generic backward kernel's limits")
        if out is None or out.numel() * 4 < nbytes or out.device != self.device:
            out = torch.empty((nbytes + 3) // 4, device=self.device, dtype=torch.float32)
        key = "bwd_in" if input_grads else "bwd"
        if self._headers.get(key) == out.data_ptr():
            _lib.check(_lib.lib().nsos_mlp_generic_repack_bwd(C.byref(self.desc), _p(out), nbytes, int(input_grads), _stream()), "nsos_mlp_generic_repack_bwd")
        else:
            _lib.check(_lib.lib().nsos_mlp_generic_pack_bwd(C.byref(self.desc), _p(out), nbytes, int(input_grads), _stream()), "nsos_mlp_generic_pack_bwd")
            self._headers[key] = out.data_ptr()
            self._keep_bufs[key] = out
        return out

    def layout(self):
        """(ld, [(module name, column block, out_dim, [(segment's column block in acts, rows, first weight column)])]) in forward
        order: where every Linear's pre-activation gradient (gbuf) and inputs (acts) sit in a point's saved row
        (nsos_mlp_generic_save_layout)."""
        if self._layout is None:
            cap = 2 + 64 * _lib.GENERIC_LAYOUT_STRIDE
            table = (C.c_int32 * cap)()
            n = int(_lib.lib().nsos_mlp_generic_save_layout(C.byref(self.desc), table, cap))
            if n < 2:
                _lib.check(n, "nsos_mlp_generic_save_layout")
            ld, n_ops, ops_ = int(table[0]), int(table[1]), []
            for i in range(n_ops):
                e = table[2 + i * _lib.GENERIC_LAYOUT_STRIDE: 2 + (i + 1) * _lib.GENERIC_LAYOUT_STRIDE]
                segs = [(int(e[4 + 3 * s]), int(e[5 + 3 * s]), int(e[6 + 3 * s])) for s in range(int(e[3]))]
                ops_.append((self.lin_names[int(e[0])], int(e[1]), int(e[2]), segs))
            self._layout = (ld, ops_)
        return self._layout


INPUT_GRADS_BIT = 1 << 31      # OR-ed into a trainable mask: "the backward also reaches the inputs" (nsos_mlp_generic_*_subset)


def mlp_generic_forward_rays(plan: GenericPlan, packed: torch.Tensor, rays_o: torch.Tensor, rays_d: torch.Tensor,
                             viewdirs: Optional[torch.Tensor], z_vals: torch.Tensor) -> torch.Tensor:
    """raw [R,S,C] for the points o + d*z of each ray through the generic-architecture kernel (models/nerf_mlp.py:67-100,179-215)."""
    rays_o, rays_d, z_vals = _dev(rays_o, "rays_o"), _dev(rays_d, "rays_d"), _dev(z_vals, "z_vals")
    if viewdirs is not None:
        viewdirs = _dev(viewdirs, "viewdirs")
    R, S = z_vals.shape
    raw = torch.empty((R, S, plan.out_channels), device=z_vals.device, dtype=torch.float32)
    ev = _ev_begin()
    _lib.check(_lib.lib().nsos_mlp_generic_forward_rays(C.byref(plan.desc), _p(packed), _p(rays_o), _p(rays_d), _p(viewdirs), _p(z_vals),
                                                        R, S, _p(raw), _stream()), "nsos_mlp_generic_forward_rays")
    _ev_end(ev, R * S)
    return raw


def mlp_generic_forward_rays_save(plan: GenericPlan, packed: torch.Tensor, rays_o: torch.Tensor, rays_d: torch.Tensor,
                                  viewdirs: Optional[torch.Tensor], z_vals: torch.Tensor, trainable: Optional[int] = None) -> Tuple[torch.Tensor, torch.Tensor]:
    """Training variant of mlp_generic_forward_rays: also returns acts [R*S, ld], every Linear's post-activation output and both
    encodings per point (nsos_mlp_generic_forward_rays_save; raw is bit-identical to the inference call's)."""
    rays_o, rays_d, z_vals = _dev(rays_o, "rays_o"), _dev(rays_d, "rays_d"), _dev(z_vals, "z_vals")
    if viewdirs is not None:
        viewdirs = _dev(viewdirs, "viewdirs")
    R, S = z_vals.shape
    ld = plan.layout()[0]
    raw = torch.empty((R, S, plan.out_channels), device=z_vals.device, dtype=torch.float32)
    acts = torch.empty((R * S, ld), device=z_vals.device, dtype=torch.float32)
    if trainable is not None:      # a GenericPlan.trainable_mask(): store only what that subset's backward reads
        _lib.check(_lib.lib().nsos_mlp_generic_forward_rays_save_subset(C.byref(plan.desc), _p(packed), _p(rays_o), _p(rays_d), _p(viewdirs), _p(z_vals),
                                                                        R, S, _p(raw), _p(acts), int(trainable), _stream()),
                   "nsos_mlp_generic_forward_rays_save_subset")
        return raw, acts
    _lib.check(_lib.lib().nsos_mlp_generic_forward_rays_save(C.byref(plan.desc), _p(packed), _p(rays_o), _p(rays_d), _p(viewdirs), _p(z_vals),
                                                             R, S, _p(raw), _p(acts), _stream()), "nsos_mlp_generic_forward_rays_save")
    return raw, acts


def mlp_generic_input_grads(plan: GenericPlan, packed_bwd: torch.Tensor, g_raw: torch.Tensor, acts: torch.Tensor, rays=None):
    """gbuf [P, ld]: every Linear's pre-activation gradient in its column block, from d loss / d raw [P, C] and the saved
    activations (nsos_mlp_generic_input_grads: the whole chain in one kernel).  rays = (rays_o, rays_d, viewdirs or None, z_vals)
    with a `packed_bwd` of run_bwd(input_grads=True): also returns d loss / d point [P,3] and / d view direction [P,3] (or None)."""
    g_raw, acts = _dev(g_raw, "g_raw"), _dev(acts, "acts")
    P_ = acts.shape[0]
    if g_raw.shape != (P_, plan.out_channels):
        raise ValueError("mlp_generic_input_grads: g_raw must be [P, out_channels]")
    gbuf = torch.empty_like(acts)
    if rays is None:
        _lib.check(_lib.lib().nsos_mlp_generic_input_grads(C.byref(plan.desc), _p(packed_bwd), _p(g_raw), _p(acts), _p(gbuf), P_, _stream()),
                   "nsos_mlp_generic_input_grads")
        return gbuf
    rays_o, rays_d, viewdirs, z_vals = rays
    rays_o, rays_d, z_vals = _dev(rays_o, "rays_o"), _dev(rays_d, "rays_d"), _dev(z_vals, "z_vals")
    R, S = z_vals.shape
    if R * S != P_:
        raise ValueError("mlp_generic_input_grads: rays do not match the saved activations")
    g_pts = torch.empty((P_, 3), device=acts.device, dtype=torch.float32)
    g_dirs = None
    if plan.desc.use_viewdirs:
        viewdirs = _dev(viewdirs, "viewdirs")
        g_dirs = torch.empty((P_, 3), device=acts.device, dtype=torch.float32)
    _lib.check(_lib.lib().nsos_mlp_generic_input_grads_rays(C.byref(plan.desc), _p(packed_bwd), _p(g_raw), _p(acts), _p(gbuf), _p(rays_o), _p(rays_d),
                                                            _p(viewdirs) if g_dirs is not None else None, _p(z_vals), R, S, _p(g_pts), _p(g_dirs), _stream()),
               "nsos_mlp_generic_input_grads_rays")
    return gbuf, g_pts, g_dirs


def mlp_generic_forward_points_save(plan: GenericPlan, packed: torch.Tensor, pts: Optional[torch.Tensor], dirs: Optional[torch.Tensor],
                                    encoded: Optional[torch.Tensor] = None, save: bool = True):
    """Point query of the generic kernels with saved activations: (raw [P,C], acts [P,ld] or None).  encoded [P, input_ch +
    input_ch_views]: MLP.forward's own pre-encoded input instead of points and directions (nsos_mlp_generic_forward_points_save)."""
    if encoded is not None:
        encoded = _dev(encoded, "encoded")
        P_ = encoded.shape[0]
    else:
        pts = _dev(pts, "pts")
        dirs = _dev(dirs, "dirs") if dirs is not None else None
        P_ = pts.shape[0]
    dev = (encoded if encoded is not None else pts).device
    raw = torch.empty((P_, plan.out_channels), device=dev, dtype=torch.float32)
    acts = torch.empty((P_, plan.layout()[0]), device=dev, dtype=torch.float32) if save else None
    _lib.check(_lib.lib().nsos_mlp_generic_forward_points_save(C.byref(plan.desc), _p(packed), _p(pts) if encoded is None else None,
                                                               _p(dirs) if encoded is None else None, _p(encoded), P_, _p(raw), _p(acts), _stream()),
               "nsos_mlp_generic_forward_points_save")
    return raw, acts


def mlp_generic_input_grads_points(plan: GenericPlan, packed_bwd: torch.Tensor, g_raw: torch.Tensor, acts: torch.Tensor,
                                   pts: Optional[torch.Tensor], dirs: Optional[torch.Tensor], encoded: bool = False):
    """(gbuf, g_pts, g_dirs) of a point query -- or, encoded=True, (gbuf, g_encoded [P, input_ch + input_ch_views], None): the
    input-gradient chain of mlp_generic_input_grads continued to the query's own inputs (`packed_bwd` of run_bwd(input_grads=True))."""
    g_raw, acts = _dev(g_raw, "g_raw"), _dev(acts, "acts")
    P_ = acts.shape[0]
    gbuf = torch.empty_like(acts)
    f32 = dict(device=acts.device, dtype=torch.float32)
    g_pts = g_dirs = g_enc = None
    if encoded:
        x_dim = 3 if plan.desc.xyz_freqs < 0 else 3 + 6 * plan.desc.xyz_freqs
        v_dim = 0 if not plan.desc.use_viewdirs else (3 if plan.desc.dir_freqs < 0 else 3 + 6 * plan.desc.dir_freqs)
        g_enc = torch.empty((P_, x_dim + v_dim), **f32)
    else:
        g_pts = torch.empty((P_, 3), **f32)
        g_dirs = torch.empty((P_, 3), **f32) if plan.desc.use_viewdirs else None
    _lib.check(_lib.lib().nsos_mlp_generic_input_grads_points(C.byref(plan.desc), _p(packed_bwd), _p(g_raw), _p(acts), _p(gbuf),
                                                              _p(pts) if not encoded else None, _p(dirs) if (not encoded and dirs is not None) else None,
                                                              P_, _p(g_pts), _p(g_dirs), _p(g_enc), _stream()), "nsos_mlp_generic_input_grads_points")
    return (gbuf, g_enc, None) if encoded else (gbuf, g_pts, g_dirs)


def ray_grad_reduce(g_pts: torch.Tensor, g_dirs: Optional[torch.Tensor], z_vals: torch.Tensor, rays_d: torch.Tensor, raw: torch.Tensor,
                    g_raw: torch.Tensor, noise: Optional[torch.Tensor], noise_std: float) -> Tuple[torch.Tensor, torch.Tensor]:
    """(g_rays_o, g_rays_d) [R,3] of one pass from its per-point gradients (nsos_ray_grad_reduce): pts = o + d z, viewdirs = d / |d| and
    the renderer's dists * |d| (g_raw = the COMPOSITING's d loss / d raw, its sigma column carries d loss / d alpha)."""
    R, S = z_vals.shape
    Cn = raw.shape[-1]
    g_o = torch.empty((R, 3), device=z_vals.device, dtype=torch.float32)
    g_d = torch.empty((R, 3), device=z_vals.device, dtype=torch.float32)
    _lib.check(_lib.lib().nsos_ray_grad_reduce(_p(_dev(g_pts, "g_pts")), _p(g_dirs), _p(_dev(z_vals, "z_vals")), _p(_dev(rays_d, "rays_d")),
                                               _p(_dev(raw, "raw")), _p(_dev(g_raw, "g_raw")), _p(noise), float(noise_std), R, S, Cn, _p(g_o), _p(g_d),
                                               _stream()), "nsos_ray_grad_reduce")
    return g_o, g_d


def mlp_generic_forward_points(plan: GenericPlan, packed: torch.Tensor, pts: torch.Tensor, dirs: Optional[torch.Tensor]) -> torch.Tensor:
    """raw [P,C] for explicit points (and per-point view directions, if the net takes them)."""
    pts = _dev(pts, "pts")
    if dirs is not None:
        dirs = _dev(dirs, "dirs")
    raw = torch.empty((pts.shape[0], plan.out_channels), device=pts.device, dtype=torch.float32)
    _lib.check(_lib.lib().nsos_mlp_generic_forward_points(C.byref(plan.desc), _p(packed), _p(pts), _p(dirs), pts.shape[0], _p(raw), _stream()),
               "nsos_mlp_generic_forward_points")
    return raw


def pack_mlp(params: Dict[str, torch.Tensor], sem_mode: int, out: Optional[torch.Tensor] = None,
             precision: str = "fp32") -> torch.Tensor:
    """Gather one net's state-dict tensors (keys relative to `<net>.mlp.`) into the MFMA-order stream
    (fp32 exact-MFMA layout, or the 16-bit layout of the reduced-precision kernel)."""
    if precision not in DTYPES:
        raise ValueError(f"precision must be one of {list(DTYPES)}, got {precision!r}")
    return PackPlan(params, sem_mode).run(out, precision)


def mlp_forward_rays(packed: torch.Tensor, sem_mode: int, rays_o: torch.Tensor, rays_d: torch.Tensor,
                     viewdirs: torch.Tensor, z_vals: torch.Tensor) -> torch.Tensor:
    """raw [R,S,C] for the points o + d*z of each ray  (models/nerf_mlp.py:67-100,179-215)."""
    rays_o, rays_d = _dev(rays_o, "rays_o"), _dev(rays_d, "rays_d")
    viewdirs, z_vals = _dev(viewdirs, "viewdirs"), _dev(z_vals, "z_vals")
    R, S = z_vals.shape
    Cn = 4 if sem_mode == SEM_NONE else 6
    raw = torch.empty((R, S, Cn), device=z_vals.device, dtype=torch.float32)
    ev = None
    if KERNEL_EVENTS is not None:
        ev = (torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True))
        ev[0].record()
    _lib.check(_lib.lib().nsos_mlp_forward_rays(_p(packed), sem_mode, _p(rays_o), _p(rays_d), _p(viewdirs),
                                                _p(z_vals), R, S, _p(raw), _stream()), "nsos_mlp_forward_rays")
    if ev is not None:
        ev[1].record()
        KERNEL_EVENTS.append((R * S, ev[0], ev[1]))
    return raw


def lp_selected_kernel() -> int:
    """3 = mlp_lp16_kernel (default), 2 = mlp_lp8_kernel, 1 = mlp_lp_kernel (NSOS_LP_KERNEL / nsos_mlp_lp_select_kernel)."""
    return int(_lib.lib().nsos_mlp_lp_selected_kernel())


def mlp_forward_rays_lp(packed: torch.Tensor, sem_mode: int, precision: str, rays_o: torch.Tensor,
                        rays_d: torch.Tensor, viewdirs: torch.Tensor, z_vals: torch.Tensor) -> torch.Tensor:
    """K2 on the 16-bit matrix pipe: raw [R,S,C] fp32.  precision "fp16" / "bf16": reduced-precision MFMA inputs with
    fp32 accumulation; "fp16x3": split-fp16 operands, three MFMAs per product, fp32-grade results.  `packed` must
    come from pack_mlp(..., precision=precision)."""
    rays_o, rays_d = _dev(rays_o, "rays_o"), _dev(rays_d, "rays_d")
    viewdirs, z_vals = _dev(viewdirs, "viewdirs"), _dev(z_vals, "z_vals")
    R, S = z_vals.shape
    Cn = 4 if sem_mode == SEM_NONE else 6
    raw = torch.empty((R, S, Cn), device=z_vals.device, dtype=torch.float32)
    ev = None
    if KERNEL_EVENTS is not None:
        ev = (torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True))
        ev[0].record()
    if precision == "fp16x3":
        _lib.check(_lib.lib().nsos_mlp_forward_rays_x3(_p(packed), sem_mode, _p(rays_o), _p(rays_d), _p(viewdirs),
                                                       _p(z_vals), R, S, _p(raw), _stream()), "nsos_mlp_forward_rays_x3")
    else:
        check_lp_stream(packed, 1 if R * S >= (1 << 31) else lp_selected_kernel())
        _lib.check(_lib.lib().nsos_mlp_forward_rays_lp(_p(packed), sem_mode, DTYPES[precision], _p(rays_o), _p(rays_d),
                                                       _p(viewdirs), _p(z_vals), R, S, _p(raw), _stream()),
                   "nsos_mlp_forward_rays_lp")
    if ev is not None:
        ev[1].record()
        KERNEL_EVENTS.append((R * S, ev[0], ev[1]))
    return raw


def mlp_forward_rays_save(packed: torch.Tensor, sem_mode: int, rays_o: torch.Tensor, rays_d: torch.Tensor,
                          viewdirs: torch.Tensor, z_vals: torch.Tensor, precision: str = "fp32", compact: bool = False):
    """Training-mode K2 (frozen backbone): raw [R,S,6] plus the semantic head's saved inputs
    sem_in [R*S,320] = [relu(h7) | x63 | 1] and sem_hid [R*S,128] (see nsos_mlp_forward_rays_save[_lp]).
    `packed` must have been packed for the same `precision`.  compact (16-bit precisions only): sem_in AND sem_hid come back
    in the precision's own 16-bit dtype (sem_in's values are 16-bit anyway; sem_hid is rounded): 896 B per point instead of
    1792 to store and to read back in sem_head_wgrad."""
    if sem_mode == SEM_NONE:
        raise ValueError("mlp_forward_rays_save needs a semantic head")
    rays_o, rays_d = _dev(rays_o, "rays_o"), _dev(rays_d, "rays_d")
    viewdirs, z_vals = _dev(viewdirs, "viewdirs"), _dev(z_vals, "z_vals")
    R, S = z_vals.shape
    dev = z_vals.device
    raw = torch.empty((R, S, 6), device=dev, dtype=torch.float32)
    compact = compact and precision in ("fp16", "bf16")
    layout = int(_lib.lib().nsos_mlp_save16_layout(R * S)) if compact else 0
    if layout & SEM_IN_TILED:
        # tile-major (what the two-waves-per-SIMD kernel stores contiguously): [group of 32 points][K][kg * 32 + point][8 channels];
        # sem_head_wgrad recognises it by its four dimensions, sem_in_rows() turns it into [P,320]
        sem_in = torch.empty(((R * S + 31) // 32, 20, 64, 8), device=dev, dtype=torch.float16 if precision == "fp16" else torch.bfloat16)
    else:
        sem_in = torch.empty((R * S, 320), device=dev,
                             dtype=(torch.float16 if precision == "fp16" else torch.bfloat16) if compact else torch.float32)
    if layout & SEM_HID_TILED:
        # tile-major like sem_in (the default 16-bit kernel): [group of 32 points][octet 0..15][point][8 channels]; sem_hid_rows() -> [P,128]
        sem_hid = torch.empty(((R * S + 31) // 32, 16, 32, 8), device=dev, dtype=sem_in.dtype)
    else:
        sem_hid = torch.empty((R * S, 128), device=dev, dtype=sem_in.dtype if compact else torch.float32)
    ev = _ev_begin()
    if precision == "fp32":
        _lib.check(_lib.lib().nsos_mlp_forward_rays_save(_p(packed), sem_mode, _p(rays_o), _p(rays_d), _p(viewdirs),
                                                         _p(z_vals), R, S, _p(raw), _p(sem_in), _p(sem_hid), _stream()),
                   "nsos_mlp_forward_rays_save")
    elif precision == "fp16x3":
        _lib.check(_lib.lib().nsos_mlp_forward_rays_save_x3(_p(packed), sem_mode, _p(rays_o), _p(rays_d), _p(viewdirs),
                                                            _p(z_vals), R, S, _p(raw), _p(sem_in), _p(sem_hid), _stream()),
                   "nsos_mlp_forward_rays_save_x3")
    elif compact:
        check_lp_stream(packed, 1 if (R * S >= (1 << 31) or lp_selected_kernel() == 1) else lp_selected_kernel())
        _lib.check(_lib.lib().nsos_mlp_forward_rays_save16_lp(_p(packed), sem_mode, DTYPES[precision], _p(rays_o),
                                                              _p(rays_d), _p(viewdirs), _p(z_vals), R, S, _p(raw),
                                                              _p(sem_in), _p(sem_hid), _stream()),
                   "nsos_mlp_forward_rays_save16_lp")
    else:
        check_lp_stream(packed, 1)                 # fp32 sem_in / sem_hid come from the round-1 kernel whatever is selected
        _lib.check(_lib.lib().nsos_mlp_forward_rays_save_lp(_p(packed), sem_mode, DTYPES[precision], _p(rays_o),
                                                            _p(rays_d), _p(viewdirs), _p(z_vals), R, S, _p(raw),
                                                            _p(sem_in), _p(sem_hid), _stream()),
                   "nsos_mlp_forward_rays_save_lp")
    _ev_end(ev, R * S)
    return raw, sem_in, sem_hid


SEM_IN_TILED = 16   # NSOS_SEM_IN_TILED (include/nerf_sos_hip.h)
SEM_HID_TILED = 32  # NSOS_SEM_HID_TILED


def sem_hid_rows(sem_hid: torch.Tensor, n_points: int) -> torch.Tensor:
    """[P,128] row-major from the tile-major sem_hid [ceil(P/32), 16, 32, 8] of mlp_forward_rays_save(compact=True) with the
    default 16-bit kernel (channel 8 o + c of point 32 g + i at [g, o, i, c]); a row-major tensor is returned as it is."""
    if sem_hid.dim() != 4:
        return sem_hid
    G = sem_hid.shape[0]
    return sem_hid.permute(0, 2, 1, 3).reshape(G * 32, 128)[:n_points]


def sem_in_rows(sem_in: torch.Tensor, n_points: int) -> torch.Tensor:
    """[P,320] row-major from the tile-major sem_in [ceil(P/32), 20, 64, 8] of mlp_forward_rays_save(compact=True)
    (channel 16 K + 8 kg + c of point 32 g + i at [g, K, 32 kg + i, c]); a row-major tensor is returned as it is.
    Rows past P of the last group are never written by the kernel."""
    if sem_in.dim() != 4:
        return sem_in
    G = sem_in.shape[0]
    return sem_in.view(G, 20, 2, 32, 8).permute(0, 3, 1, 2, 4).reshape(G * 32, 320)[:n_points]


def sem_hid_tiled(rows: torch.Tensor) -> torch.Tensor:
    """The inverse of sem_hid_rows: a [P,128] 16-bit matrix in the tile-major layout (zero rows up to a multiple of 32)."""
    P = rows.shape[0]
    G = (P + 31) // 32
    pad = torch.zeros((G * 32, 128), device=rows.device, dtype=rows.dtype)
    pad[:P] = rows
    return pad.view(G, 32, 16, 8).permute(0, 2, 1, 3).contiguous()


def sem_in_tiled(rows: torch.Tensor) -> torch.Tensor:
    """The inverse of sem_in_rows: a [P,320] 16-bit matrix in the tile-major layout (zero rows up to a multiple of 32)."""
    P = rows.shape[0]
    G = (P + 31) // 32
    pad = torch.zeros((G * 32, 320), device=rows.device, dtype=rows.dtype)
    pad[:P] = rows
    return pad.view(G, 32, 20, 2, 8).permute(0, 2, 3, 1, 4).reshape(G, 20, 64, 8).contiguous()


def sem_head_backward(weights: torch.Tensor, g_semantics: torch.Tensor, sem2_w: torch.Tensor,
                      sem_hid: torch.Tensor) -> Tuple[torch.Tensor, torch.Tensor]:
    """Element-wise part of the semantic head's backward: (g_hid [P,128], g_logits [P,2])."""
    weights, g_semantics = _dev(weights, "weights"), _dev(g_semantics, "g_semantics")
    sem2_w, sem_hid = _dev(sem2_w, "semantic_linear.2.weight"), _dev(sem_hid, "sem_hid")
    R, S = weights.shape
    if tuple(g_semantics.shape) != (R, 2) or tuple(sem_hid.shape) != (R * S, 128) or tuple(sem2_w.shape) != (2, 128):
        raise ValueError("sem_head_backward: inconsistent shapes")
    g_hid = torch.empty((R * S, 128), device=weights.device, dtype=torch.float32)
    g_logits = torch.empty((R * S, 2), device=weights.device, dtype=torch.float32)
    _lib.check(_lib.lib().nsos_sem_head_backward(_p(weights), _p(g_semantics), _p(sem2_w), _p(sem_hid), R, S,
                                                 _p(g_hid), _p(g_logits), _stream()), "nsos_sem_head_backward")
    return g_hid, g_logits


_WGRAD_WS: Dict[torch.device, torch.Tensor] = {}


def sem_head_wgrad(weights: torch.Tensor, g_semantics: torch.Tensor, sem2_w: torch.Tensor, sem_hid: torch.Tensor,
                   sem_in: torch.Tensor, split_fp16: bool = False, in_dim: Optional[int] = None):
    """Backward of the semantic head in one pass (nsos_sem_head_wgrad): returns
    (gw1_aug [128,320] = [dW1 | . | db1 in column 319], dW2 [2,128], db2 [2]) -- or, with `in_dim` (semantic_linear.0's fan-in:
    256 or 319), (dW1 [128,in_dim], db1 [128], dW2, db2) as contiguous tensors straight from the reduction.
    split_fp16: the big reduction on the 16-bit matrix pipe with split operands (nsos_sem_head_wgrad_x3; needs S >= 8 and
    fewer than 2^31 points, else the exact kernel runs)."""
    weights, g_semantics = _dev(weights, "weights"), _dev(g_semantics, "g_semantics")
    sem2_w = _dev(sem2_w, "semantic_linear.2.weight")
    x_dtype = {torch.float32: 0, torch.float16: 1, torch.bfloat16: 2}.get(sem_in.dtype)
    if x_dtype is None or not sem_in.is_cuda:
        raise TypeError(f"sem_in must be a float32 / float16 / bfloat16 GPU tensor, got {sem_in.dtype} on {sem_in.device}")
    if not sem_hid.is_cuda or sem_hid.dtype not in (torch.float32, sem_in.dtype):
        raise TypeError(f"sem_hid must be a GPU tensor, float32 or of sem_in's dtype {sem_in.dtype}; got {sem_hid.dtype}")
    sem_in, sem_hid = sem_in.contiguous(), sem_hid.contiguous()
    R, S = weights.shape
    tiled = sem_in.dim() == 4                      # the tile-major layout of mlp_forward_rays_save(compact=True), see sem_in_rows
    if tiled and x_dtype == 0:
        raise TypeError("sem_head_wgrad: the tile-major sem_in layout is a 16-bit one")
    htiled = sem_hid.dim() == 4                    # tile-major hidden activations (the default 16-bit kernel), see sem_hid_rows
    if htiled and not tiled:
        raise TypeError("sem_head_wgrad: a tile-major sem_hid comes with a tile-major sem_in")
    if (tuple(g_semantics.shape) != (R, 2) or tuple(sem_hid.shape) != (((R * S + 31) // 32, 16, 32, 8) if htiled else (R * S, 128)) or tuple(sem2_w.shape) != (2, 128)
            or tuple(sem_in.shape) != (((R * S + 31) // 32, 20, 64, 8) if tiled else (R * S, 320))):
        raise ValueError("sem_head_wgrad: inconsistent shapes")
    dev = weights.device
    if dev not in _WGRAD_WS:
        _WGRAD_WS[dev] = torch.empty(_lib.lib().nsos_sem_head_wgrad_workspace_bytes() // 4, device=dev, dtype=torch.float32)
    ws = _WGRAD_WS[dev]
    gw1 = torch.empty((128, 320 if in_dim is None else int(in_dim)), device=dev, dtype=torch.float32)
    gb1 = None if in_dim is None else torch.empty((128,), device=dev, dtype=torch.float32)
    gw2 = torch.empty((2, 128), device=dev, dtype=torch.float32)
    gb2 = torch.empty((2,), device=dev, dtype=torch.float32)
    out = (gw1, gw2, gb2) if in_dim is None else (gw1, gb1, gw2, gb2)
    use_split = split_fp16 and S >= 8 and R * S < 2 ** 31
    if not use_split and x_dtype != 0:
        sem_in, x_dtype, tiled = sem_in_rows(sem_in, R * S).float(), 0, False          # the exact kernel reads fp32 rows
        sem_hid, htiled = sem_hid_rows(sem_hid, R * S), False
    if x_dtype == 0:
        sem_hid = sem_hid.float()                    # fp32 sem_in: fp32 hid (the 16-bit kernel reads both matrices in one format)
    elif sem_hid.dtype != sem_in.dtype:
        sem_hid = sem_hid.to(sem_in.dtype)
    if use_split:   # the power of two that keeps g_hid in fp16 range is derived (and divided out again) on the device
        _lib.check(_lib.lib().nsos_sem_head_wgrad_x3(_p(weights), _p(g_semantics), _p(sem2_w), _p(sem_hid), _p(sem_in),
                                                     x_dtype | (SEM_IN_TILED if tiled else 0) | (SEM_HID_TILED if htiled else 0),
                                                     R, S, None, _p(gw1), _p(gw2), _p(gb2), _p(ws), ws.numel() * 4,
                                                     _p(gb1), 0 if in_dim is None else int(in_dim), _stream()),
                   "nsos_sem_head_wgrad_x3")
        return out
    _lib.check(_lib.lib().nsos_sem_head_wgrad(_p(weights), _p(g_semantics), _p(sem2_w), _p(sem_hid), _p(sem_in), R, S,
                                              _p(gw1), _p(gw2), _p(gb2), _p(ws), ws.numel() * 4,
                                              _p(gb1), 0 if in_dim is None else int(in_dim), _stream()),
               "nsos_sem_head_wgrad")
    return out


def mlp_forward_points(packed: torch.Tensor, sem_mode: int, pts: torch.Tensor, dirs: torch.Tensor) -> torch.Tensor:
    """raw [P,C] for explicit points / per-point directions  (NeRFMLP.forward, models/nerf_mlp.py:179)."""
    pts, dirs = _dev(pts, "pts"), _dev(dirs, "viewdirs")
    P = pts.shape[0]
    if tuple(pts.shape) != (P, 3) or tuple(dirs.shape) != (P, 3):
        raise ValueError(f"pts / viewdirs must both be [P,3], got {tuple(pts.shape)} / {tuple(dirs.shape)}")
    Cn = 4 if sem_mode == SEM_NONE else 6
    raw = torch.empty((P, Cn), device=pts.device, dtype=torch.float32)
    _lib.check(_lib.lib().nsos_mlp_forward_points(_p(packed), sem_mode, _p(pts), _p(dirs), P, _p(raw), _stream()),
               "nsos_mlp_forward_points")
    return raw


# ------------------------------------------------------------------------------------------ train-mode draws
def render_draws(seed: int, call, n_rays: int, n_coarse: int, n_importance: int, device, jitter: bool = True,
                 noise: bool = True, importance: bool = True):
    """The four random tensors of one train-mode ray chunk in one launch (nsos_render_draws): (t_rand [R,S] or None,
    noise0 [R,S] or None, u [R,N] or None, noise1 [R,S+N] or None) from a Philox stream keyed by `seed`, block `call`.
    `call` may be a 1-element int64 device tensor holding the number of calls made SO FAR: the kernel then uses that + 1
    and advances the tensor (nsos_render_draws_counted) -- what a captured graph needs, same values as call = 1, 2, ..."""
    dev = torch.device(device)
    f = lambda *shape: torch.empty(shape, device=dev, dtype=torch.float32)  # noqa: E731
    fine = n_importance > 0
    t = f(n_rays, n_coarse) if jitter else None
    n0 = f(n_rays, n_coarse) if noise else None
    u = f(n_rays, n_importance) if (importance and fine) else None
    n1 = f(n_rays, n_coarse + n_importance) if (noise and fine) else None
    if isinstance(call, torch.Tensor):
        if call.dtype != torch.int64 or call.numel() != 1 or call.device != dev:
            raise TypeError("render_draws: a device call counter is a 1-element int64 tensor on the rays' device")
        _lib.check(_lib.lib().nsos_render_draws_counted(int(seed) & (2 ** 64 - 1), _p(call), n_rays, n_coarse, n_importance,
                                                        _p(t), _p(n0), _p(u), _p(n1), _stream()), "nsos_render_draws_counted")
        return t, n0, u, n1
    _lib.check(_lib.lib().nsos_render_draws(int(seed) & (2 ** 64 - 1), int(call) & (2 ** 64 - 1), n_rays, n_coarse, n_importance,
                                            _p(t), _p(n0), _p(u), _p(n1), _stream()), "nsos_render_draws")
    return t, n0, u, n1


# ------------------------------------------------------------------------------------------ K3
def composite(raw: torch.Tensor, z_vals: torch.Tensor, rays_d: torch.Tensor, noise: Optional[torch.Tensor] = None,
              noise_std: float = 0.0, white_bkgd: bool = False) -> Dict[str, torch.Tensor]:
    """VolumetricRenderer.forward (models/renderer.py:35-85); returns the reference's dict."""
    raw, z_vals, rays_d = _dev(raw, "raw"), _dev(z_vals, "z_vals"), _dev(rays_d, "rays_d")
    R, S, Cn = raw.shape
    if Cn > 6:
        # sem_dim > 2 (generic architectures only): the kernel composites two semantic channels per launch -- the same weights
        # every time, the extra maps two channels at a time (device-side copies of the channel slices; no shipped config comes here)
        ret = composite(raw[..., :6].contiguous(), z_vals, rays_d, noise, noise_std, white_bkgd)
        sems = [ret["semantics"]]
        for c in range(6, Cn, 2):
            sub = torch.cat([raw[..., :4], raw[..., c:min(c + 2, Cn)]], -1)
            sems.append(composite(sub, z_vals, rays_d, noise, noise_std, white_bkgd)["semantics"])
        ret["semantics"] = torch.cat(sems, -1)
        return ret
    if noise is not None:
        noise = _dev(noise, "noise")
        if tuple(noise.shape) != (R, S):
            raise ValueError(f"noise must be [{R},{S}]")
    dev = raw.device
    f = lambda *s: torch.empty(s, device=dev, dtype=torch.float32)  # noqa: E731
    weights, rgb, depth, acc, disp = f(R, S), f(R, 3), f(R, 1), f(R, 1), f(R, 1)
    sem = f(R, Cn - 4) if Cn > 4 else None
    _lib.check(_lib.lib().nsos_composite(_p(raw), _p(z_vals), _p(rays_d), _p(noise), float(noise_std), R, S, Cn,
                                         int(bool(white_bkgd)), _p(weights), _p(rgb), _p(sem), _p(depth), _p(acc),
                                         _p(disp), _stream()), "nsos_composite")
    ret = dict(rgb=rgb, disp=disp, acc=acc, weights=weights, depth=depth)
    if sem is not None:
        ret["semantics"] = sem
    return ret


def composite_importance(raw: torch.Tensor, z_vals: torch.Tensor, rays_d: torch.Tensor, n_importance: int,
                         noise: Optional[torch.Tensor] = None, noise_std: float = 0.0, white_bkgd: bool = False,
                         u: Optional[torch.Tensor] = None):
    """composite(...) of the coarse pass + importance_sample(...) on its weights in one launch (nsos_composite_importance).
    Returns (the composite dict, z_fine [R,S+N], z_samples [R,N], z_std [R]); bit-identical to the two separate calls."""
    raw, z_vals, rays_d = _dev(raw, "raw"), _dev(z_vals, "z_vals"), _dev(rays_d, "rays_d")
    R, S, Cn = raw.shape
    N = int(n_importance)
    if noise is not None:
        noise = _dev(noise, "noise")
        if tuple(noise.shape) != (R, S):
            raise ValueError(f"noise must be [{R},{S}]")
    if u is not None:
        u = _dev(u, "u")
        if tuple(u.shape) != (R, N):
            raise ValueError(f"u must be [{R},{N}]")
    dev = raw.device
    f = lambda *s: torch.empty(s, device=dev, dtype=torch.float32)  # noqa: E731
    weights, rgb, depth, acc, disp = f(R, S), f(R, 3), f(R, 1), f(R, 1), f(R, 1)
    sem = f(R, Cn - 4) if Cn > 4 else None
    z_fine, z_samples, z_std = f(R, S + N), f(R, N), f(R)
    _lib.check(_lib.lib().nsos_composite_importance(_p(raw), _p(z_vals), _p(rays_d), _p(noise), float(noise_std), R, S, Cn,
                                                    int(bool(white_bkgd)), _p(weights), _p(rgb), _p(sem), _p(depth), _p(acc),
                                                    _p(disp), _p(u), N, _p(z_fine), _p(z_samples), _p(z_std), _stream()),
               "nsos_composite_importance")
    ret = dict(rgb=rgb, disp=disp, acc=acc, weights=weights, depth=depth)
    if sem is not None:
        ret["semantics"] = sem
    return ret, z_fine, z_samples, z_std


def composite_backward(raw: torch.Tensor, z_vals: torch.Tensor, rays_d: torch.Tensor, noise: Optional[torch.Tensor] = None,
                       noise_std: float = 0.0, white_bkgd: bool = False, g_rgb=None, g_sem=None, g_depth=None, g_acc=None,
                       g_disp=None, g_weights=None) -> torch.Tensor:
    """d loss / d raw [R,S,C] given the gradients of VolumetricRenderer.forward's outputs (any may be None)."""
    raw, z_vals, rays_d = _dev(raw, "raw"), _dev(z_vals, "z_vals"), _dev(rays_d, "rays_d")
    R, S, Cn = raw.shape
    opt = lambda t, n, shape: None if t is None else _dev(t.reshape(shape), n)  # noqa: E731
    noise = opt(noise, "noise", (R, S))
    g_rgb, g_sem = opt(g_rgb, "g_rgb", (R, 3)), (opt(g_sem, "g_sem", (R, Cn - 4)) if Cn > 4 else None)
    g_depth, g_acc, g_disp = opt(g_depth, "g_depth", (R,)), opt(g_acc, "g_acc", (R,)), opt(g_disp, "g_disp", (R,))
    g_weights = opt(g_weights, "g_weights", (R, S))
    g_raw = torch.empty_like(raw)
    _lib.check(_lib.lib().nsos_composite_backward(_p(raw), _p(z_vals), _p(rays_d), _p(noise), float(noise_std), R, S, Cn,
                                                  int(bool(white_bkgd)), _p(g_rgb), _p(g_sem), _p(g_depth), _p(g_acc),
                                                  _p(g_disp), _p(g_weights), _p(g_raw), _stream()), "nsos_composite_backward")
    return g_raw


# ------------------------------------------------------------------------------------------ K4
def importance_sample(z_vals: torch.Tensor, weights: torch.Tensor, n_importance: int,
                      u: Optional[torch.Tensor] = None, cdf_in: Optional[torch.Tensor] = None,
                      debug: bool = False):
    """ImportanceSampler.forward (models/sampler.py:91-167) + z_std (models/nerf_net.py:124).
    Returns (z_fine [R,S+N], z_samples [R,N], z_std [R]) and, with debug=True, also (cdf [R,S-1], inds int64 [R,N])."""
    z_vals, weights = _dev(z_vals, "z_vals"), _dev(weights, "weights")
    R, S = z_vals.shape
    N = int(n_importance)
    if u is not None:
        u = _dev(u, "u")
        if tuple(u.shape) != (R, N):
            raise ValueError(f"u must be [{R},{N}]")
    if cdf_in is not None:
        cdf_in = _dev(cdf_in, "cdf_in")
    dev = z_vals.device
    z_fine = torch.empty((R, S + N), device=dev, dtype=torch.float32)
    z_samples = torch.empty((R, N), device=dev, dtype=torch.float32)
    z_std = torch.empty((R,), device=dev, dtype=torch.float32)
    cdf = torch.empty((R, S - 1), device=dev, dtype=torch.float32) if debug else None
    inds = torch.empty((R, N), device=dev, dtype=torch.int64) if debug else None
    _lib.check(_lib.lib().nsos_importance_sample(_p(z_vals), _p(weights), _p(u), _p(cdf_in), R, S, N, _p(z_fine),
                                                 _p(z_samples), _p(z_std), _p(cdf), _p(inds), _stream()),
               "nsos_importance_sample")
    if debug:
        return z_fine, z_samples, z_std, cdf, inds
    return z_fine, z_samples, z_std


# ------------------------------------------------------------------------------------------ evaluation post-processing
def eval_postprocess(semantics: Optional[torch.Tensor] = None, rgb: Optional[torch.Tensor] = None,
                     target: Optional[torch.Tensor] = None) -> Dict[str, torch.Tensor]:
    """engines/eval.py:44-57,79-86 without the trip to the host: ``sem_prob = softmax(semantics, -1)``,
    ``sem = argmax(sem_prob, -1)[..., None]`` (int32, as the reference's ``astype(np.int32)``), and
    ``mse = img2mse(rgb, target)``, ``psnr = mse2psnr(mse)`` (utils/image.py:125-137) as 1-element tensors.
    Leading dimensions are kept ([H,W,C] images or [R,C] ray lists)."""
    if semantics is None and (rgb is None or target is None):
        raise ValueError("eval_postprocess needs semantics and/or (rgb, target)")
    out: Dict[str, torch.Tensor] = {}
    first = semantics if semantics is not None else rgb
    dev = first.device
    lead = tuple(first.shape[:-1])
    n = 1
    for d in lead:
        n *= int(d)
    sem2d = prob = pred = None
    C_ = 0
    if semantics is not None:
        C_ = int(semantics.shape[-1])
        sem2d = _dev(semantics.reshape(n, C_), "semantics")
        prob = torch.empty((n, C_), device=dev, dtype=torch.float32)
        pred = torch.empty((n,), device=dev, dtype=torch.int32)
    rgb2d = tgt2d = metrics = None
    if rgb is not None and target is not None:
        if tuple(rgb.shape) != tuple(target.shape) or rgb.shape[-1] != 3 or tuple(rgb.shape[:-1]) != lead:
            raise ValueError(f"rgb {tuple(rgb.shape)} and target {tuple(target.shape)} must both be [..., 3] over the same rays")
        rgb2d, tgt2d = _dev(rgb.reshape(n, 3), "rgb"), _dev(target.reshape(n, 3), "target")
        metrics = torch.empty((2,), device=dev, dtype=torch.float32)
    ws = torch.empty((_lib.lib().nsos_eval_workspace_bytes() // 8,), device=dev, dtype=torch.float64)
    _lib.check(_lib.lib().nsos_eval_postprocess(_p(sem2d), _p(rgb2d), _p(tgt2d), n, C_, _p(prob), _p(pred), _p(metrics),
                                                _p(ws), _stream()), "nsos_eval_postprocess")
    if semantics is not None:
        out["sem_prob"] = prob.reshape(*lead, C_)
        out["sem"] = pred.reshape(*lead, 1)
    if metrics is not None:
        out["mse"], out["psnr"] = metrics[0:1], metrics[1:2]
    return out


# ------------------------------------------------------------------------------------------ K7: full backward
ACTS_FEAT, ACTS_VIEWS, ACTS_SEM, ACTS_X, ACTS_D, ACTS_DIM = 2048, 2304, 2432, 2560, 2624, 2656   # nerf_sos_hip.h


def mlp_forward_rays_save_all(packed: torch.Tensor, sem_mode: int, rays_o: torch.Tensor, rays_d: torch.Tensor,
                              viewdirs: torch.Tensor, z_vals: torch.Tensor, precision: str = "fp32", acts16: bool = False):
    """Training-mode K2 with every layer's activations stored: (raw [R,S,C], acts [R*S, ACTS_DIM], relu bit masks or None).
    precision "fp32" (exact kernel) or "fp16x3" (split-fp16 kernel, fp32-accurate); `packed` must match."""
    if precision not in ("fp32", "fp16x3"):
        raise NotImplementedError("the full backward needs fp32-accurate activations: precision 'fp32' or 'fp16x3'")
    rays_o, rays_d = _dev(rays_o, "rays_o"), _dev(rays_d, "rays_d")
    viewdirs, z_vals = _dev(viewdirs, "viewdirs"), _dev(z_vals, "z_vals")
    R, S = z_vals.shape
    dev = z_vals.device
    raw = torch.empty((R, S, 4 if sem_mode == SEM_NONE else 6), device=dev, dtype=torch.float32)
    if acts16 and precision != "fp16x3":
        raise NotImplementedError("16-bit saved activations exist for the split-fp16 kernels (precision 'fp16x3')")
    # acts16: the split kernel's hi parts as they are (IEEE half): 5.3 KB per point instead of 10.6
    acts = torch.empty((R * S, ACTS_DIM), device=dev, dtype=torch.float16 if acts16 else torch.float32)
    ev = _ev_begin()
    if precision == "fp16x3":   # also returns the trunk layers' ReLU patterns as bit masks (input of mlp_input_grads_x3)
        masks = torch.empty(int(_lib.lib().nsos_mlp_relu_masks_bytes_x3(R * S)) // 4, device=dev, dtype=torch.int32)
        fn = _lib.lib().nsos_mlp_forward_rays_save_all16_x3 if acts16 else _lib.lib().nsos_mlp_forward_rays_save_all_x3
        _lib.check(fn(_p(packed), sem_mode, _p(rays_o), _p(rays_d), _p(viewdirs), _p(z_vals), R, S, _p(raw), _p(acts), _p(masks), _stream()),
                   "nsos_mlp_forward_rays_save_all_x3")
        _ev_end(ev, R * S)
        return raw, acts, masks
    _lib.check(_lib.lib().nsos_mlp_forward_rays_save_all(_p(packed), sem_mode, _p(rays_o), _p(rays_d), _p(viewdirs), _p(z_vals),
                                                         R, S, _p(raw), _p(acts), _stream()), "nsos_mlp_forward_rays_save_all")
    _ev_end(ev, R * S)
    return raw, acts, None


GBUF_DIM = 2560


def mlp_input_grads_x3(packed_bwd: torch.Tensor, sem_mode: int, g_raw: torch.Tensor, acts: torch.Tensor,
                       scale: torch.Tensor, masks: Optional[torch.Tensor] = None) -> torch.Tensor:
    """Fused split-fp16 input-gradient chain of the full backward (K7-X3): gbuf [P, GBUF_DIM] = scale * (d loss / d every
    layer's pre-activation), columns as in `acts` (256 l | ACTS_FEAT | ACTS_VIEWS | ACTS_SEM).  `packed_bwd` comes from
    pack_mlp(..., precision="fp16x3_bwd"); `scale` is a 3-element device tensor of powers of two -- trunk scale, colour-branch
    factor, semantic-branch factor (include/nerf_sos_hip.h; a 1-element tensor means factors of 1) --; `masks` = the bit
    masks mlp_forward_rays_save_all(..., "fp16x3") returned for the same points (None: trunk masks are read from `acts`)."""
    g_raw = _dev(g_raw, "g_raw")
    a16 = isinstance(acts, torch.Tensor) and acts.dtype == torch.float16
    if a16:
        if not acts.is_cuda or masks is None:
            raise ValueError("mlp_input_grads_x3: 16-bit activations are a GPU tensor and come with the forward's bit masks")
    else:
        acts = _dev(acts, "acts")
    P_, C_ = g_raw.shape
    if C_ != (4 if sem_mode == SEM_NONE else 6) or acts.shape != (P_, ACTS_DIM) or not acts.is_contiguous():
        raise ValueError(f"mlp_input_grads_x3: g_raw {tuple(g_raw.shape)} / acts {tuple(acts.shape)} do not fit sem_mode {sem_mode}")
    scale = scale.reshape(-1)
    if scale.numel() == 1:
        scale = torch.cat([scale, torch.ones(2, device=scale.device, dtype=scale.dtype)])
    if scale.numel() != 3:
        raise ValueError("mlp_input_grads_x3: `scale` holds one or three powers of two")
    scale = _dev(scale.float(), "scale")
    gbuf = torch.empty((P_, GBUF_DIM), device=acts.device, dtype=torch.float32)
    if masks is not None and (not masks.is_cuda or masks.numel() * 4 < int(_lib.lib().nsos_mlp_relu_masks_bytes_x3(P_))):
        raise ValueError("mlp_input_grads_x3: `masks` does not belong to these points")
    fn = _lib.lib().nsos_mlp_input_grads_x3_a16 if a16 else _lib.lib().nsos_mlp_input_grads_x3
    _lib.check(fn(_p(packed_bwd), sem_mode, _p(g_raw), _p(acts), _p(masks), P_, _p(scale), _p(gbuf), _stream()), "nsos_mlp_input_grads_x3")
    return gbuf


_WG_WS: Dict[torch.device, torch.Tensor] = {}


def _rows(t: torch.Tensor, name: str, half_ok: bool = False):
    """(pointer-bearing tensor, row stride) of a 2-D row-major slice with unit column stride."""
    if not t.is_cuda or (t.dtype != torch.float32 and not (half_ok and t.dtype == torch.float16)) or t.dim() != 2 or t.stride(1) != 1:
        raise ValueError(f"{name}: need a float32 GPU matrix with contiguous rows, got {t.dtype} {tuple(t.shape)} {t.stride()}")
    return t, t.stride(0)


def wgrad(G: torch.Tensor, X: torch.Tensor, dW: torch.Tensor, db: Optional[torch.Tensor] = None, split_fp16: bool = False) -> None:
    """dW [M,N] = G^T X, db [M] = column sums of G (nsos_wgrad); G [P,M], X [P,N] may be column slices of wider
    buffers, dW a column slice of a weight-gradient matrix.  M, N in {32,64,128,256}.
    split_fp16: for M = N = 256 run on the 16-bit matrix pipe with split operands (nsos_wgrad_x3); the caller guarantees
    |G|, |X| < 65504 (the fused input-gradient kernel's output is scaled into that range)."""
    G, ldg = _rows(G, "G")
    X, ldx = _rows(X, "X", half_ok=True)      # 16-bit saved activations (mlp_forward_rays_save_all(..., acts16=True)): widened exactly
    xh = X.dtype == torch.float16
    dW, ldw = _rows(dW, "dW")
    P_, M = G.shape
    N = X.shape[1]
    if X.shape[0] != P_ or tuple(dW.shape) != (M, N) or (db is not None and (tuple(db.shape) != (M,) or not db.is_contiguous())):
        raise ValueError("wgrad: inconsistent shapes")
    dev = G.device
    if dev not in _WG_WS:
        _WG_WS[dev] = torch.empty(_lib.lib().nsos_wgrad_workspace_bytes() // 4, device=dev, dtype=torch.float32)
    ws = _WG_WS[dev]
    if split_fp16 and M == 256 and N == 256:
        fn = _lib.lib().nsos_wgrad_x3_xh if xh else _lib.lib().nsos_wgrad_x3
        _lib.check(fn(G.data_ptr(), ldg, X.data_ptr(), ldx, P_, dW.data_ptr(), ldw, _p(db), _p(ws), ws.numel() * 4, _stream()), "nsos_wgrad_x3")
        return
    fn = _lib.lib().nsos_wgrad_xh if xh else _lib.lib().nsos_wgrad
    _lib.check(fn(G.data_ptr(), ldg, X.data_ptr(), ldx, P_, M, N, dW.data_ptr(), ldw, _p(db), _p(ws), ws.numel() * 4, _stream()), "nsos_wgrad")


def wgrad_batch(items, n_items: int, G: torch.Tensor, X: torch.Tensor, out: torch.Tensor) -> None:
    """A prepared list of nsos_wgrad reductions over column blocks of G [P, ldg] and X [P, ldx] into the flat fp32 buffer `out`
    (nsos_wgrad_batch; backward.generic_weight_grads builds the list)."""
    G, X, out = _dev(G, "G"), _dev(X, "X"), _dev(out, "out")
    if G.dim() != 2 or X.dim() != 2 or G.shape[0] != X.shape[0] or G.dtype != torch.float32 or X.dtype != torch.float32 or out.dtype != torch.float32:
        raise ValueError("wgrad_batch: G [P, ldg] and X [P, ldx] must be fp32 matrices over the same points")
    dev = G.device
    if dev not in _WG_WS:
        _WG_WS[dev] = torch.empty(_lib.lib().nsos_wgrad_workspace_bytes() // 4, device=dev, dtype=torch.float32)
    ws = _WG_WS[dev]
    _lib.check(_lib.lib().nsos_wgrad_batch(items, int(n_items), G.data_ptr(), G.shape[1], X.data_ptr(), X.shape[1], G.shape[0],
                                          out.data_ptr(), _p(ws), ws.numel() * 4, _stream()), "nsos_wgrad_batch")


def relu_mask_(g: torch.Tensor, h: torch.Tensor) -> torch.Tensor:
    """In place g *= (h > 0) for row-major matrices / column slices with the same shape."""
    g, ldg = _rows(g, "g")
    h, ldh = _rows(h, "h")
    if g.shape != h.shape:
        raise ValueError("relu_mask_: shape mismatch")
    _lib.check(_lib.lib().nsos_relu_mask(g.data_ptr(), ldg, h.data_ptr(), ldh, g.shape[0], g.shape[1], _stream()), "nsos_relu_mask")
    return g
