"""Drop-in modules for the reference's render path, backed by the gfx950 HIP kernels.

Interface mirrored (reference file:line):
  * ``NeRFNet(**ctor kwargs)``, ``NeRFNet.forward(ray_batch, bound_batch, **kwargs)``,
    ``NeRFNet.render_rays(...)``                                   -- models/nerf_net.py:22-195
  * ``NeRFMLP(...)(pts, viewdirs=...)`` point query                -- models/nerf_mlp.py:132-215
  * ``MLP`` parameter container, same names / shapes / init order  -- models/nerf_mlp.py:24-64
so that ``state_dict()`` keys (``nerf.mlp.pts_linears.0.weight`` ...) and reference checkpoints are
interchangeable (engines/trainer.py:216-222, run_nerf.py:350-360), and the callers
``model(batch_rays, (near, far), radii=radii)`` (engines/trainer.py:68, engines/eval.py:40) and
``model.nerf_fine(pts, viewdirs=viewdirs)`` (engines/eval.py:297) work unchanged.

What is NOT here on purpose: any arithmetic.  Modules only hold parameters and sequence kernel
launches (ops.py -> include/nerf_sos_hip.h).  Non-GPU tensors raise; unsupported architectures raise.
"""
from __future__ import annotations

import math
from collections import OrderedDict
from typing import Dict, Optional

import torch
import torch.nn as nn

from . import ops

# The architecture every shipped NeRF-SOS config uses (netdepth=8, netwidth=256, skips=[4], viewdirs=True, use_embed=True,
# multires=10, multires_views=4, sem_layer<=2, sem_dim=2, sem_with_geo=False) has hand-scheduled kernels in three precisions and
# backward kernels; every other architecture the reference's constructors accept renders through the generic fp32 kernel
# (MLP.fast below; csrc/mlp_generic.hip): inference, training of any parameter subset, gradients to the rays.  conv_embed=True is the one constructor argument refused outright.


def _named_params(module: nn.Module):
    """[(qualified name, parameter)] in `named_parameters()` order, without its per-call recursion (prefix strings, de-duplication
    set, generator frames: ~0.4 ms per training step over the path's call sites).  The module tree of this package is fixed
    after construction, so the (owner module, attribute) slots are resolved once; the parameter OBJECTS are looked up on every
    call, so a replaced or re-typed parameter is seen."""
    slots = module.__dict__.get("_nsos_param_slots")
    if slots is None:
        seen, slots = set(), []
        for mod_name, mod in module.named_modules():
            for attr, prm in mod._parameters.items():
                if prm is not None and id(prm) not in seen:
                    seen.add(id(prm))
                    slots.append(((mod_name + "." if mod_name else "") + attr, mod, attr))
        module.__dict__["_nsos_param_slots"] = slots
    return [(n, m._parameters[a]) for n, m, a in slots if m._parameters.get(a) is not None]


def fc_block(in_f, out_f):
    """models/nerf_mlp.py:18-22 (the deep semantic head's middle layers)."""
    return nn.Sequential(nn.Linear(in_f, out_f), nn.ReLU())


class MLP(nn.Module):
    """Parameter container of the NeRF MLP with the semantic head, for every architecture the reference's constructor builds.
    Layers are created in the reference's order (models/nerf_mlp.py:40-64) so a given torch seed yields identical initial
    weights, with the reference's module names (state_dict keys).  `fast` marks the architecture every shipped config uses
    (8 x 256, skips [4], 63 / 27 encoded inputs, view directions, the two-Linear head with sem_dim 2): that one runs on the
    hand-scheduled kernels (exact fp32, 16-bit, split-fp16; training); everything else renders through the generic fp32
    kernels (csrc/mlp_generic.hip): inference and training."""

    def __init__(self, D=8, W=256, input_ch=63, input_ch_views=27, output_ch=4, skips=(4,), use_viewdirs=True,
                 use_semantics=True, sem_layer=2, sem_dim=2, sem_with_coord=False, sem_with_geo=False):
        super().__init__()
        if output_ch != 4:
            raise NotImplementedError("nerf_sos_amd.MLP: output_ch must be 4 (rgb + sigma; NeRFNet never builds anything else, models/nerf_net.py:44-56)")
        self.D, self.W = D, W
        self.input_ch, self.input_ch_views = input_ch, input_ch_views
        self.skips = list(skips)
        self.use_viewdirs, self.use_semantics, self.sem_with_coord = use_viewdirs, use_semantics, sem_with_coord
        self.fast = (D == 8 and W == 256 and tuple(skips) == (4,) and bool(use_viewdirs) and input_ch == 63 and input_ch_views == 27
                     and (not use_semantics or (sem_layer <= 2 and sem_dim == 2 and not sem_with_geo)))
        self.pts_linears = nn.ModuleList(
            [nn.Linear(input_ch, W)] +
            [nn.Linear(W + input_ch, W) if i in self.skips else nn.Linear(W, W) for i in range(D - 1)])
        if use_viewdirs:
            self.alpha_linear = nn.Linear(W, 1)
            self.feature_linear = nn.Linear(W, W)
            self.views_linears = nn.ModuleList([nn.Linear(input_ch_views + W, W // 2)])
            self.rgb_linear = nn.Linear(W // 2, output_ch - 1)
        else:
            self.output_linear = nn.Linear(W, output_ch)                                       # models/nerf_mlp.py:55
        if use_semantics:
            sem_in = W + input_ch if sem_with_coord else W
            if sem_layer <= 2:
                self.semantic_linear = nn.Sequential(nn.Linear(sem_in, W // 2), nn.ReLU(), nn.Linear(W // 2, sem_dim))
            else:                                                                              # models/nerf_mlp.py:63
                self.semantic_linear = nn.Sequential(nn.Linear(sem_in, W), nn.ReLU(), *[fc_block(W, W) for _ in range(sem_layer - 3)],
                                                     nn.Linear(W, W // 2), nn.ReLU(), nn.Linear(W // 2, sem_dim))
            self.geo_map_sem = nn.Sequential(nn.Linear(1, W // 2), nn.ReLU(), nn.Linear(W // 2, sem_dim)) if sem_with_geo else None

    @property
    def sem_mode(self) -> int:
        return ops.sem_mode_of(self.use_semantics, self.sem_with_coord)

    # ---- MLP.forward on PRE-ENCODED inputs (models/nerf_mlp.py:67-100): no in-scope caller uses it (NeRFMLP / NeRFNet own the encoding
    #      and fuse it into the kernels), but it is the reference's signature: the generic kernels read the encoded row as given.
    def _generic_plan(self):
        ptrs = tuple(p.data_ptr() for p in self.parameters())
        if self.__dict__.get("_enc_plan") is None or self._enc_plan.param_ptrs != ptrs:
            def octaves(ch):
                if ch == 3:
                    return None
                if ch < 3 or (ch - 3) % 6:
                    raise NotImplementedError(f"nerf_sos_amd.MLP.forward: an input of {ch} channels is not 3 + 6 L (models/embedder.py:21-32)")
                return (ch - 3) // 6
            self.__dict__["_enc_plan"] = ops.GenericPlan(self, octaves(self.input_ch), octaves(self.input_ch_views) if self.use_viewdirs else None)
            self._enc_plan.param_ptrs = ptrs
            self.__dict__["_enc_packed"] = {}
        return self._enc_plan

    def packed_weights(self, precision: str = "generic") -> torch.Tensor:
        plan = self._generic_plan()
        self._enc_packed["generic"] = plan.run(self._enc_packed.get("generic"))       # (every call: see NeRFMLP.packed_weights)
        return self._enc_packed["generic"]

    def packed_bwd_generic(self, input_grads: bool = False) -> torch.Tensor:
        plan = self._generic_plan()
        mask = plan.trainable_mask(self)
        key = ("generic_bwd_in" if mask is None else "generic_bwd_in_sub") if input_grads else ("generic_bwd" if mask is None else "generic_bwd_sub")
        self._enc_packed[key] = plan.run_bwd(self._enc_packed.get(key), input_grads=input_grads, trainable=mask)
        return self._enc_packed[key]

    def __getstate__(self):  # copy.deepcopy / pickling: the plan holds raw device pointers, the streams are derived data
        state = self.__dict__.copy()
        state.pop("_enc_plan", None), state.pop("_enc_packed", None)
        return state

    def forward(self, x):
        """outputs [..., 4 (+ sem_dim)] for pre-encoded inputs x [..., input_ch + input_ch_views] (models/nerf_mlp.py:67-100), fp32 on the
        generic kernels; differentiable w.r.t. the parameters and x."""
        width = self.input_ch + (self.input_ch_views if self.use_viewdirs else 0)
        if x.shape[-1] != width:
            raise ValueError(f"nerf_sos_amd.MLP.forward: expected {width} encoded channels, got {x.shape[-1]}")
        lead = x.shape[:-1]
        flat = x.reshape(-1, width).float()
        if torch.is_grad_enabled() and (flat.requires_grad or any(p.requires_grad for p in self.parameters())):
            raw = _GenericQuery.apply(self, self, "encoded", flat, None, *[p for _, p in _named_params(self)])
        else:
            raw = ops.mlp_generic_forward_points_save(self._generic_plan(), self.packed_weights(), None, None, encoded=flat.contiguous(), save=False)[0]
        return raw.reshape(list(lead) + [raw.shape[-1]])


class NeRFMLP(nn.Module):
    """Point query with embedding (models/nerf_mlp.py:132-215): ``raw = nerf(pts[...,3], viewdirs[...,3])``."""

    def __init__(self, input_dim=3, output_dim=4, net_depth=8, net_width=256, skips=(4,), viewdirs=True,
                 use_embed=True, multires=10, multires_views=4, conv_embed=False, netchunk=1024 * 64,
                 use_semantics=False, sem_layer=2, sem_dim=2, sem_with_coord=False, sem_with_geo=False):
        super().__init__()
        if input_dim != 3:
            raise NotImplementedError("nerf_sos_amd.NeRFMLP: input_dim must be 3 (NeRFNet never builds anything else, models/nerf_net.py:44-56)")
        if conv_embed:
            # a Conv1d over the SAMPLES of a ray between the encoding and the MLP (models/nerf_mlp.py:152-158,196-209): couples
            # neighbouring points, no shipped config turns it on
            raise NotImplementedError("nerf_sos_amd.NeRFMLP: conv_embed=True is not implemented")
        if viewdirs and not use_embed:
            # the reference builds this net and then fails in its first forward: `embeddirs` stays None, so the directions are never
            # appended (models/nerf_mlp.py:142-150,203) while MLP.forward still splits 3 view channels off its input (:68)
            raise ValueError("nerf_sos_amd.NeRFMLP: use_embed=False needs viewdirs=False (the reference's own forward fails otherwise)")
        self.chunk = netchunk  # kept for interface parity; the fused kernels need no point chunking
        self.use_viewdirs = bool(viewdirs)
        self.multires = int(multires) if use_embed else None
        self.multires_views = (int(multires_views) if use_embed else None) if viewdirs else None
        input_ch = 3 + 6 * multires if use_embed else 3                                       # PositionEncoder.out_dim (models/embedder.py:21-32)
        input_ch_views = (3 + 6 * multires_views if use_embed else 3) if viewdirs else 0
        self.mlp = MLP(net_depth, net_width, skips=skips, input_ch=input_ch, output_ch=output_dim,
                       input_ch_views=input_ch_views, use_viewdirs=viewdirs, use_semantics=use_semantics,
                       sem_layer=sem_layer, sem_dim=sem_dim, sem_with_coord=sem_with_coord, sem_with_geo=sem_with_geo)
        self.fast = self.mlp.fast
        # arithmetic of point queries (forward): "fp32" exact MFMA; "fp16x3" / "fp16" / "bf16" as NeRFNet.mlp_precision, which sets it
        self.mlp_precision = "fp32"
        self._gplan = None     # ops.GenericPlan of the current parameter storages (generic architectures)
        self._frozen_key = {}  # precision -> keys of the frozen (non-head) parameters the packed stream was built from
        self._packed = {}      # precision -> packed stream
        self._packed_key = {}  # precision -> (data_ptr, version) of every parameter when it was packed
        self._plan = None      # ops.PackPlan of the current parameter storages

    @property
    def sem_mode(self) -> int:
        return self.mlp.sem_mode

    def packed_weights(self, precision: str = "fp32") -> torch.Tensor:
        """The MFMA-order weight stream for `precision`, packed on device from the parameters.

        Trainable parameters (any ``requires_grad``) are re-packed on EVERY call -- one ~6 us launch per net.  Nothing
        cheaper is safe: ``torch.optim.Adam(fused=True)`` (and every other fused/`.data`-style update) changes the values
        without bumping ``Tensor._version``, so a version-keyed cache would keep rendering -- and differentiating -- the
        initial weights while the optimizer moves the parameters.  A fully frozen net is packed once and re-packed when
        (data_ptr, _version) of a parameter changes (``load_state_dict``, in-place ops); after a ``p.data`` edit of a
        frozen net call invalidate_packed()."""
        named = _named_params(self.mlp)
        params = [p for _, p in named]
        if not self.fast or precision == "generic":     # ("generic": the shipped architecture on the generic kernels -- ray gradients)
            if precision not in ("fp32", "generic"):
                raise NotImplementedError(f"nerf_sos_amd: mlp_precision {precision!r} exists for the shipped architecture only "
                                          "(8 x 256, skips [4], multires 10 / 4, view directions, two-Linear head); this net renders in fp32")
            # (trainable: re-packed on every call, for the reason in the docstring; frozen: keyed by (data_ptr, _version))
            key = None if any(p.requires_grad for p in params) else tuple((p.data_ptr(), p._version) for p in params)
            ptrs = tuple(p.data_ptr() for p in params)
            if self._gplan is None or self._gplan.param_ptrs != ptrs:
                self._gplan = ops.GenericPlan(self.mlp, self.multires, self.multires_views)
                self._gplan.param_ptrs = ptrs
                self._packed_key.pop("generic", None)
            if key is None or "generic" not in self._packed or self._packed_key.get("generic") != key:
                self._packed["generic"] = self._gplan.run(self._packed.get("generic"))
                self._packed_key["generic"] = key
            return self._packed["generic"]
        ptrs = tuple(p.data_ptr() for p in params)
        if self._plan is None or self._plan.ptrs != ptrs:
            self._plan = ops.PackPlan(dict(named), self.sem_mode)
        trainable = any(p.requires_grad for p in params)
        key = None if trainable else tuple((p.data_ptr(), p._version) for p in params)
        if trainable or precision not in self._packed or key != self._packed_key.get(precision):
            # the shipped recipe trains the semantic heads alone (run_nerf.py:307-318): with the 16-bit streams, re-pack only their
            # chunks while the frozen trunk's (data_ptr, _version) keys stand (3 chunks per stream instead of 37-40 per step and net)
            heads_only = False
            if trainable and precision in ("fp16", "bf16") and self.sem_mode != ops.SEM_NONE:
                # (+ the kernel selection: a heads-only re-pack touches the selected kernel's stream only)
                frozen = (ops.lp_selected_kernel(),) + tuple((p.data_ptr(), p._version) for n, p in named if "semantic_linear" not in n)
                only_heads = all(("semantic_linear" in n) or not p.requires_grad for n, p in named)
                heads_only = only_heads and precision in self._packed and self._frozen_key.get(precision) == frozen
                self._frozen_key[precision] = frozen if only_heads else None
            self._packed[precision] = self._plan.run(self._packed.get(precision), precision, heads_only=heads_only)
            self._packed_key[precision] = key
        return self._packed[precision]

    def __getstate__(self):  # copy.deepcopy / pickling: the plan holds raw device pointers, the streams are derived data
        state = self.__dict__.copy()
        state["_plan"], state["_packed"], state["_packed_key"], state["_gplan"], state["_frozen_key"] = None, {}, {}, None, {}
        return state

    def invalidate_packed(self) -> None:
        """Drop the cached weight streams of a frozen net (see packed_weights: edits made through ``p.data`` bump neither
        data_ptr nor _version).  Trainable nets never need it."""
        self._packed.clear()
        self._packed_key.clear()
        self._frozen_key.clear()
        self._plan = self._gplan = None

    def query_rays(self, rays_o, rays_d, viewdirs, z_vals, save: bool = False, subset: bool = False, input_grads: bool = False):
        """raw [R,S,C] of a generic-architecture net for the points o + d z (NeRFNet's ray path); save=True: the training variant,
        (raw, acts) with every Linear's output saved per point (raw bit-identical)."""
        packed = self.packed_weights("generic")
        dirs = viewdirs if self.use_viewdirs else None
        if save:   # subset: store only what the backward of the parameters that require grad reads; input_grads: ... of a backward that
            # also reaches the rays (the ReLU patterns always travel as bits: a frozen net then stores no activation block at all)
            mask = self._gplan.trainable_mask(self.mlp) if (subset or input_grads) else None
            if mask is not None and input_grads:
                mask |= ops.INPUT_GRADS_BIT
            return ops.mlp_generic_forward_rays_save(self._gplan, packed, rays_o, rays_d, dirs, z_vals, trainable=mask)
        return ops.mlp_generic_forward_rays(self._gplan, packed, rays_o, rays_d, dirs, z_vals)

    def packed_bwd_generic(self, input_grads: bool = False) -> torch.Tensor:
        """The transposed weight streams of the generic input-gradient chain, packed from the CURRENT parameters (every call:
        training moves them, see packed_weights); input_grads: the program that also reaches the encodings (ray gradients).
        Without input gradients the chain is cut to what the parameters that require grad need (frozen backbone: the head alone)."""
        if self._gplan is None:
            self.packed_weights("generic")
        mask = self._gplan.trainable_mask(self.mlp)
        key = ("generic_bwd_in" if mask is None else "generic_bwd_in_sub") if input_grads else ("generic_bwd" if mask is None else "generic_bwd_sub")
        self._packed[key] = self._gplan.run_bwd(self._packed.get(key), input_grads=input_grads, trainable=mask)
        return self._packed[key]

    def forward(self, inputs, viewdirs=None):
        if self.use_viewdirs and viewdirs is None:
            raise ValueError("nerf_sos_amd.NeRFMLP: this net was built with viewdirs=True: pass the view directions "
                             "(the reference fails in embeddirs(None) here, models/nerf_mlp.py:203)")
        lead = inputs.shape[:-1]
        pts = inputs.reshape(-1, inputs.shape[-1]).float()
        dirs = viewdirs.expand(inputs.shape).reshape(-1, viewdirs.shape[-1]).float() if self.use_viewdirs else None
        if torch.is_grad_enabled() and (_trainable(self) or pts.requires_grad or (dirs is not None and dirs.requires_grad)):
            # the reference's point query is an ordinary differentiable module: here it runs on the generic fp32 kernels (any
            # architecture, the shipped one included), with gradients to the parameters, the points and the directions
            if self.mlp_precision != "fp32":
                raise NotImplementedError(f"nerf_sos_amd.NeRFMLP.forward: gradients of a point query exist in fp32 only (mlp_precision = {self.mlp_precision!r}): "
                                          "run it under torch.no_grad() or set mlp_precision = 'fp32'")
            raw = _GenericQuery.apply(self, self.mlp, "points", pts.contiguous(), None if dirs is None else dirs.contiguous(),
                                      *[p for _, p in _named_params(self.mlp)])
            return raw.reshape(list(lead) + [raw.shape[-1]])
        if self.fast and self.mlp_precision != "fp32":
            # the 16-bit / split-fp16 kernels take rays: every point is a ray of one sample with o = the point, d = 0, z = 0
            # (o + 0 * 0 is the point, bit for bit), its direction the ray's view direction
            raw = ops.mlp_forward_rays_lp(self.packed_weights(self.mlp_precision), self.sem_mode, self.mlp_precision, pts.contiguous(),
                                          torch.zeros_like(pts), dirs.contiguous(), pts.new_zeros((pts.shape[0], 1)))[:, 0]
        elif self.fast:
            raw = ops.mlp_forward_points(self.packed_weights(), self.sem_mode, pts, dirs)
        else:
            packed = self.packed_weights()
            raw = ops.mlp_generic_forward_points(self._gplan, packed, pts, dirs)
        return raw.reshape(list(lead) + [raw.shape[-1]])


def _trainable(module: nn.Module):
    if not torch.is_grad_enabled():
        return []
    return [n for n, p in _named_params(module) if p.requires_grad]


class _GenericQuery(torch.autograd.Function):
    """A point query (NeRFMLP.forward: mode "points", a = pts [P,3], b = dirs [P,3] or None) or MLP.forward on pre-encoded inputs (mode
    "encoded", a = x [P, input_ch + input_ch_views]) under autograd, on the generic kernels: forward = nsos_mlp_generic_forward_points_save
    (saved activations), backward = the input-gradient chain + the weight-gradient reductions (backward.generic_mlp_backward),
    continued to the query's own inputs when they ask for a gradient.  `owner` holds the streams (packed_weights("generic"),
    packed_bwd_generic, the plan), `mlp` the parameters."""

    @staticmethod
    def forward(ctx, owner, mlp, mode, a, b, *params):
        packed = owner.packed_weights("generic")
        plan = owner._gplan if hasattr(owner, "_gplan") else owner._generic_plan()
        with torch.no_grad():
            if mode == "encoded":
                raw, acts = ops.mlp_generic_forward_points_save(plan, packed, None, None, encoded=a.contiguous())
            else:
                raw, acts = ops.mlp_generic_forward_points_save(plan, packed, a, b)
        ctx.owner, ctx.mlp, ctx.mode, ctx.plan, ctx.acts, ctx.a, ctx.b = owner, mlp, mode, plan, acts, a, b
        ctx.in_grad = bool(ctx.needs_input_grad[3] or ctx.needs_input_grad[4])
        return raw

    @staticmethod
    def backward(ctx, g_raw):
        from .backward import generic_mlp_backward
        if ctx.acts is None:
            raise RuntimeError("nerf_sos_amd: backward through the same point query twice (the saved activations were released)")
        g_a = g_b = None
        packed_bwd = ctx.owner.packed_bwd_generic(input_grads=ctx.in_grad)
        g_raw = g_raw.contiguous()
        if ctx.in_grad:
            points = ("encoded",) if ctx.mode == "encoded" else (ctx.a, ctx.b)
            by_name, g_a, g_b = generic_mlp_backward(ctx.mlp, ctx.plan, ctx.acts, g_raw, packed_bwd, points=points)
        else:
            by_name = generic_mlp_backward(ctx.mlp, ctx.plan, ctx.acts, g_raw, packed_bwd)
        ctx.acts = None
        return (None, None, None, g_a, g_b) + tuple(by_name.get(n) for n, _ in _named_params(ctx.mlp))


_SEM_KEYS = ("semantic_linear.0.weight", "semantic_linear.0.bias", "semantic_linear.2.weight", "semantic_linear.2.bias")


class _FrozenBackboneRender(torch.autograd.Function):
    """render_rays with gradients for the semantic heads only -- the reference's shipped training recipe
    (run_nerf.py:307-318 + scripts/train_*_node0.sh --fix_backbone; engines/trainer.py:201).

    forward : the normal kernel sequence, with the SAVE variant of the fused MLP kernel that also stores the
              head's inputs; every output except `semantics` / `semantics0` is non-differentiable (they do not
              depend on the semantic parameters: weights/rgb/depth come from the frozen backbone).
    backward: ONE kernel per pass, nsos_sem_head_wgrad: g_logits = w G and g_hid = (hid > 0) (g_logits W2) are formed in
              registers and reduced over the points on the exact-fp32 MFMA:
              dW2 = g_logits^T hid, db2 = g_logits^T 1, [dW1 | db1] = g_hid^T [h7, x63, 1]."""


    @staticmethod
    def forward(ctx, net, args, kwargs, *sem_params):
        with torch.no_grad():
            ret, saved = net._render_rays_impl(*args, save=True, **kwargs)
        keys = list(ret.keys())
        outs = tuple(ret[k] for k in keys)
        ctx.mark_non_differentiable(*[o for k, o in zip(keys, outs) if k not in ("semantics", "semantics0")])
        # autograd hands backward() a freshly ZERO-FILLED tensor for every output that received no gradient -- the eleven
        # non-differentiable maps included ([R,192] weights ...): eleven fill launches per step for values nobody reads
        ctx.set_materialize_grads(False)
        ctx.keys, ctx.saved, ctx.net = keys, saved, net
        net._last_keys = keys
        return outs

    @staticmethod
    def backward(ctx, *gouts):
        net, saved = ctx.net, ctx.saved
        if saved is None:
            raise RuntimeError("nerf_sos_amd: backward through the same render twice (the saved operands were released)")
        g = dict(zip(ctx.keys, gouts))
        grads = []
        has_fine = "fine" in saved  # then the coarse pass's outputs carry the '0' suffix (models/nerf_net.py:126-128)
        for tag, mlp in net._sem_nets():
            sv = saved.get(tag)
            gs = g.get("semantics" if (tag == "fine" or not has_fine) else "semantics0") if sv else None
            if sv is None or gs is None:
                grads += [None] * 4
                continue
            w2 = mlp.mlp.semantic_linear[2].weight.detach()
            in_dim = mlp.mlp.semantic_linear[0].weight.shape[1]
            gw1, gb1, gw2, gb2 = ops.sem_head_wgrad(sv["weights"], gs.reshape(-1, 2).contiguous(), w2, sv["sem_hid"],
                                                    sv["sem_in"], split_fp16=sv["precision"] != "fp32",   # exact MFMA only on the exact path
                                                    in_dim=in_dim)     # dW1 / db1 as their own contiguous tensors: no slicing copies
            grads += [gw1, gb1, gw2, gb2]
        ctx.saved = None   # release the saved operands now: the node itself lives as long as the caller keeps the loss
        return (None, None, None) + tuple(grads)


class _FullRender(torch.autograd.Function):
    """render_rays with gradients for every MLP parameter (K7; the reference trains the whole model in
    configs/*_full.txt unless --fix_backbone is given, engines/trainer.py:201-203).

    forward : the normal kernel sequence with the SAVE==2 variant of the fused MLP kernel, which stores every layer's
              activations (10.4 KB per point); outputs are bit-identical to inference.  z_std and pts carry no
              gradient (the importance samples are detached, models/sampler.py:159).
    backward: per pass, nsos_composite_backward (d loss / d raw from the gradients of all rendered maps) then the MLP
              backward over the saved activations (backward.mlp_backward: the fused input-gradient chain and the
              weight-gradient reductions, all HIP kernels; no library GEMM)."""

    @staticmethod
    def forward(ctx, net, args, kwargs, rays_o, rays_d, *params):
        # rays_o / rays_d are args[0:2] again, as direct inputs: autograd hands their gradients back through this node when they
        # ask for one (pose refinement); then BOTH nets run on the generic kernels, whose chain reaches the encodings
        ctx.rays_grad = bool(ctx.needs_input_grad[3] or ctx.needs_input_grad[4])
        with torch.no_grad():
            ret, saved = net._render_rays_impl(*args, save="all", force_generic=ctx.rays_grad, **kwargs)
        keys = list(ret.keys())
        outs = tuple(ret[k] for k in keys)
        # `pts` = o + d z (retpts=True) carries a gradient to the rays in the reference (models/sampler.py:70,166; z is detached): it
        # stays differentiable when the rays ask for one, and carries none otherwise
        nodiff = ("z_std",) if ctx.rays_grad else ("z_std", "pts")
        ctx.mark_non_differentiable(*[o for k, o in zip(keys, outs) if k.rstrip("0") in nodiff])
        ctx.set_materialize_grads(False)     # outputs without a gradient arrive as None (backward filters them), not as zero fills
        ctx.keys, ctx.saved, ctx.net = keys, saved, net
        net._last_keys = keys
        return outs

    @staticmethod
    def backward(ctx, *gouts):
        from .backward import mlp_backward
        net, saved = ctx.net, ctx.saved
        if saved is None:
            raise RuntimeError("nerf_sos_amd: backward through the same render twice (the saved activations were released)")
        g = {k: v for k, v in zip(ctx.keys, gouts) if v is not None}
        has_fine = "fine" in saved
        grads = []
        g_rays_o = g_rays_d = None
        for tag, mlp in net._sem_nets():
            sv = saved.get(tag)
            names = [n for n, _ in _named_params(mlp.mlp)]
            if sv is None:
                grads += [None] * len(names)
                continue
            sfx = "0" if (tag == "coarse" and has_fine) else ""
            get = lambda k: g.get(k + sfx)  # noqa: E731
            if not any(get(k) is not None for k in ("rgb", "semantics", "depth", "acc", "disp", "weights", "raw", "pts")):
                # no output of this network reached the loss (e.g. a loss on the fine maps only: the fine samples are detached from the
                # coarse weights, models/sampler.py:159): autograd would not visit its branch either -- no compositing backward, no chain
                grads += [None] * len(names)
                continue
            g_raw = ops.composite_backward(sv["raw"], sv["z"], saved["rays_d"], sv["noise"], saved["noise_std"],
                                           net.white_bkgd, g_rgb=get("rgb"), g_sem=get("semantics"), g_depth=get("depth"),
                                           g_acc=get("acc"), g_disp=get("disp"), g_weights=get("weights"))
            if ctx.rays_grad and get("pts") is not None:            # d pts / d o = 1, d pts / d d = z
                gp = get("pts").reshape(sv["z"].shape + (3,))
                go, gd = gp.sum(1), (gp * sv["z"].unsqueeze(-1)).sum(1)
                g_rays_o = go if g_rays_o is None else g_rays_o + go
                g_rays_d = gd if g_rays_d is None else g_rays_d + gd
            g_raw_comp = g_raw                  # the compositing's own part: its sigma column carries d loss / d alpha (ray gradients)
            if get("raw") is not None:
                g_raw = g_raw + get("raw").reshape(g_raw.shape)
            if sv.get("generic"):
                from .backward import generic_mlp_backward
                if ctx.rays_grad:
                    by_name, g_pts, g_dirs = generic_mlp_backward(mlp.mlp, mlp._gplan, sv["acts"], g_raw.reshape(-1, g_raw.shape[-1]),
                                                                  mlp.packed_bwd_generic(input_grads=True),
                                                                  rays=(saved["rays_o"], saved["rays_d"], saved["viewdirs"], sv["z"]))
                    go, gd = ops.ray_grad_reduce(g_pts, g_dirs, sv["z"], saved["rays_d"], sv["raw"], g_raw_comp, sv["noise"], saved["noise_std"])
                    g_rays_o = go if g_rays_o is None else g_rays_o + go
                    g_rays_d = gd if g_rays_d is None else g_rays_d + gd
                else:
                    by_name = generic_mlp_backward(mlp.mlp, mlp._gplan, sv["acts"], g_raw.reshape(-1, g_raw.shape[-1]), mlp.packed_bwd_generic())
                grads += [by_name.get(n) for n in names]
                continue
            # fused input-gradient chain (K7-X3) for both precisions; "fp32": exact-fp32 weight-gradient reductions and the
            # trunk masks from the saved fp32 activations, "fp16x3": split-fp16 reductions and the forward's bit masks
            by_name = mlp_backward(mlp.mlp, mlp.sem_mode, sv["acts"], g_raw.reshape(-1, g_raw.shape[-1]),
                                   mlp.packed_weights("fp16x3_bwd"), sv["masks"],
                                   split_wgrad=net.mlp_precision != "fp32" or net.exact_weight_gradients is False)
            grads += [by_name.get(n) for n in names]
        ctx.saved = None   # release 10 KB/point of activations now (the node lives as long as the caller keeps the loss)
        return (None, None, None, g_rays_o, g_rays_d) + tuple(grads)


class NeRFNet(nn.Module):
    """Coarse + fine volumetric renderer with the reference's constructor and call contract
    (models/nerf_net.py:22-195).  Note the reference's spelling ``pts_chuck``."""

    _warned_full_16bit = False

    def __init__(self, netdepth=8, netwidth=256, netdepth_fine=8, netwidth_fine=256, N_samples=64, N_importance=64,
                 viewdirs=True, use_embed=True, multires=10, multires_views=4, conv_embed=False,
                 ray_chunk=1024 * 32, pts_chuck=1024 * 64, perturb=1., raw_noise_std=0., white_bkgd=False,
                 use_semantics=False, sem_layer=2, sem_dim=2, sem_with_coord=False, sem_with_geo=False):
        super().__init__()
        self.use_semantics = use_semantics
        self.N_samples, self.N_importance = N_samples, N_importance
        self.perturb, self.raw_noise_std, self.white_bkgd = perturb, raw_noise_std, white_bkgd
        self.chunk = ray_chunk
        self.use_viewdirs = viewdirs
        common = dict(input_dim=3, output_dim=4, skips=(4,), viewdirs=viewdirs, use_embed=use_embed,
                      multires=multires, multires_views=multires_views, conv_embed=conv_embed, netchunk=pts_chuck,
                      use_semantics=use_semantics, sem_layer=sem_layer, sem_dim=sem_dim,
                      sem_with_coord=sem_with_coord, sem_with_geo=sem_with_geo)
        self.nerf = NeRFMLP(net_depth=netdepth, net_width=netwidth, **common)
        self.nerf_fine = self.nerf  # models/nerf_net.py:49
        if N_importance > 0:
            self.nerf_fine = NeRFMLP(net_depth=netdepth_fine, net_width=netwidth_fine, **common)
        self.render_kwargs_train = {'N_importance': N_importance, 'N_samples': N_samples, 'perturb': perturb,
                                    'raw_noise_std': raw_noise_std, 'retraw': True, 'retpts': False}
        self.render_kwargs_test = dict(self.render_kwargs_train, perturb=0., raw_noise_std=0.)
        # Not in the reference (which is fp32 only): "fp32" = exact-fp32 MFMA (parity path, default);
        # "fp16" / "bf16" = 16-bit MFMA inputs with fp32 accumulation (BASELINE configs C5 / C3), inference only.
        # Point queries (NeRFMLP.forward, forward_pts, export_density) follow it: the setter hands it to both nets.
        self.mlp_precision = "fp32"
        # Precision of the COARSE network's pass alone (round 6; None = `mlp_precision`).  The hierarchical sampler places the 128
        # fine samples from the coarse weights, and on a trained field the image of a silhouette pixel depends on where they fall:
        # with the coarse pass in 16 bit, 0.11 % (fp16) / 0.19 % (bf16) of a 1008x756 image's rays move by more than 0.01 in rgb (up to
        # 0.4: the ray lands on the other side of a depth edge), ALL of it through the sample positions -- the fine network's own 16-bit
        # error on given positions is 9 rays / max 0.026 (scripts/diag/lp_outliers.py, profiles/r06/a_lp_outliers.json).
        # `coarse_precision = "fp16x3"` (split fp16, fp32-grade) with `mlp_precision = "fp16"` renders the fine pass -- 3/4 of the
        # points -- on the fast kernel and removes that tail at ~1.6x the time of an all-fp16 render.  Inference and the
        # frozen-backbone recipe follow it; a trainable backbone (full backward) ignores it.
        self.coarse_precision = None
        # Full backward (every parameter trainable), mlp_precision == "fp32": the 256x256 weight-gradient reductions run on the
        # exact-fp32 MFMA -- the precision whose name promises the reference's arithmetic gets it in the backward too
        # (VERDICT r03 weak-2).  False opts into the split-fp16 reductions on the 16-bit matrix pipe (fp32-grade: <= 1e-6 of scale
        # against fp64, HBM-bound, 5-6.6 ms less per 4096-ray step; what "fp16x3" always uses).  The forward is exact either way.
        self.exact_weight_gradients = True
        # Full backward on the split-fp16 kernels ("fp16x3", or a 16-bit precision with a trainable backbone): True keeps the saved
        # activations as 16-bit floats -- the hi parts of the split values the MFMAs consumed -- 5.3 KB per point instead of 10.6
        # (5.6 instead of 11.1 GB per 4096-ray step); the weight-gradient reductions widen them exactly.  Opt-in: the rounding of
        # the weight gradients' X operand to 11 bits shows as 2-5e-4 of a gradient's scale on a ten-ray batch (it averages out over
        # larger ones), outside the 1e-4 bar the default (fp32 activations) is held to; test_full_backward_compact_activations: 1.5e-3.
        self.compact_activations = False
        # Train-mode random draws.  "torch" (default): the reference's four torch.rand / torch.randn calls per ray chunk, in
        # its order, from torch's global generator (what the parity tests inject into).  "philox": ONE launch of the
        # package's counter-based generator per chunk (ops.render_draws), keyed by `rng_seed`, advanced per chunk.
        self.rng = "torch"
        self.rng_seed = 0
        self._rng_calls = 0
        # With rng == "philox": None keeps the call counter on the host (`_rng_calls`, passed by value); a 1-element int64
        # device tensor moves it into device memory (`use_device_rng_counter()`), which a captured graph of the training step
        # needs -- a by-value counter would be baked in and every replay would draw the same numbers.  Same draws either way.
        self.rng_counter: Optional[torch.Tensor] = None
        # fp16 range guard (round 5; VERDICT r04 missing-6).  fp16 tops out at 65 504 and loses weights below 6e-5: a field whose
        # hidden activations leave that range renders garbage under "fp16" / "fp16x3" -- and a ReLU can turn the resulting NaNs back
        # into finite numbers, so looking at the outputs alone does not catch it.  With `validate_precision` (default True) the FIRST
        # eval-mode render of a frozen net under such a precision, and the first one after its weights change, also renders up to
        # 1024 of the call's own rays (a constant stride over the whole call) with the exact fp32 kernels and raises FloatingPointError
        # if the two images disagree on more than 5 % of them (`check_numerics`).  Two extra renders of the sample (~2 ms) and a few host
        # syncs per weight version; nothing per step afterwards; set it to False to opt out.  Never during
        # stream capture, never for trainable nets (their weights move every step: call `check_numerics()` when it matters).
        self.validate_precision = True
        self._validated: Dict[str, tuple] = {}
        self._last_sample = None

    # Precisions whose range needs guarding (bf16 has fp32's exponent range: nothing to guard), and the share of the sampled rays
    # that may differ from the exact render by more than `_GUARD_RGB` before `check_numerics` raises.  An overflowed field differs on
    # most rays (a few dB); an in-range field differs on silhouette rays only -- 0.1-0.2 % of a trained image,
    # single rays by up to 0.4, which is why the verdict is a SHARE of rays and not a PSNR (round 6; ADVICE r05: a PSNR floor on
    # a small batch can be crossed by one importance-sample flip).
    _GUARDED = ("fp16", "fp16x3")
    _GUARD_RGB, _GUARD_SHARE = 0.05, 0.05

    @staticmethod
    def _strided(t, n_max: int):
        """At most `n_max` rows of t [R, ...] taken with a constant stride over the WHOLE call (the first rows of an image are
        often background: ADVICE r05)."""
        R = t.shape[0]
        if R <= n_max:
            return t
        step = -(-R // n_max)
        return t[::step]

    def check_numerics(self, ray_batch=None, bound_batch=None, max_rays: int = 1024) -> Dict[str, float]:
        """Render up to `max_rays` of `ray_batch` (a constant stride over the batch; default: the sample of the last validated render)
        in eval mode with the exact fp32 kernels and with `mlp_precision` / `coarse_precision`; raise FloatingPointError when the
        reduced-precision image (fine or coarse: `rgb`, `rgb0`) is non-finite or more than 5 % of the rays (and at least two) differ by more than 0.05 -- the
        signature of activations or weights outside fp16's range.  Returns the measured figures.  Costs two renders of the sample and
        host synchronisations; `NeRFNet.validate_precision` runs it once per weight version."""
        prec, cprec = self.mlp_precision, self.coarse_precision
        if ray_batch is None:
            if self._last_sample is None:
                raise ValueError("check_numerics: no rays given and no render to take them from yet")
            ray_batch, bound_batch = self._last_sample
        o, d = ray_batch
        o, d = self._strided(o.reshape(-1, 3).detach(), max_rays), self._strided(d.reshape(-1, 3).detach(), max_rays)
        near, far = bound_batch
        near = near if isinstance(near, (int, float)) else self._strided(near.reshape(-1), max_rays)
        far = far if isinstance(far, (int, float)) else self._strided(far.reshape(-1), max_rays)
        was_training, was_validating = self.training, self.validate_precision
        self.validate_precision = False
        try:
            self.eval()
            with torch.no_grad():
                got = self.forward((o, d), (near, far), retraw=False)
                self.mlp_precision, self.coarse_precision = "fp32", None
                want = self.forward((o, d), (near, far), retraw=False)
        finally:
            self.mlp_precision, self.coarse_precision = prec, cprec
            self.train(was_training)
            self.validate_precision = was_validating
        # both images: an overflowing COARSE pass alone leaves the fine image plausible (its 64 stratified samples are still there) while
        # the importance samples -- and `rgb0` -- are garbage
        maps = [k for k in ("rgb", "rgb0") if k in got and k in want]
        finite = all(bool(torch.isfinite(got[k]).all()) for k in maps) and bool(torch.isfinite(got["depth"]).all())
        n = int(o.shape[0])
        if finite:
            err = (got["rgb"].double() - want["rgb"].double()).reshape(n, -1)
            mse = float((err ** 2).mean())
            off = torch.zeros(n, dtype=torch.bool, device=err.device)
            for k in maps:
                off |= (got[k].double() - want[k].double()).reshape(n, -1).abs().amax(-1) > self._GUARD_RGB
            n_off = int(off.sum())
        else:
            mse, n_off = float("inf"), n
        psnr = -10.0 * math.log10(max(mse, 1e-30)) if finite else float("-inf")
        res = {"precision": prec, "coarse_precision": cprec, "rays": n, "finite": finite, "psnr_vs_fp32_db": psnr,
               "rays_off_by_more_than_0.05": n_off}          # (in `rgb` or `rgb0`)
        if not bool(torch.isfinite(want["rgb"]).all()):
            return res                      # the exact render itself is non-finite (inf / nan inputs propagate, as in the reference): no verdict
        guarded = prec in self._GUARDED or cprec in self._GUARDED
        if guarded and (not finite or (n_off >= 2 and n_off > self._GUARD_SHARE * n)):
            raise FloatingPointError(
                f"nerf_sos_amd: mlp_precision={prec!r}" + (f" / coarse_precision={cprec!r}" if cprec else "") + " does not reproduce this field: "
                + ("non-finite outputs" if not finite else f"{n_off} of {n} rays differ by more than {self._GUARD_RGB} in rgb / rgb0 (fine image {psnr:.1f} dB)")
                + " against the exact fp32 render of the same rays.  An activation beyond fp16's 65 504 or weights below its 6e-5 are the "
                "usual cause; use mlp_precision='bf16' (fp32's exponent range) or 'fp32' for this checkpoint.")
        return res

    def _maybe_validate(self, ray_batch, bound_batch) -> None:
        prec, cprec = self.mlp_precision, self.coarse_precision
        if (not self.validate_precision or not (prec in self._GUARDED or cprec in self._GUARDED) or self.training
                or torch.is_grad_enabled() and _trainable(self) or (ray_batch[0].is_cuda and torch.cuda.is_current_stream_capturing())):
            return
        key = tuple((p.data_ptr(), p._version) for p in self.parameters())
        if self._validated.get((prec, cprec)) == key:
            return
        o, d = (self._strided(t.detach().reshape(-1, 3), 1024).clone() for t in ray_batch)
        b = tuple(v if isinstance(v, (int, float)) else self._strided(v.detach().reshape(-1), 1024).clone() for v in bound_batch)
        self._last_sample = ((o, d), b)                   # (what an argument-less check_numerics() re-renders)
        self.check_numerics((o, d), b)
        self._validated[(prec, cprec)] = key

    def use_device_rng_counter(self, device=None) -> torch.Tensor:
        """Move the Philox call counter of `rng = "philox"` into device memory, continuing from the host count (see
        `rng_counter`) -- or from the device counter, if the module has one already.  Returns the counter tensor (number of
        draw launches so far); set `rng_counter = None` first to restart from the host count deliberately."""
        dev = torch.device(device if device is not None else next(self.parameters()).device)
        if self.rng_counter is not None:
            # a device counter exists already and is the truth (graph replays advance it without the host seeing them:
            # re-seeding from `_rng_calls` would rewind the Philox stream): continue from it
            if self.rng_counter.device != dev:
                self.rng_counter = self.rng_counter.to(dev)
            return self.rng_counter
        self.rng_counter = torch.full((1,), int(self._rng_calls), dtype=torch.int64, device=dev)
        return self.rng_counter

    def invalidate_packed(self) -> None:
        """Forget the packed weight streams of both networks (needed only after edits through ``param.data``, which
        autograd's version counters do not see; see NeRFMLP.invalidate_packed)."""
        self.nerf.invalidate_packed()
        self.nerf_fine.invalidate_packed()

    # ---------------------------------------------------------------------------------------------
    def _sem_nets(self):
        """(tag, NeRFMLP) pairs whose semantic heads can receive gradients, in the order their parameters are
        passed to _FrozenBackboneRender."""
        nets = [("coarse", self.nerf)]
        if self.nerf_fine is not self.nerf:
            nets.append(("fine", self.nerf_fine))
        return nets

    @property
    def mlp_precision(self) -> str:
        return self._mlp_precision

    @mlp_precision.setter
    def mlp_precision(self, value: str) -> None:
        if value not in ("fp32", "fp16x3", "fp16", "bf16"):
            raise ValueError(f"NeRFNet.mlp_precision must be 'fp32', 'fp16x3', 'fp16' or 'bf16', got {value!r}")
        self._mlp_precision = value
        for net in (self.nerf, self.nerf_fine):
            if net.fast:                  # generic architectures render in fp32 (render_rays refuses anything else)
                net.mlp_precision = value

    @property
    def coarse_precision(self) -> Optional[str]:
        return self._coarse_precision

    @coarse_precision.setter
    def coarse_precision(self, value: Optional[str]) -> None:
        if value not in (None, "fp32", "fp16x3", "fp16", "bf16"):
            raise ValueError(f"NeRFNet.coarse_precision must be None, 'fp32', 'fp16x3', 'fp16' or 'bf16', got {value!r}")
        self._coarse_precision = value

    def pass_precision(self, tag: str) -> str:
        """The arithmetic of one pass ("coarse" / "fine") of the shipped architecture's kernels: `coarse_precision` for the coarse
        pass when set, `mlp_precision` otherwise."""
        if tag == "coarse" and self._coarse_precision is not None:
            return self._coarse_precision
        return self._mlp_precision

    def render_rays(self, rays_o, rays_d, near, far, viewdirs=None, raw_noise_std=0., verbose=False,
                    retraw=False, retpts=False, pytest=False, **kwargs) -> Dict[str, torch.Tensor]:
        """One ray chunk: coarse sample -> MLP -> composite -> importance sample -> fine MLP -> composite
        (models/nerf_net.py:71-130).  Random tensors are drawn on the rays' device in the reference's
        order (rand[R,S], randn[R,S], rand[R,N], randn[R,S+N]) and handed to the kernels.
        Under autograd: only semantic heads trainable (the shipped --fix_backbone recipe) -> _FrozenBackboneRender;
        anything else trainable, any trainable generic-architecture net, or rays that require grad -> _FullRender (every
        parameter; the latter two on the generic fp32 kernels)."""
        args = (rays_o, rays_d, near, far, viewdirs, raw_noise_std, retraw, retpts)
        trainable = _trainable(self)
        generic = not (self.nerf.fast and self.nerf_fine.fast)
        if generic and (self.mlp_precision != "fp32" or self.coarse_precision not in (None, "fp32")):
            raise NotImplementedError(f"nerf_sos_amd.NeRFNet: mlp_precision {self.mlp_precision!r} / coarse_precision {self.coarse_precision!r} exist for the "
                                      "shipped architecture only; this one renders in fp32")
        rays_grad = torch.is_grad_enabled() and (rays_o.requires_grad or rays_d.requires_grad)
        if rays_grad:
            # the reference's autograd differentiates through pts = o + d z, viewdirs = d / |d| and dists * |d| (pose refinement):
            # both nets then run on the generic fp32 kernels, whose input-gradient chain reaches the encodings (_FullRender)
            if self.mlp_precision != "fp32" or self.coarse_precision not in (None, "fp32"):
                raise NotImplementedError("nerf_sos_amd.NeRFNet: gradients with respect to the rays exist in fp32 only (mlp_precision = 'fp32')")
            if viewdirs is not None:
                raise NotImplementedError("nerf_sos_amd.NeRFNet: gradients with respect to the rays need viewdirs=None (derived from rays_d, as NeRFNet.forward does)")
            params = [p_ for _, m in self._sem_nets() for _, p_ in _named_params(m.mlp)]
            outs = _FullRender.apply(self, args, kwargs, rays_o, rays_d, *params)
            return dict(zip(self._last_keys, outs))
        if not trainable:
            return self._render_rays_impl(*args, save=False, **kwargs)[0]
        other = [n for n in trainable if "semantic_linear" not in n]
        if other or generic:
            # a generic-architecture net trains through the generic backward kernels whatever subset of its parameters is trainable
            # (K7-G: every gradient is computed, autograd keeps those of the parameters that ask for one)
            # any backbone parameter trainable (e.g. configs/flower_full.txt trains everything): full backward
            if not generic and self.mlp_precision not in ("fp32", "fp16x3") and not NeRFNet._warned_full_16bit:
                # there is no 16-bit full backward (the reference has no 16-bit path at all): a trainable backbone under
                # "fp16" / "bf16" trains on the split-fp16 kernels -- 16-bit matrix pipe, fp32-grade values, the same gradient
                # tests as "fp16x3" -- instead of raising; inference and frozen-backbone steps keep the 16-bit kernels
                import warnings
                warnings.warn(f"nerf_sos_amd.NeRFNet: mlp_precision={self.mlp_precision!r} with a trainable backbone runs the "
                              "split-fp16 ('fp16x3') forward and backward kernels: there is no 16-bit full backward", stacklevel=2)
                NeRFNet._warned_full_16bit = True
            params = [p_ for _, m in self._sem_nets() for _, p_ in _named_params(m.mlp)]
            outs = _FullRender.apply(self, args, kwargs, rays_o.detach(), rays_d.detach(), *params)
            return dict(zip(self._last_keys, outs))
        params = []
        for _, m in self._sem_nets():
            by_name = dict(_named_params(m.mlp))
            params += [by_name[k] for k in _SEM_KEYS]
        outs = _FrozenBackboneRender.apply(self, args, kwargs, *params)
        return dict(zip(self._last_keys, outs))

    def _render_rays_impl(self, rays_o, rays_d, near, far, viewdirs, raw_noise_std, retraw, retpts, save=False,
                          force_generic=False, **kwargs):
        perturb = kwargs.get('perturb', self.perturb)
        n_samples = kwargs.get('N_samples', self.N_samples)
        R, dev = rays_d.shape[0], rays_d.device
        # stage-wise pin (like `cdf_in` of the importance kernel): fine-pass sample positions handed in by the caller
        # replace the importance sampler's -- the parity tests feed the REFERENCE's own z_fine to hold the fine
        # network + compositing to the 1e-4 bar separately from last-ulp index flips of the sampler
        z_fine_override = kwargs.get('z_fine_override')
        saved = {}

        def query(net, z, tag):
            if not net.fast or force_generic:     # any other architecture (or ray gradients): the generic fp32 kernels
                if save:         # (training a generic net always takes the full backward: _FullRender)
                    raw, acts = net.query_rays(rays_o, rays_d, viewdirs, z, save=True, subset=not force_generic, input_grads=force_generic)
                    saved[tag] = dict(acts=acts, raw=raw, z=z, generic=True)
                    return raw
                return net.query_rays(rays_o, rays_d, viewdirs, z)
            pp = self.pass_precision(tag)
            if not save:
                if pp != "fp32":
                    return ops.mlp_forward_rays_lp(net.packed_weights(pp), net.sem_mode, pp, rays_o, rays_d, viewdirs, z)
                return ops.mlp_forward_rays(net.packed_weights(), net.sem_mode, rays_o, rays_d, viewdirs, z)
            if save == "all":   # full backward (K7): every layer's activations (exact-fp32 or split-fp16 kernel)
                prec = self.mlp_precision if self.mlp_precision in ("fp32", "fp16x3") else "fp16x3"    # (see render_rays)
                raw, acts, masks = ops.mlp_forward_rays_save_all(net.packed_weights(prec), net.sem_mode, rays_o, rays_d, viewdirs, z, prec,
                                                                 acts16=prec == "fp16x3" and self.compact_activations)
                saved[tag] = dict(acts=acts, raw=raw, z=z, masks=masks)
                return raw
            raw, sem_in, sem_hid = ops.mlp_forward_rays_save(net.packed_weights(pp), net.sem_mode, rays_o, rays_d, viewdirs, z, pp, compact=True)
            saved[tag] = dict(sem_in=sem_in, sem_hid=sem_hid, precision=pp)
            return raw

        n_importance = kwargs.get('N_importance', self.N_importance)
        fine = self.N_importance > 0 and n_importance > 0
        pre = None
        if self.rng == "philox" and (perturb != 0. or raw_noise_std > 0.) and R > 0:
            self._rng_calls += 1
            call = self._rng_calls if self.rng_counter is None else self.rng_counter
            pre = ops.render_draws(self.rng_seed, call, R, n_samples, self.N_importance if fine else 0, dev,
                                   jitter=perturb > 0., noise=raw_noise_std > 0., importance=perturb != 0.)
        elif self.rng not in ("torch", "philox"):
            raise ValueError(f"NeRFNet.rng must be 'torch' or 'philox', got {self.rng!r}")
        t_rand = (pre[0] if pre else torch.rand((R, n_samples), device=dev)) if perturb > 0. else None   # sampler.py:61
        z_vals, unit_dirs = ops.ray_setup(rays_d, near, far, n_samples, t_rand)
        if viewdirs is None:
            viewdirs = unit_dirs
        raw = query(self.nerf, z_vals, "coarse")
        noise = (pre[1] if pre else torch.randn((R, n_samples), device=dev)) if raw_noise_std > 0. else None   # renderer.py:47
        sampled = None
        if fine and 2 <= n_samples <= 64 and R > 0 and raw.shape[-1] <= 6:
            # coarse compositing and hierarchical resampling in ONE launch (the draws keep the reference's order:
            # sigma noise of the coarse pass, then the importance u -- renderer.py:47, sampler.py:103)
            u = (pre[2] if pre else torch.rand((R, self.N_importance), device=dev)) if perturb != 0.0 else None
            ret, *sampled = ops.composite_importance(raw, z_vals, rays_d, self.N_importance, noise, raw_noise_std, self.white_bkgd, u)
        else:
            ret = ops.composite(raw, z_vals, rays_d, noise, raw_noise_std, self.white_bkgd)
        if self.use_semantics and "semantics" not in ret:
            # viewdirs=False: output_linear has no semantic channels; the reference's renderer still returns sum(w * raw[..., 4:]) = [R, 0]
            ret["semantics"] = raw.new_zeros((R, 0))
        if save:
            saved["coarse"]["weights"] = ret['weights']
            saved["coarse"]["noise"] = noise
            saved["rays_d"], saved["noise_std"] = rays_d, raw_noise_std
            saved["rays_o"], saved["viewdirs"] = rays_o, viewdirs
        if retraw:
            ret['raw'] = raw
        if retpts:
            ret['pts'] = ops.ray_points(rays_o, rays_d, z_vals)

        if fine:
            ret0 = ret
            # the sample count is the constructor's, not the per-call kwarg (sampler.py:100,103)
            N = self.N_importance
            if sampled is not None:
                z_fine, z_samples, z_std = sampled
            else:
                u = (pre[2] if pre else torch.rand((R, N), device=dev)) if perturb != 0.0 else None   # sampler.py:103,158
                z_fine, z_samples, z_std = ops.importance_sample(z_vals, ret0['weights'], N, u)
            if z_fine_override is not None:
                z_fine = z_fine_override.to(device=dev, dtype=torch.float32).reshape(R, n_samples + N).contiguous()
            raw = query(self.nerf_fine, z_fine, "fine")
            noise = (pre[3] if pre else torch.randn((R, n_samples + N), device=dev)) if raw_noise_std > 0. else None
            ret = ops.composite(raw, z_fine, rays_d, noise, raw_noise_std, self.white_bkgd)
            if self.use_semantics and "semantics" not in ret:
                ret["semantics"] = raw.new_zeros((R, 0))
            if save:
                saved["fine"]["weights"] = ret['weights']
                saved["fine"]["noise"] = noise
            if retraw:
                ret['raw'] = raw
            if retpts:
                ret['pts'] = ops.ray_points(rays_o, rays_d, z_fine)
            ret['z_std'] = z_std
            for k in ret0:
                ret[k + '0'] = ret0[k]
        return ret, saved

    def forward(self, ray_batch, bound_batch, **kwargs) -> Dict[str, torch.Tensor]:
        """models/nerf_net.py:132-195: kwargs selection by mode, flatten, per-chunk render, un-flatten.
        Unknown kwargs (``radii``) are accepted and ignored like the reference does."""
        render_kwargs = dict(self.render_kwargs_train if self.training else self.render_kwargs_test)
        render_kwargs.update(kwargs)

        rays_o, rays_d = ray_batch
        assert rays_o.shape == rays_d.shape
        if self._mlp_precision in self._GUARDED or self._coarse_precision in self._GUARDED:
            self._maybe_validate((rays_o, rays_d), bound_batch)
        # (rays that require a gradient -- pose refinement -- get one: render_rays routes them through _FullRender on the generic kernels)
        old_shape = rays_d.shape
        rays_o = rays_o.reshape(-1, rays_o.shape[-1]).float().contiguous()
        rays_d = rays_d.reshape(-1, rays_d.shape[-1]).float().contiguous()
        R = rays_d.shape[0]

        near, far = bound_batch
        near = self._bound(near, rays_d)
        far = self._bound(far, rays_d)

        zf = render_kwargs.pop('z_fine_override', None)
        if zf is not None:
            zf = zf.reshape(R, -1)
        all_ret: Dict[str, list] = {}
        for i in range(0, R, self.chunk):
            e = min(i + self.chunk, R)
            if zf is not None:
                render_kwargs['z_fine_override'] = zf[i:e]
            ret = self.render_rays(rays_o[i:e], rays_d[i:e], near[i:e], far[i:e], viewdirs=None, **render_kwargs)
            for k, v in ret.items():
                all_ret.setdefault(k, []).append(v)
        out = {k: (v[0] if len(v) == 1 else torch.cat(v, 0)) for k, v in all_ret.items()}
        return {k: v.reshape(list(old_shape[:-1]) + list(v.shape[1:])) for k, v in out.items()}

    _BOUNDS: "OrderedDict[tuple, torch.Tensor]" = OrderedDict()

    @staticmethod
    def _bound(b, rays_d) -> torch.Tensor:
        """Scalar near / far -> one value per ray (models/nerf_net.py:168-173).  Constant fills are cached per
        (device, ray count, value): the kernels only read them, and a fill launch per bound per call was a visible
        share of the host-side floor at 16-bit rates.  The cache is LRU (one entry leaves at a time, never a clear),
        and it is bypassed during stream capture: a HIP graph must own every tensor whose address it bakes in
        (GraphedRender passes its own bound tensors; any other capture gets a fill node in its private pool)."""
        if isinstance(b, (int, float)):
            if rays_d.is_cuda and torch.cuda.is_current_stream_capturing():
                return torch.full((rays_d.shape[0],), float(b), device=rays_d.device, dtype=torch.float32)
            key = (rays_d.device, rays_d.shape[0], float(b))
            t = NeRFNet._BOUNDS.get(key)
            if t is None:
                while len(NeRFNet._BOUNDS) >= 64:
                    NeRFNet._BOUNDS.popitem(last=False)
                t = torch.full((rays_d.shape[0],), float(b), device=rays_d.device, dtype=torch.float32)
                NeRFNet._BOUNDS[key] = t
            else:
                NeRFNet._BOUNDS.move_to_end(key)
            return t
        return b.to(device=rays_d.device, dtype=torch.float32).reshape(-1).contiguous()


def export_density(model: NeRFNet, extents=(2.0, 2.0, 2.0), voxel_size: float = 2. / 256., device=None, slab: int = 64) -> torch.Tensor:
    """sigma [W/v, H/v, D/v] of the fine net on the reference's export grid (engines/eval.py:285-300: linspace grid x 14, ZERO view
    directions, `model.nerf_fine(pts, viewdirs=0)`, sigma = max(raw[..., -1], 0)) -- the one caller of the point-query entry.
    Returned on the device (the reference's .cpu().numpy() and its mrc / ply writers are the harness's business).  The grid is
    queried `slab` x-planes at a time: 256^3 points x (3 + 3 + C) floats at once is 0.7-0.8 GB for nothing."""
    model.eval()
    dev = torch.device(device) if device is not None else next(model.parameters()).device
    h, w, d = extents
    with torch.no_grad():
        xs = torch.linspace(-w / 2, w / 2, int(w / voxel_size), device=dev)
        ys = torch.linspace(-h / 2, h / 2, int(h / voxel_size), device=dev)
        zs = torch.linspace(-d / 2, d / 2, int(d / voxel_size), device=dev)
        out = torch.empty((xs.numel(), ys.numel(), zs.numel()), device=dev, dtype=torch.float32)
        for i in range(0, xs.numel(), slab):
            pts = torch.stack(torch.meshgrid(xs[i:i + slab], ys, zs, indexing="ij"), dim=-1).float() * 14
            raw = model.nerf_fine(pts, viewdirs=torch.zeros_like(pts))
            out[i:i + slab] = raw[..., -1].clamp_min(0)
    return out
