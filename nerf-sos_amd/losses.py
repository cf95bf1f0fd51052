"""Correlation losses on the rendered patches -- host-side mirror of the reference's loss modules
(utils/image.py:263-487) over `nsos_app_correlation_loss` / `nsos_geo_correlation_loss`.

Same constructor (`args` namespace with `rand_neg`, `self_corr_w`, `use_sim_matrix`, `app_corr_params` /
`geo_corr_params`, `patch_stride`), same `forward` signatures, same random draws in the same order from torch's
global generator on the inputs' device (`rand` for coords1 then coords2; `randperm` for `super_perm` / `rand_neg`).
Only `orig_code` carries gradient, as in the reference (the feature / depth side is under `no_grad` there).
No CPU path: inputs must be GPU tensors and the HIP library must be present.
"""
from __future__ import annotations

from typing import Optional, Sequence

import torch
import torch.nn as nn

from . import _lib
from .ops import _dev, _p, _stream


def _params_of(values: Sequence, defaults):
    vals = [d if v is None else float(v) for v, d in zip(values, defaults)]
    return tuple(vals)


def exchange_floats(batch: int, n_points: int) -> int:
    """fp32 elements one row-partitioned loss evaluation contributes to the step's `sums` reduction."""
    return int(_lib.lib().nsos_corr_exchange_floats(int(batch), int(n_points)))


def _run_phases(run, ws: torch.Tensor, batch: int, n_points: int, group) -> None:
    """A row-partitioned geometric evaluation on its own: phase 3 (everything in one call) without a process group; with one,
    phases 0..2 and the two sum-all-reduces of its workspace slots in between.  Every rank issues both reductions (a rank that
    owns no patch contributes zeros): the sequence of collectives does not depend on the data."""
    import ctypes as C_
    import torch.distributed as dist
    from .sharding import collective, multi_process
    if not multi_process(group):
        run(3)
        return
    so, go, gn = C_.c_int64(), C_.c_int64(), C_.c_int64()
    _lib.check(_lib.lib().nsos_corr_workspace_slots(batch, n_points, C_.byref(so), C_.byref(go), C_.byref(gn)), "nsos_corr_workspace_slots")
    means = ws[so.value // 8: so.value // 8 + 4]
    sums = ws.view(torch.float32)[go.value // 4: go.value // 4 + gn.value]
    run(0)
    collective("loss_means_all_reduce", lambda async_op: dist.all_reduce(means, group=group, async_op=async_op), group)
    run(1)
    collective("loss_sums_all_reduce", lambda async_op: dist.all_reduce(sums, group=group, async_op=async_op), group)
    run(2)


def _is_channel_last_view(t: torch.Tensor) -> bool:
    """[B,C,H,W] that is a permuted view of a dense [B,H,W,C] tensor (what `ret['semantics'].permute(0,3,1,2)` is)."""
    return t.dim() == 4 and t.shape[1] > 1 and not t.is_contiguous() and t.permute(0, 2, 3, 1).is_contiguous()


class _PairFn(torch.autograd.Function):
    """loss = f(code0, code1); the forward launches already produce both gradients, backward only scales them."""

    @staticmethod
    def forward(ctx, code0, code1, launch):
        loss, g0, g1 = launch(code0, code1, code0.requires_grad or code1.requires_grad)
        ctx.has_grad = g0 is not None
        if ctx.has_grad:
            ctx.save_for_backward(g0, g1)
        return loss

    @staticmethod
    def backward(ctx, g):
        if not ctx.has_grad:
            return None, None, None
        g0, g1 = ctx.saved_tensors
        return g0 * g, g1 * g, None


class _CorrFn(torch.autograd.Function):
    """loss = f(code); the forward launch already produces d loss / d code, backward only scales it."""

    @staticmethod
    def forward(ctx, code, launch):
        loss, grad = launch(code, code.requires_grad)
        ctx.save_for_backward(grad if grad is not None else torch.empty(0, device=code.device))
        ctx.has_grad = grad is not None
        return loss

    @staticmethod
    def backward(ctx, g):
        (grad,) = ctx.saved_tensors
        return (grad * g if ctx.has_grad else None), None


class CorrelationLoss(nn.Module):
    """utils/image.py:263-370.  forward(orig_feats [B,Cf,Hf,Wf], orig_code [B,C,P,P], sim_matrix [B,B] or None)."""

    _DEFAULTS = (0.18, 0.67, 0.46, 0.63)   # self_shift, self_weight, neg_shift, neg_weight (:271-274)

    def __init__(self, args=None):
        super().__init__()
        self.zero_clamp = True
        self.stabalize = False
        self.pointwise = True
        self.feature_samples = 11
        self.rand_neg = bool(getattr(args, "rand_neg", False))
        self.self_corr_w = getattr(args, "self_corr_w", 1)
        self.use_sim_matrix = getattr(args, "use_sim_matrix", True)
        # None = torch's global generator on the inputs' device, the reference's behaviour.  Sharded training sets a
        # per-step generator seeded identically on every rank (sharding.loss_generator): all ranks must draw the same
        # coordinates / permutations for the summed gradients to equal the single-process gradient.
        self.generator: Optional[torch.Generator] = None
        raw = getattr(args, self._PARAM_ATTR, None) if args is not None else None
        self.self_shift, self.self_weight, self.neg_shift, self.neg_weight = _params_of(raw or [None] * 4, self._DEFAULTS)

    _PARAM_ATTR = "app_corr_params"

    def super_perm(self, size: int, device: torch.device):
        perm = torch.randperm(size, device=device, dtype=torch.long, generator=self.generator)   # :306-309
        perm[perm == torch.arange(size, device=device)] += 1
        return perm % size

    def _neg_index(self, sim_matrix: Optional[torch.Tensor], B: int, device) -> torch.Tensor:
        if sim_matrix is None:
            neg = self.super_perm(B, device)                                   # :351-352
        else:
            assert len(sim_matrix.shape) == 2
            neg = torch.min(sim_matrix, dim=0)[1]                              # :354
        if self.rand_neg:
            neg = torch.randperm(sim_matrix.shape[0], device=device, dtype=torch.long, generator=self.generator)   # :357
        return neg.to(device=device, dtype=torch.int64).contiguous()

    def draw_coords(self, n_calls: int, B: int, device) -> torch.Tensor:
        """[n_calls, 2, B, S, S, 2]: the sample coordinates of `n_calls` evaluations in ONE launch of the module's generator (a
        training step scores two maps: four torch.rand launches otherwise); pass [i] as `coords` to value_and_grad / rows_phased."""
        S = self.feature_samples
        return torch.rand([n_calls, 2, B, S, S, 2], device=device, generator=self.generator)

    def queue_coords(self, coords) -> None:
        """Sample coordinates for the NEXT len(coords) evaluations (each item [2, B, S, S, 2] from draw_coords), consumed in order by
        forward / value_and_grad calls that are not handed `coords` themselves."""
        self._queued = list(coords)

    def _launcher(self, orig_feats: torch.Tensor, code_shape, sim_matrix: Optional[torch.Tensor], weight: float = 1.0,
                  neg: Optional[torch.Tensor] = None, coords: Optional[torch.Tensor] = None, loss_out: Optional[torch.Tensor] = None):
        if coords is None and getattr(self, "_queued", None):
            coords = self._queued.pop(0)
        """Draws the sample coordinates and the negatives (the reference's order: rand1, rand2, negatives) and returns
        launch(code, want_grad) -> (weight * loss, weight * d loss / d code or None).  `weight` rides on the kernel's own
        self / negative weights (no extra launch); `neg`: negatives computed by the caller (one argmin per step, not per call)."""
        feats = _dev(orig_feats.detach(), "orig_feats")
        B, Cf, Hf, Wf = feats.shape
        Bc, C, Hc, Wc = code_shape
        if Bc != B:
            raise ValueError(f"orig_feats has {B} patches, orig_code {Bc}")
        S = self.feature_samples
        dev = feats.device
        if coords is not None:                                                                           # drawn by the caller (draw_coords)
            rand1, rand2 = _dev(coords[0], "coords"), _dev(coords[1], "coords")
        else:
            rand1 = torch.rand([B, S, S, 2], device=dev, generator=self.generator)                       # :343 (the kernel applies *2-1)
            rand2 = torch.rand([B, S, S, 2], device=dev, generator=self.generator)                       # :344
        if neg is None:
            neg = self._neg_index(sim_matrix, B, dev)
        lib = _lib.lib()
        w = float(weight)
        prm = (self.self_shift, self.self_weight * w, self.neg_shift, self.neg_weight * w)

        def launch(code, want_grad):
            code = code.detach()
            nbytes = lib.nsos_corr_workspace_bytes(0, B, S * S, Cf)
            ws = torch.empty((nbytes + 15) // 16 * 2, device=dev, dtype=torch.float64)
            loss = loss_out if loss_out is not None else torch.empty((), device=dev, dtype=torch.float32)   # (loss_out: a slot of the step's loss buffer)
            if _is_channel_last_view(code):
                # the renderer's `semantics` [B,P,P,C] seen through .permute(0,3,1,2): read (and differentiated) in place
                nhwc = _dev(code.permute(0, 2, 3, 1), "orig_code")
                g = torch.empty_like(nhwc) if want_grad else None
                _lib.check(lib.nsos_app_correlation_loss_nhwc(_p(feats), _p(nhwc), neg.data_ptr(), _p(rand1), _p(rand2), B, Cf, Hf, Wf,
                                                              C, Hc, Wc, S, *prm, _p(loss), _p(g), ws.data_ptr(), ws.numel() * 8,
                                                              _stream()), "nsos_app_correlation_loss_nhwc")
                return loss, (g.permute(0, 3, 1, 2) if want_grad else None)
            code = _dev(code, "orig_code")
            grad = torch.empty_like(code) if want_grad else None
            _lib.check(lib.nsos_app_correlation_loss(_p(feats), _p(code), neg.data_ptr(), _p(rand1), _p(rand2), B, Cf, Hf, Wf, C,
                                                     Hc, Wc, S, *prm, _p(loss), _p(grad), ws.data_ptr(), ws.numel() * 8,
                                                     _stream()), "nsos_app_correlation_loss")
            return loss, grad

        return launch

    def rows_phased(self, orig_feats: torch.Tensor, orig_code: torch.Tensor, sim_matrix: Optional[torch.Tensor], rows: Sequence[int],
                    exchange, weight: float = 1.0, neg: Optional[torch.Tensor] = None, coords: Optional[torch.Tensor] = None,
                    loss_out: Optional[torch.Tensor] = None):
        """The row-partitioned evaluation for the patch-sharded step (`nsos_app_correlation_loss_rows`): every rank passes the
        whole batch and ITS patches `rows`, draws the same coordinates (a generator seeded alike on every rank), and runs
        phase 0 -> [sum `means` over the ranks] -> phase 1 -> [sum `sums`] -> phase 2.  Returns (run(phase), (loss, grad)):
        after phase 2 `loss` is the batch-wide value on every rank and `grad` [B,C,P,P] holds weight * d loss / d code for the
        patches in `rows` (zeros elsewhere).  exchange = (means [8] fp64, sums [exchange_floats(B, S S)] fp32): slices of the two
        buffers the step reduces once per phase for all of its evaluations."""
        feats = _dev(orig_feats.detach(), "orig_feats")
        B, Cf, Hf, Wf = feats.shape
        Bc, C, Hc, Wc = orig_code.shape
        if Bc != B:
            raise ValueError(f"orig_feats has {B} patches, orig_code {Bc}")
        S = self.feature_samples
        dev = feats.device
        if coords is not None:
            rand1, rand2 = _dev(coords[0], "coords"), _dev(coords[1], "coords")
        else:
            rand1 = torch.rand([B, S, S, 2], device=dev, generator=self.generator)                       # :343
            rand2 = torch.rand([B, S, S, 2], device=dev, generator=self.generator)                       # :344
        if neg is None:
            neg = self._neg_index(sim_matrix, B, dev)
        lib = _lib.lib()
        w = float(weight)
        prm = (self.self_shift, self.self_weight * w, self.neg_shift, self.neg_weight * w)
        code = orig_code.detach()
        nhwc = _is_channel_last_view(code)
        code = _dev(code.permute(0, 2, 3, 1), "orig_code") if nhwc else _dev(code, "orig_code")
        grad = torch.empty_like(code)
        loss = loss_out if loss_out is not None else torch.empty((), device=dev, dtype=torch.float32)
        nbytes = lib.nsos_corr_workspace_bytes(0, B, S * S, Cf)
        ws = torch.empty((nbytes + 15) // 16 * 2, device=dev, dtype=torch.float64)
        from .sharding import device_index
        rows_t = device_index(rows, torch.int32, dev)
        xm, xs = exchange

        def run(phase):
            _lib.check(lib.nsos_app_correlation_loss_rows(phase, _p(feats), _p(code), neg.data_ptr(), _p(rand1), _p(rand2),
                                                          rows_t.data_ptr() if len(rows) else None, len(rows), B, 1 if nhwc else 0, Cf, Hf, Wf,
                                                          C, Hc, Wc, S, *prm, _p(loss), _p(grad), ws.data_ptr(), ws.numel() * 8,
                                                          _p(xm), _p(xs), _stream()), "nsos_app_correlation_loss_rows")

        return run, (loss, grad.permute(0, 3, 1, 2) if nhwc else grad)

    def forward(self, orig_feats: torch.Tensor, orig_code: torch.Tensor, sim_matrix: Optional[torch.Tensor]):
        return _CorrFn.apply(orig_code, self._launcher(orig_feats, orig_code.shape, sim_matrix))

    def value_and_grad(self, orig_feats: torch.Tensor, orig_code: torch.Tensor, sim_matrix: Optional[torch.Tensor],
                       weight: float = 1.0, neg: Optional[torch.Tensor] = None, want_grad: bool = True,
                       coords: Optional[torch.Tensor] = None, loss_out: Optional[torch.Tensor] = None):
        """(weight * loss, weight * d loss / d orig_code) straight from the launch, outside autograd -- for a training step that
        sums the gradients of several losses itself and enters autograd once (sharding._losses_and_backward)."""
        return self._launcher(orig_feats, orig_code.shape, sim_matrix, weight, neg, coords, loss_out)(orig_code, want_grad)


class GeoCorrelationLoss(CorrelationLoss):
    """utils/image.py:373-487.  forward(orig_feats = depth [B,1,P,P], orig_code [B,C,P,P],
    batch_rays = [ray_o, ray_d, rgbs] each [B,3,P,P], sim_matrix).  As in the reference, depth values above
    `max_depth` are replaced IN PLACE (the caller's tensor changes) before back-projection."""

    _DEFAULTS = (3.0, 0.67, 10.0, 0.63)    # :379-382
    _PARAM_ATTR = "geo_corr_params"

    def __init__(self, args=None):
        super().__init__(args)
        self.max_depth = 15
        self.ps = getattr(args, "patch_stride", 8)

    def forward(self, orig_feats: torch.Tensor, orig_code: torch.Tensor, batch_rays, sim_matrix: Optional[torch.Tensor],
                rows: Optional[Sequence[int]] = None, group=None):
        """`rows` (not in the reference): the row patches THIS rank evaluates, for the patch-sharded multi-GPU step.  Every
        rank passes the whole batch (depth, code and rays of all B patches, as gathered by sharding.all_gather_patches) and its
        own patch ids; the O(P^4) pair sets are then evaluated once across the ranks instead of once per rank, and four tiny
        sum-all-reduces over `group` (the global means of fd / fd1, the loss sums, and the [B,P*P,4] role sums of the gradient)
        make the returned loss and d loss / d code of EVERY patch the batch-wide, single-process values."""
        depth = orig_feats
        B, one, H, W = depth.shape
        if one != 1:
            raise ValueError("depth must be [B,1,P,P]")
        ray_o, ray_d = batch_rays[0], batch_rays[1]
        dev = depth.device
        # The kernel filters a PRIVATE copy of the depth (values above max_depth -> the largest value below it); the
        # caller's tensor is then updated the way the reference does it, with an in-place masked assignment under
        # autograd (utils/image.py:454): the version counter moves, and only the filtered elements lose their gradient.
        dbuf = depth.detach().float().contiguous().clone()
        _dev(dbuf, "depth")
        ro = _dev(ray_o.detach().expand(B, 3, H, W), "ray_o")
        rd = _dev(ray_d.detach().expand(B, 3, H, W), "ray_d")
        neg = self._neg_index(sim_matrix, B, dev)
        C = orig_code.shape[1]
        lib = _lib.lib()
        prm = (self.self_shift, self.self_weight, self.neg_shift, self.neg_weight)

        grad_mode = torch.is_grad_enabled()       # (inside the autograd Function's forward it reads False)

        def launch(code, want_grad):
            code = _dev(code.detach(), "orig_code")
            nbytes = lib.nsos_corr_workspace_bytes(1, B, H * W, 0)
            ws = torch.empty((nbytes + 15) // 16 * 2, device=dev, dtype=torch.float64)
            loss = torch.empty((), device=dev, dtype=torch.float32)
            grad = torch.empty_like(code) if want_grad else None
            if rows is None:
                _lib.check(lib.nsos_geo_correlation_loss(dbuf.data_ptr(), _p(code), _p(ro), _p(rd), neg.data_ptr(), B, C, H, W, *prm,
                                                         float(self.max_depth), 1, _p(loss), _p(grad), ws.data_ptr(),
                                                         ws.numel() * 8, _stream()), "nsos_geo_correlation_loss")
                return loss, grad
            from .sharding import device_index
            rows_t = device_index(rows, torch.int32, dev)     # uploaded once (a fresh torch.tensor(..., device=) synchronises)

            def run(phase):
                _lib.check(lib.nsos_geo_correlation_loss_rows(phase, dbuf.data_ptr(), _p(code), _p(ro), _p(rd), neg.data_ptr(),
                                                              rows_t.data_ptr() if len(rows) else None, len(rows), B, C, H, W, *prm,
                                                              float(self.max_depth), 1, _p(loss), _p(grad), ws.data_ptr(),
                                                              ws.numel() * 8, None, None, _stream()), "nsos_geo_correlation_loss_rows")

            _run_phases(run, ws, B, H * W, group)
            return loss, grad

        out = _CorrFn.apply(orig_code, launch)
        with torch.no_grad():
            changed = dbuf != depth
        depth.copy_(torch.where(changed, dbuf.to(depth.dtype), depth))   # masked assign without the host sync of a bool index
        return out


    def _pair_launcher(self, depth, code0, code1, ray_o, ray_d, sim_matrix, rows, group, weight: float = 1.0,
                       neg: Optional[torch.Tensor] = None, grad_mode: Optional[bool] = None, loss_out: Optional[torch.Tensor] = None):
        """launch(code0, code1, want_grad) -> (weight * stacked-mean loss, its gradients) of forward_pair's evaluation.
        `grad_mode`: the group-wide switch for the gradient's role-sum all-reduce (default: the caller's autograd mode)."""
        if self.rand_neg:
            raise NotImplementedError("forward_pair: with rand_neg every call draws its own negatives -- call forward twice")
        B, H, W = depth.shape[0], depth.shape[1], depth.shape[2]
        C = code0.shape[-1]
        if tuple(code0.shape) != (B, H, W, C) or tuple(code1.shape) != (B, H, W, C) or tuple(ray_o.shape) != (B, H, W, 3):
            raise ValueError("forward_pair takes channel-last tensors: depth [B,P,P,1], codes [B,P,P,C], rays [B,P,P,3]")
        dev = depth.device
        d = _dev(depth.detach().reshape(B, H * W), "depth")
        ro, rd = _dev(ray_o.detach(), "ray_o"), _dev(ray_d.detach(), "ray_d")
        if neg is None:
            neg = self._neg_index(sim_matrix, B, dev)
        neg2 = neg if neg.numel() == 2 * B else torch.cat([neg, neg + B])      # (similarity_negatives(copies=2) is already stacked)
        own = list(range(B)) if rows is None else [int(r) for r in rows]
        rows2 = own + [B + r for r in own]
        lib = _lib.lib()
        w = float(weight)
        prm = (self.self_shift, self.self_weight * w, self.neg_shift, self.neg_weight * w)
        if grad_mode is None:
            grad_mode = torch.is_grad_enabled()

        def launch(c0, c1, want_grad, exchange=None):
            """exchange = (means, sums): slices of the step's two reduction buffers -- returns the phase runner instead of running
            (the caller interleaves the phases of several evaluations with ONE all-reduce per phase: sharding._losses_direct)."""
            from .sharding import device_index
            c0, c1 = _dev(c0.detach(), "code0"), _dev(c1.detach(), "code1")
            nbytes = lib.nsos_corr_workspace_bytes(1, 2 * B, H * W, 0)
            ws = torch.empty((nbytes + 15) // 16 * 2, device=dev, dtype=torch.float64)
            loss = loss_out if loss_out is not None else torch.empty((), device=dev, dtype=torch.float32)
            g0 = torch.empty_like(c0) if want_grad else None
            g1 = torch.empty_like(c1) if want_grad else None
            rows_t = device_index(rows2, torch.int32, dev)
            xm, xs = exchange if exchange is not None else (None, None)

            def run(phase):
                _lib.check(lib.nsos_geo_correlation_loss_pair(phase, _p(d), _p(c0), _p(c1), _p(ro), _p(rd), neg2.data_ptr(),
                                                              rows_t.data_ptr() if len(rows2) else None, len(rows2), B, 1, C, H, W, *prm,
                                                              float(self.max_depth), _p(loss), _p(g0), _p(g1), ws.data_ptr(),
                                                              ws.numel() * 8, _p(xm), _p(xs), _stream()), "nsos_geo_correlation_loss_pair")

            if exchange is not None:
                return run, (loss, g0, g1)
            _run_phases(run, ws, 2 * B, H * W, group)
            return loss, g0, g1

        return launch


    def forward_pair(self, depth: torch.Tensor, code0: torch.Tensor, code1: torch.Tensor, ray_o: torch.Tensor, ray_d: torch.Tensor,
                     sim_matrix: Optional[torch.Tensor], rows: Optional[Sequence[int]] = None, group=None) -> torch.Tensor:
        """forward(depth, code0, ...) + forward(depth, code1, ...) -- the coarse and the fine semantic map against the same
        geometry, as `train_one_step` scores them (engines/trainer.py:147-166) -- in ONE evaluation over the stacked batch
        [code0; code1] that is never materialised (`nsos_geo_correlation_loss_pair`), on the renderer's own channel-last
        tensors: depth [B,P,P,1], code0 / code1 [B,P,P,C], ray_o / ray_d [B,P,P,3].  The stacked mean is exactly
        (L0 + L1) / 2 (every batch-wide quantity of the loss depends on the geometry only), so 2 x it is returned; with
        `sim_matrix` the negatives stay inside their half.  `rows` / `group` as in forward (row-partitioned over the ranks; rows
        index the B geometry patches).  Not for rand_neg (each call of the reference draws its own permutation there).
        The depth is only read: the reference's in-place filter of the caller's tensor (utils/image.py:454) is left to forward."""
        return 2.0 * _PairFn.apply(code0, code1, self._pair_launcher(depth, code0, code1, ray_o, ray_d, sim_matrix, rows, group))

    def pair_value_and_grads(self, depth, code0, code1, ray_o, ray_d, sim_matrix, rows=None, group=None, weight: float = 1.0,
                             neg: Optional[torch.Tensor] = None, grad_mode: Optional[bool] = None, loss_out: Optional[torch.Tensor] = None):
        """(weight * forward_pair(...), weight * d / d code0, weight * d / d code1) straight from the launches, outside autograd
        (see CorrelationLoss.value_and_grad); the factor 2 of the stacked mean and `weight` ride on the kernel's weights."""
        launch = self._pair_launcher(depth, code0, code1, ray_o, ray_d, sim_matrix, rows, group, 2.0 * float(weight), neg, grad_mode, loss_out)
        want = torch.is_grad_enabled() if grad_mode is None else bool(grad_mode)
        return launch(code0, code1, want)

    def pair_phased(self, depth, code0, code1, ray_o, ray_d, sim_matrix, rows, exchange, weight: float = 1.0,
                    neg: Optional[torch.Tensor] = None, loss_out: Optional[torch.Tensor] = None):
        """pair_value_and_grads split at its two reductions: (run(phase), (loss, grad0, grad1)) with the reduced slots in
        `exchange` = (means [8] fp64, sums [exchange_floats(2 B, P P)] fp32) -- see rows_phased."""
        launch = self._pair_launcher(depth, code0, code1, ray_o, ray_d, sim_matrix, rows, None, 2.0 * float(weight), neg, True, loss_out)
        return launch(code0, code1, True, exchange)


def similarity_negatives(cls_tokens: torch.Tensor, copies: int = 1, want_similarity: bool = False):
    """negatives [copies * B] int64 (copy c = negatives + c * B) and, if asked for, the [B,B] cosine-similarity matrix of the
    class tokens -- get_similarity_matrix (utils/image.py:186-189) followed by torch.min(sim, dim=0)[1] (:354) as one launch
    (`nsos_similarity_negatives`)."""
    x = _dev(cls_tokens.detach().reshape(cls_tokens.shape[0], -1), "cls_tokens")
    B, D = x.shape
    neg = torch.empty(copies * B, device=x.device, dtype=torch.int64)
    sim = torch.empty((B, B), device=x.device, dtype=torch.float32) if want_similarity else None
    _lib.check(_lib.lib().nsos_similarity_negatives(_p(x), B, D, _p(sim), neg.data_ptr(), copies, _stream()), "nsos_similarity_negatives")
    return (neg, sim) if want_similarity else neg


class NeRFContrastive(nn.Module):
    """utils/image.py:192-218 -- the contrastive loss on the batch's DINO class tokens (`contrast_loss(cls_)`,
    engines/trainer.py:168-170; `--use_contrast`).  Same constructor; `temperature` is registered and unused, and
    `min_max_contrast=False` raises, both as in the reference.  forward(embeddings [B,D]) -> 0-dim loss
    = -log(max / (max + min)) over the off-diagonal cosine similarities; gradient to the embeddings (they come out of the
    feature extractor that consumed the rendered rgb) from the same single launch (`nsos_contrastive_loss`)."""

    def __init__(self, temperature=1, device=None, verbose=False, min_max_contrast=True):
        super().__init__()
        self.device = device
        self.verbose = verbose
        self.min_max_contrast = min_max_contrast
        self.register_buffer("temperature", torch.tensor(temperature).to(device))

    def forward(self, embeddings: torch.Tensor) -> torch.Tensor:
        self.batch_size = embeddings.shape[0]
        if not self.min_max_contrast:
            raise NotImplementedError                                   # utils/image.py:215-216
        if embeddings.dim() != 2:
            raise ValueError(f"NeRFContrastive: embeddings must be [B,D], got {tuple(embeddings.shape)}")

        def launch(emb, want_grad):
            e = _dev(emb.detach(), "embeddings")
            loss = torch.empty((1,), device=e.device, dtype=torch.float32)
            grad = torch.empty_like(e) if want_grad else None
            _lib.check(_lib.lib().nsos_contrastive_loss(_p(e), e.shape[0], e.shape[1], _p(loss), _p(grad), _stream()),
                       "nsos_contrastive_loss")
            return loss.reshape(()), grad

        return _CorrFn.apply(embeddings, launch)
