"""Ray-sharded multi-GPU rendering: one process per GPU, weights replicated, rays partitioned.

The reference is single-process; rays are independent through the whole render path
(models/nerf_net.py:177-187 already chunks them), so the path shards with NO data-path collective.
The only exchange the training recipe needs is an all-gather of the *rendered patch tensors* that
the correlation losses compare across the batch (negative patch = argmin of the DINO similarity,
utils/image.py:354,359-360,468,473-474): ~0.5 MiB per 64x64 patch, latency-bound over xGMI, so it is
one flat all-gather (RCCL picks the direct algorithm at this size), never a hand-rolled ring.

``torch.distributed`` backend "nccl" is RCCL on ROCm; the same code runs on "gloo" for CPU tests.
"""
from __future__ import annotations

import os
from typing import Callable, Dict, Iterable, List, Optional, Sequence, Tuple

import torch
import torch.distributed as dist

# What the batch-wide losses read from every patch of the global batch (engines/trainer.py:101-109,125-166):
# rendered maps (`semantics`, `semantics0`, fine `depth`), the DINO tensors of the patch's ground-truth crop
# (`feat` [384,14,14] -- the largest item, 294 KiB -- and the class token `cls_` [384] the similarity matrix is built
# from) and the patch's rays (`ray_o`, `ray_d`: the geometric loss back-projects depth along them, utils/image.py:404-438).
PATCH_KEYS = ("semantics", "semantics0", "depth", "feat", "cls_", "ray_o", "ray_d")


def shard_bounds(n: int, rank: int, world: int) -> Tuple[int, int]:
    """Contiguous balanced split of n items: the first n % world ranks get one extra item."""
    if not (0 <= rank < world):
        raise ValueError(f"rank {rank} outside world of {world}")
    q, r = divmod(n, world)
    start = rank * q + min(rank, r)
    return start, start + q + (1 if rank < r else 0)


def patch_owner(b: int, world: int) -> int:
    """Training: patch b of the global batch is rendered by GPU b mod world."""
    return b % world


def local_patches(n_patches: int, rank: int, world: int) -> List[int]:
    return [b for b in range(n_patches) if patch_owner(b, world) == rank]


# ---- collectives: counted, and never allowed to hang silently ---------------------------------------------------------
# Every collective of the sharded paths goes through `collective()`: it counts the calls by kind (bench.py reports them
# per step, the geometric loss's phase reductions included) and puts a watchdog on each one.  A rank that skips a
# collective its peers issue (a mismatch of the call sequences -- e.g. a rank that owns no patch taking a short cut) is a
# hang, not an error, in torch.distributed: with COLLECTIVE_TIMEOUT_S set, the host waits on the work handle for at most that
# long and then raises with the name of the collective and the rank (gloo only: with RCCL a timed wait would block the host on
# every collective, so there the wait only orders streams and the process group's own timeout -- set it at init_process_group --
# aborts the job through its watchdog).
COLLECTIVE_TIMEOUT_S: Optional[float] = float(os.environ.get("NSOS_COLLECTIVE_TIMEOUT_S", "0")) or None
COLLECTIVE_COUNTS: Dict[str, int] = {}
# diagnostics (bench.py): a dict here makes every collective record (HIP event before, HIP event after, host seconds) under its
# kind -- the events on the CURRENT stream (the RCCL work is joined to it by wait()), so that one bench line of the first real
# multi-GPU run says where a step's time went
COLLECTIVE_EVENTS: Optional[Dict[str, list]] = None


def reset_collective_counts() -> Dict[str, int]:
    """Returns the counts so far (kind -> calls on this rank) and clears them."""
    out = dict(COLLECTIVE_COUNTS)
    COLLECTIVE_COUNTS.clear()
    return out


def collective(kind: str, launch: Callable[..., "dist.Work"], group=None):
    """Run `launch(async_op=True)` (a torch.distributed collective) under the watchdog and count it under `kind`."""
    COLLECTIVE_COUNTS[kind] = COLLECTIVE_COUNTS.get(kind, 0) + 1
    if COLLECTIVE_EVENTS is not None and torch.cuda.is_available() and not torch.cuda.is_current_stream_capturing():
        import time
        ev = (torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True))
        ev[0].record()
        t0 = time.perf_counter()
        try:
            _collective(kind, launch, group)
        finally:
            ev[1].record()
            COLLECTIVE_EVENTS.setdefault(kind, []).append((ev[0], ev[1], time.perf_counter() - t0))
        return
    _collective(kind, launch, group)


def _collective(kind: str, launch: Callable[..., "dist.Work"], group=None):
    work = launch(async_op=True)
    if work is None:
        return
    # RCCL: wait() without a timeout only makes the current stream wait for the collective's stream -- the host runs on.  A wait
    # WITH a timeout blocks the host thread until the collective has completed (ProcessGroupNCCL::WorkNCCL::wait): four host
    # synchronisations per training step.  There the process group's own timeout (init_process_group(timeout=...)) and its
    # watchdog abort a diverged job; the per-collective host wait is for gloo (CPU tests, ranks sharing one GPU).
    if COLLECTIVE_TIMEOUT_S is None or dist.get_backend(group) == "nccl":
        work.wait()
        return
    import datetime
    try:
        done = work.wait(datetime.timedelta(seconds=COLLECTIVE_TIMEOUT_S))
    except RuntimeError as e:   # gloo raises on timeout
        raise RuntimeError(f"nerf_sos_amd.sharding: collective `{kind}` did not complete within {COLLECTIVE_TIMEOUT_S:.0f} s on rank "
                           f"{dist.get_rank(group)} of {dist.get_world_size(group)} -- the ranks' collective sequences have diverged "
                           f"(a rank skipped or added a call)") from e
    if done is False:
        raise RuntimeError(f"nerf_sos_amd.sharding: collective `{kind}` timed out after {COLLECTIVE_TIMEOUT_S:.0f} s on rank {dist.get_rank(group)}")


def _world(group) -> Tuple[int, int]:
    if dist.is_available() and dist.is_initialized():
        return dist.get_rank(group), dist.get_world_size(group)
    return 0, 1


# The N > 1 code path on ONE process (round 6): with FORCE_COLLECTIVES (env NSOS_FORCE_COLLECTIVES=1, or
# `sharding.FORCE_COLLECTIVES = True`) and an initialised process group of world size 1, the sharded step takes every branch an
# N-rank job takes -- the flat all-gather with its re-ordering, the row-partitioned loss phases with their two reductions, the flat
# gradient all-reduce -- and issues its four collectives on the group.  A `backend="nccl"`, `world_size=1` group makes them real RCCL
# launches on a single GPU: the only way to execute RCCL work inside a HIP-graph capture (graphs.GraphedPatchStep) on a one-GPU box.
FORCE_COLLECTIVES: bool = os.environ.get("NSOS_FORCE_COLLECTIVES", "") not in ("", "0")


def multi_process(group=None) -> bool:
    """Whether the sharded paths issue their collectives: a process group of more than one rank -- or any initialised group under
    FORCE_COLLECTIVES."""
    if not (dist.is_available() and dist.is_initialized()):
        return False
    return dist.get_world_size(group) > 1 or FORCE_COLLECTIVES


def render_image_sharded(render: Callable[..., Dict[str, torch.Tensor]], rays_o: torch.Tensor,
                         rays_d: torch.Tensor, bounds, group=None, gather: bool = False,
                         keys: Optional[Iterable[str]] = None, **kwargs) -> Dict[str, torch.Tensor]:
    """Eval: rank r renders the contiguous ray block shard_bounds(R, r, world) of a flattened image
    with ``render((o, d), bounds, **kwargs)``.  With gather=False each rank keeps (and would write) its
    own rows; with gather=True the selected keys are all-gathered so every rank holds the full image."""
    rank, world = _world(group)
    flat_o, flat_d = rays_o.reshape(-1, 3), rays_d.reshape(-1, 3)
    R = flat_d.shape[0]
    s, e = shard_bounds(R, rank, world)
    near, far = bounds
    near = near.reshape(-1)[s:e] if isinstance(near, torch.Tensor) else near
    far = far.reshape(-1)[s:e] if isinstance(far, torch.Tensor) else far
    out = render((flat_o[s:e], flat_d[s:e]), (near, far), **kwargs)
    if keys is not None:
        out = {k: out[k] for k in keys}
    if not gather or not multi_process(group):
        return out
    sizes = [shard_bounds(R, r, world) for r in range(world)]
    return {k: all_gather_rows(v, [b - a for a, b in sizes], group) for k, v in out.items()}


def all_gather_rows(t: torch.Tensor, rows_per_rank: Sequence[int], group=None) -> torch.Tensor:
    """All-gather along dim 0 with (possibly) unequal row counts: pad to the max, one collective, trim."""
    rank, world = _world(group)
    if not multi_process(group):
        return t
    m = max(rows_per_rank)
    pad = t
    if t.shape[0] < m:
        pad = torch.cat([t, t.new_zeros((m - t.shape[0],) + tuple(t.shape[1:]))], 0)
    pad = pad.contiguous()
    buf = pad.new_empty((world * m,) + tuple(pad.shape[1:]))
    collective("all_gather", lambda async_op: dist.all_gather_into_tensor(buf, pad, group=group, async_op=async_op), group)
    if all(r == m for r in rows_per_rank):
        return buf
    return torch.cat([buf[r * m:r * m + rows_per_rank[r]] for r in range(world)], 0)


def all_gather_patches(local: Dict[str, torch.Tensor], n_patches: int, group=None,
                       keys: Iterable[str] = PATCH_KEYS, stats: Optional[dict] = None) -> Dict[str, torch.Tensor]:
    """Training: ``local[k]`` is [n_local, ...] for the patches this rank owns (in increasing global index; any
    trailing shape, e.g. [n_local,P,P,2] semantics, [n_local,384,14,14] features).  Returns [n_patches, ...] per key in
    GLOBAL patch order on every rank.  All keys of one dtype travel in ONE flat buffer -> one collective per step
    (fp32 everywhere in practice): the payload is ~0.5 MiB per patch, latency-bound over xGMI.  The gathered tensors
    carry no autograd history (remote patches act as the losses' detached negatives; see `splice_local_patches`).
    `stats`, if given, receives ``bytes_per_patch`` and ``collectives``."""
    rank, world = _world(group)
    present = [k for k in keys if k in local]
    if not multi_process(group):
        return {k: local[k].detach() for k in present}
    counts = [len(local_patches(n_patches, r, world)) for r in range(world)]
    order = [b for r in range(world) for b in local_patches(n_patches, r, world)]  # rank-major -> global id
    inv = [0] * n_patches
    for pos, b in enumerate(order):
        inv[b] = pos
    n_local = counts[rank]
    out: Dict[str, torch.Tensor] = {}
    by_dtype: Dict[torch.dtype, List[str]] = {}
    for k in present:
        if local[k].shape[0] != n_local:
            raise ValueError(f"all_gather_patches: `{k}` holds {local[k].shape[0]} patches, this rank owns {n_local}")
        by_dtype.setdefault(local[k].dtype, []).append(k)
    nbytes = 0
    for dt, ks in by_dtype.items():
        widths = [int(torch.Size(local[k].shape[1:]).numel()) for k in ks]
        flat = torch.cat([local[k].detach().reshape(n_local, wd) for k, wd in zip(ks, widths)], 1)
        nbytes += flat.shape[1] * flat.element_size()
        g = all_gather_rows(flat, counts, group)
        g = g[device_index(inv, torch.long, g.device)]      # cached on the device: no per-step upload (= synchronisation)
        off = 0
        for k, wd in zip(ks, widths):
            out[k] = g[:, off:off + wd].reshape((n_patches,) + tuple(local[k].shape[1:]))
            off += wd
    if stats is not None:
        stats["bytes_per_patch"] = nbytes
        stats["collectives"] = len(by_dtype)
    return {k: out[k] for k in present}


def splice_local_patches(gathered: Dict[str, torch.Tensor], local: Dict[str, torch.Tensor], n_patches: int,
                         group=None) -> Dict[str, torch.Tensor]:
    """Put this rank's own (gradient-carrying) patches back into the gathered, detached batch.  Every rank then
    evaluates the batch-wide correlation losses on the same values, and back-propagation reaches exactly the patches the
    rank rendered -- in both of their roles (as patch n and as the negative of other patches, utils/image.py:359-360).
    Summing the parameter gradients over the ranks (`all_reduce_grads`) gives the single-process gradient."""
    rank, world = _world(group)
    own = local_patches(n_patches, rank, world)
    out = dict(gathered)
    for k, g in gathered.items():
        if k in local and local[k].requires_grad and len(own):
            # one process: the "gathered" batch IS the local one (no copy, no index kernel)
            out[k] = local[k] if not multi_process(group) else g.index_put((device_index(own, torch.long, g.device),), local[k])
    return out


_INDEX_CACHE: Dict[tuple, torch.Tensor] = {}


def device_index(values, dtype, device) -> torch.Tensor:
    """A small constant index list as a device tensor, uploaded once: `torch.tensor(list, device=gpu)` is a pageable
    host-to-device copy, i.e. a full synchronisation of the stream -- per step, per call site."""
    key = (tuple(int(v) for v in values), dtype, str(device))
    t = _INDEX_CACHE.get(key)
    if t is None:
        if len(_INDEX_CACHE) > 256:
            _INDEX_CACHE.clear()
        t = _INDEX_CACHE[key] = torch.tensor(list(key[0]), dtype=dtype, device=device)
    return t


def all_reduce_grads(params: Iterable[torch.nn.Parameter], group=None, average: bool = False) -> None:
    """ONE flat all-reduce (sum, or mean with average=True) of the gradients of `params`, in place.  82 436 floats for
    the frozen-backbone recipe, 1.27 M for the full model: latency-bound, so a single bucket."""
    if not multi_process(group):
        return
    ps = [p for p in params if p.requires_grad]
    if not ps:
        return
    for p in ps:
        if p.grad is None:
            p.grad = torch.zeros_like(p)
    flat = torch.cat([p.grad.reshape(-1) for p in ps])
    collective("grad_all_reduce", lambda async_op: dist.all_reduce(flat, op=dist.ReduceOp.SUM, group=group, async_op=async_op), group)
    if average:
        flat /= dist.get_world_size(group)
    views, off = [], 0
    for p in ps:
        n = p.numel()
        views.append(flat[off:off + n].view_as(p.grad))
        off += n
    torch._foreach_copy_([p.grad for p in ps], views)        # one launch for all tensors


_SIDE_STREAMS: Dict[str, "torch.cuda.Stream"] = {}


def _side_stream(device) -> "torch.cuda.Stream":
    key = str(device)
    if key not in _SIDE_STREAMS:
        _SIDE_STREAMS[key] = torch.cuda.Stream(device=device)
    return _SIDE_STREAMS[key]


def geo_loss_both(geo_loss, depth, code0, code1, ray_o, ray_d, sim, rows, group=None):
    """geo_loss(depth, code0, ...) + geo_loss(depth, code1, ...) -- the coarse and the fine semantic maps against the same
    geometry (engines/trainer.py:147-166) -- as ONE evaluation over the 2B stacked patches [code0; code1] with the geometry
    repeated and a block-diagonal similarity matrix (negatives stay inside their half).  Every term of the loss is a mean over
    the batch and the batch-wide quantities it subtracts (utils/image.py:316-319) depend on the geometry only, which both halves
    share, so the stacked mean is exactly (L0 + L1) / 2: half the launches, and in the sharded step four collectives per step
    instead of eight.  (Not used for the appearance loss: its two calls draw their own sample coordinates, so their batch means
    differ, and not with rand_neg, where every call draws its own permutation.)"""
    B = code0.shape[0]
    big = torch.full_like(sim, float("inf"))
    sim2 = torch.cat([torch.cat([sim, big], 1), torch.cat([big, sim], 1)], 0)
    rows2 = list(rows) + [B + int(r) for r in rows]
    both = geo_loss(depth.repeat(2, 1, 1, 1), torch.cat([code0, code1], 0), [ray_o.repeat(2, 1, 1, 1), ray_d.repeat(2, 1, 1, 1), None],
                    sim2, rows=rows2, group=group)
    return 2.0 * both


def loss_generator(device, step: int, base_seed: int = 0) -> torch.Generator:
    """The correlation losses draw sample coordinates and permutations (utils/image.py:306-309,343-344,357); in the
    sharded step every rank evaluates the batch-wide losses, and the summed gradients equal the single-process
    gradient only if all ranks draw the SAME values.  Give the loss modules a generator of their own
    (``loss.generator = loss_generator(dev, step)``), seeded identically on every rank and advanced per step, instead
    of re-seeding the global generator (which would also freeze the render's perturbation / noise draws)."""
    g = torch.Generator(device=device)
    g.manual_seed((int(base_seed) * 1_000_003 + int(step)) & 0x7FFFFFFFFFFFFFFF)
    return g


def similarity_matrix(cls_tokens: torch.Tensor) -> torch.Tensor:
    """[B,B] cosine similarity of the patches' class tokens (utils/image.py:187-190, engines/trainer.py:125): the
    correlation losses take each patch's negative as the argmin of its column.  B x 384 values: host-side glue."""
    x = cls_tokens.reshape(cls_tokens.shape[0], -1)
    return torch.nn.functional.cosine_similarity(x.unsqueeze(0), x.unsqueeze(1), dim=2)


def _direct_losses(corr_loss, geo_loss) -> bool:
    """Whether the step takes `_losses_direct` -- a function of the modules and the autograd mode only: the same on every rank."""
    return (corr_loss is not None and geo_loss is not None and torch.is_grad_enabled() and hasattr(corr_loss, "value_and_grad")
            and hasattr(geo_loss, "pair_value_and_grads") and not getattr(geo_loss, "rand_neg", False)
            and not getattr(corr_loss, "rand_neg", False) and getattr(corr_loss, "use_sim_matrix", True)
            and os.environ.get("NSOS_STEP_AUTOGRAD_LOSSES", "") in ("", "0"))


def _losses_direct(full, sim, s0, s1, own, gen, corr_loss, geo_loss, correlation_w, geo_w, dev, group, overlap_losses,
                   contrast_loss=None, contrast_w=0.0, loss_out=None):
    """The loss section with the gradient bookkeeping done here instead of by autograd: every loss launch already returns
    d loss / d code, the loss weights ride on the kernels' own weights, the negatives are found once, the four gradients are
    summed by one multi-tensor launch, and autograd is entered ONCE, at the two rendered semantic maps
    (torch.autograd.backward(tensors, grad_tensors)).  The same numbers as the autograd formulation below up to the rounding
    of `weight * (self_weight, neg_weight)`, with sixteen element-wise launches fewer per step (three scalar products and sums
    of the total, their three backward products, ones_like, four gradient scalings, two gradient accumulations, two argmin
    reductions).  The condition for this path is the same on every rank (modules + autograd mode), as is the sequence of
    collectives inside the geometric loss."""
    side = _side_stream(dev) if (overlap_losses and dev.type == "cuda") else None
    with torch.no_grad():
        for m in (corr_loss, geo_loss):
            if gen is not None:
                m.generator = gen
        f = full["feat"]
        B = s0.shape[0]
        # utils/image.py:354, once for the three evaluations: [neg, neg + B] straight from the class tokens (one launch)
        from .losses import similarity_negatives
        neg2 = similarity_negatives(full["cls_"], copies=2) if full["cls_"].shape[0] <= 120 else None
        if neg2 is None:
            n_ = corr_loss._neg_index(similarity_matrix(full["cls_"]) if sim is None else sim, B, dev)
            neg2 = torch.cat([n_, n_ + B])
        neg = neg2[:B]
        xy = corr_loss.draw_coords(2, B, dev)      # both evaluations' coordinates in one launch (rand1, rand2 of s0, then of s1)
        # the three evaluations write their loss into slots of ONE buffer: the step's total is one reduction launch (straight into
        # `loss_out` when the caller has a place for it) instead of a stack (cat), a sum and a copy
        lbuf = torch.empty(3, device=dev, dtype=torch.float32)
        if multi_process(group):
            # N > 1: all three evaluations row-partitioned (each rank the pair sets of its own patches), their phases interleaved
            # so that the step issues ONE all-reduce per phase for all of them: the means (3 x 8 doubles) and the sums (the
            # gradients' role sums + the split loss sums, fp32)
            from .losses import exchange_floats
            P2 = full["depth"].shape[1] * full["depth"].shape[2]
            S2 = corr_loss.feature_samples ** 2
            sizes = [exchange_floats(2 * B, P2), exchange_floats(B, S2), exchange_floats(B, S2)]
            means = torch.zeros(24, device=dev, dtype=torch.float64)
            sums = torch.empty(sum(sizes), device=dev, dtype=torch.float32)
            xs = torch.split(sums, sizes)
            run_a0, (la0, ga0) = corr_loss.rows_phased(f, s0, sim, own, (means[0:8], xs[1]), correlation_w, neg, xy[0], loss_out=lbuf[0])
            run_a1, (la1, ga1) = corr_loss.rows_phased(f, s1, sim, own, (means[8:16], xs[2]), correlation_w, neg, xy[1], loss_out=lbuf[1])
            run_g, (lg, gg0, gg1) = geo_loss.pair_phased(full["depth"], full["semantics0"], full["semantics"], full["ray_o"], full["ray_d"],
                                                        sim, own, (means[16:24], xs[0]), geo_w, neg2, loss_out=lbuf[2])
            runs = (run_g, run_a0, run_a1)
            for r_ in runs:
                r_(0)
            collective("loss_means_all_reduce", lambda async_op: dist.all_reduce(means, group=group, async_op=async_op), group)
            for r_ in runs:
                r_(1)
            collective("loss_sums_all_reduce", lambda async_op: dist.all_reduce(sums, group=group, async_op=async_op), group)
            for r_ in runs:
                r_(2)
            side = None
        elif side is not None:
            main = torch.cuda.current_stream(dev)
            side.wait_stream(main)
            with torch.cuda.stream(side):
                for t in (f, s0, s1, neg2, lbuf):
                    t.record_stream(side)
                xy.record_stream(side)
                la0, ga0 = corr_loss.value_and_grad(f, s0, sim, correlation_w, neg, coords=xy[0], loss_out=lbuf[0])
                la1, ga1 = corr_loss.value_and_grad(f, s1, sim, correlation_w, neg, coords=xy[1], loss_out=lbuf[1])
        else:
            la0, ga0 = corr_loss.value_and_grad(f, s0, sim, correlation_w, neg, coords=xy[0], loss_out=lbuf[0])
            la1, ga1 = corr_loss.value_and_grad(f, s1, sim, correlation_w, neg, coords=xy[1], loss_out=lbuf[1])
        if not multi_process(group):
            lg, gg0, gg1 = geo_loss.pair_value_and_grads(full["depth"], full["semantics0"], full["semantics"], full["ray_o"], full["ray_d"],
                                                          sim, rows=own, group=group, weight=geo_w, neg=neg2, grad_mode=True, loss_out=lbuf[2])
        if side is not None:
            torch.cuda.current_stream(dev).wait_stream(side)
            for t in (ga0, ga1):
                t.record_stream(torch.cuda.current_stream(dev))
        # gradients w.r.t. the channel-last maps: the appearance loss hands back [B,C,P,P] views of [B,P,P,C] buffers
        torch._foreach_add_([gg0, gg1], [ga0.permute(0, 2, 3, 1), ga1.permute(0, 2, 3, 1)])
    # the contrastive term (engines/trainer.py:168-170) is the one loss here that autograd differentiates itself: the reference
    # back-propagates it into the feature extractor through `cls_`, so it is evaluated with the graph on (under
    # no_grad that gradient was silently dropped on this path only)
    c = None
    if contrast_loss is not None:
        with torch.enable_grad():
            c = (contrast_w * contrast_loss(full["cls_"])).reshape(())
    with torch.no_grad():
        if c is None:
            loss = torch.sum(lbuf, dim=0, out=loss_out) if loss_out is not None else lbuf.sum()
        else:
            loss = torch.cat([lbuf, c.detach().reshape(1)]).sum()
            if loss_out is not None:
                loss = loss_out.copy_(loss)
    roots = [(t, g) for t, g in ((full["semantics0"], gg0), (full["semantics"], gg1)) if t.requires_grad]
    if c is not None and c.requires_grad:
        roots.append((c, torch.ones_like(c)))
    if roots:
        torch.autograd.backward([t for t, _ in roots], [g for _, g in roots])
    return loss


def _losses_and_backward(net, full, sim, s0, s1, own, gen, corr_loss, geo_loss, correlation_w, geo_w, dev, group, overlap_losses,
                         contrast_loss=None, contrast_w=0.0, loss_out=None):
    """The loss section of `sharded_patch_step` (engines/trainer.py:127-166) and the backward through this rank's patches."""
    if _direct_losses(corr_loss, geo_loss):
        return _losses_direct(full, sim, s0, s1, own, gen, corr_loss, geo_loss, correlation_w, geo_w, dev, group, overlap_losses,
                              contrast_loss, contrast_w, loss_out)
    loss = None
    # The appearance loss is a train of small launches (121 sample points per patch: grids of a few hundred threads), the
    # geometric one a few chip-filling ones with one workgroup per CU: on a stream of its own the former runs in the latter's
    # shadow (forward here, and backward too -- autograd runs a node on the stream its forward ran on).  The draws of both
    # come from the host-side generator state in program order, so the values do not depend on the overlap.
    side = _side_stream(dev) if (overlap_losses and dev.type == "cuda" and corr_loss is not None and geo_loss is not None) else None
    app = None
    if corr_loss is not None:
        if gen is not None:
            corr_loss.generator = gen
        f = full["feat"]
        if hasattr(corr_loss, "queue_coords"):     # the same one-launch draw as the direct path: both formulations see the same samples
            xy = corr_loss.draw_coords(2, s0.shape[0], dev)
            corr_loss.queue_coords([xy[0], xy[1]])
        if side is not None:
            main = torch.cuda.current_stream(dev)
            side.wait_stream(main)
            with torch.cuda.stream(side):
                for t in (f, s0, s1, sim) + ((xy,) if hasattr(corr_loss, "queue_coords") else ()):
                    t.record_stream(side)
                app = correlation_w * (corr_loss(f, s0, sim) + corr_loss(f, s1, sim))
        else:
            loss = correlation_w * (corr_loss(f, s0, sim) + corr_loss(f, s1, sim))
    if geo_loss is not None:
        if gen is not None:
            geo_loss.generator = gen
        # the O(P^4) geometric loss is evaluated ONCE across the ranks: each rank its own row patches (losses.py); it uses the
        # FINE depth for both terms (engines/trainer.py:159-160)
        if getattr(geo_loss, "rand_neg", False):
            depth = full["depth"].detach().permute(0, 3, 1, 2).contiguous()
            ro, rd = full["ray_o"].permute(0, 3, 1, 2), full["ray_d"].permute(0, 3, 1, 2)
            g = geo_w * (geo_loss(depth, s0, [ro, rd, None], sim, rows=own, group=group) +
                         geo_loss(depth, s1, [ro, rd, None], sim, rows=own, group=group))
        elif hasattr(geo_loss, "forward_pair"):
            # both codes in one evaluation on the renderer's own channel-last tensors: no stacked copies, no layout copies
            g = geo_w * geo_loss.forward_pair(full["depth"], full["semantics0"], full["semantics"], full["ray_o"], full["ray_d"],
                                              sim, rows=own, group=group)
        else:
            depth = full["depth"].detach().permute(0, 3, 1, 2).contiguous()
            ro, rd = full["ray_o"].permute(0, 3, 1, 2), full["ray_d"].permute(0, 3, 1, 2)
            g = geo_w * geo_loss_both(geo_loss, depth, s0, s1, ro, rd, sim, own, group)
        loss = g if loss is None else loss + g
    if app is not None:
        torch.cuda.current_stream(dev).wait_stream(side)
        app.record_stream(torch.cuda.current_stream(dev))
        loss = app + loss
    if contrast_loss is not None:
        c = contrast_w * contrast_loss(full["cls_"])                        # engines/trainer.py:168-170
        loss = c if loss is None else loss + c
    if loss is None:
        raise ValueError("sharded_patch_step: give at least one of corr_loss / geo_loss / contrast_loss")
    if loss.requires_grad:
        loss.backward()
    if loss_out is not None:
        loss_out.copy_(loss.detach())
        return loss_out
    return loss


def sharded_patch_step(net, rays: torch.Tensor, bounds, n_patches: int, feat: torch.Tensor, cls_tokens: torch.Tensor,
                       corr_loss=None, geo_loss=None, correlation_w: float = 1.0, geo_w: float = 0.01, step: int = 0,
                       seed: Optional[int] = 0, group=None, timings: Optional[dict] = None,
                       overlap_losses: bool = True, contrast_loss=None, contrast_w: float = 0.0,
                       generator: Optional[torch.Generator] = None, loss_out: Optional[torch.Tensor] = None) -> torch.Tensor:
    """One patch-mode training step of the path with the patch batch sharded over the ranks -- the loss section of
    `train_one_step` (engines/trainer.py:101-166) re-stated for one process per GPU:

      render this rank's patches (`rays` [2, n_local, P, P, 3], train mode)
      -> ONE flat all-gather of what the batch-wide losses read from every patch: semantics0 / semantics / fine depth,
         DINO `feat` [n_local,384,14,14] and `cls_tokens` [n_local,384] of the rank's own crops, ray_o / ray_d
      -> similarity matrix of the class tokens -> negatives (utils/image.py:354)
      -> the rank's own gradient-carrying patches are spliced back into the detached batch
      -> appearance + geometric correlation losses on `semantics0` and `semantics` (engines/trainer.py:127-166); the
         O(P^4) geometric one row-partitioned: each rank its own patches' pair sets, four tiny sum-all-reduces
      -> `contrast_loss` (NeRFContrastive, engines/trainer.py:168-170), if given: contrast_w * contrast_loss(cls_) on the
         gathered class tokens -- every rank evaluates the same B x 384 values; its gradient goes to the feature extractor's
         input, which is outside this path, so under the frozen-backbone recipe it only moves the loss value
      -> backward through the rank's own patches -> ONE flat all-reduce (sum) of the parameter gradients.

    Returns the (batch-wide) loss; `.grad` of the trainable parameters then holds the single-process gradient.
    With no process group it is the plain single-GPU step over `n_patches` local patches.  The losses' random draws come
    from `loss_generator(device, step, seed)` -- identical on every rank; seed=None uses torch's global generator like the
    reference (single process only: ranks would draw different coordinates).  `generator`, if given, is used INSTEAD (a
    persistent generator the caller advances step after step -- what a captured graph of this step needs, graphs.py:
    a fresh per-step generator cannot be registered with a graph).
    `timings`, if a dict, receives HIP event pairs under 'render', 'gather', 'losses_backward' and 'allreduce' (recorded on the current stream;
    RCCL's own stream is joined by the non-async collectives before the second event) and the gather `stats`."""
    rank, world = _world(group)
    own = local_patches(n_patches, rank, world)
    if rays.shape[1] != len(own):
        raise ValueError(f"sharded_patch_step: rank {rank} owns {len(own)} of {n_patches} patches, got rays for {rays.shape[1]}")
    dev = rays.device
    ev_r = None
    if timings is not None and dev.type == "cuda":
        ev_r = torch.cuda.Event(enable_timing=True)
        ev_r.record()
    if len(own):
        ret = net(rays, bounds, retraw=False)
    else:   # more ranks than patches: this rank renders nothing but still takes part in every collective
        P_ = tuple(rays.shape[2:4])
        ret = {"semantics": rays.new_zeros((0,) + P_ + (2,)), "semantics0": rays.new_zeros((0,) + P_ + (2,)),
               "depth": rays.new_zeros((0,) + P_ + (1,))}
    local = {"semantics": ret["semantics"], "semantics0": ret["semantics0"], "depth": ret["depth"],
             "feat": feat, "cls_": cls_tokens, "ray_o": rays[0], "ray_d": rays[1]}
    ev = None
    if timings is not None and dev.type == "cuda":
        ev = [torch.cuda.Event(enable_timing=True) for _ in range(4)]
        ev[0].record()
    stats: dict = {}
    full = all_gather_patches(local, n_patches, group, stats=stats)
    if ev:
        ev[1].record()
    full = splice_local_patches(full, local, n_patches, group)
    sim = None if _direct_losses(corr_loss, geo_loss) else similarity_matrix(full["cls_"])   # (the direct path finds the negatives itself)
    s0 = full["semantics0"].permute(0, 3, 1, 2)
    s1 = full["semantics"].permute(0, 3, 1, 2)
    loss = None
    gen = generator if generator is not None else (loss_generator(dev, step, seed) if seed is not None else None)
    # (None: the global generator, single process only.)  The generator is lent to the caller's loss modules for the
    # duration of this step only: a validation loss or a
    # single-GPU step that uses the same modules afterwards draws from whatever generator they had before
    lent = [(m, getattr(m, "generator", None)) for m in (corr_loss, geo_loss) if m is not None and gen is not None]
    try:
        loss = _losses_and_backward(net, full, sim, s0, s1, own, gen, corr_loss, geo_loss, correlation_w, geo_w, dev, group,
                                    overlap_losses, contrast_loss, contrast_w, loss_out)   # (loss_out: a 0-dim tensor the total is written into)
    finally:
        for m, g0 in lent:
            m.generator = g0
    if ev:
        ev[2].record()
    all_reduce_grads(net.parameters(), group)
    if ev:
        ev[3].record()
        timings.setdefault("render", []).append((ev_r, ev[0]))
        timings.setdefault("gather", []).append((ev[0], ev[1]))
        timings.setdefault("losses_backward", []).append((ev[1], ev[2]))      # (with their own two reductions when N > 1)
        timings.setdefault("allreduce", []).append((ev[2], ev[3]))
        timings["stats"] = stats
    return loss.detach()
