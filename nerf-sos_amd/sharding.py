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

from typing import Callable, Dict, Iterable, List, Optional, Sequence, Tuple

import torch
import torch.distributed as dist

# outputs the losses read across patches (engines/trainer.py:127-166)
PATCH_KEYS = ("semantics", "semantics0", "depth")


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


def _world(group) -> Tuple[int, int]:
    if dist.is_available() and dist.is_initialized():
        return dist.get_rank(group), dist.get_world_size(group)
    return 0, 1


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
    if not gather or world == 1:
        return out
    sizes = [shard_bounds(R, r, world) for r in range(world)]
    return {k: all_gather_rows(v, [b - a for a, b in sizes], group) for k, v in out.items()}


def all_gather_rows(t: torch.Tensor, rows_per_rank: Sequence[int], group=None) -> torch.Tensor:
    """All-gather along dim 0 with (possibly) unequal row counts: pad to the max, one collective, trim."""
    rank, world = _world(group)
    if world == 1:
        return t
    m = max(rows_per_rank)
    pad = t
    if t.shape[0] < m:
        pad = torch.cat([t, t.new_zeros((m - t.shape[0],) + tuple(t.shape[1:]))], 0)
    pad = pad.contiguous()
    buf = pad.new_empty((world * m,) + tuple(pad.shape[1:]))
    dist.all_gather_into_tensor(buf, pad, group=group)
    if all(r == m for r in rows_per_rank):
        return buf
    return torch.cat([buf[r * m:r * m + rows_per_rank[r]] for r in range(world)], 0)


def all_gather_patches(local: Dict[str, torch.Tensor], n_patches: int, group=None,
                       keys: Iterable[str] = PATCH_KEYS) -> Dict[str, torch.Tensor]:
    """Training: ``local[k]`` is [n_local, P, P, C] for the patches this rank owns (in increasing global
    index).  Returns [n_patches, P, P, C] per key in GLOBAL patch order on every rank.  The gathered
    tensors carry no autograd history (the remote patches act as the losses' detached negatives)."""
    rank, world = _world(group)
    out = {}
    counts = [len(local_patches(n_patches, r, world)) for r in range(world)]
    order = [b for r in range(world) for b in local_patches(n_patches, r, world)]  # rank-major -> global id
    inv = torch.empty(n_patches, dtype=torch.long)
    inv[torch.tensor(order, dtype=torch.long)] = torch.arange(n_patches)
    for k in keys:
        if k not in local:
            continue
        g = all_gather_rows(local[k].detach(), counts, group)
        out[k] = g[inv.to(g.device)]
    return out


def splice_local_patches(gathered: Dict[str, torch.Tensor], local: Dict[str, torch.Tensor], n_patches: int,
                         group=None) -> Dict[str, torch.Tensor]:
    """Put this rank's own (gradient-carrying) patches back into the gathered, detached batch.  Every rank then
    evaluates the batch-wide correlation losses on the same values, and back-propagation reaches exactly the patches the
    rank rendered -- in both of their roles (as patch n and as the negative of other patches, utils/image.py:359-360).
    Summing the parameter gradients over the ranks (`all_reduce_grads`) gives the single-process gradient."""
    rank, world = _world(group)
    own = torch.tensor(local_patches(n_patches, rank, world), dtype=torch.long)
    out = dict(gathered)
    for k, g in gathered.items():
        if k in local and local[k].requires_grad and len(own):
            out[k] = g.index_put((own.to(g.device),), local[k])
    return out


def all_reduce_grads(params: Iterable[torch.nn.Parameter], group=None, average: bool = False) -> None:
    """ONE flat all-reduce (sum, or mean with average=True) of the gradients of `params`, in place.  82 436 floats for
    the frozen-backbone recipe, 1.27 M for the full model: latency-bound, so a single bucket."""
    ps = [p for p in params if p.requires_grad]
    if not ps or not (dist.is_available() and dist.is_initialized()):
        return
    for p in ps:
        if p.grad is None:
            p.grad = torch.zeros_like(p)
    flat = torch.cat([p.grad.reshape(-1) for p in ps])
    dist.all_reduce(flat, op=dist.ReduceOp.SUM, group=group)
    if average:
        flat /= dist.get_world_size(group)
    off = 0
    for p in ps:
        n = p.numel()
        p.grad.copy_(flat[off:off + n].view_as(p.grad))
        off += n
