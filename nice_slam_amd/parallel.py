"""Ray-sharded execution of ``render_batch_ray`` across the GPUs of one node.

One process per GPU (torch.distributed, backend "nccl" = RCCL over xGMI).  Every rank holds the same
ray batch, grids and decoders (replicated), renders a contiguous block of N/W rays and
  * all-gathers the 28 B/ray outputs so the caller's loss code runs unchanged on the full batch,
  * sums the feature-grid gradients (dense all-reduce, one collective per grid, issued as soon as the
    local backward kernel is enqueued) and the flat decoder-parameter gradients,
  * all-gathers the per-ray gradients (pose optimisation in BA).
The two batch-global scalars of the path (max(gt_depth), Renderer.py:109,144) are taken over the FULL
batch before slicing, so shard results are identical to the single-GPU result.
Tracking batches are 200-1000 rays with pose-only gradients: "replicas only", do not wrap the tracker.
"""
from __future__ import annotations

from typing import List, Optional, Tuple

import torch
import torch.distributed as dist


def shard_range(n: int, world: int, rank: int) -> Tuple[int, int]:
    """Contiguous block partition of n rays; the first (n % world) ranks get one extra ray."""
    base, rem = divmod(n, world)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


class _ShardRows(torch.autograd.Function):
    """x[lo:hi]; backward all-gathers the shard gradients into the full-size gradient."""

    @staticmethod
    def forward(ctx, x, lo, hi, sizes, group):
        ctx.sizes, ctx.group = sizes, group
        return x[lo:hi].contiguous()

    @staticmethod
    def backward(ctx, g):
        return _all_gather_rows(g.contiguous(), ctx.sizes, ctx.group), None, None, None, None


class _GatherRows(torch.autograd.Function):
    """all-gather of row blocks; backward keeps this rank's block (every rank computes the same loss)."""

    @staticmethod
    def forward(ctx, x, lo, hi, sizes, group):
        ctx.lo, ctx.hi = lo, hi
        return _all_gather_rows(x.contiguous(), sizes, group)

    @staticmethod
    def backward(ctx, g):
        return g[ctx.lo:ctx.hi].contiguous(), None, None, None, None


class _SumGrad(torch.autograd.Function):
    """identity; backward all-reduces (SUM) the gradient -- used on the replicated feature grids."""

    @staticmethod
    def forward(ctx, x, group):
        ctx.group = group
        return x.view_as(x)

    @staticmethod
    def backward(ctx, g):
        if g.is_contiguous():
            buf = g
        elif g.dim() == 5 and g.is_contiguous(memory_format=torch.channels_last_3d):
            buf = g.permute(0, 2, 3, 4, 1)            # the same dense memory, viewed as a standard-contiguous tensor
        else:
            g = g.contiguous()
            buf = g
        dist.all_reduce(buf, op=dist.ReduceOp.SUM, group=ctx.group)
        return g, None


def _all_gather_rows(x: torch.Tensor, sizes: List[int], group) -> torch.Tensor:
    """all-gather of row blocks whose sizes differ by at most one: pad to the largest block, gather into one
    buffer (a single collective), drop the padding."""
    m = max(sizes)
    if x.shape[0] < m:
        x = torch.cat([x, x.new_zeros((m - x.shape[0],) + tuple(x.shape[1:]))], 0)
    buf = torch.empty((len(sizes) * m,) + tuple(x.shape[1:]), dtype=x.dtype, device=x.device)
    dist.all_gather_into_tensor(buf, x.contiguous(), group=group)
    if all(s == m for s in sizes):
        return buf
    return torch.cat([buf[r * m:r * m + s] for r, s in enumerate(sizes)], 0)


class ShardedRenderer:
    """Wraps a Renderer-like object (``render_batch_ray`` + the ``_gt_max`` / ``_reduce_hook`` protocol)."""

    def __init__(self, renderer, group=None):
        self.renderer = renderer
        self.group = group

    def __getattr__(self, name):
        return getattr(self.renderer, name)

    def _reduce_flat(self, _d_grids, gflat: Optional[torch.Tensor]):
        if gflat is not None:
            dist.all_reduce(gflat, op=dist.ReduceOp.SUM, group=self.group)

    def render_batch_ray(self, c, decoders, rays_d, rays_o, device, stage, gt_depth=None):
        world, rank = dist.get_world_size(self.group), dist.get_rank(self.group)
        n = rays_o.shape[0]
        sizes = [shard_range(n, world, r)[1] - shard_range(n, world, r)[0] for r in range(world)]
        lo, hi = shard_range(n, world, rank)
        if stage == "coarse":
            gt_depth = None
        o_s = _ShardRows.apply(rays_o, lo, hi, sizes, self.group)
        d_s = _ShardRows.apply(rays_d, lo, hi, sizes, self.group)
        gt_s = None
        c_s = {k: (_SumGrad.apply(v, self.group) if (torch.is_grad_enabled() and v.requires_grad) else v) for k, v in c.items()}
        self.renderer._gt_max = None
        if gt_depth is not None:
            gt_depth = gt_depth.reshape(-1)
            self.renderer._gt_max = torch.max(gt_depth.detach().to(torch.float32)).reshape(1)    # batch-global, before slicing
            gt_s = gt_depth[lo:hi]
        self.renderer._reduce_hook = self._reduce_flat
        try:
            depth, unc, col = self.renderer.render_batch_ray(c_s, decoders, d_s, o_s, device, stage, gt_depth=gt_s)
        finally:
            self.renderer._gt_max = None
            self.renderer._reduce_hook = None
        return tuple(_GatherRows.apply(x, lo, hi, sizes, self.group) for x in (depth, unc, col))
