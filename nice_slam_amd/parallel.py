"""Ray-sharded execution of ``render_batch_ray`` across the GPUs of one node.

One process per GPU (torch.distributed, backend "nccl" = RCCL over xGMI).  Every rank holds the same
ray batch, grids and decoders (replicated), renders a contiguous block of N/W rays and
  * all-gathers the 28 B/ray outputs (one collective) so the caller's loss code runs unchanged on the full batch,
  * sums the feature-grid gradients and the flat decoder-parameter gradients: dense in-place all-reduces, or -- with
    the frame's frustum voxel masks set -- ONE all-reduce over the compacted rows of the selected voxels,
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


class _GatherOutputs(torch.autograd.Function):
    """(depth fp64, uncertainty fp64, colour fp32) of this rank's rays -> the same three for the whole batch, with ONE
    all-gather: the 28 bytes of a ray travel as five fp64 columns (fp32 -> fp64 -> fp32 is exact).  Backward keeps this
    rank's block of each gradient (every rank computes the same loss on the full batch)."""

    @staticmethod
    def forward(ctx, depth, unc, rgb, lo, hi, sizes, group):
        ctx.lo, ctx.hi = lo, hi
        packed = torch.cat([depth.reshape(-1, 1).to(torch.float64), unc.reshape(-1, 1).to(torch.float64),
                            rgb.to(torch.float64)], 1)
        full = _all_gather_rows(packed, sizes, group)
        return (full[:, 0].to(depth.dtype).contiguous(), full[:, 1].to(unc.dtype).contiguous(),
                full[:, 2:5].to(rgb.dtype).contiguous())

    @staticmethod
    def backward(ctx, g_depth, g_unc, g_rgb):
        lo, hi = ctx.lo, ctx.hi
        cut = lambda g: None if g is None else g[lo:hi].contiguous()
        return cut(g_depth), cut(g_unc), cut(g_rgb), None, None, None, None


class _SumGrads(torch.autograd.Function):
    """identity on the replicated feature grids; backward hands ALL their gradients (they become available together, from
    one backward kernel) to ``ShardedRenderer._reduce_grid_grads`` -- one exchange per render call."""

    @staticmethod
    def forward(ctx, owner, keys, *grids):
        ctx.owner, ctx.keys = owner, keys
        ctx.set_materialize_grads(False)        # a grid the stage does not read gets no gradient, and no exchange
        return tuple(g.view_as(g) for g in grids)

    @staticmethod
    def backward(ctx, *gs):
        return (None, None, *ctx.owner._reduce_grid_grads(ctx.keys, gs))


def _voxel_rows(g: torch.Tensor):
    """(g', 2-D view of g' with one axis = voxels, that axis): a dense view suitable for in-place collectives and for
    row compaction.  Channels-last grids (the product layout) give [V,32] rows of 128 B; NCDHW gives [32,V]."""
    if g.dim() == 5 and not g.is_contiguous() and g.is_contiguous(memory_format=torch.channels_last_3d):
        return g, g.permute(0, 2, 3, 4, 1).reshape(-1, g.shape[1]), 0
    if not g.is_contiguous():
        g = g.contiguous()
    return g, g.reshape(g.shape[1], -1), 1


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
    """Wraps a Renderer-like object (``render_batch_ray`` + the ``_gt_max`` / ``_reduce_hook`` protocol).

    Gradient exchange.  Without voxel masks every feature-grid gradient is all-reduced densely, in place (Replica colour
    stage: 48.4 MB per iteration).  The mapper only ever steps the voxels inside the current frame's
    frustum mask (Mapper.py:315-333: ``val_grad = val[mask]`` is the optimised leaf), and that mask is identical on every
    rank (same pose, same depth image), so after ``set_voxel_masks`` only the masked voxel rows are exchanged: they are
    compacted (128-B rows), packed together with the decoder-parameter gradients into ONE buffer, summed with one
    all-reduce and scattered back.  Gradients of voxels outside the mask stay rank-local partial sums -- nothing reads
    them (MaskedGridAdam / the reference's masked leaf skip those voxels)."""

    def __init__(self, renderer, group=None):
        self.renderer = renderer
        self.group = group
        self._rows = {}                 # grid key -> int64 indices of the selected voxels ([Z,Y,X] raster order)
        self._pending_flat = None       # decoder-gradient blob waiting to ride with the packed grid rows
        self._pending_publish = None    # ... and the renderer's callback that publishes it as Parameter.grad afterwards
        self.last_exchange_floats = 0   # size of the most recent gradient exchange (diagnostics / bench)

    def __getattr__(self, name):
        return getattr(self.renderer, name)

    def set_voxel_masks(self, masks):
        """``masks``: dict grid key -> bool/uint8 [Z,Y,X] voxel mask (FrustumSelector.voxel_mask, or the reference's
        ``get_mask_from_c2w(...)`` after its ``permute(2,1,0)``), None / missing key = exchange that grid densely.
        Must be called with identical masks on every rank; ``set_voxel_masks(None)`` returns to dense exchange."""
        self._rows = {}
        for k, m in (masks or {}).items():
            if m is not None:
                self._rows[k] = torch.as_tensor(m).reshape(-1).ne(0).nonzero().squeeze(1)

    def _all_reduce(self, t):
        dist.all_reduce(t, op=dist.ReduceOp.SUM, group=self.group)

    def _reduce_flat(self, d_grids, gflat: Optional[torch.Tensor], publish=None) -> bool:
        """Renderer hook, called right after the backward kernel is enqueued, before the grid gradients are returned to
        autograd (so before ``_SumGrads.backward`` of the same call).  Returns True when the decoder-gradient blob was
        parked to ride with the packed voxel rows: ``publish`` (which turns the blob into ``Parameter.grad``s) is then
        called by ``_reduce_grid_grads`` / ``_flush_pending`` AFTER the exchange -- publishing earlier would let an
        accumulation into already existing ``.grad`` tensors read rank-local values."""
        if gflat is None:
            return False
        if d_grids and self._rows:
            self._flush_pending()
            self._pending_flat, self._pending_publish = gflat, publish
            return True
        self._all_reduce(gflat)
        return False

    def _flush_pending(self):
        if self._pending_flat is not None:
            self._all_reduce(self._pending_flat)
            self._publish_pending()

    def _publish_pending(self):
        publish, self._pending_flat, self._pending_publish = self._pending_publish, None, None
        if publish is not None:
            publish()

    def _reduce_grid_grads(self, keys, gs):
        out, packed, floats = [], [], 0
        for k, g in zip(keys, gs):
            if g is None:
                out.append(None)
                continue
            g, v, axis = _voxel_rows(g)
            out.append(g)
            rows = self._rows.get(k)
            if rows is None:
                self._all_reduce(v)
                floats += v.numel()
            else:
                if rows.device != v.device:
                    rows = self._rows[k] = rows.to(v.device)
                packed.append((v, axis, rows))
        flat = self._pending_flat
        if packed:
            sizes = [rows.numel() * v.shape[1 - axis] for v, axis, rows in packed]
            total = sum(sizes) + (flat.numel() if flat is not None else 0)
            buf = torch.empty((total,), dtype=packed[0][0].dtype, device=packed[0][0].device)
            off, pieces = 0, []
            for (v, axis, rows), n in zip(packed, sizes):
                piece = buf[off:off + n].view((rows.numel(), v.shape[1]) if axis == 0 else (v.shape[0], rows.numel()))
                torch.index_select(v, axis, rows, out=piece)
                pieces.append(piece)
                off += n
            if flat is not None:
                buf[off:].copy_(flat)
            self._all_reduce(buf)
            for (v, axis, rows), piece in zip(packed, pieces):
                v.index_copy_(axis, rows, piece)
            if flat is not None:
                flat.copy_(buf[off:])
            floats += total
        elif flat is not None:
            self._all_reduce(flat)
            floats += flat.numel()
        self._publish_pending()                 # the decoder blob is reduced now: expose it as Parameter.grad
        self.last_exchange_floats = floats
        return out

    def render_batch_ray(self, c, decoders, rays_d, rays_o, device, stage, gt_depth=None, gt_max=None):
        world, rank = dist.get_world_size(self.group), dist.get_rank(self.group)
        n = rays_o.shape[0]
        if n < world:            # fewer rays than ranks: every rank renders the whole (tiny) batch itself -- identical results
            return self.renderer.render_batch_ray(c, decoders, rays_d, rays_o, device, stage, gt_depth=gt_depth, **(
                {"gt_max": gt_max} if gt_max is not None else {}))     # everywhere, no empty shard, no collective to mismatch
        sizes = [shard_range(n, world, r)[1] - shard_range(n, world, r)[0] for r in range(world)]
        lo, hi = shard_range(n, world, rank)
        if stage == "coarse":
            gt_depth = None
        self._flush_pending()
        o_s = _ShardRows.apply(rays_o, lo, hi, sizes, self.group)
        d_s = _ShardRows.apply(rays_d, lo, hi, sizes, self.group)
        gt_s = None
        c_s = dict(c)
        keys = tuple(k for k, v in c.items() if torch.is_grad_enabled() and v.requires_grad)
        if keys:
            c_s.update(zip(keys, _SumGrads.apply(self, keys, *[c[k] for k in keys])))
        self.renderer._gt_max = None
        if gt_depth is not None:
            gt_depth = gt_depth.reshape(-1)
            self.renderer._gt_max = (gt_max.detach().to(torch.float32).reshape(1) if gt_max is not None else
                                     torch.max(gt_depth.detach().to(torch.float32)).reshape(1))   # batch-global, before slicing
            gt_s = gt_depth[lo:hi]
        self.renderer._reduce_hook = self._reduce_flat
        try:
            depth, unc, col = self.renderer.render_batch_ray(c_s, decoders, d_s, o_s, device, stage, gt_depth=gt_s)
        finally:
            self.renderer._gt_max = None
            self.renderer._reduce_hook = None
        return _GatherOutputs.apply(depth, unc, col, lo, hi, sizes, self.group)


class ShardedMapping:
    """The mapping iteration sharded over the ranks of a process group (one process per GPU), built on ``mapping_loss``:

    * every rank draws ITS OWN ``pixs_per_image`` pixels per keyframe from a generator seeded per rank (independent draws even
      when every process seeds torch identically, as the reference does: the union over ranks is the iteration's batch),
      renders them and forms its partial loss -- nothing is gathered, the loss is a sum over rays;
    * the bounding-box pre-filter's kept rays' maximum depth is a scalar of the WHOLE batch (Renderer.py:109,144) and must agree
      on every rank: with the in-kernel pixel draw (default) every rank's window kernel re-draws the other ranks' pixels and takes
      the maximum over the union itself -- no collective (round 6); with explicit indices / torch draws / ``peer_draw=False`` it
      is one MAX all-reduce of a single float between the sampling kernel and the render;
    * ONE packed SUM all-reduce per iteration carries everything the backward produced: the frustum-selected voxel rows of
      every grid gradient (``set_voxel_masks``; dense grid gradients without masks), the flat decoder-gradient blobs, the pose
      gradients (local BA) and the partial loss.  Gather and scatter around it are one kernel each (``nsr_pack_rows``).

    The grids, decoders and poses are replicated; after ``loss.backward()`` every rank holds the full-batch gradients (inside
    the voxel masks for the grids), so the replicated optimiser steps stay identical."""

    def __init__(self, renderer, group=None, seed: int = 0, peer_draw: bool = True, split_exchange: bool = False):
        self.renderer, self.group = renderer, group
        self.split_exchange = bool(split_exchange)   # see finish_exchange(): pack inside backward(), collective + scatter behind it
        self._deferred, self._leafs = None, None
        self.sum_collectives = 0             # packed SUM all-reduces issued so far
        # round 6: with the in-kernel pixel draw every rank can repeat every other rank's draw (philox keyed by the rank's seed, the
        # call counter advances in lock step), so the batch-global depth cap needs NO collective: the window kernel re-draws the
        # peers' pixels and takes the maximum over the union itself (nsr_get_samples_window_sharded).  False, explicit `indices`,
        # or more than 16 ranks: the 4-byte MAX all-reduce between the sampling and the render, as before.
        self.peer_draw = bool(peer_draw)
        self.max_collectives = 0             # MAX all-reduces issued so far (diagnostics / tests)
        self._rows = {}
        self._pending = None
        self.last_exchange_floats = 0
        self.last_total_loss = None          # 1-element fp32 tensor: the all-rank loss of the last backward (logging)
        # The reference seeds every process identically (setup_seed, run.py), so the global generators of all ranks would
        # draw the SAME pixels and the "union over ranks" would be one batch repeated.  The shard draws therefore come from
        # a generator of this object, seeded per rank.
        self.seed = int(seed)
        self._gen = None
        self._draw_state = None

    def generator(self, dev) -> torch.Generator:
        """This rank's pixel generator.  A hipGraph that captures ``mapping_loss`` must know it:
        ``graph.register_generator_state(sharder.generator(device))`` before the capture (every replay then draws afresh)."""
        dev = torch.device(dev)
        if self._gen is None or self._gen.device != dev:
            self._gen = torch.Generator(device=dev)
            self._gen.manual_seed(self.seed * 1000003 + 7919 * (dist.get_rank(self.group) + 1))
        return self._gen

    def _draw(self, n_total: int, n_pixels_crop: int, dev) -> torch.Tensor:
        return torch.randint(n_pixels_crop, (n_total,), device=dev, generator=self.generator(dev))

    def draw_state(self, dev) -> torch.Tensor:
        """This rank's state of the in-kernel pixel draw (mapping.PIXEL_DRAW = "kernel"): seeded per rank like ``generator``,
        advanced by the window kernel itself -- nothing to register with a capturing graph."""
        dev = torch.device(dev)
        if self._draw_state is None:
            self._draw_state = {}                    # one state tensor per device, kept for the life of the object: a graph
        st = self._draw_state.get(dev)               # captured on a device has that tensor's address baked in
        if st is None:
            st = self._draw_state[dev] = torch.tensor([self.rank_seed(dist.get_rank(self.group)), 0, 0, 0], dtype=torch.int64, device=dev)
        return st

    def rank_seed(self, rank: int) -> int:
        """seed of rank ``rank``'s in-kernel draw state (every rank can compute every other rank's)"""
        return (self.seed * 1000003 + 7919 * (int(rank) + 1)) & ((1 << 63) - 1)

    def peer_seeds(self):
        world, me = dist.get_world_size(self.group), dist.get_rank(self.group)
        if not self.peer_draw or world < 2 or world > 16:
            return None
        return [self.rank_seed(r) for r in range(world) if r != me]

    def set_voxel_masks(self, masks):
        """dict grid key -> bool/uint8 [Z,Y,X] voxel mask (``FrustumSelector.voxel_mask``), identical on every rank;
        None returns to the dense exchange."""
        self._rows = {}
        for k, m in (masks or {}).items():
            if m is not None:
                self._rows[k] = torch.as_tensor(m).reshape(-1).ne(0).nonzero().squeeze(1).contiguous()

    def reduce_max(self, kmax: torch.Tensor):
        self.max_collectives += 1
        dist.all_reduce(kmax, op=dist.ReduceOp.MAX, group=self.group)

    def collect(self, d_grids, gflat, publish) -> bool:          # renderer.render_backward hook: park, exchange() follows
        self._pending = (gflat, publish)
        return True

    def exchange(self, named_grads, pose_base, loss32):
        """``named_grads``: [(grid key, channels-last gradient)] of this backward; ``pose_base``: [K,4,4] pose gradients or
        None; ``loss32``: 1-element fp32 tensor holding this rank's partial loss.  All are summed over the ranks in place --
        unless ``split_exchange`` is set (and the iteration qualifies, see ``finish_exchange``): then this call only PACKS, and the
        caller runs the collective and the scatter after ``backward()`` has returned."""
        from . import _capi
        from .common import _stream
        gflat, publish = self._pending if self._pending is not None else (None, None)
        self._pending = None
        self._deferred = None                 # (a record of an earlier split iteration -- e.g. one a graph segment keeps -- is not THIS backward's)
        lib = _capi.get_lib()
        dev = loss32.device
        leafs, self._leafs = self._leafs, None
        split = self.split_exchange and pose_base is None and leafs is not None and \
            all(k in leafs and leafs[k].grad is None for k, _ in named_grads)
        keys, rows_of, dense, n_rows_total = [], [], [], 0
        for k, g in named_grads:
            rows = self._rows.get(k)
            if rows is None:
                dense.append((k, g))
                continue
            if rows.device != dev:
                rows = self._rows[k] = rows.to(dev)
            if not g.is_contiguous(memory_format=torch.channels_last_3d):
                raise RuntimeError("ShardedMapping: grid gradients must be channels-last (what render backward produces)")
            keys.append(k)
            rows_of.append((g, rows))
            n_rows_total += rows.numel()
        spans = [t for t in (gflat, None if pose_base is None else pose_base.view(-1), loss32) if t is not None]
        total = n_rows_total * 32 + sum(t.numel() for t in spans)
        buf = torch.empty((total,), dtype=torch.float32, device=dev)
        rec = {"keys": keys, "rows": [r for _, r in rows_of], "spans": spans, "buf": buf, "dense": dense, "publish": publish,
               "leafs": leafs if split else None, "loss32": loss32, "dev": dev}
        with _capi.on_device(dev):                               # two launches + a collective on the gradients' device
            self._pack_rows(lib, rec, [g for g, _ in rows_of], 0, _stream(dev))
            if split:
                # the kernels up to here and the scatter behind the collective can be two captured graph segments with the
                # collective eager between them (bench.py NSR_DIST_GRAPH=segments): finish_exchange() = reduce_deferred() + scatter_deferred()
                self._deferred = rec
                return
            self._reduce(rec, [g for _, g in dense])
            self._pack_rows(lib, rec, [g for g, _ in rows_of], 1, _stream(dev))
        self._finish(rec)

    @staticmethod
    def _pack_rows(lib, rec, grid_tensors, mode, stream):
        from . import _capi
        ng = len(grid_tensors)
        rows_arr = (_capi.NsrRows * max(1, ng))()
        for i, (g, rows) in enumerate(zip(grid_tensors, rec["rows"])):
            rows_arr[i].grid, rows_arr[i].rows, rows_arr[i].n_rows = g.data_ptr(), rows.data_ptr(), rows.numel()
        span_arr = (_capi.NsrSpan * len(rec["spans"]))()
        for i, t in enumerate(rec["spans"]):
            span_arr[i].ptr, span_arr[i].n = t.data_ptr(), t.numel()
        lib.check(lib.nsr_pack_rows(rows_arr, ng, span_arr, len(rec["spans"]), rec["buf"].data_ptr(), mode, stream), "nsr_pack_rows")

    def _reduce(self, rec, dense_tensors):
        self.sum_collectives += 1
        dist.all_reduce(rec["buf"], op=dist.ReduceOp.SUM, group=self.group)
        floats = rec["buf"].numel()
        for g in dense_tensors:                                  # grids without a mask: dense, in place (one more collective each)
            _, v, _ = _voxel_rows(g)
            dist.all_reduce(v, op=dist.ReduceOp.SUM, group=self.group)
            floats += v.numel()
        self.last_exchange_floats = floats

    def _finish(self, rec):
        self.last_total_loss = rec["loss32"]
        if rec["publish"] is not None:
            rec["publish"]()

    # -- split exchange -------------------------------------------------------------------------------------------------------
    # `split_exchange = True`: backward() returns with the iteration's gradients PACKED but not yet summed; the caller then runs
    #     rec = sharder.deferred(); sharder.reduce_deferred(rec); sharder.scatter_deferred(rec)      (= finish_exchange())
    # Everything before and everything behind the collective is kernels only, so both halves replay from hipGraphs while the
    # collective itself stays an ordinary eager call -- the middle path between "collectives captured in the graph" and a
    # fully eager iteration (a capture that fails with N > 1 RCCL ranks then costs one graph boundary, not the 4x of eager
    # launches).  The scatter writes into the `.grad` tensors autograd has produced by then (it may have cloned the backward's
    # buffers), so an iteration qualifies only if those were None before (no accumulation) and no pose is optimised (local BA:
    # the blocking exchange runs instead, inside backward(), as without the flag).
    def deferred(self):
        return self._deferred

    def _grad_targets(self, rec, keys):
        out = []
        for k in keys:
            g = rec["leafs"][k].grad
            if g is None or not g.is_contiguous(memory_format=torch.channels_last_3d):
                raise RuntimeError(f"ShardedMapping.split_exchange: {k}.grad is missing or not channels-last after backward()")
            out.append(g)
        return out

    def reduce_deferred(self, rec=None):
        rec = rec if rec is not None else self._deferred
        if rec is not None:
            self._reduce(rec, self._grad_targets(rec, [k for k, _ in rec["dense"]]))

    def scatter_deferred(self, rec=None):
        from . import _capi
        from .common import _stream
        rec = rec if rec is not None else self._deferred
        if rec is None:
            return
        with _capi.on_device(rec["dev"]):
            self._pack_rows(_capi.get_lib(), rec, self._grad_targets(rec, rec["keys"]), 1, _stream(rec["dev"]))
        self._finish(rec)

    def finish_exchange(self):
        """after ``backward()`` of a ``split_exchange`` iteration: the collective + the scatter (no-op if backward() ran the blocking exchange)"""
        rec, self._deferred = self._deferred, None
        if rec is not None:
            self.reduce_deferred(rec)
            self.scatter_deferred(rec)

    def mapping_loss(self, c, decoders, frames, pixs_per_image, stage, w_color: float = 0.2, indices=None, out=None):
        """This rank's share of one mapping iteration (``pixs_per_image`` pixels per frame HERE); returns the rank's partial
        loss -- ``backward()`` leaves the all-rank gradients on every rank (and the all-rank loss in ``last_total_loss``)."""
        from . import mapping
        state, peers = None, None
        if indices is None:                  # this rank's own draw (see __init__): never the global generator / the device's state
            dev = frames[0][1].device
            if mapping.PIXEL_DRAW == "kernel" and dev.type == "cuda":
                state = self.draw_state(dev)
                peers = self.peer_seeds()
            else:
                indices = self._draw(len(frames) * int(pixs_per_image), self.renderer.H * self.renderer.W, dev)
        self._leafs = {k: v for k, v in c.items() if torch.is_tensor(v)} if self.split_exchange else None
        return mapping.mapping_loss(self.renderer, c, decoders, frames, pixs_per_image, stage, w_color=w_color, indices=indices,
                                    coarse_mapper=(stage == "coarse"), out=out, sharder=self, draw_state=state, peer_seeds=peers)
