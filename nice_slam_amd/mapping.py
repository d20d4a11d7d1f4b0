"""The mapper's iteration as three launches (src/Mapper.py:437-503), and the tracker's (src/Tracker.py:86-125).

The reference's mapping iteration is: ``get_samples`` per keyframe of the window + ``torch.cat`` (Mapper.py:437-468), the
bounding-box pre-filter (:471-481), ``render_batch_ray`` (:482), the L1 losses (:487-493) and ``loss.backward()`` (:503) --
about forty small ATen launches around two big kernels.  Here:

* ``get_samples_window``  : all frames of the window in ONE kernel (rays, depth / colour gathers, the pre-filter as a byte
  mask, the kept rays' maximum depth), differentiable w.r.t. the poses (local BA) through ``nsr_pose_grad``;
* ``mapping_loss``        : sampling + render + loss as one autograd node.  The forward kernel accumulates the loss and
  writes its derivative w.r.t. every ray's outputs, so the loss and its backward add no launch; the backward is the one
  render-backward launch (+ the partial-sum kernel) writing dense grid gradients, the decoder-gradient blob and -- with BA --
  the pose gradients.

Same numbers as the unfused path (tests/test_hip_mapping.py); per iteration: the window kernel (which draws the pixels and
zero-fills the iteration's gradient buffer beside its sampling blocks), render forward, render backward, the partial sum.
"""
from __future__ import annotations

import ctypes as C
import os
from typing import List, Optional, Sequence, Tuple

import torch

from . import _capi
from .common import _as_f32c, _require_cuda, _stream
from .layout import param_count, stage_slots
from .renderer import _fill_common, _gate, _prep_grids, render_backward


class WindowSamples:
    """Rays of one mapping iteration: the concatenation over the window's frames (frame-major, like ``torch.cat``)."""
    __slots__ = ("rays_o", "rays_d", "gt_depth", "gt_color", "keep", "kept_max", "indices", "geom")


def _frames_block(c2ws, depths, colors, dev):
    K = len(depths)
    fr = (_capi.NsrFrame * K)()
    hold = []
    for k in range(K):
        d = _as_f32c(depths[k], dev)
        col = _as_f32c(colors[k], dev)
        p = _as_f32c(c2ws[k].detach(), dev)
        hold += [d, col, p]
        fr[k].depth, fr[k].color, fr[k].c2w, fr[k].c2w_stride = d.data_ptr(), col.data_ptr(), p.data_ptr(), p.stride(0)
    return fr, hold


def _bound_arrays(bound):
    lo = (C.c_double * 3)(*[float(bound[a][0]) for a in range(3)])
    hi = (C.c_double * 3)(*[float(bound[a][1]) for a in range(3)])
    return lo, hi


# Where the pixel indices of a window come from when the caller passes none (src/common.py:99: `torch.randint(h * w, (n,))`
# per keyframe): "kernel" -- drawn inside the window kernel (philox4x32-10 keyed by torch's seed, nsr_get_samples_window_draw):
# no launch of its own and, under graph capture, none of the fills with which torch keeps a captured generator's offset -- or
# "torch" -- one `torch.randint` call for the window.  Same distribution either way; neither is the reference's stream (it
# draws once per keyframe).  `get_samples` (the drop-in of common.py:91-106) always uses torch.randint like the reference.
PIXEL_DRAW = os.environ.get("NSR_PIXEL_DRAW", "kernel")
# The fused iterations' one zero fill (loss accumulator, kept max, every gradient buffer of the backward) inside the window
# kernel's launch (nsr_get_samples_window_fused) instead of a `torch.zeros` launch in front of it; "0": the separate fill (A/B).
FUSED_FILL = os.environ.get("NSR_FUSED_FILL", "1") != "0"
_DRAW_STATE = {}


def _capturing() -> bool:
    return torch.cuda.is_available() and torch.cuda.is_current_stream_capturing()


def _set_state(st: torch.Tensor, seed: int):
    """(Re)start a draw state IN PLACE: a hipGraph captured earlier has the tensor's address baked into its window kernel, so
    the tensor must live -- at that address -- as long as the process; the replays then see the new seed."""
    st.copy_(torch.tensor([int(seed) & ((1 << 63) - 1), 0, 0, 0], dtype=torch.int64), non_blocking=False)


def _draw_state(dev) -> torch.Tensor:
    """Device-side state of the in-kernel draw: [seed, calls so far, internal]; ONE tensor per device for the life of the process.
    It follows torch.manual_seed (a new seed restarts the sequence, in place) and is advanced by the kernel itself, so replays
    of a captured graph draw afresh.  Never re-seeded while a stream is capturing (the copy would become part of the graph and
    reset the sequence on every replay): a capture keeps the state it finds."""
    key = (dev.type, dev.index if dev.index is not None else torch.cuda.current_device())
    seed = torch.initial_seed() & ((1 << 63) - 1)
    st = _DRAW_STATE.get(key)
    if st is None:
        if _capturing():
            raise RuntimeError("nice_slam_amd: the first in-kernel pixel draw on a device cannot happen under graph capture "
                               "(run one eager iteration first)")
        st = [seed, torch.tensor([seed, 0, 0, 0], dtype=torch.int64, device=dev)]
        _DRAW_STATE[key] = st
    elif st[0] != seed and not _capturing():
        _set_state(st[1], seed)
        st[0] = seed
    return st[1]


def seed_pixel_draws(seed: int, device=None):
    """Restart the in-kernel pixel draw of ``device`` (default: every device that has drawn so far) from ``seed``, in place:
    graphs captured before keep working and draw the new sequence.  (A NEW ``torch.manual_seed`` value restarts it too;
    re-seeding torch with the SAME value cannot be seen from here.)  Not allowed while a stream is capturing."""
    if _capturing():
        raise RuntimeError("nice_slam_amd.seed_pixel_draws: not under graph capture")
    if device is None:
        keys = list(_DRAW_STATE)
    else:
        d = torch.device(device)
        keys = [(d.type, d.index if d.index is not None else torch.cuda.current_device())]
    for key in keys:
        st = _DRAW_STATE.get(key)
        if st is None:
            st = _DRAW_STATE[key] = [torch.initial_seed() & ((1 << 63) - 1), torch.zeros(4, dtype=torch.int64, device=torch.device(*key))]
        _set_state(st[1], seed)


DEBUG_PTRS = {} if os.environ.get("NSR_DEBUG_PTRS") == "1" else None


def _launch_window(indices, K, n, crop, intr, frames, bound6, sbuf, keep, kmax_ptr, dev, fused=None):
    """``fused``: None, or (header tensor [4] fp32, zero span tensor) of a fused iteration -- the launch then also zero-fills the
    span and writes the header {loss = 0 (fp64), kept max, 0} itself (nsr_get_samples_window_fused): no fill launch before it."""
    lib = _capi.get_lib()
    H0, H1, W0, W1, W_full = crop
    fx, fy, cx, cy = intr
    N = K * n
    if fused is not None:
        hdr, zero = fused
        draw = getattr(indices, "_nsr_draw", False)
        indices._nsr_draw = False
        state = getattr(indices, "_nsr_state", None)
        if state is None:
            state = _draw_state(dev)
        peers = getattr(indices, "_nsr_peers", None)
        if peers and draw:
            # one rank of a ray-sharded iteration: the other ranks' draws are repeated for the batch-global depth cap (no collective)
            seeds = (C.c_uint64 * len(peers))(*[int(v) for v in peers])
            lib.check(lib.nsr_get_samples_window_sharded(indices.data_ptr(), state.data_ptr(), seeds, len(peers), K, n, H0, H1, W0, W1, W_full,
                                                         fx, fy, cx, cy, frames, sbuf.data_ptr(), sbuf.data_ptr() + 12 * N, sbuf.data_ptr() + 24 * N,
                                                         sbuf.data_ptr() + 28 * N, bound6[0], bound6[1], keep.data_ptr(), hdr.data_ptr(),
                                                         zero.data_ptr() if zero.numel() else None, zero.numel(), _stream(dev)),
                      "nsr_get_samples_window_sharded")
            return
        lib.check(lib.nsr_get_samples_window_fused(None if draw else indices.data_ptr(), indices.data_ptr() if draw else None, state.data_ptr(),
                                                   K, n, H0, H1, W0, W1, W_full, fx, fy, cx, cy, frames, sbuf.data_ptr(),
                                                   sbuf.data_ptr() + 12 * N, sbuf.data_ptr() + 24 * N, sbuf.data_ptr() + 28 * N, bound6[0],
                                                   bound6[1], keep.data_ptr(), hdr.data_ptr(), zero.data_ptr() if zero.numel() else None,
                                                   zero.numel(), _stream(dev)), "nsr_get_samples_window_fused")
        return
    if getattr(indices, "_nsr_draw", False):               # drawn by the kernel, written to `indices` for the backward / the caller
        indices._nsr_draw = False
        state = getattr(indices, "_nsr_state", None)
        if state is None:
            state = _draw_state(dev)
        lib.check(lib.nsr_get_samples_window_draw(indices.data_ptr(), state.data_ptr(), K, n, H0, H1, W0, W1, W_full,
                                                  fx, fy, cx, cy, frames, sbuf.data_ptr(), sbuf.data_ptr() + 12 * N,
                                                  sbuf.data_ptr() + 24 * N, sbuf.data_ptr() + 28 * N, bound6[0], bound6[1],
                                                  keep.data_ptr(), kmax_ptr, _stream(dev)), "nsr_get_samples_window_draw")
        return
    lib.check(lib.nsr_get_samples_window(indices.data_ptr(), K, n, H0, H1, W0, W1, W_full, fx, fy, cx, cy, frames,
                                         sbuf.data_ptr(), sbuf.data_ptr() + 12 * N, sbuf.data_ptr() + 24 * N, sbuf.data_ptr() + 28 * N,
                                         bound6[0], bound6[1], keep.data_ptr(), kmax_ptr, _stream(dev)), "nsr_get_samples_window")


class _WindowFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, meta, *c2ws):
        with _capi.on_device(meta[8]):
            return _WindowFn._forward_impl(ctx, meta, *c2ws)

    @staticmethod
    def _forward_impl(ctx, meta, *c2ws):
        indices, K, n, crop, intr, depths, colors, bound, dev = meta
        N = K * n
        frames, hold = _frames_block(c2ws, depths, colors, dev)
        sbuf = torch.empty((10 * N + (N + 3) // 4,), dtype=torch.float32, device=dev)      # o | d | depth | colour | keep bytes
        keep = sbuf[10 * N:].view(torch.uint8)[:N]
        kmax = torch.zeros((1,), dtype=torch.float32, device=dev)
        _launch_window(indices, K, n, crop, intr, frames, _bound_arrays(bound), sbuf, keep, kmax.data_ptr(), dev)
        ctx.meta = (indices, K, n, crop, intr, [tuple(c.shape) for c in c2ws], [c.dtype for c in c2ws], [c.device for c in c2ws])
        outs = (sbuf[:3 * N].view(N, 3), sbuf[3 * N:6 * N].view(N, 3), sbuf[6 * N:7 * N], sbuf[7 * N:10 * N].view(N, 3), keep, kmax)
        ctx.mark_non_differentiable(*outs[2:])
        return outs

    @staticmethod
    @torch.autograd.function.once_differentiable
    def backward(ctx, g_o, g_d, *_):
        indices, K, n, crop, intr, shapes, dtypes, devs = ctx.meta
        grads, _ = pose_grads(indices, K, n, crop, intr, g_o.contiguous(), g_d.contiguous(), shapes)
        return (None, *[g.to(device=dv, dtype=dt) for g, dv, dt in zip(grads, devs, dtypes)])


def pose_grads(indices, K, n, crop, intr, g_o, g_d, shapes, out=None) -> List[torch.Tensor]:
    """d c2w[k] (shape of the pose, rows 0..2 filled) from the gradients of the window's rays: one launch.  ``out``: an already
    zero-filled [K, 4, 4] fp32 tensor (the fused iteration's one zero-filled buffer has room for it), else allocated here."""
    lib = _capi.get_lib()
    dev = g_o.device
    H0, H1, W0, W1, _ = crop
    fx, fy, cx, cy = intr
    if out is None:
        out = torch.zeros((K, 4, 4), dtype=torch.float32, device=dev)
    lib.check(lib.nsr_pose_grad(indices.data_ptr(), K, n, H0, H1, W0, W1, fx, fy, cx, cy, g_o.data_ptr(), g_d.data_ptr(),
                                out.data_ptr(), 16, _stream(dev)), "nsr_pose_grad")
    return [out[k, :shp[0], :] for k, shp in enumerate(shapes)], out


def _window_meta(H0, H1, W0, W1, n, W, fx, fy, cx, cy, c2ws, depths, colors, bound, device, indices, draw_state=None, peer_seeds=None):
    K = len(depths)
    dev = torch.device(device)
    if indices is None and (PIXEL_DRAW == "kernel" or draw_state is not None) and dev.type == "cuda":
        indices = torch.empty((K * n,), dtype=torch.int64, device=dev)            # filled by the window kernel (see PIXEL_DRAW)
        indices._nsr_draw = True
        indices._nsr_state = draw_state                                           # None: the device's default state
        indices._nsr_peers = list(peer_seeds) if (peer_seeds and FUSED_FILL) else None   # (ShardedMapping: the other ranks' seeds)
    elif indices is None:
        indices = torch.randint((H1 - H0) * (W1 - W0), (K * n,), device=dev)      # one draw for the window (common.py:99 per frame)
    else:
        indices = indices.to(dev).reshape(-1).contiguous().view(-1)             # (a tensor object of our own: it carries the state below)
        indices._nsr_state = draw_state
    c2ws = [c if isinstance(c, torch.Tensor) else torch.as_tensor(c) for c in c2ws]
    crop = (int(H0), int(H1), int(W0), int(W1), int(W))
    intr = (float(fx), float(fy), float(cx), float(cy))
    return (indices, K, int(n), crop, intr, list(depths), list(colors), bound, dev), c2ws


def get_samples_window(H0, H1, W0, W1, n, H, W, fx, fy, cx, cy, c2ws: Sequence[torch.Tensor], depths: Sequence[torch.Tensor],
                       colors: Sequence[torch.Tensor], bound, device, indices: Optional[torch.Tensor] = None) -> WindowSamples:
    """``n`` pixels from each of the K frames (pose ``c2ws[k]``: 3x4 or 4x4, may require grad; ``depths[k]`` [H,W],
    ``colors[k]`` [H,W,3] on the device), concatenated in frame order -- the sampling loop of Mapper.py:437-468 -- plus the
    bounding-box pre-filter of :471-481 as ``keep`` (bool per ray) and ``kept_max`` (1-element tensor: maximum depth over
    the kept rays, to be passed as ``render_batch_ray(..., gt_max=kept_max)``).  ``indices``: optional [K*n] flat crop
    indices (default: drawn inside the kernel, see ``PIXEL_DRAW``)."""
    meta, c2ws = _window_meta(H0, H1, W0, W1, n, W, fx, fy, cx, cy, c2ws, depths, colors, bound, device, indices)
    _require_cuda(depths[0] if depths[0].is_cuda else torch.empty(0, device=meta[-1]), "get_samples_window: frames")
    ro, rd, gd, gc, keep, kmax = _WindowFn.apply(meta, *c2ws)
    w = WindowSamples()
    w.rays_o, w.rays_d, w.gt_depth, w.gt_color, w.keep, w.kept_max, w.indices, w.geom = ro, rd, gd, gc, keep.bool(), kmax, meta[0], meta[1:5]
    return w


# --------------------------------------------------------------------------------------------------------------------
# sampling + render + mapping loss as one autograd node
# --------------------------------------------------------------------------------------------------------------------
class _MappingLossFn(torch.autograd.Function):
    """inputs: K poses, one grid per decoder of the stage, one gate per decoder (see renderer._RenderFn)."""

    @staticmethod
    def forward(ctx, meta, *tensors):
        with _capi.on_device(meta[3][8]):
            return _MappingLossFn._forward_impl(ctx, meta, *tensors)

    @staticmethod
    def _forward_impl(ctx, meta, *tensors):
        renderer, decoders, stage, wmeta, w_color, sharder, out, track = meta      # track: None | (handle_dynamic, use_color)
        indices, K, n, crop, intr, depths, colors, bound, dev = wmeta
        lib = _capi.get_lib()
        slots = stage_slots(stage)
        c2ws = tensors[:K]
        grids = dict(zip(slots, tensors[K:K + len(slots)]))
        stream = _stream(dev)
        N = K * n
        guided = stage != "coarse"
        S = renderer.N_samples + (renderer.N_surface if guided else 0)
        need_pose = any(ctx.needs_input_grad[1:1 + K])
        need_grid = ctx.needs_input_grad[1 + K:1 + K + len(slots)]
        need_par = ctx.needs_input_grad[1 + K + len(slots):1 + K + 2 * len(slots)]
        need_bwd = need_pose or any(need_grid) or any(need_par)
        # ONE zero-filled buffer: loss (fp64) | kept_max | pad | every gradient of the backward (renderer.render_backward)
        n_grad = 0
        if need_bwd:
            n_grad = sum(grids[s].numel() for s, nd in zip(slots, need_grid) if nd) + (6 * N if need_pose else 0) + \
                sum(param_count(s) for s, nd in zip(slots, need_par) if nd)
        n_pose = 16 * K if need_pose else 0                     # d c2w of the window (pose_grads), behind the gradients
        # (round 5: not a fill launch -- the window kernel zero-fills it beside its sampling blocks and writes the header)
        fuse_fill = FUSED_FILL and N > 0
        if fuse_fill and not getattr(indices, "_nsr_draw", False) and getattr(indices, "_nsr_state", None) is None and _capturing() \
                and (dev.type, dev.index if dev.index is not None else torch.cuda.current_device()) not in _DRAW_STATE:
            # explicit indices, nothing is drawn -- but the fused launch borrows the device's draw state for its hand-off words, and that
            # tensor cannot be created under graph capture (it must outlive every graph): the separate fill + plain window launch instead
            fuse_fill = False
        Z = (torch.empty if fuse_fill else torch.zeros)((4 + n_grad + n_pose,), dtype=torch.float32, device=dev)
        loss = Z[:2].view(torch.float64)
        kmax = Z[2:3]
        frames, hold = _frames_block(c2ws, depths, colors, dev)
        # ONE allocation for everything the iteration writes besides the gradients: forward results (fp64 part, then fp32
        # part) and, behind them, the sampled rays (measured: no faster than two allocations, one launch-side call fewer)
        n64 = 3 * N + N * S
        nf32 = N * S * 4 + 6 * N
        n_s = 10 * N + (N + 3) // 4
        FS = torch.empty((n64 + (nf32 + 1) // 2 + (n_s + 1) // 2,), dtype=torch.float64, device=dev)
        F = FS[:n64 + (nf32 + 1) // 2]
        sbuf = FS[n64 + (nf32 + 1) // 2:].view(torch.float32)[:n_s]
        keep = sbuf[10 * N:].view(torch.uint8)[:N]
        _launch_window(indices, K, n, crop, intr, frames, _bound_arrays(bound), sbuf, keep, kmax.data_ptr(), dev,
                       fused=(Z[:4], Z[4:]) if fuse_fill else None)
        if sharder is not None and not (fuse_fill and getattr(indices, "_nsr_peers", None)):
            # the depth cap is a scalar of the WHOLE batch (Renderer.py:109,144): one 4-byte MAX all-reduce -- unless the window kernel
            # has just re-drawn the other ranks' pixels itself and its header already holds the maximum over the union (peer seeds)
            sharder.reduce_max(kmax)
        rays_o, rays_d = sbuf[:3 * N].view(N, 3), sbuf[3 * N:6 * N].view(N, 3)
        gt_depth, gt_color = sbuf[6 * N:7 * N], sbuf[7 * N:10 * N].view(N, 3)
        # forward results: depth | var | dl_depth | zvals (fp64), then raw | rgb | dl_rgb (fp32)
        f32 = F[n64:].view(torch.float32)
        depth, var, dl_depth, zvals = F[:N], F[N:2 * N], F[2 * N:3 * N], F[3 * N:n64].view(N, S)
        raw, rgb, dl_rgb = f32[:N * S * 4].view(N, S, 4), f32[N * S * 4:N * S * 4 + 3 * N].view(N, 3), f32[N * S * 4 + 3 * N:N * S * 4 + 6 * N].view(N, 3)
        flats = {s: decoders.sub(s).flat_params() for s in slots}
        packed = {s: decoders.sub(s).packed_params(lib, stream) for s in slots}
        a = _capi.NsrRenderArgs.from_buffer_copy(renderer._arg_template)
        a.n_samples, a.n_surface, a.n_rays = renderer.N_samples, renderer.N_surface, N
        a.rays_o, a.rays_d = rays_o.data_ptr(), rays_d.data_ptr()
        a.gt_depth, a.gt_max = gt_depth.data_ptr(), kmax.data_ptr()
        _fill_common(a, stage, renderer.bound, decoders, grids, packed, flats)
        if not guided:
            a.n_surface = 0
        a.depth, a.var, a.rgb, a.raw, a.zvals = depth.data_ptr(), var.data_ptr(), rgb.data_ptr(), raw.data_ptr(), zvals.data_ptr()
        if track is None:                                       # the mapper's L1 loss is accumulated by the forward kernel itself
            a.gt_color, a.loss, a.w_color = gt_color.data_ptr(), loss.data_ptr(), float(w_color)
            a.dl_depth, a.dl_rgb = dl_depth.data_ptr(), dl_rgb.data_ptr()
        # the rays the bounding-box pre-filter rejects: masked out of the loss, and -- like the reference, which removes them from
        # the batch (Mapper.py:471-481, Tracker.py:95-104) -- not rendered at all unless Renderer.skip_masked_rays is off
        a.keep = keep.data_ptr()
        a.skip_masked = 1 if (renderer.skip_masked_rays and need_bwd) else 0
        if need_bwd and renderer.profile_fwd_events is not None:
            a.ev_pass_start, a.ev_pass_stop = renderer.profile_fwd_events(stage)
        acts = renderer._attach_acts(a, stage, N, S, dev, masks_only=[not g_ for g_ in need_par]) if need_bwd else None
        if DEBUG_PTRS is not None:                              # measurement (bench.py NSR_DEBUG_PTRS=1): where the iteration's buffers landed
            DEBUG_PTRS[stage] = {"Z": Z.data_ptr(), "Z_bytes": 4 * Z.numel(), "FS": FS.data_ptr(), "acts": None if acts is None else acts.data_ptr(),
                                 "acts_bytes": None if acts is None else 4 * acts.numel(), "grids": {s: grids[s].data_ptr() for s in slots}}
        if need_bwd and acts is None:
            raise _capi.NsrError("nice_slam_amd: the activation buffer of a %d-ray fused iteration does not fit (Renderer."
                                 "max_saved_activation_bytes / free device memory); use smaller batches or render_batch_ray" % N)
        lib.check(lib.nsr_render_fwd(C.byref(a), stream), "nsr_render_fwd")
        if track is not None:                                   # the tracker's loss needs the batch median of the rendered outputs
            lib.check(lib.nsr_tracking_loss(N, gt_depth.data_ptr(), gt_color.data_ptr(), keep.data_ptr(), depth.data_ptr(), var.data_ptr(),
                                            rgb.data_ptr(), int(track[0]), int(track[1]), float(w_color), loss.data_ptr(),
                                            dl_depth.data_ptr(), dl_rgb.data_ptr(), stream), "nsr_tracking_loss")
        if out is not None:
            out.update(rays_o=rays_o, rays_d=rays_d, gt_depth=gt_depth, gt_color=gt_color, keep=keep, kept_max=kmax, depth=depth,
                       uncertainty=var, color=rgb, indices=indices)
        if need_bwd:
            ctx.sharder, ctx.loss32 = sharder, Z[3:4]
            ctx.from_forward = track is None            # the mapper's loss epilogue wrote dl_* AND d raw; nobody touches them in between
            ctx.pose_buf = Z[4 + n_grad:4 + n_grad + n_pose].view(K, 4, 4) if need_pose else None
            ctx.state = (a, (renderer, decoders, stage, S, None if sharder is None else sharder.collect),
                         ([kmax, F, sbuf, Z, hold, acts], rays_o, rays_d, gt_depth, grids, flats, packed, raw, depth),
                         (need_pose, need_grid, need_par), dl_depth,
                         dl_rgb if (stage == "color" and (track is None or track[1])) else None, Z[4:],
                         (indices, K, n, crop, intr, [tuple(c.shape) for c in c2ws], [c.dtype for c in c2ws], [c.device for c in c2ws]))
        return loss[0]

    @staticmethod
    @torch.autograd.function.once_differentiable
    def backward(ctx, g_loss):
        with _capi.on_device(g_loss.device):
            return _MappingLossFn._backward_impl(ctx, g_loss)

    @staticmethod
    def _backward_impl(ctx, g_loss):
        if ctx.state is None:
            raise RuntimeError("nice_slam_amd: backward through mapping_loss / tracking_loss a second time is not supported (the "
                               "saved buffers are released after the first backward)")
        a, meta, kept, (need_pose, need_grid, need_par), dl_depth, dl_rgb, zero_buf, wm = ctx.state
        # d loss / d outputs were written by the forward for an incoming gradient of 1 (loss.backward(), Mapper.py:503); whatever
        # autograd hands over (loss * w, loss / n, a GradScaler ...) multiplies them inside the backward kernel: device scalar,
        # no host sync, no extra launch
        gs = g_loss.detach().to(device=dl_depth.device, dtype=torch.float64).reshape(1)
        d_o, d_d, d_grids = render_backward(a, meta, kept, (need_pose, need_pose, need_grid, need_par), dl_depth, None, dl_rgb,
                                            zero_buf=zero_buf, grad_scale=gs, loss_grads_from_forward=ctx.from_forward)
        indices, K, n, crop, intr, shapes, dtypes, devs = wm
        g_pose = [None] * K
        gp, pose_base = None, None
        if need_pose:
            gp, pose_base = pose_grads(indices, K, n, crop, intr, d_o, d_d, shapes, out=ctx.pose_buf)
        if ctx.sharder is not None:                            # multi-GPU: ONE packed all-reduce of everything this iteration produced
            ctx.loss32.copy_(kept[0][3][:2].view(torch.float64).to(torch.float32))
            ctx.sharder.exchange([("grid_" + s_, g) for s_, g in zip(stage_slots(meta[2]), d_grids) if g is not None], pose_base, ctx.loss32)
        if need_pose:
            g_pose = [g.to(device=dv, dtype=dt) if nd else None for g, dv, dt, nd in zip(gp, devs, dtypes, ctx.needs_input_grad[1:1 + K])]
        ctx.state = None
        nslots = len(d_grids)
        return (None, *g_pose, *d_grids, *([None] * nslots))


def mapping_loss(renderer, c, decoders, frames: Sequence[Tuple[torch.Tensor, torch.Tensor, torch.Tensor]], pixs_per_image: int,
                 stage: str, w_color: float = 0.2, device=None, indices: Optional[torch.Tensor] = None, coarse_mapper: bool = False,
                 crop: Optional[Tuple[int, int, int, int]] = None, out: Optional[dict] = None, sharder=None,
                 draw_state: Optional[torch.Tensor] = None, peer_seeds=None) -> torch.Tensor:
    """One mapping iteration's loss (src/Mapper.py:437-493) as a single autograd node.

    ``frames``: ``(c2w, depth [H,W], color [H,W,3])`` per frame of the window, in the reference's order; a pose that requires
    grad gets its gradient (local BA).  Samples ``pixs_per_image`` pixels per frame, applies the bounding-box pre-filter as a
    mask, renders ``stage`` and returns ``sum_{kept, gt>0} |gt - depth| (+ w_color * sum_kept |gt_rgb - rgb|`` in the colour
    stage) as an fp64 scalar, an ordinary autograd node (an incoming gradient other than 1 scales every gradient, on the device).
    ``out`` (optional dict) receives the sampled rays, masks and rendered outputs.  ``sharder``: a
    ``nice_slam_amd.parallel.ShardedMapping`` (multi-GPU; use its ``mapping_loss`` method).
    One stream per device draw state: the fused window launch uses hand-off words of the device's draw state (also with explicit
    ``indices``), so two iterations of one process that run on DIFFERENT streams at the same time (a coarse mapper beside the mapper)
    must each pass their own ``draw_state`` tensor (4 int64 on the device: ``[seed, 0, 0, 0]``); iterations on one stream need nothing."""
    if coarse_mapper and stage != "coarse":
        raise ValueError("the coarse mapper optimises in stage 'coarse' (Mapper.py:403-404)")
    dev = torch.device(device) if device is not None else frames[0][1].device
    H0, H1, W0, W1 = crop if crop is not None else (0, renderer.H, 0, renderer.W)
    wmeta, c2ws = _window_meta(H0, H1, W0, W1, pixs_per_image, renderer.W, renderer.fx, renderer.fy, renderer.cx, renderer.cy,
                               [f[0] for f in frames], [f[1] for f in frames], [f[2] for f in frames], renderer.bound, dev, indices,
                               draw_state=draw_state, peer_seeds=peer_seeds)
    slots = stage_slots(stage)
    grids = _prep_grids(c, stage, dev)
    gates = [_gate(dev, torch.is_grad_enabled() and decoders.sub(s).wants_grad() and
                   (renderer.decoder_grads is None or s in renderer.decoder_grads)) for s in slots]
    meta = (renderer, decoders, stage, wmeta, w_color, sharder, out, None)
    return _MappingLossFn.apply(meta, *c2ws, *[grids[s] for s in slots], *gates)


def tracking_loss(renderer, c, decoders, c2w: torch.Tensor, depth: torch.Tensor, color: torch.Tensor, n_pixels: int,
                  ignore_edge_H: int = 0, ignore_edge_W: int = 0, w_color: float = 0.5, handle_dynamic: bool = True,
                  use_color: bool = True, device=None, indices: Optional[torch.Tensor] = None, out: Optional[dict] = None) -> torch.Tensor:
    """One tracking iteration's loss (Tracker.optimize_cam_in_batch, src/Tracker.py:86-124) as a single autograd node:
    ``n_pixels`` samples from the frame's ``[ignore_edge_H, H - ignore_edge_H) x [ignore_edge_W, W - ignore_edge_W)`` crop
    under the pose ``c2w`` (3x4 or 4x4; gets its gradient), the bounding-box pre-filter as a mask, the colour-stage render with
    depth-guided samples, and ``sum_mask |gt - depth| / sqrt(var + 1e-10) (+ w_color * sum_mask |gt_rgb - rgb|)`` with
    ``mask = kept & (gt > 0) (& tmp < 10 * median(tmp))`` -- five launches forward (the window kernel, which draws the pixels and
    zero-fills the gradient buffer; the render forward's three; ``nsr_tracking_loss``) and three in the backward (compositor backward,
    dX, pose gradient; + dW / the partial sum only if a decoder wants parameter gradients) instead of ~80.  Returns an fp64 scalar (an ordinary autograd node: an incoming gradient other than 1 scales the pose gradient)."""
    dev = torch.device(device) if device is not None else depth.device
    H0, H1, W0, W1 = int(ignore_edge_H), renderer.H - int(ignore_edge_H), int(ignore_edge_W), renderer.W - int(ignore_edge_W)
    wmeta, c2ws = _window_meta(H0, H1, W0, W1, n_pixels, renderer.W, renderer.fx, renderer.fy, renderer.cx, renderer.cy,
                               [c2w], [depth], [color], renderer.bound, dev, indices)
    slots = stage_slots("color")
    grids = _prep_grids(c, "color", dev)
    gates = [_gate(dev, torch.is_grad_enabled() and decoders.sub(s).wants_grad() and
                   (renderer.decoder_grads is None or s in renderer.decoder_grads)) for s in slots]
    meta = (renderer, decoders, "color", wmeta, w_color, None, out, (bool(handle_dynamic), bool(use_color)))
    return _MappingLossFn.apply(meta, *c2ws, *[grids[s] for s in slots], *gates)


_ONES = {}


def backward(loss: torch.Tensor, retain_graph: bool = False):
    """``loss.backward()`` for a scalar loss without the fill kernel with which autograd creates the root gradient on every call
    (``torch.ones_like(loss)``: one launch, ~5 us of a 230 us mapping iteration): the root gradient is a constant 1 kept per
    (device, dtype).  Same gradients as ``loss.backward()``."""
    key = (loss.device, loss.dtype)
    one = _ONES.get(key)
    if one is None:
        if torch.cuda.is_available() and loss.is_cuda and torch.cuda.is_current_stream_capturing():
            return loss.backward(retain_graph=retain_graph)          # first use under capture: the plain path (allocates inside the graph)
        one = _ONES[key] = torch.ones((), dtype=loss.dtype, device=loss.device)
    torch.autograd.backward(loss, grad_tensors=one.expand_as(loss) if loss.dim() else one, retain_graph=retain_graph)

