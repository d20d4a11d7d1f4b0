"""Drop-in ``Renderer`` (reference: src/utils/Renderer.py) on top of the fused HIP kernels.

Same constructor, same method names, same argument meaning.  ``render_batch_ray`` is one forward kernel -- three under autograd
(sample placement, decoder passes, compositor; csrc/nsr_fwd2.h), followed by the split backward (compositor backward, dX, dW,
finalize; csrc/nsr_bwd2.h) -- instead of the ~250 ATen launches of the reference.  The decoders object must be a
``nice_slam_amd.NICE``.
"""
from __future__ import annotations

import ctypes as C
from typing import Dict, Optional

import torch

from . import _capi
from .common import _as_f32c, _require_cuda, _stream, get_rays, to_channels_last, warn_ncdhw_once
from .layout import param_count, stage_slots

_SLOT_IDX = {s: i for i, s in enumerate(_capi.SLOT_NAMES)}


def _bound6(t: Optional[torch.Tensor]) -> tuple:
    """(lo_x, lo_y, lo_z, hi_x, hi_y, hi_z) as python floats.  The conversion is cached ON the tensor object
    (keyed by its version counter), so a CUDA-resident bound costs one host sync ever and a recycled id() /
    data_ptr() can never alias another tensor's values."""
    if t is None:
        return (float("-inf"),) * 3 + (float("inf"),) * 3
    cached = getattr(t, "_nsr_bound6", None)
    if cached is not None and cached[0] == t._version:
        return cached[1]
    b = t.detach().cpu().to(torch.float64)
    v = tuple(float(b[i, 0]) for i in range(3)) + tuple(float(b[i, 1]) for i in range(3))
    try:
        t._nsr_bound6 = (t._version, v)
    except Exception:
        pass
    return v


def _fill_common(a: _capi.NsrRenderArgs, stage: str, bound: Optional[torch.Tensor], decoders, grids: Dict[str, torch.Tensor],
                 packed: Dict[str, torch.Tensor], flats: Dict[str, torch.Tensor]):
    a.stage = _capi.STAGE_ID[stage]
    b6 = _bound6(bound)
    for i in range(3):
        a.bound_lo[i], a.bound_hi[i] = b6[i], b6[3 + i]
    for s in stage_slots(stage):
        i = _SLOT_IDX[s]
        g = grids[s]
        a.grid[i].feat = g.data_ptr()
        a.grid[i].Z, a.grid[i].Y, a.grid[i].X = g.shape[2], g.shape[3], g.shape[4]
        db = decoders.sub(s).bound
        if db is None:
            raise _capi.NsrError(f"{s}_decoder.bound is not set (reference: NICE_SLAM.load_bound)")
        d6 = _bound6(db)
        for d in range(3):
            a.grid[i].lo[d], a.grid[i].hi[d] = d6[d], d6[3 + d]
        a.dec[i].params = flats[s].data_ptr()
        a.dec[i].packed = packed[s].data_ptr()


def _prep_grids(c: Dict[str, torch.Tensor], stage: str, device) -> Dict[str, torch.Tensor]:
    out = {}
    for s in stage_slots(stage):
        g = c["grid_" + s]
        if g.device != device:
            g = g.to(device)
        if g.dtype != torch.float32 or g.dim() != 5 or g.shape[0] != 1 or g.shape[1] != 32:
            raise _capi.NsrError(f"grid_{s}: expected fp32 [1,32,Z,Y,X], got {g.dtype} {tuple(g.shape)}")
        warn_ncdhw_once(g, f"grid_{s}")
        out[s] = to_channels_last(g)           # differentiable; no copy when the grid already is channels-last
    return out


def eval_points_raw(p: torch.Tensor, decoders, c: Dict[str, torch.Tensor], stage: str, bound: Optional[torch.Tensor]):
    """Forward-only point query: (M,3) world points -> (M,4) fp32 [r,g,b,occ]; occ := 100 outside the open
    ``bound`` box when ``bound`` is given (src/utils/Renderer.py:43-46,57)."""
    lib = _capi.get_lib()
    _require_cuda(p, "eval_points: points")
    dev = p.device
    with torch.no_grad(), _capi.on_device(dev):
        pts = p.detach().to(torch.float64).contiguous()
        grids = _prep_grids({k: v.detach() for k, v in c.items()}, stage, dev)
        stream = _stream(dev)
        flats = {s: decoders.sub(s).flat_params() for s in stage_slots(stage)}
        packed = {s: decoders.sub(s).packed_params(lib, stream) for s in stage_slots(stage)}
        a = _capi.NsrRenderArgs()
        a.n_samples, a.n_surface, a.n_rays = 1, 0, 0
        _fill_common(a, stage, bound, decoders, grids, packed, flats)
        out = torch.empty((pts.shape[0], 4), dtype=torch.float32, device=dev)
        lib.check(lib.nsr_eval_points_fwd(C.byref(a), pts.data_ptr(), pts.shape[0], out.data_ptr(), stream), "nsr_eval_points_fwd")
    return out


_GATES = {}


def _gate(dev, want: bool) -> torch.Tensor:
    """0-dim tensor whose only job is to tell _RenderFn (via needs_input_grad) whether a decoder wants parameter
    gradients; one per (device, flag) for the life of the process instead of a fill kernel per decoder per call."""
    key = (dev, want)
    g = _GATES.get(key)
    if g is None:
        g = _GATES[key] = torch.zeros((), device=dev, requires_grad=want)
    return g


class _RenderFn(torch.autograd.Function):
    """inputs: rays_o, rays_d, then one grid per decoder of the stage, then one 0-dim 'gate' tensor per
    decoder (requires_grad iff that decoder's parameters do).  Parameter gradients are published straight
    into ``Parameter.grad`` as views of one flat buffer (no per-tensor kernels)."""

    @staticmethod
    def forward(ctx, meta, rays_o, rays_d, *tensors):
        with _capi.on_device(rays_o.device):
            return _RenderFn._forward_impl(ctx, meta, rays_o, rays_d, *tensors)

    @staticmethod
    def _forward_impl(ctx, meta, rays_o, rays_d, *tensors):
        renderer, decoders, stage, gt_depth, reduce_hook = meta
        lib = _capi.get_lib()
        slots = stage_slots(stage)
        grids = dict(zip(slots, tensors[:len(slots)]))
        dev = rays_o.device
        stream = _stream(dev)
        n = rays_o.shape[0]
        guided = gt_depth is not None and stage != "coarse"
        S = renderer.N_samples + (renderer.N_surface if guided else 0)
        flats = {s: decoders.sub(s).flat_params() for s in slots}
        packed = {s: decoders.sub(s).packed_params(lib, stream) for s in slots}
        a = _capi.NsrRenderArgs.from_buffer_copy(renderer._arg_template)     # sample fractions pre-filled
        a.n_samples, a.n_surface, a.n_rays = renderer.N_samples, renderer.N_surface, n
        a.rays_o, a.rays_d = rays_o.data_ptr(), rays_d.data_ptr()
        keep = []
        if guided:
            gmax = renderer._gt_max if renderer._gt_max is not None else torch.max(gt_depth).reshape(1)
            keep.append(gmax)
            a.gt_depth, a.gt_max = gt_depth.data_ptr(), gmax.data_ptr()
        _fill_common(a, stage, renderer.bound, decoders, grids, packed, flats)
        depth = torch.empty((n,), dtype=torch.float64, device=dev)
        var = torch.empty((n,), dtype=torch.float64, device=dev)
        rgb = torch.empty((n, 3), dtype=torch.float32, device=dev)
        need_bwd = any(ctx.needs_input_grad)
        raw = torch.empty((n, S, 4), dtype=torch.float32, device=dev) if need_bwd else None
        a.depth, a.var, a.rgb = depth.data_ptr(), var.data_ptr(), rgb.data_ptr()
        a.raw = raw.data_ptr() if raw is not None else None
        zsave = torch.empty((n, S), dtype=torch.float64, device=dev) if need_bwd else None
        a.zvals = zsave.data_ptr() if zsave is not None else None
        keep.append(zsave)
        chunk = 0
        if need_bwd:
            n_sl = len(slots)
            masks_only = [not g_ for g_ in ctx.needs_input_grad[3 + n_sl:3 + 2 * n_sl]]     # per decoder: one nobody differentiates saves its relu masks only
            acts = renderer._attach_acts(a, stage, n, S, dev, masks_only=masks_only)
            keep.append(acts)
            if acts is None:
                # The activation buffer of the whole batch does not fit (Renderer.max_saved_activation_bytes / free memory / 2^25
                # sample points): this forward runs without one and the backward goes through the batch in chunks, each chunk a
                # forward that saves its activations followed by the split backward (_chunked_backward).
                chunk = renderer.acts_chunk_rays(stage, S, dev)
                a.raw = a.zvals = None
                raw = zsave = None
        lib.check(lib.nsr_render_fwd(C.byref(a), stream), "nsr_render_fwd")
        ctx.chunk = chunk
        if need_bwd:
            # `depth` is an OUTPUT: kept as a detached alias (same storage, different tensor object), so that no reference
            # cycle output -> grad_fn -> ctx -> output forms (a forward whose backward never runs is then freed normally)
            ctx.args, ctx.keep = a, (keep, rays_o, rays_d, gt_depth, grids, flats, packed, raw, depth.detach())
            ctx.meta = (renderer, decoders, stage, S, reduce_hook)
        return depth, var, rgb

    @staticmethod
    @torch.autograd.function.once_differentiable
    def backward(ctx, g_depth, g_var, g_rgb):
        with _capi.on_device(g_depth.device):
            return _RenderFn._backward_impl(ctx, g_depth, g_var, g_rgb)

    @staticmethod
    def _backward_impl(ctx, g_depth, g_var, g_rgb):
        slots = stage_slots(ctx.meta[2])
        need = (ctx.needs_input_grad[1], ctx.needs_input_grad[2], ctx.needs_input_grad[3:3 + len(slots)],
                ctx.needs_input_grad[3 + len(slots):3 + 2 * len(slots)])
        if ctx.keep is None:
            raise RuntimeError("nice_slam_amd: backward through render_batch_ray a second time is not supported (the saved "
                               "buffers are released after the first backward; render again instead of retain_graph=True)")
        g_depth = g_depth.to(torch.float64).contiguous()
        g_var = g_var.to(torch.float64).contiguous()
        g_rgb = g_rgb.to(torch.float32).contiguous()
        if ctx.chunk:
            d_o, d_d, d_grids = _chunked_backward(ctx.args, ctx.meta, ctx.keep, need, g_depth, g_var, g_rgb, ctx.chunk)
        else:
            d_o, d_d, d_grids = render_backward(ctx.args, ctx.meta, ctx.keep, need, g_depth, g_var, g_rgb)
        ctx.keep = ctx.args = None
        return (None, d_o, d_d, *d_grids, *([None] * len(slots)))


def _chunked_backward(a, meta, kept, need, g_depth, g_var, g_rgb, chunk):
    """Backward of a batch whose activation buffer was too large to keep: per chunk of ``chunk`` rays the forward is run again
    with an activation buffer (same kernels, same batch-global depth cap) and the split backward follows; grid and decoder
    gradients accumulate over the chunks, the ray gradients are written per chunk."""
    lib = _capi.get_lib()
    renderer, decoders, stage, S, reduce_hook = meta
    keep, rays_o, rays_d, gt_depth, grids, flats, packed, _, _ = kept
    if reduce_hook is not None:
        raise _capi.NsrError("nice_slam_amd: a sharded render call whose activation buffer does not fit is not supported (use smaller batches)")
    slots = stage_slots(stage)
    dev = rays_o.device
    stream = _stream(dev)
    n = rays_o.shape[0]
    need_o, need_d, need_grid, need_par = need
    d_o = torch.zeros_like(rays_o) if (need_o or need_d) else None
    d_d = torch.zeros_like(rays_d) if (need_o or need_d) else None
    d_grids = [None] * len(slots)
    for i0 in range(0, n, chunk):
        i1 = min(n, i0 + chunk)
        m = i1 - i0
        ac = _capi.NsrRenderArgs.from_buffer_copy(a)
        ac.n_rays = m
        ro, rd = rays_o[i0:i1].contiguous(), rays_d[i0:i1].contiguous()
        ac.rays_o, ac.rays_d = ro.data_ptr(), rd.data_ptr()
        gdc = None
        if gt_depth is not None and a.gt_depth:
            gdc = gt_depth[i0:i1].contiguous()
            ac.gt_depth = gdc.data_ptr()
        depth = torch.empty((m,), dtype=torch.float64, device=dev)
        var = torch.empty((m,), dtype=torch.float64, device=dev)
        rgb = torch.empty((m, 3), dtype=torch.float32, device=dev)
        raw = torch.empty((m, S, 4), dtype=torch.float32, device=dev)
        zs = torch.empty((m, S), dtype=torch.float64, device=dev)
        ac.depth, ac.var, ac.rgb, ac.raw, ac.zvals = depth.data_ptr(), var.data_ptr(), rgb.data_ptr(), raw.data_ptr(), zs.data_ptr()
        acts = renderer._attach_acts(ac, stage, m, S, dev, masks_only=[not g_ for g_ in need_par])
        if acts is None:
            raise _capi.NsrError("nice_slam_amd: no room for the activation buffer of a %d-ray chunk" % m)
        lib.check(lib.nsr_render_fwd(C.byref(ac), stream), "nsr_render_fwd(chunk)")
        kc = (keep + [acts, zs, gdc], ro, rd, gdc, grids, flats, packed, raw, depth)
        co, cd, cg = render_backward(ac, meta, kc, need, g_depth[i0:i1].contiguous(), g_var[i0:i1].contiguous(), g_rgb[i0:i1].contiguous())
        if co is not None:
            d_o[i0:i1] = co
        if cd is not None:
            d_d[i0:i1] = cd
        for k, g in enumerate(cg):
            if g is not None:
                d_grids[k] = g if d_grids[k] is None else d_grids[k].add_(g)
    return (d_o if need_o else None, d_d if need_d else None, d_grids)


def render_backward(a, meta, kept, need, g_depth, g_var, g_rgb, zero_buf=None, grad_scale=None, loss_grads_from_forward=False):
    """The backward launch of one render call (shared by ``_RenderFn`` and the fused mapping loss, mapping.py).
    ``a``: the forward's argument block; ``need`` = (rays_o, rays_d, per-grid, per-decoder) gradient requests;
    ``g_*``: gradients of the outputs (contiguous, fp64 / fp64 / fp32) or None; ``zero_buf``: an already zero-filled fp32
    buffer of the size ``backward_buffer_floats`` returns (saves the fill launch); ``grad_scale``: optional 1-element fp64 device
    tensor every ``g_*`` is multiplied by inside the kernel (the incoming gradient of a fused loss node);
    ``loss_grads_from_forward``: ``g_depth`` / ``g_rgb`` are the forward's own ``dl_depth`` / ``dl_rgb``, untouched (the backward
    then starts from the ``d raw`` the forward's loss epilogue precomputed, nsr_bwd_args.loss_grads_from_forward).
    -> (d_rays_o, d_rays_d, [d_grid ...])."""
    lib = _capi.get_lib()
    renderer, decoders, stage, S, reduce_hook = meta
    keep, rays_o, rays_d, gt_depth, grids, flats, packed, raw, depth = kept
    slots = stage_slots(stage)
    dev = rays_o.device
    stream = _stream(dev)
    n = rays_o.shape[0]
    need_o, need_d, need_grid, need_par = need
    b = _capi.NsrBwdArgs()
    b.d_depth = g_depth.data_ptr() if g_depth is not None else None
    b.d_var = g_var.data_ptr() if g_var is not None else None
    b.d_rgb = g_rgb.data_ptr() if g_rgb is not None else None
    b.depth = depth.data_ptr()
    b.grad_scale = grad_scale.data_ptr() if grad_scale is not None else None
    b.loss_grads_from_forward = 1 if loss_grads_from_forward else 0
    # every gradient this call produces lives in ONE zero-filled buffer (a single fill kernel): channels-last views for
    # the dense grid gradients, then the ray gradients, then the flat decoder-gradient blob
    need_ray = need_o or need_d
    n_grid = [grids[s].numel() if nd else 0 for s, nd in zip(slots, need_grid)]
    # decoder gradients: straight into each decoder's persistent blob when possible (see _FlatDecoder.grad_target);
    # the multi-GPU path and foreign .grad tensors use a temporary blob inside the fused buffer
    direct = {}
    if reduce_hook is None:
        for s, nd in zip(slots, need_par):
            if nd:
                tgt, mode = decoders.sub(s).grad_target()
                if tgt is not None:
                    direct[s] = (tgt, mode)
        modes = {m for _, m in direct.values()}
        if len(modes) > 1:                                    # mixed: make everything "accumulate"
            for s, (tgt, mode) in list(direct.items()):
                if mode == "overwrite":
                    tgt.zero_()
        b.overwrite_dparams = 1 if modes == {"overwrite"} else 0
    n_par = [param_count(s) if (nd and s not in direct) else 0 for s, nd in zip(slots, need_par)]
    total = sum(n_grid) + (6 * n if need_ray else 0) + sum(n_par)
    buf = zero_buf if (zero_buf is not None and zero_buf.numel() >= total) else torch.zeros((total,), dtype=torch.float32, device=dev)
    off = 0
    d_grids = []
    for s, cnt in zip(slots, n_grid):
        i = _SLOT_IDX[s]
        if cnt:
            _, ch, Z, Y, X = grids[s].shape
            dg = buf[off:off + cnt].view(1, Z, Y, X, ch).permute(0, 4, 1, 2, 3)      # [1,C,Z,Y,X], channels-last strides
            off += cnt
            a.grid[i].dfeat = dg.data_ptr()
            a.grad_voxel_mask[i] = renderer._grad_voxel_mask_ptr(s, grids[s])
            d_grids.append(dg)
        else:
            a.grid[i].dfeat = None
            a.grad_voxel_mask[i] = None
            d_grids.append(None)
    d_o = d_d = None
    if need_ray:
        d_od = buf[off:off + 6 * n].view(2, n, 3)
        off += 6 * n
        d_o, d_d = d_od[0], d_od[1]
        b.d_rays_o, b.d_rays_d = d_o.data_ptr(), d_d.data_ptr()
    gflat = None
    offs = {}
    if any(need_par):
        if sum(n_par):
            gflat = buf[off:off + sum(n_par)]
        poff = 0
        for s, cnt, nd in zip(slots, n_par, need_par):
            i = _SLOT_IDX[s]
            if s in direct:
                a.dec[i].dparams = direct[s][0].data_ptr()
            elif cnt:
                offs[s] = poff
                a.dec[i].dparams = gflat.data_ptr() + 4 * poff
                poff += cnt
            else:
                a.dec[i].dparams = None
        nws = lib.nsr_bwd_workspace_floats(_capi.STAGE_ID[stage], n, S, renderer.bwd_max_blocks)
        ws = renderer._workspace(nws, dev)
        b.workspace, b.workspace_floats = ws.data_ptr(), nws
    else:
        for s in slots:
            a.dec[_SLOT_IDX[s]].dparams = None
    b.max_blocks = renderer.bwd_max_blocks
    if renderer.profile_events is not None:           # bench.py: hipEvents around the backward (start, stop[, behind dX, behind dW])
        evs = renderer.profile_events(stage)
        b.ev_start, b.ev_stop = evs[0], evs[1]
        if len(evs) >= 4:
            b.ev_dx_done, b.ev_dw_done = evs[2], evs[3]
    lib.check(lib.nsr_render_bwd(C.byref(a), C.byref(b), stream), "nsr_render_bwd")

    def publish():
        for s, nd in zip(slots, need_par):
            if nd:
                if s in direct:
                    decoders.sub(s).grad_done(direct[s][1])
                else:
                    decoders.sub(s).publish_grads(gflat[offs[s]:offs[s] + param_count(s)])

    # multi-GPU (parallel.py): the hook sums the shard gradients.  It may DEFER the decoder blob (to let it ride in the
    # same collective as the grid rows, which autograd hands over a moment later); the `.grad` tensors are then
    # published by the hook after that exchange -- never before, or an accumulation into pre-existing `.grad`s would
    # consume rank-local values
    deferred = reduce_hook is not None and bool(reduce_hook([g for g in d_grids if g is not None], gflat, publish))
    if not deferred:
        publish()
    return (d_o if need_o else None, d_d if need_d else None, d_grids)


class Renderer(object):
    """src/utils/Renderer.py:5-21.  ``slam`` only needs ``nice, bound, H, W, fx, fy, cx, cy``."""

    def __init__(self, cfg, args, slam, points_batch_size=500000, ray_batch_size=100000):
        self.ray_batch_size = ray_batch_size
        self.points_batch_size = points_batch_size       # kept for API parity; the fused kernel needs no chunking
        self.lindisp = cfg["rendering"]["lindisp"]
        self.perturb = cfg["rendering"]["perturb"]
        self.N_samples = cfg["rendering"]["N_samples"]
        self.N_surface = cfg["rendering"]["N_surface"]
        self.N_importance = cfg["rendering"]["N_importance"]
        self.scale = cfg["scale"]
        self.occupancy = cfg["occupancy"]
        self.nice = slam.nice
        self.bound = slam.bound
        self.H, self.W, self.fx, self.fy, self.cx, self.cy = slam.H, slam.W, slam.fx, slam.fy, slam.cx, slam.cy
        if self.lindisp or self.perturb > 0 or self.N_importance > 0 or not self.occupancy or not self.nice:
            raise NotImplementedError("nice_slam_amd implements the NICE-SLAM configuration of the render path "
                                      "(occupancy, perturb=0, N_importance=0, lindisp=False); iMAP* is out of scope")
        if self.N_samples + self.N_surface > _capi.MAX_SAMPLES:
            raise NotImplementedError("N_samples + N_surface must not exceed 64")
        # sample fractions exactly as torch produces them (Renderer.py:132,152)
        self._t_uniform = torch.linspace(0.0, 1.0, steps=self.N_samples).tolist()
        self._t_surface = torch.linspace(0.0, 1.0, steps=max(self.N_surface, 1)).double().tolist()[:self.N_surface]
        tmpl = _capi.NsrRenderArgs()
        for i, v in enumerate(self._t_uniform):
            tmpl.t_uniform[i] = v
        for i, v in enumerate(self._t_surface):
            tmpl.t_surface[i] = v
        self._arg_template = bytes(tmpl)
        self.bwd_max_blocks = 0                 # 0 = library default persistent-grid cap
        # Saved activations: the forward of a call that will be differentiated also writes the decoders' hidden states, relu
        # masks and grid features (832 B per sample point and decoder) and the backward runs as the split dX / dW kernels over
        # them (csrc/nsr_bwd2.h; +640 B per point and decoder of dY scratch in the same buffer: 214 MB per 1000 colour-stage
        # rays).  The buffer may take `max_saved_activation_bytes` at most, and never more than `acts_memory_fraction` of the
        # device memory that is free when the call is made; a batch that needs more is differentiated in chunks
        # (_chunked_backward: per chunk a forward that saves, then the split backward).
        self.max_saved_activation_bytes = 64 << 30
        self.acts_memory_fraction = 0.6
        # Fused iterations (mapping_loss / tracking_loss): the rays the callers' bounding-box pre-filter rejects are removed from the
        # batch like the reference's compaction does (Mapper.py:471-481, Tracker.py:95-104) -- no decoder evaluation, outputs 0, no
        # gradient -- instead of being rendered and masked out of the loss (False: render them, e.g. to look at their outputs).
        self.skip_masked_rays = True
        # Optional: restrict parameter gradients to these decoders, e.g. ("color",).  The reference's autograd
        # produces dW for every decoder in every stage although src/Mapper.py:335-341 only ever steps the colour
        # decoder (and the fine one when fix_fine is False); None = reference semantics (requires_grad decides).
        self.decoder_grads = None
        # Optional, opt-in: {"grid_middle": uint8 [Z,Y,X] tensor, ...} (what FrustumSelector.voxel_mask returns, frustum.py) -- the
        # voxels whose gradient the caller will consume.  With `frustum_feature_selection` the mapper's optimiser only holds
        # `val[mask]` (src/Mapper.py:315-333,394-401); given the same masks, the backward's scatter skips every other voxel
        # (their gradient stays zero instead of being computed and thrown away).  None = the reference's dense gradient.
        self.grad_voxel_masks = None
        self.profile_events = None              # optional callable(stage) -> (hipEvent_t start, stop[, behind dX, behind dW]) for the backward
        self.profile_fwd_events = None          # optional callable(stage) -> (hipEvent_t start, stop) around the forward's decoder-pass kernel
        self._gt_max = None                     # set by the multi-GPU wrapper: batch-global max(gt_depth)
        self._reduce_hook = None
        self._ws = {}                           # per-device backward workspace (not pickled)

    def __getstate__(self):                      # Renderer objects are pickled into spawned processes
        d = dict(self.__dict__)
        d["_ws"] = {}
        d["_reduce_hook"] = None
        d["profile_events"] = None
        d["profile_fwd_events"] = None
        return d

    def _grad_voxel_mask_ptr(self, slot: str, grid: torch.Tensor):
        m = None if not self.grad_voxel_masks else (self.grad_voxel_masks.get("grid_" + slot, self.grad_voxel_masks.get(slot)))
        if m is None:
            return None
        if m.dtype != torch.uint8 or tuple(m.shape) != tuple(grid.shape[2:]) or m.device != grid.device or not m.is_contiguous():
            raise _capi.NsrError(f"grad_voxel_masks[{slot}]: expected a contiguous uint8 {tuple(grid.shape[2:])} tensor on {grid.device}, "
                                 f"got {m.dtype} {tuple(m.shape)} on {m.device}")
        return m.data_ptr()

    def _workspace(self, nfloats: int, dev) -> torch.Tensor:
        ws = self._ws.get(dev)
        if ws is None or ws.numel() < nfloats:
            ws = torch.empty((max(nfloats, 1),), dtype=torch.float32, device=dev)
            self._ws[dev] = ws
        return ws

    # ---------------------------------------------------------------------------------------------
    def eval_points(self, p, decoders, c=None, stage="color", device="cuda:0"):
        """Renderer.py:23-61 (forward only)."""
        return eval_points_raw(p, decoders, c, stage, self.bound)

    def render_batch_ray(self, c, decoders, rays_d, rays_o, device, stage, gt_depth=None, gt_max=None):
        """Renderer.py:63-198: returns (depth fp64 (N,), uncertainty fp64 (N,), color fp32 (N,3)).
        ``gt_max`` (optional, not in the reference): a 1-element device tensor to use as the batch-global
        ``max(gt_depth)`` of Renderer.py:109,144 instead of the maximum over ``gt_depth`` -- for callers that keep the
        bounding-box-rejected rays in the batch and mask their loss (``nice_slam_amd.aabb_keep``)."""
        _require_cuda(rays_o, "render_batch_ray: rays")
        dev = rays_o.device
        if stage == "coarse":
            gt_depth = None
        slots = stage_slots(stage)
        rays_o = _as_f32c(rays_o)
        rays_d = _as_f32c(rays_d, dev)
        if gt_depth is not None:
            gt_depth = _as_f32c(gt_depth.detach().reshape(-1), dev)
        if rays_o.shape[0] == 0:                 # empty batch (e.g. every ray removed by the caller's AABB pre-filter)
            z = (rays_o.sum() + rays_d.sum()) * 0.0  # keeps the graph connected
            return (torch.zeros((0,), dtype=torch.float64, device=dev) + z, torch.zeros((0,), dtype=torch.float64, device=dev) + z,
                    torch.zeros((0, 3), dtype=torch.float32, device=dev) + z)
        grids = _prep_grids(c, stage, dev)
        gates = []
        for s in slots:
            want = torch.is_grad_enabled() and decoders.sub(s).wants_grad() and \
                (self.decoder_grads is None or s in self.decoder_grads)
            gates.append(_gate(dev, want))
        meta = (self, decoders, stage, gt_depth, self._reduce_hook)
        if gt_max is not None and self._gt_max is None:
            self._gt_max = gt_max.detach().to(device=dev, dtype=torch.float32).reshape(1)
            try:
                return _RenderFn.apply(meta, rays_o, rays_d, *[grids[s] for s in slots], *gates)
            finally:
                self._gt_max = None
        return _RenderFn.apply(meta, rays_o, rays_d, *[grids[s] for s in slots], *gates)

    def _acts_budget(self, dev) -> int:
        """bytes an activation buffer may take right now (not queried while a stream is capturing: the cap alone applies)"""
        cap = int(self.max_saved_activation_bytes)
        if not torch.cuda.is_current_stream_capturing():
            free, _ = torch.cuda.mem_get_info(dev)
            try:                                            # + what the caching allocator holds but has not handed out
                st = torch.cuda.memory_stats(dev)
                free += int(st.get("reserved_bytes.all.current", 0)) - int(st.get("allocated_bytes.all.current", 0))
            except Exception:
                pass
            cap = min(cap, int(self.acts_memory_fraction * free))
        return cap

    def acts_chunk_rays(self, stage, S, dev) -> int:
        """rays per chunk of a batch whose activation buffer does not fit: the budget and the 2^25-point limit of one call"""
        per_ray = 4 * max(1, _capi.get_lib().nsr_acts_floats(_capi.STAGE_ID[stage], 1024, S)) / 1024.0
        rays = int(min(self._acts_budget(dev) / per_ray, ((1 << 25) - 16) // S))
        rays -= rays % 64
        if rays < 64:
            raise _capi.NsrError("nice_slam_amd: not enough device memory for the activation buffer of even 64 rays")
        return rays

    def _attach_acts(self, a, stage, n, S, dev, masks_only=False):
        """allocate the activation buffer of a differentiable forward and point the argument block at it; None when it does not
        fit (budget, 2^25 sample points per call, or the allocation fails): the caller then differentiates in chunks.
        ``masks_only``: no decoder will want parameter gradients (tracking) -- the forward then only writes the relu masks"""
        nfl = _capi.get_lib().nsr_acts_floats(_capi.STAGE_ID[stage], n, S)
        if nfl <= 0:
            raise _capi.NsrError("nsr_acts_floats: bad arguments")
        if n * S > (1 << 25) - 16 or 4 * nfl > self.max_saved_activation_bytes:
            return None
        if 4 * nfl > (256 << 20) and 4 * nfl > self._acts_budget(dev):      # small buffers: no query on the per-iteration path
            return None
        try:
            acts = torch.empty((nfl,), dtype=torch.float32, device=dev)
        except torch.cuda.OutOfMemoryError:
            return None
        a.acts = acts.data_ptr()
        # masks_only: bool (every decoder) or one bool per decoder pass of the stage, in slot order (nsr_render_args.acts_masks_only)
        if isinstance(masks_only, (list, tuple)):
            masks_only = list(masks_only)
            if len(masks_only) >= 2 and not masks_only[1]:
                masks_only[0] = False               # the fine decoder's dW reads the middle pass's saved features ([c_fine | c_mid], decoder.py:182-187)
            a.acts_masks_only = 1 if all(masks_only) else sum(2 << i for i, m in enumerate(masks_only) if m)
        else:
            a.acts_masks_only = 1 if masks_only else 0
        return acts

    def render_img(self, c, decoders, c2w, device, stage, gt_depth=None):
        """Renderer.py:200-255 (the reference requires gt_depth here; we also accept None)."""
        with torch.no_grad():
            H, W = self.H, self.W
            rays_o, rays_d = get_rays(H, W, self.fx, self.fy, self.cx, self.cy, c2w, device)
            rays_o = rays_o.reshape(-1, 3).contiguous()
            rays_d = rays_d.reshape(-1, 3).contiguous()
            gd = None if gt_depth is None else gt_depth.reshape(-1)
            depth_l, unc_l, col_l = [], [], []
            for i in range(0, rays_d.shape[0], self.ray_batch_size):
                sl = slice(i, i + self.ray_batch_size)
                d, u, col = self.render_batch_ray(c, decoders, rays_d[sl], rays_o[sl], device, stage,
                                                  gt_depth=None if gd is None else gd[sl])
                depth_l.append(d.double()); unc_l.append(u.double()); col_l.append(col)
            depth = torch.cat(depth_l).reshape(H, W)
            unc = torch.cat(unc_l).reshape(H, W)
            color = torch.cat(col_l).reshape(H, W, 3)
            return depth, unc, color

    def regulation(self, c, decoders, rays_d, rays_o, gt_depth, device, stage="color"):
        raise NotImplementedError("Renderer.regulation is only used by iMAP* (src/Mapper.py:496-501), out of scope")
