"""Host-side mirror of the reference's ray helpers and scene-geometry setup.

get_samples / get_rays : src/common.py:74-134, 248-266
load_bound / grid_init : src/NICE_SLAM.py:137-157, 192-250  (grids are allocated channels-last)
"""
from __future__ import annotations

from typing import Dict, Optional, Tuple

import warnings

import numpy as np

import torch

from . import _capi


def _stream(device) -> int:
    """HIP stream handle for the next libnsr launch on `device`.  The library launches on the CURRENT device (its kernels
    carry no device guard): a caller working on tensors of another GPU of the same process (the reference lets tracking and
    mapping name different devices, configs/nice_slam.yaml:31,44) gets that device made current here, like every torch
    operator does for the duration of its launch."""
    device = torch.device(device)
    idx = device.index
    cur = torch.cuda.current_device()
    if idx is not None and idx != cur:
        # restored by Lib.check(), which every launch site calls right after the library call (`lib.check(lib.nsr_x(...,
        # _stream(dev)), ...)`): the caller's current device is the same before and after, like around a torch operator
        if getattr(_capi.restore_device, "idx", None) is None:
            _capi.restore_device.idx = cur
        torch.cuda.set_device(idx)
    return torch.cuda.current_stream(device).cuda_stream


def _as_f32c(t: torch.Tensor, device=None) -> torch.Tensor:
    """fp32, contiguous, on `device` -- without touching tensors that already are (each no-op `.to()` / `.contiguous()`
    is microseconds of host time, and the mapping loop makes dozens of them per iteration)."""
    if t.dtype is not torch.float32 or (device is not None and t.device != device):
        t = t.to(device=device if device is not None else t.device, dtype=torch.float32)
    return t if t.is_contiguous() else t.contiguous()


def _require_cuda(t: torch.Tensor, what: str):
    if not t.is_cuda:
        raise _capi.NsrError(f"{what} must live on an AMD GPU (got {t.device}); nice_slam_amd has no CPU path")


# --------------------------------------------------------------------------------------------------
# scene geometry
# --------------------------------------------------------------------------------------------------
def load_bound(cfg: dict, scale: Optional[float] = None) -> torch.Tensor:
    """src/NICE_SLAM.py:137-150: scene bound (3,2) fp64, upper corner enlarged to a multiple of
    ``grid_len.bound_divisible``."""
    scale = cfg["scale"] if scale is None else scale
    bound = torch.from_numpy(np.array(cfg["mapping"]["bound"], dtype=np.float64) * scale)
    div = cfg["grid_len"]["bound_divisible"]
    bound[:, 1] = (((bound[:, 1] - bound[:, 0]) / div).int() + 1) * div + bound[:, 0]
    return bound


def set_decoder_bounds(decoders, bound: torch.Tensor, coarse_bound_enlarge: float = 2.0):
    """src/NICE_SLAM.py:151-157."""
    decoders.bound = bound
    decoders.middle_decoder.bound = bound
    decoders.fine_decoder.bound = bound
    decoders.color_decoder.bound = bound
    if hasattr(decoders, "coarse_decoder"):
        decoders.coarse_decoder.bound = bound * coarse_bound_enlarge


def grid_shapes(cfg: dict, bound: torch.Tensor) -> Dict[str, Tuple[int, int, int]]:
    """Cells per axis, (Z,Y,X) order (src/NICE_SLAM.py:211-246)."""
    xyz_len = bound[:, 1] - bound[:, 0]
    out = {}
    names = (("coarse",) if cfg.get("coarse", True) else ()) + ("middle", "fine", "color")
    for name in names:
        ext = xyz_len * cfg["model"]["coarse_bound_enlarge"] if name == "coarse" else xyz_len
        nx, ny, nz = (int(v) for v in (ext / cfg["grid_len"][name]).tolist())
        out["grid_" + name] = (nz, ny, nx)
    return out


_WARNED_NCDHW = False


def warn_ncdhw_once(grid: torch.Tensor, what: str):
    """A grid handed to a per-iteration entry point in NCDHW order is re-laid out (a full copy forward, a permuted
    gradient backward) on EVERY call: legal, slow, and said once."""
    global _WARNED_NCDHW
    if not _WARNED_NCDHW and not grid.is_contiguous(memory_format=torch.channels_last_3d):
        _WARNED_NCDHW = True
        warnings.warn(f"nice_slam_amd: {what} is not in torch.channels_last_3d memory order; it is copied into that order on "
                      "every call.  Allocate grids with grid_init() / to_channels_last() once (INTEGRATION.md section 2).",
                      RuntimeWarning, stacklevel=4)


def to_channels_last(grid: torch.Tensor) -> torch.Tensor:
    """Logical [1,C,Z,Y,X] tensor whose memory is [Z][Y][X][C] (what the kernels read).  No-op if it already is."""
    return grid.contiguous(memory_format=torch.channels_last_3d)


def grid_init(cfg: dict, bound: torch.Tensor, device="cpu") -> Dict[str, torch.Tensor]:
    """src/NICE_SLAM.py:192-250 with the same init statistics (normal std 0.01, fine 1e-4), allocated in
    torch.channels_last_3d so that a voxel's 32 channels are one 128-byte line."""
    c_dim = cfg["model"]["c_dim"]
    out = {}
    for key, zyx in grid_shapes(cfg, bound).items():
        std = 1e-4 if key == "grid_fine" else 1e-2
        val = torch.zeros((1, c_dim) + tuple(zyx)).normal_(mean=0, std=std)
        out[key] = val.to(device).contiguous(memory_format=torch.channels_last_3d)
    return out


# --------------------------------------------------------------------------------------------------
# rays
# --------------------------------------------------------------------------------------------------
class _GetSamplesFn(torch.autograd.Function):
    """indices -> (rays_o, rays_d, depth, color); differentiable w.r.t. c2w (tracking / BA)."""

    @staticmethod
    def run(c2w, indices, depth, color, H0, H1, W0, W1, fx, fy, cx, cy):
        lib = _capi.get_lib()
        n = indices.shape[0]
        dev = depth.device
        buf = torch.empty((10 * n,), dtype=torch.float32, device=dev)       # one allocation: o | d | depth | colour
        rays_o, rays_d = buf[:3 * n].view(n, 3), buf[3 * n:6 * n].view(n, 3)
        s_depth, s_color = buf[6 * n:7 * n], buf[7 * n:].view(n, 3)
        c2w_c = _as_f32c(c2w.detach(), dev)
        lib.check(lib.nsr_get_samples(indices.data_ptr(), n, H0, H1, W0, W1, depth.shape[1], fx, fy, cx, cy,
                                      c2w_c.data_ptr(), c2w_c.stride(0), depth.data_ptr(), color.data_ptr(),
                                      rays_o.data_ptr(), rays_d.data_ptr(), s_depth.data_ptr(), s_color.data_ptr(),
                                      _stream(dev)), "nsr_get_samples")
        return rays_o, rays_d, s_depth, s_color

    @staticmethod
    def forward(ctx, c2w, indices, depth, color, H0, H1, W0, W1, fx, fy, cx, cy):
        rays_o, rays_d, s_depth, s_color = _GetSamplesFn.run(c2w, indices, depth, color, H0, H1, W0, W1, fx, fy, cx, cy)
        ctx.geom = (H0, W0, W1 - W0, fx, fy, cx, cy, tuple(c2w.shape), c2w.dtype, c2w.device)
        ctx.save_for_backward(indices)
        ctx.mark_non_differentiable(s_depth, s_color)
        return rays_o, rays_d, s_depth, s_color

    @staticmethod
    @torch.autograd.function.once_differentiable
    def backward(ctx, g_o, g_d, _gd, _gc):
        (indices,) = ctx.saved_tensors
        H0, W0, w, fx, fy, cx, cy, shape, dtype, dev = ctx.geom
        col = (indices % w + W0).to(torch.float32)
        row = (torch.div(indices, w, rounding_mode="floor") + H0).to(torch.float32)
        dirs = torch.stack([(col - cx) / fx, -(row - cy) / fy, -torch.ones_like(col)], -1)
        g = torch.zeros(shape, dtype=torch.float32, device=g_d.device)
        g[:3, :3] = g_d.t() @ dirs                       # rays_d[n,a] = sum_k dirs[n,k] * c2w[a,k]
        g[:3, 3] = g_o.sum(0)                            # rays_o[n,a] = c2w[a,3]
        return (g.to(device=dev, dtype=dtype),) + (None,) * 11


def get_samples(H0, H1, W0, W1, n, H, W, fx, fy, cx, cy, c2w, depth, color, device):
    """Drop-in for src/common.py:125-134.  The index draw is the reference's own call
    (``torch.randint(h*w, (n,), device=device)``, common.py:99) so the RNG stream is consumed identically;
    everything after the draw is one fused kernel."""
    if isinstance(c2w, np.ndarray):
        c2w = torch.from_numpy(c2w).to(device)
    if not depth.is_cuda:
        depth = depth.to(device=device)
    if not color.is_cuda:
        color = color.to(device=device)
    _require_cuda(depth, "get_samples: depth image")
    indices = torch.randint((H1 - H0) * (W1 - W0), (n,), device=depth.device)
    return samples_from_indices(indices, H0, H1, W0, W1, fx, fy, cx, cy, c2w, depth, color)


def samples_from_indices(indices, H0, H1, W0, W1, fx, fy, cx, cy, c2w, depth, color):
    depth = _as_f32c(depth)
    color = _as_f32c(color, depth.device)
    args = (c2w, indices if indices.is_contiguous() else indices.contiguous(), depth, color, int(H0), int(H1), int(W0), int(W1),
            float(fx), float(fy), float(cx), float(cy))
    if not (torch.is_grad_enabled() and isinstance(c2w, torch.Tensor) and c2w.requires_grad):
        return _GetSamplesFn.run(*args)               # mapping without BA: nothing to differentiate, skip the autograd node
    return _GetSamplesFn.apply(*args)


def aabb_keep(rays_o: torch.Tensor, rays_d: torch.Tensor, gt_depth: torch.Tensor, bound) -> Tuple[torch.Tensor, torch.Tensor]:
    """The callers' bounding-box pre-filter (src/Mapper.py:471-481, src/Tracker.py:95-104) as a mask instead of a
    compaction:  keep = min_axis max((bound - o) / d) >= gt_depth  in fp64, one kernel, no host sync.
    Returns ``(keep bool (N,), kept_max fp32 (1,))``; ``kept_max`` = max of ``gt_depth`` over the kept rays, to be passed as
    ``render_batch_ray(..., gt_max=kept_max)``.  Rendering the full batch and multiplying each loss term by ``keep`` gives
    the same loss and the same gradients as the reference's ``batch_rays_o[inside_mask]`` compaction."""
    import ctypes as C
    _require_cuda(rays_o, "aabb_keep: rays")
    dev = rays_o.device
    o, d, gd = _as_f32c(rays_o.detach()), _as_f32c(rays_d.detach(), dev), _as_f32c(gt_depth.detach().reshape(-1), dev)
    n = o.shape[0]
    keep = torch.empty((n,), dtype=torch.uint8, device=dev)
    kmax = torch.zeros((1,), dtype=torch.float32, device=dev)
    lo = (C.c_double * 3)(*[float(bound[a][0]) for a in range(3)])
    hi = (C.c_double * 3)(*[float(bound[a][1]) for a in range(3)])
    lib = _capi.get_lib()
    lib.check(lib.nsr_aabb_keep(o.data_ptr(), d.data_ptr(), gd.data_ptr(), n, lo, hi, keep.data_ptr(), kmax.data_ptr(),
                                _stream(dev)), "nsr_aabb_keep")
    return keep.bool(), kmax


class _CameraFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, cam):
        lib = _capi.get_lib()
        c = cam.detach().reshape(-1, 7).to(torch.float32).contiguous()
        rt = torch.empty((c.shape[0], 3, 4), dtype=torch.float32, device=c.device)
        lib.check(lib.nsr_camera_from_tensor(c.data_ptr(), c.shape[0], rt.data_ptr(), None, None, _stream(c.device)), "nsr_camera_from_tensor")
        ctx.save_for_backward(c)
        ctx.meta = (cam.shape, cam.dtype)
        return rt[0] if cam.dim() == 1 else rt

    @staticmethod
    @torch.autograd.function.once_differentiable
    def backward(ctx, g):
        lib = _capi.get_lib()
        (c,) = ctx.saved_tensors
        g = g.reshape(-1, 3, 4).to(torch.float32).contiguous()
        d = torch.empty_like(c)
        lib.check(lib.nsr_camera_from_tensor(c.data_ptr(), c.shape[0], None, g.data_ptr(), d.data_ptr(), _stream(c.device)), "nsr_camera_from_tensor")
        shape, dtype = ctx.meta
        return d.reshape(shape).to(dtype)


def get_camera_from_tensor(inputs: torch.Tensor) -> torch.Tensor:
    """src/common.py:163-176 (with quad2rotation, :137-160): ``[quaternion (w,x,y,z), translation]`` (7,) or (B,7) -> the 3x4
    (or B x 3 x 4) matrix ``[R | T]``; one launch forward, one backward (the reference spends ~25 + ~45 ATen launches per
    tracking / BA iteration here)."""
    _require_cuda(inputs, "get_camera_from_tensor: inputs")
    if inputs.shape[-1] != 7 or inputs.dim() not in (1, 2):
        raise _capi.NsrError(f"get_camera_from_tensor: expected (7,) or (B,7), got {tuple(inputs.shape)}")
    return _CameraFn.apply(inputs)


def get_rays(H, W, fx, fy, cx, cy, c2w, device):
    """src/common.py:248-266: rays for a whole image (used by render_img, forward only)."""
    if isinstance(c2w, np.ndarray):
        c2w = torch.from_numpy(c2w)
    c2w = c2w.to(device)
    i, j = torch.meshgrid(torch.linspace(0, W - 1, W, device=device), torch.linspace(0, H - 1, H, device=device), indexing="ij")
    i, j = i.t(), j.t()
    dirs = torch.stack([(i - cx) / fx, -(j - cy) / fy, -torch.ones_like(i)], -1).reshape(H, W, 1, 3)
    rays_d = torch.sum(dirs * c2w[:3, :3], -1)
    rays_o = c2w[:3, -1].expand(rays_d.shape)
    return rays_o, rays_d
