"""Frustum feature selection on the device.

``Mapper.get_mask_from_c2w`` (src/Mapper.py:93-164) projects every voxel centre of a feature grid into the current frame
with numpy, looks its depth up with ``cv2.remap`` on the CPU and returns a boolean voxel mask, once per grid per
``optimize_map`` call.  ``FrustumSelector`` keeps that on the GPU: one ``nsr_frustum_mask`` call per grid, the mask
stays in HBM as the [Z,Y,X] byte array ``MaskedGridAdam`` and ``ShardedRenderer.set_voxel_masks`` consume.
"""
from __future__ import annotations

import ctypes as C
from typing import Dict, Tuple

import numpy as np
import torch

from . import _capi
from .common import _require_cuda, _stream


class FrustumSelector:
    """``bound``: (3,2) un-enlarged scene bound (``slam.bound``); H, W, fx, fy, cx, cy: the cropped-frame intrinsics
    the mapper holds (Mapper.py:91)."""

    def __init__(self, bound, H: int, W: int, fx: float, fy: float, cx: float, cy: float):
        self.bound = [[float(bound[a][0]), float(bound[a][1])] for a in range(3)]
        self.H, self.W, self.fx, self.fy, self.cx, self.cy = int(H), int(W), float(fx), float(fy), float(cx), float(cy)
        self._axes: Dict[Tuple[Tuple[int, int, int], str], torch.Tensor] = {}

    def _voxel_axes(self, shape, device) -> torch.Tensor:
        key = (tuple(shape), str(device))
        ax = self._axes.get(key)
        if ax is None:                                 # Mapper.py:111-113: fp32 torch.linspace per axis, X | Y | Z
            nz, ny, nx = shape
            ax = torch.cat([torch.linspace(self.bound[0][0], self.bound[0][1], nx),
                            torch.linspace(self.bound[1][0], self.bound[1][1], ny),
                            torch.linspace(self.bound[2][0], self.bound[2][1], nz)]).to(device)
            self._axes[key] = ax
        return ax

    def voxel_mask(self, c2w, key: str, val_shape, depth: torch.Tensor) -> torch.Tensor:
        """uint8 [Z,Y,X] mask on ``depth.device``.  ``c2w``: (4,4) camera-to-world (tensor or array; read on the host,
        like the reference's ``c2w.cpu().numpy()``); ``val_shape`` = grid.shape[2:]; ``depth``: (H,W) fp32 device tensor."""
        nz, ny, nx = (int(v) for v in val_shape)
        _require_cuda(depth, "FrustumSelector: depth")
        dev = depth.device
        if key == "grid_coarse":                       # Mapper.py:116-118: the coarse grid is always fully selected
            return torch.ones((nz, ny, nx), dtype=torch.uint8, device=dev)
        if tuple(depth.shape) != (self.H, self.W):
            raise _capi.NsrError(f"FrustumSelector: depth shape {tuple(depth.shape)} != ({self.H}, {self.W})")
        depth = depth.detach().to(torch.float32).contiguous()
        c2w = np.asarray(c2w.detach().cpu().numpy() if isinstance(c2w, torch.Tensor) else c2w, dtype=np.float32)
        if c2w.shape != (4, 4):
            raise _capi.NsrError(f"FrustumSelector: c2w must be 4x4 (got {c2w.shape})")
        w2c = np.ascontiguousarray(np.linalg.inv(c2w)[:3], dtype=np.float32)         # Mapper.py:119-120
        o = np.ascontiguousarray(c2w[:3, 3], dtype=np.float32)
        ax = self._voxel_axes((nz, ny, nx), dev)
        lib = _capi.get_lib()
        n = nx * ny * nz
        ws = torch.empty((lib.nsr_frustum_workspace_floats(n),), dtype=torch.float32, device=dev)
        mask = torch.empty((nz, ny, nx), dtype=torch.uint8, device=dev)
        fp = lambda a: a.ctypes.data_as(C.POINTER(C.c_float))
        p0 = ax.data_ptr()
        lib.check(lib.nsr_frustum_mask(fp(w2c), fp(o), self.fx, self.fy, self.cx, self.cy, self.H, self.W, depth.data_ptr(),
                                       p0, p0 + 4 * nx, p0 + 4 * (nx + ny), nx, ny, nz, ws.data_ptr(), mask.data_ptr(),
                                       _stream(dev)), "nsr_frustum_mask")
        return mask

    def get_mask_from_c2w(self, c2w, key: str, val_shape, depth: torch.Tensor) -> torch.Tensor:
        """Reference-shaped result: bool (X,Y,Z) tensor (a transposed view of the device mask), so the caller's
        ``mask.permute(2,1,0)`` (Mapper.py:318) lands on the grid's own [Z,Y,X] order."""
        return self.voxel_mask(c2w, key, val_shape, depth).permute(2, 1, 0).bool()
