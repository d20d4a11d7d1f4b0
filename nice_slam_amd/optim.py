"""Fused grid update for the mapping loop.

The reference optimises, per feature grid, a 1-D leaf ``val_grad = val[mask]`` with torch.optim.Adam and copies it into /
out of the dense grid with boolean-mask ``index_put`` twice per iteration (src/Mapper.py:303-333, 368-379, 394-401, 504,
511-519); each of those is a ``nonzero`` (host sync) plus a gather/scatter over millions of elements.
``MaskedGridAdam`` keeps the dense channels-last grid as the parameter and applies the same Adam arithmetic in place to
the masked voxels only -- one HBM-bound kernel per grid per step, no sync, no compaction.
"""
from __future__ import annotations

from typing import Dict, Optional

import torch

from . import _capi
from .common import _require_cuda, _stream, to_channels_last


class MaskedGridAdam:
    """``grids``: dict key -> [1,32,Z,Y,X] fp32 channels-last tensor (updated in place).
    ``masks``: dict key -> bool/uint8 voxel mask [Z,Y,X] (what ``Mapper.get_mask_from_c2w`` returns after the
    ``permute(2,1,0)`` of Mapper.py:318), or None for "every voxel".  Adam hyper-parameters default to torch's."""

    def __init__(self, grids: Dict[str, torch.Tensor], masks: Optional[Dict[str, Optional[torch.Tensor]]] = None,
                 betas=(0.9, 0.999), eps: float = 1e-8):
        self.grids = grids
        self.betas, self.eps = betas, eps
        self.state = {}
        self.masks = {}
        for k, g in grids.items():
            _require_cuda(g, f"MaskedGridAdam: {k}")
            if not g.is_contiguous(memory_format=torch.channels_last_3d) or g.dtype != torch.float32 or g.shape[1] != 32:
                raise _capi.NsrError(f"{k}: expected an fp32 channels-last [1,32,Z,Y,X] grid (nice_slam_amd.grid_init)")
            m = None if masks is None else masks.get(k)
            if m is not None:
                if tuple(m.shape) != tuple(g.shape[2:]):
                    raise _capi.NsrError(f"{k}: voxel mask shape {tuple(m.shape)} != grid {tuple(g.shape[2:])}")
                m = m.to(device=g.device, dtype=torch.uint8).contiguous()
            self.masks[k] = m
            self.state[k] = {"step": 0, "exp_avg": torch.zeros_like(g, memory_format=torch.preserve_format),
                             "exp_avg_sq": torch.zeros_like(g, memory_format=torch.preserve_format)}

    def step(self, lrs: Dict[str, float], grads: Optional[Dict[str, Optional[torch.Tensor]]] = None):
        """One Adam step for every grid that has a gradient (``grads[key]`` or ``grid.grad``); grids without one are
        skipped entirely, like torch.optim.Adam skips parameters whose ``.grad`` is None."""
        lib = _capi.get_lib()
        b1, b2 = self.betas
        for k, g in self.grids.items():
            grad = (grads or {}).get(k) if grads is not None else g.grad
            if grad is None:
                continue
            grad = to_channels_last(grad.detach())
            st = self.state[k]
            st["step"] += 1
            t = st["step"]
            mask = self.masks[k]
            n_vox = g.shape[2] * g.shape[3] * g.shape[4]
            lib.check(lib.nsr_masked_adam(g.data_ptr(), grad.data_ptr(), st["exp_avg"].data_ptr(), st["exp_avg_sq"].data_ptr(),
                                          None if mask is None else mask.data_ptr(), n_vox,
                                          float(lrs.get(k, 0.0)) / (1.0 - b1 ** t), b1, b2, self.eps, (1.0 - b2 ** t) ** 0.5,
                                          _stream(g.device)), "nsr_masked_adam")
