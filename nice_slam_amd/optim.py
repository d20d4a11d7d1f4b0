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
                 betas=(0.9, 0.999), eps: float = 1e-8, capturable: bool = False):
        """``capturable=True``: the step counts live on the device and ALL grids of a step are updated by one launch pair
        (``nsr_masked_adam_multi``), so ``step`` changes no host-side scalar and a whole mapping iteration can be replayed
        from a hipGraph (``nice_slam_amd.graphs.CapturedStep``); the learning rates passed to ``step`` are baked into the
        capture (one captured graph per stage, like the reference's per-stage learning rates, Mapper.py:412-416)."""
        self.grids = grids
        self.betas, self.eps = betas, eps
        self.capturable = capturable
        self.state = {}
        self.masks = {}
        self._dev_steps = None
        self._scratch = None
        for k, g in grids.items():
            _require_cuda(g, f"MaskedGridAdam: {k}")
            if not g.is_contiguous(memory_format=torch.channels_last_3d) or g.dtype != torch.float32 or g.shape[1] != 32:
                raise _capi.NsrError(f"{k}: expected an fp32 channels-last [1,32,Z,Y,X] grid (nice_slam_amd.grid_init)")
            m = None if masks is None else masks.get(k)
            if m is not None:
                if tuple(m.shape) != tuple(g.shape[2:]):
                    raise _capi.NsrError(f"{k}: voxel mask shape {tuple(m.shape)} != grid {tuple(g.shape[2:])}")
                m = m.to(device=g.device, dtype=torch.uint8).contiguous()
            elif capturable:                 # a mask buffer from the start ("every voxel" = all ones): a captured graph holds its
                m = torch.ones(tuple(g.shape[2:]), dtype=torch.uint8, device=g.device)      # address, set_masks() only rewrites it
            self.masks[k] = m
            self.state[k] = {"step": 0, "exp_avg": torch.zeros_like(g, memory_format=torch.preserve_format),
                             "exp_avg_sq": torch.zeros_like(g, memory_format=torch.preserve_format)}

    def reset_state(self):
        """Zero the moments and the step counts in place (buffers and captured graphs stay valid)."""
        for st in self.state.values():
            st["step"] = 0
            st["exp_avg"].zero_()
            st["exp_avg_sq"].zero_()
        if self._dev_steps is not None:
            self._dev_steps.zero_()

    def set_masks(self, masks: Optional[Dict[str, Optional[torch.Tensor]]], reset_state: bool = True):
        """New voxel masks for a new mapping frame (its frustum, Mapper.py:315-318).  The reference builds a fresh
        ``torch.optim.Adam`` inside every ``optimize_map`` call (Mapper.py:368-379), so the Adam state starts from zero with
        every frame: ``reset_state=True`` (default) does the same in place; ``False`` keeps the moments.  A persistent
        gradient buffer that is cleared by ``step(zero_grad=True)`` only inside the OLD mask must be zeroed by the caller when
        the mask changes."""
        for k, g in self.grids.items():
            m = None if masks is None else masks.get(k)
            if m is not None and tuple(m.shape) != tuple(g.shape[2:]):
                raise _capi.NsrError(f"{k}: voxel mask shape {tuple(m.shape)} != grid {tuple(g.shape[2:])}")
            if self.capturable:                                                         # same buffer: captured graphs stay valid
                if m is None:
                    self.masks[k].fill_(1)
                else:
                    self.masks[k].copy_(m.to(device=g.device, dtype=torch.uint8))
                continue
            self.masks[k] = None if m is None else m.to(device=g.device, dtype=torch.uint8).contiguous()
        if reset_state:
            self.reset_state()

    def _step_multi(self, lrs, grads, zero_grad):
        lib = _capi.get_lib()
        keys = list(self.grids)
        dev = self.grids[keys[0]].device
        if self._dev_steps is None:
            self._dev_steps = torch.zeros((len(keys),), dtype=torch.int32, device=dev)
            self._scratch = torch.zeros((8,), dtype=torch.float32, device=dev)
        todo = []
        for i, k in enumerate(keys):
            g = self.grids[k]
            grad = (grads or {}).get(k) if grads is not None else g.grad
            if grad is None:
                continue
            if not grad.is_contiguous(memory_format=torch.channels_last_3d):
                raise _capi.NsrError(f"{k}: the capturable step needs a channels-last gradient (what render backward produces)")
            todo.append((i, k, grad))
        for lo in range(0, len(todo), 4):
            part = todo[lo:lo + 4]
            arr = (_capi.NsrAdamGrid * len(part))()
            for j, (i, k, grad) in enumerate(part):
                g, st, m = self.grids[k], self.state[k], self.masks[k]
                arr[j].p, arr[j].g, arr[j].m, arr[j].v = g.data_ptr(), grad.data_ptr(), st["exp_avg"].data_ptr(), st["exp_avg_sq"].data_ptr()
                arr[j].voxel_mask = None if m is None else m.data_ptr()
                arr[j].n_voxels = g.shape[2] * g.shape[3] * g.shape[4]
                arr[j].step = self._dev_steps.data_ptr() + 4 * i
                arr[j].lr = float(lrs.get(k, 0.0))
            lib.check(lib.nsr_masked_adam_multi(arr, len(part), self.betas[0], self.betas[1], self.eps, 1 if zero_grad else 0,
                                                self._scratch.data_ptr(), _stream(dev)), "nsr_masked_adam_multi")

    def step(self, lrs: Dict[str, float], grads: Optional[Dict[str, Optional[torch.Tensor]]] = None, zero_grad: bool = False):
        """One Adam step for every grid that has a gradient (``grads[key]`` or ``grid.grad``); grids without one are
        skipped entirely, like torch.optim.Adam skips parameters whose ``.grad`` is None.  ``zero_grad`` (capturable mode):
        also clear the gradient of the voxels that were updated."""
        if self.capturable:
            return self._step_multi(lrs, grads, zero_grad)
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
