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


class FlatAdam:
    """torch.optim.Adam (defaults: no weight decay, no amsgrad) for the dense rest of the callers' optimiser -- the decoder
    parameters and the camera tensors (src/Mapper.py:368-387,504; src/Tracker.py:214-222,127) -- as ONE launch pair per step with
    the step counts on the device (capturable: nothing on the host changes between replays of a captured iteration).

    ``entries``: up to four of
      * a decoder of ``nice_slam_amd.NICE`` (``decoders.color_decoder`` ...): its flat parameter blob is stepped with the flat gradient
        blob the render backward writes (all of its parameters at once, like a param group holding ``decoder.parameters()``);
      * a contiguous fp32 leaf tensor on the GPU (a pose tensor ``[7]`` / ``[n, 7]``): stepped with its ``.grad``.
    ``lr``: one float or one per entry (host values: a captured step has them baked in, one captured graph per stage like the
    reference's per-stage learning rates).  An entry without a gradient is skipped, like torch skips parameters whose ``.grad`` is
    None; ``lr = 0`` still updates the moments.  Same numbers as torch.optim.Adam to rounding (tests/test_hip_callers.py)."""

    def __init__(self, entries, lr=1e-3, betas=(0.9, 0.999), eps: float = 1e-8):
        from .decoders import _FlatDecoder
        self.entries = list(entries)
        if not 1 <= len(self.entries) <= 4:
            raise _capi.NsrError("FlatAdam: one to four entries (use several FlatAdam objects for more)")
        self.betas, self.eps = (float(betas[0]), float(betas[1])), float(eps)
        self.lr = self._lrs(lr)
        self._is_dec = [isinstance(e, _FlatDecoder) for e in self.entries]
        flats = [e.flat_params() if d else e for e, d in zip(self.entries, self._is_dec)]
        for f in flats:
            _require_cuda(f, "FlatAdam: parameter")
            if f.dtype != torch.float32 or not f.is_contiguous():
                raise _capi.NsrError("FlatAdam: contiguous fp32 tensors only")
        dev = flats[0].device
        if any(f.device != dev for f in flats):
            raise _capi.NsrError("FlatAdam: all entries must live on one device")
        self._ptrs = [f.data_ptr() for f in flats]
        self.state = [{"exp_avg": torch.zeros_like(f), "exp_avg_sq": torch.zeros_like(f)} for f in flats]
        self._steps = torch.zeros((len(flats),), dtype=torch.int32, device=dev)
        self._scratch = torch.zeros((8,), dtype=torch.float32, device=dev)

    def _lrs(self, lr):
        lrs = [float(v) for v in lr] if isinstance(lr, (list, tuple)) else [float(lr)] * len(self.entries)
        if len(lrs) != len(self.entries):
            raise _capi.NsrError("FlatAdam: %d learning rates for %d entries" % (len(lrs), len(self.entries)))
        return lrs

    def reset_state(self):
        """Zero moments and step counts in place (a fresh optimiser per frame, Tracker.py:214-222, without new buffers)."""
        torch._foreach_zero_([t for st in self.state for t in st.values()] + [self._steps])

    def _grad(self, i):
        e = self.entries[i]
        if not self._is_dec[i]:
            return e.grad
        tgt, mode = e.grad_target()
        if mode == "accumulate":                     # every .grad is the cached view of the decoder's gradient blob
            return tgt
        if mode == "overwrite":                      # every .grad is None: nothing to step
            return None
        gs = [p.grad for p in e.parameters()]        # foreign gradient tensors: one concatenation (not capturable-stable)
        if any(g is None for g in gs):
            return None
        if torch.cuda.is_available() and torch.cuda.is_current_stream_capturing():
            raise _capi.NsrError("FlatAdam: a decoder whose .grad tensors are not the views of its gradient blob cannot be stepped under "
                                 "graph capture (the concatenated copy gets a new address every step)")
        self._foreign = gs
        return torch.cat([g.reshape(-1) for g in gs])

    def step(self, lr=None, zero_grad: bool = False):
        lib = _capi.get_lib()
        lrs = self.lr if lr is None else self._lrs(lr)
        arr = (_capi.NsrAdamSpan * len(self.entries))()
        hold, n = [], 0
        dev = self._steps.device
        foreign = []
        for i, e in enumerate(self.entries):
            if self._is_dec[i]:
                # the backward writes -- and this step consumes -- the gradient of the WHOLE blob: torch would skip a frozen
                # parameter (its .grad stays None), a blob cannot
                rg = {p.requires_grad for p in e.parameters()}
                if len(rg) > 1:
                    raise _capi.NsrError("FlatAdam: entry %d mixes parameters with and without requires_grad; a decoder is stepped as "
                                         "one blob (freeze it as a whole, or leave it out of the optimiser)" % i)
            self._foreign = None
            g = self._grad(i)
            if g is None:
                continue
            if self._foreign is not None:
                foreign += self._foreign
            p = e.flat_params() if self._is_dec[i] else e
            if g.device != p.device or g.numel() != p.numel():
                raise _capi.NsrError("FlatAdam: gradient of entry %d does not match its parameter (device / size)" % i)
            if p.data_ptr() != self._ptrs[i]:
                raise _capi.NsrError("FlatAdam: the storage of entry %d moved since the optimiser was built (.to() / .data = ...): "
                                     "build a new FlatAdam" % i)
            g = g if (g.dtype == torch.float32 and g.is_contiguous()) else g.to(torch.float32).contiguous()
            hold.append(g)
            st = self.state[i]
            arr[n].p, arr[n].g, arr[n].m, arr[n].v = p.data_ptr(), g.data_ptr(), st["exp_avg"].data_ptr(), st["exp_avg_sq"].data_ptr()
            arr[n].n, arr[n].step, arr[n].lr = p.numel(), self._steps.data_ptr() + 4 * i, float(lrs[i])
            n += 1
            if self._is_dec[i]:
                e.mark_dirty()                       # the packed operand streams are re-packed before the next render
        if n:
            with _capi.on_device(dev):
                lib.check(lib.nsr_flat_adam(arr, n, self.betas[0], self.betas[1], self.eps, 1 if zero_grad else 0,
                                            self._scratch.data_ptr(), _stream(dev)), "nsr_flat_adam")
            if zero_grad and foreign:                # the kernel cleared the concatenated COPY: clear what the caller holds
                torch._foreach_zero_(foreign)

    def zero_grad(self, set_to_none: bool = True):
        for e, d in zip(self.entries, self._is_dec):
            for p in (e.parameters() if d else [e]):
                if set_to_none:
                    p.grad = None
                elif p.grad is not None:
                    p.grad.zero_()

