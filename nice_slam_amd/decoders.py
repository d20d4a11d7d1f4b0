"""Decoder modules with the reference's attribute / state_dict surface, stored as ONE flat fp32 blob
per decoder so the kernels can read them without any per-call gathering.

Reference: src/conv_onet/models/decoder.py (MLP :91-203, MLP_no_xyz :206-274, NICE :277-342).
The arithmetic lives in libnsr.so; these classes only own parameters.
"""
from __future__ import annotations

import math
from typing import List, Optional

import torch
import torch.nn as nn

from . import _capi
from .layout import param_spec


class _Holder(nn.Module):
    """A leaf module that only carries Parameters (stands in for nn.Linear / DenseLayer / the embedder)."""

    def forward(self, *a, **k):
        raise RuntimeError("nice_slam_amd decoders are evaluated by the fused HIP kernels; sub-layers are not callable")


class _FlatDecoder(nn.Module):
    """Common machinery: every Parameter is a view into ``self._flat`` (reference named_parameters() order)."""

    slot = ""

    def __init__(self, name: str):
        super().__init__()
        self.name = name
        self.bound: Optional[torch.Tensor] = None          # set by the system, src/NICE_SLAM.py:152-157
        self._spec = param_spec(self.slot)
        self._flat = torch.zeros(sum(math.prod(s) for _, s in self._spec), dtype=torch.float32)
        self._grad_flat: Optional[torch.Tensor] = None
        self._grad_views = None
        self._packed = None                                 # (key, tensor) cache of the MFMA operand stream
        self._cross_process = False                         # parameters may be written by another process (see packed_params)
        self._build_modules()
        self.reset_parameters()

    # -- structure ---------------------------------------------------------------------------------
    def _build_modules(self):
        names = [n for n, _ in self._spec]
        # attribute order == reference registration order, so named_parameters() matches decoder.py
        if any(n.startswith("fc_c.") for n in names):
            self.fc_c = nn.ModuleList([_Holder() for _ in range(5)])
        if "embedder._B" in names:
            self.embedder = _Holder()
        self.pts_linears = nn.ModuleList([_Holder() for _ in range(5)])
        self.output_linear = _Holder()
        off = 0
        self._views: List[nn.Parameter] = []
        self._offsets: List[int] = []
        for pname, shape in self._spec:
            n = math.prod(shape)
            p = nn.Parameter(self._flat[off:off + n].view(shape))
            mod = self
            parts = pname.split(".")
            for part in parts[:-1]:
                mod = mod[int(part)] if part.isdigit() else getattr(mod, part)
            mod.register_parameter(parts[-1], p)
            self._views.append(p)
            self._offsets.append(off)
            off += n

    def reset_parameters(self):
        """Same init statistics as the reference (DenseLayer xavier/zero bias decoder.py:75-79, default
        nn.Linear init for fc_c, randn*25 for the Fourier matrix :21-22)."""
        with torch.no_grad():
            for (pname, shape), p in zip(self._spec, self._views):
                if pname.startswith("fc_c"):
                    fan_in = self._spec_shape(pname.rsplit(".", 1)[0] + ".weight")[1]
                    bound = 1.0 / math.sqrt(fan_in)
                    if pname.endswith("weight"):
                        nn.init.kaiming_uniform_(p, a=math.sqrt(5))
                    else:
                        nn.init.uniform_(p, -bound, bound)
                elif pname == "embedder._B":
                    p.copy_(torch.randn(shape) * 25.0)
                elif pname.endswith("weight"):
                    gain = 1.0 if pname.startswith("output_linear") else nn.init.calculate_gain("relu")
                    nn.init.xavier_uniform_(p, gain=gain)
                else:
                    p.zero_()

    def _spec_shape(self, pname):
        return dict(self._spec)[pname]

    # -- flat storage ------------------------------------------------------------------------------
    def flat_params(self) -> torch.Tensor:
        """The flat blob; re-establishes the view relationship if something (``.to()``, deepcopy,
        ``share_memory``) replaced the Parameters' storage since the last call."""
        f = self._flat
        p0, pl = self._views[0], self._views[-1]
        ok = (p0.device == f.device and p0.data_ptr() == f.data_ptr()
              and pl.data_ptr() == f.data_ptr() + 4 * self._offsets[-1])
        if not ok:
            dev = p0.device
            with torch.no_grad():
                f = torch.cat([p.detach().reshape(-1).to(device=dev, dtype=torch.float32) for p in self._views])
            for p, off in zip(self._views, self._offsets):
                p.data = f[off:off + p.numel()].view(p.shape)
            self._flat = f
            self._grad_flat = None
            self._grad_views = None
            self._packed = None
        return self._flat

    def _pack_key(self):
        f = self.flat_params()
        # in-place updates (optimizer steps, load_state_dict) bump the version counter of the Parameter they touch --
        # not necessarily the one of the flat buffer (``p.data = view`` gives p its own counter) -- so the cache key
        # is the tuple of all of them
        return (f.data_ptr(), f.device, f._version, tuple(p._version for p in self._views))

    def packed_params(self, lib, stream) -> torch.Tensor:
        f = self.flat_params()
        key = self._pack_key()
        # a decoder shared with other processes (share_memory(), or unpickled into a spawned process: the reference's
        # tracker / mapper / coarse-mapper processes, src/NICE_SLAM.py:82-90,288-305) can be updated in place by a process
        # whose writes never touch THIS process's version counters: re-pack on every call (one small kernel)
        if self._packed is None or self._packed[0] != key or self._cross_process:
            slot = _capi.SLOT_NAMES.index(self.slot)
            # re-pack INTO the buffer of the last pack when there is one: a captured hipGraph keeps reading that address
            # (the pack launch is then either part of the graph -- parameters stepped inside it -- or issued by the caller
            # between replays: NICE.repack())
            pk = self._packed[1] if (self._packed is not None and self._packed[1].device == f.device) else \
                torch.empty(lib.nsr_packed_count(slot), dtype=torch.float32, device=f.device)
            lib.check(lib.nsr_pack_params(slot, f.data_ptr(), pk.data_ptr(), stream), "nsr_pack_params")
            self._packed = (key, pk)
        return self._packed[1]

    def mark_dirty(self):
        """The flat blob was written behind autograd's back (nice_slam_amd.optim.FlatAdam: a kernel through the raw pointer, no
        version counter moves): the next ``packed_params`` re-packs -- into the same buffer, so captured graphs stay valid."""
        if self._packed is not None:
            self._packed = (None, self._packed[1])

    def share_memory(self):
        self._cross_process = True
        return super().share_memory()

    def __getstate__(self):                                 # pickled into a spawned process: caches stay behind
        d = dict(self.__dict__)
        d["_packed"], d["_grad_flat"], d["_grad_views"] = None, None, None
        d["_cross_process"] = True
        return d

    def wants_grad(self) -> bool:
        return any(p.requires_grad for p in self._views)

    # -- parameter gradients ---------------------------------------------------------------------------
    # The backward kernel reduces a decoder's gradient into one flat blob.  Exposing it as ~30 fresh `.grad` views per
    # decoder per iteration (slice + view + assignment each) costs more host time than the kernel takes, so the blob is a
    # persistent buffer with its per-Parameter views built once:
    #   * every .grad is None (optimizer.zero_grad(set_to_none=True), the torch default): the kernel OVERWRITES the blob,
    #     the cached views are assigned afterwards;
    #   * every .grad already is its cached view (zero_grad(set_to_none=False), or deliberate accumulation): the kernel
    #     ACCUMULATES into the blob -- autograd's semantics -- and nothing is assigned;
    #   * anything else (foreign .grad tensors): the caller falls back to a temporary blob + publish_grads().
    def grad_target(self):
        """-> (flat gradient buffer, mode) with mode in {"overwrite", "accumulate"}, or (None, None) for the fallback.
        ``self.persistent_grads = False`` (default True) always takes the fallback: fresh gradient tensors per backward, like
        stock autograd -- for callers that keep references to ``.grad`` tensors across ``zero_grad(set_to_none=True)``."""
        if not getattr(self, "persistent_grads", True):
            return None, None
        f = self.flat_params()
        if self._grad_flat is None or self._grad_flat.device != f.device:
            self._grad_flat = torch.zeros_like(f)
            self._grad_views = [self._grad_flat[off:off + p.numel()].view(p.shape) for p, off in zip(self._views, self._offsets)]
        n = n_none = n_ours = 0
        for p, v in zip(self._views, self._grad_views):
            if p.requires_grad:
                n += 1
                g = p.grad
                if g is None:
                    n_none += 1
                elif g is v:
                    n_ours += 1
        if n_none == n:
            return self._grad_flat, "overwrite"
        if n_ours == n:
            return self._grad_flat, "accumulate"
        return None, None

    def grad_done(self, mode: str):
        if mode == "overwrite":
            for p, v in zip(self._views, self._grad_views):
                if p.requires_grad:
                    p.grad = v

    def publish_grads(self, gflat: torch.Tensor):
        """Expose a freshly computed flat gradient as the ``.grad`` of every Parameter (views, no copies)."""
        for p, off in zip(self._views, self._offsets):
            if not p.requires_grad:
                continue
            g = gflat[off:off + p.numel()].view(p.shape)
            if p.grad is None:
                p.grad = g
            else:
                p.grad = p.grad + g

    def __deepcopy__(self, memo):
        # Tracker.update_para_from_mapping deep-copies the decoders (src/Tracker.py:138)
        new = self.__class__.__new__(self.__class__)
        nn.Module.__init__(new)
        new.name, new.bound = self.name, (None if self.bound is None else self.bound.clone())
        new._spec = self._spec
        new._flat = self.flat_params().detach().clone()
        new._grad_flat, new._grad_views, new._packed, new._cross_process = None, None, None, False
        new._build_modules()
        for a, b in zip(new._views, self._views):
            a.requires_grad_(b.requires_grad)
        for k, v in self.__dict__.items():
            if k not in new.__dict__ and not k.startswith("_"):
                new.__dict__[k] = v
        return new

    def forward(self, p, c_grid=None, **kwargs):
        raise RuntimeError("call NICE.forward / Renderer.eval_points: single decoders are evaluated inside the fused kernels")


class MLP(_FlatDecoder):
    """decoder.py:91-203 (c_dim 32 or 64, 93-wide Gaussian Fourier embedding, 5 blocks, skip at 2)."""

    def __init__(self, name="", dim=3, c_dim=32, hidden_size=32, n_blocks=5, leaky=False, sample_mode="bilinear",
                 color=False, skips=(2,), grid_len=0.16, pos_embedding_method="fourier", concat_feature=False):
        if (dim, hidden_size, n_blocks, tuple(skips), pos_embedding_method, sample_mode, leaky) != \
                (3, 32, 5, (2,), "fourier", "bilinear", False):
            raise NotImplementedError("only the NICE-SLAM decoder configuration is implemented (SURVEY §2: iMAP* out of scope)")
        self.slot = "color" if color else ("fine" if concat_feature else "middle")
        if c_dim != (64 if concat_feature else 32):
            raise NotImplementedError("c_dim must be 32 (64 for the concat-feature fine decoder)")
        self.color, self.c_dim, self.grid_len, self.concat_feature = color, c_dim, grid_len, concat_feature
        self.n_blocks, self.skips, self.no_grad_feature, self.sample_mode = n_blocks, list(skips), False, sample_mode
        super().__init__(name)


class MLP_no_xyz(_FlatDecoder):
    """decoder.py:206-274 (the coarse decoder)."""
    slot = "coarse"

    def __init__(self, name="", dim=3, c_dim=32, hidden_size=32, n_blocks=5, leaky=False, sample_mode="bilinear",
                 color=False, skips=(2,), grid_len=0.16):
        if (dim, c_dim, hidden_size, n_blocks, tuple(skips), color, leaky) != (3, 32, 32, 5, (2,), False, False):
            raise NotImplementedError("only the NICE-SLAM coarse decoder configuration is implemented")
        self.color, self.c_dim, self.grid_len = color, c_dim, grid_len
        self.n_blocks, self.skips, self.no_grad_feature, self.sample_mode = n_blocks, list(skips), False, sample_mode
        super().__init__(name)


class NICE(nn.Module):
    """decoder.py:277-342.  ``forward(p, c_grid, stage)`` is the point query used by the reference's
    Renderer.eval_points / Mesher.eval_points: (1,M,3) or (M,3) points -> (M,4) [rgb, occupancy]."""

    def __init__(self, dim=3, c_dim=32, coarse_grid_len=2.0, middle_grid_len=0.16, fine_grid_len=0.16,
                 color_grid_len=0.16, hidden_size=32, coarse=False, pos_embedding_method="fourier"):
        super().__init__()
        self.bound: Optional[torch.Tensor] = None
        if coarse:
            self.coarse_decoder = MLP_no_xyz(name="coarse", dim=dim, c_dim=c_dim, color=False,
                                             hidden_size=hidden_size, grid_len=coarse_grid_len)
        self.middle_decoder = MLP(name="middle", dim=dim, c_dim=c_dim, color=False, hidden_size=hidden_size,
                                  grid_len=middle_grid_len, pos_embedding_method=pos_embedding_method)
        self.fine_decoder = MLP(name="fine", dim=dim, c_dim=c_dim * 2, color=False, hidden_size=hidden_size,
                                grid_len=fine_grid_len, concat_feature=True, pos_embedding_method=pos_embedding_method)
        self.color_decoder = MLP(name="color", dim=dim, c_dim=c_dim, color=True, hidden_size=hidden_size,
                                 grid_len=color_grid_len, pos_embedding_method=pos_embedding_method)

    def sub(self, slot: str) -> _FlatDecoder:
        return getattr(self, slot + "_decoder")

    def repack(self):
        """Refresh the packed operand streams after the parameters were written by something the version counters of this
        process do not see, or between replays of a captured graph that does not contain the pack launch (a tracker-side
        copy refreshed from the mapper's decoders, src/Tracker.py:130-142).  In place: captured graphs stay valid."""
        from .common import _stream
        lib = _capi.get_lib()
        for m in self.children():
            if isinstance(m, _FlatDecoder) and m.flat_params().is_cuda:
                with _capi.on_device(m.flat_params().device):   # (a cached pack makes no library call: nothing else would restore the device)
                    m.packed_params(lib, _stream(m.flat_params().device))

    def share_memory(self):                                  # src/NICE_SLAM.py:88-90
        for m in self.children():
            if isinstance(m, _FlatDecoder):
                m._cross_process = True
        return super().share_memory()

    def forward(self, p, c_grid, stage="middle", **kwargs):
        from .renderer import eval_points_raw          # local import: renderer imports this module
        if torch.is_grad_enabled() and (p.requires_grad or any(v.requires_grad for v in c_grid.values())):
            raise RuntimeError("NICE.forward is forward-only; differentiate through Renderer.render_batch_ray")
        return eval_points_raw(p.reshape(-1, 3), self, c_grid, stage, None)
