"""TEST INFRASTRUCTURE: drive the C ABI with host (numpy) buffers.

Used with tests/emu/libnsr_emu.so -- the kernel sources compiled for the CPU against the fiber shim
in tests/emu/ -- so that kernel logic is unit-tested without a GPU.  The product package never
imports this.
"""
import ctypes as C
import os
import subprocess

import numpy as np
import torch

from nice_slam_amd import _capi
from nice_slam_amd.layout import param_spec, stage_slots

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
EMU_DIR = os.path.join(ROOT, "tests", "emu")
EMU_LIB = os.path.join(EMU_DIR, "libnsr_emu.so")


def build_emu(force=False):
    srcs = [os.path.join(ROOT, "nice_slam_amd", "csrc", f) for f in ("nsr_api.cpp", "nsr_kernels.h", "nsr_bwd2.h", "nsr_fwd2.h", "nsr_layout.h")]
    srcs += [os.path.join(EMU_DIR, f) for f in ("nsr_dev.h", "nsr_rt.h", "emu_runtime.cpp", "build_emu.sh")]
    srcs += [os.path.join(ROOT, "include", "nsr.h")]
    if not force and os.path.exists(EMU_LIB) and all(os.path.getmtime(EMU_LIB) >= os.path.getmtime(s) for s in srcs):
        return EMU_LIB
    subprocess.run([os.path.join(EMU_DIR, "build_emu.sh")], check=True, capture_output=True)
    return EMU_LIB


_emu = None


def emu_lib():
    global _emu
    if _emu is None:
        _emu = _capi.Lib(build_emu())
    return _emu


def ptr(a):
    return None if a is None else a.ctypes.data_as(C.c_void_p)


def flat_params(P, slot):
    return np.concatenate([P[f"{slot}_decoder.{n}"].detach().numpy().reshape(-1).astype(np.float32)
                           for n, _ in param_spec(slot)])


def unflat_grads(flat, slot):
    out, off = {}, 0
    for n, shp in param_spec(slot):
        k = int(np.prod(shp))
        out[f"{slot}_decoder.{n}"] = flat[off:off + k].reshape(shp)
        off += k
    return out


class HostScene:
    """Host-side mirror of what nice_slam_amd.renderer does with torch tensors, on numpy arrays."""

    def __init__(self, lib, grids, P, bound, coarse_enlarge=2.0, n_samples=32, n_surface=16):
        self.lib = lib
        self.bound = np.asarray(bound, dtype=np.float64)
        self.enl = float(coarse_enlarge)
        self.n_samples, self.n_surface = n_samples, n_surface
        self.save_z = True          # the forward saves the sample depths (required by the backward)
        # True (default): the forward gets an activation buffer (three-launch forward, nsr_fwd2.h); False: the one-launch forward
        # kernel, as for calls that will not be differentiated -- backward() then runs the saving forward first, like
        # nice_slam_amd.renderer._chunked_backward does for batches whose buffer does not fit
        self.save_acts = os.environ.get("NSR_EMU_SAVE_ACTS", "1") == "1"
        self.grids = {}
        for k, v in grids.items():          # [1,32,Z,Y,X] -> [Z,Y,X,32] contiguous
            a = v.detach().numpy()[0].transpose(1, 2, 3, 0)
            self.grids[k[len("grid_"):]] = np.ascontiguousarray(a, dtype=np.float32)
        self.flat = {s: flat_params(P, s) for s in _capi.SLOT_NAMES if f"{s}_decoder.output_linear.weight" in P}
        self.packed = {}
        for s, f in self.flat.items():
            slot = _capi.SLOT_NAMES.index(s)
            assert lib.nsr_param_count(slot) == f.size, (s, f.size)
            pk = np.full(lib.nsr_packed_count(slot), np.nan, dtype=np.float32)
            lib.check(lib.nsr_pack_params(slot, ptr(f), ptr(pk), None), "pack")
            self.packed[s] = pk

    def _args(self, stage, rays_o, rays_d, gt_depth, keep):
        a = _capi.NsrRenderArgs()
        a.stage = _capi.STAGE_ID[stage]
        a.n_samples, a.n_surface = self.n_samples, self.n_surface
        n = rays_o.shape[0]
        a.n_rays = n
        a.rays_o, a.rays_d = ptr(rays_o), ptr(rays_d)
        if gt_depth is not None:
            gmax = np.array([gt_depth.max()], dtype=np.float32)
            keep.append(gmax)
            a.gt_depth, a.gt_max = ptr(gt_depth), ptr(gmax)
        for i in range(3):
            a.bound_lo[i], a.bound_hi[i] = self.bound[i, 0], self.bound[i, 1]
        tu = torch.linspace(0.0, 1.0, self.n_samples).numpy()
        ts = torch.linspace(0.0, 1.0, max(self.n_surface, 1)).double().numpy()
        for i in range(self.n_samples):
            a.t_uniform[i] = float(tu[i])
        for i in range(self.n_surface):
            a.t_surface[i] = float(ts[i])
        for s in stage_slots(stage):
            i = _capi.SLOT_NAMES.index(s)
            g = self.grids[s]
            a.grid[i].feat = ptr(g)
            a.grid[i].Z, a.grid[i].Y, a.grid[i].X = g.shape[:3]
            b = self.bound * (self.enl if s == "coarse" else 1.0)
            for d in range(3):
                a.grid[i].lo[d], a.grid[i].hi[d] = b[d, 0], b[d, 1]
            a.dec[i].params, a.dec[i].packed = ptr(self.flat[s]), ptr(self.packed[s])
        return a

    def forward(self, stage, rays_o, rays_d, gt_depth, fused_loss=None):
        """fused_loss: None or {"gt_color": [n,3], "keep": uint8 [n] or None, "w_color": float}: the mapper's L1 loss in the
        forward (out["loss"], out["dl_depth"], out["dl_rgb"]); with an activation buffer the forward then also leaves d raw for
        the split backward (backward(..., from_forward=True) hands the same arrays back)."""
        keep = []
        rays_o = np.ascontiguousarray(rays_o, dtype=np.float32)
        rays_d = np.ascontiguousarray(rays_d, dtype=np.float32)
        gt = None if gt_depth is None else np.ascontiguousarray(gt_depth, dtype=np.float32)
        a = self._args(stage, rays_o, rays_d, gt, keep)
        n = rays_o.shape[0]
        guided = gt is not None and stage != "coarse"
        S = self.n_samples + (self.n_surface if guided else 0)
        out = {"depth": np.full(n, np.nan), "var": np.full(n, np.nan),
               "rgb": np.full((n, 3), np.nan, dtype=np.float32), "raw": np.full((n, S, 4), np.nan, dtype=np.float32)}
        a.depth, a.var, a.rgb, a.raw = ptr(out["depth"]), ptr(out["var"]), ptr(out["rgb"]), ptr(out["raw"])
        if self.save_z:
            out["zvals"] = np.full((n, S), np.nan)
            a.zvals = ptr(out["zvals"])
        if self.save_acts:                                 # nsr_render_args.acts: split backward over saved activations
            nfl = self.lib.nsr_acts_floats(_capi.STAGE_ID[stage], n, S)
            out["acts"] = np.full((max(1, nfl),), np.nan, dtype=np.float32)
            a.acts = ptr(out["acts"])
            m = getattr(self, "acts_masks_only", False)         # True / False, or the bit field of nsr_render_args.acts_masks_only
            a.acts_masks_only = int(m) if not isinstance(m, bool) else (1 if m else 0)
        if fused_loss is not None:
            out["loss"], out["dl_depth"], out["dl_rgb"] = np.zeros(1), np.full(n, np.nan), np.full((n, 3), np.nan, dtype=np.float32)
            gcol = np.ascontiguousarray(fused_loss["gt_color"], dtype=np.float32)
            kp = None if fused_loss.get("keep") is None else np.ascontiguousarray(fused_loss["keep"], dtype=np.uint8)
            keep += [gcol, kp]
            a.gt_color, a.keep, a.loss, a.w_color = ptr(gcol), ptr(kp), ptr(out["loss"]), float(fused_loss.get("w_color", 0.2))
            a.dl_depth, a.dl_rgb = ptr(out["dl_depth"]), ptr(out["dl_rgb"])
            a.skip_masked = 1 if fused_loss.get("skip_masked") else 0        # masked rays removed from the batch (nsr_render_args.skip_masked)
        self.lib.check(self.lib.nsr_render_fwd(C.byref(a), None), "fwd")
        out["_ctx"] = (a, keep, rays_o, rays_d, gt, S)
        return out

    def backward(self, stage, fwd, d_depth, d_var, d_rgb, want_grid=True, want_params=True, want_rays=True, max_blocks=0,
                 overwrite_dparams=False, grad_scale=None, from_forward=False, in_place=False, grad_voxel_masks=None):
        """grad_voxel_masks: None or {slot: uint8 [Z][Y][X]} -- nsr_render_args.grad_voxel_mask (consumed-gradient masks)"""
        if "acts" not in fwd:                              # forward without an activation buffer: run the saving forward now
            assert not from_forward and not in_place
            was, self.save_acts = self.save_acts, True
            try:
                fwd2 = self.forward(stage, fwd["_ctx"][2], fwd["_ctx"][3], fwd["_ctx"][4])
            finally:
                self.save_acts = was
            for k in ("depth", "var", "rgb", "raw"):       # the two forward implementations share every expression
                assert np.array_equal(fwd2[k], fwd[k]), k
            fwd = fwd2
        a, keep, rays_o, rays_d, gt, S = fwd["_ctx"]
        n = rays_o.shape[0]
        res = {}
        for s in stage_slots(stage):
            i = _capi.SLOT_NAMES.index(s)
            a.grid[i].dfeat = None
            a.dec[i].dparams = None
            gm = None if grad_voxel_masks is None else grad_voxel_masks.get(s)
            if gm is not None:
                gm = np.ascontiguousarray(gm, dtype=np.uint8)
                assert gm.shape == self.grids[s].shape[:3], (gm.shape, self.grids[s].shape)
                keep.append(gm)
            a.grad_voxel_mask[i] = ptr(gm)
            if want_grid:
                res["d_grid_" + s] = np.zeros_like(self.grids[s])
                a.grid[i].dfeat = ptr(res["d_grid_" + s])
            if want_params is True or (want_params and s in want_params):       # True / False, or the slots whose decoders want them
                res["d_flat_" + s] = np.full_like(self.flat[s], np.nan) if overwrite_dparams else np.zeros_like(self.flat[s])
                a.dec[i].dparams = ptr(res["d_flat_" + s])
        b = _capi.NsrBwdArgs()
        if from_forward:            # exactly the derivative arrays the fused forward wrote (same pointers: no comp_bwd launch)
            dd, dv, dr = fwd["dl_depth"], None, (fwd["dl_rgb"] if stage == "color" else None)
            b.loss_grads_from_forward = 1
        elif in_place:              # the forward's own arrays, same pointers, but EDITED by the caller: the flag stays 0
            dd, dv, dr = fwd["dl_depth"], None, (fwd["dl_rgb"] if stage == "color" else None)
        else:
            dd = np.ascontiguousarray(d_depth, dtype=np.float64)
            dv = None if d_var is None else np.ascontiguousarray(d_var, dtype=np.float64)
            dr = None if d_rgb is None else np.ascontiguousarray(d_rgb, dtype=np.float32)
        b.d_depth, b.d_var, b.d_rgb, b.depth = ptr(dd), ptr(dv), ptr(dr), ptr(fwd["depth"])
        if want_rays:
            res["d_rays_o"] = np.zeros((n, 3), dtype=np.float32)
            res["d_rays_d"] = np.zeros((n, 3), dtype=np.float32)
            b.d_rays_o, b.d_rays_d = ptr(res["d_rays_o"]), ptr(res["d_rays_d"])
        nws = self.lib.nsr_bwd_workspace_floats(_capi.STAGE_ID[stage], n, S, max_blocks)
        ws = np.full(max(nws, 1), np.nan, dtype=np.float32)
        self.last_ws = ws
        b.workspace, b.workspace_floats, b.max_blocks = ptr(ws), nws, max_blocks
        b.overwrite_dparams = 1 if overwrite_dparams else 0
        gs = None if grad_scale is None else np.array([grad_scale], dtype=np.float64)
        b.grad_scale = ptr(gs)
        self.lib.check(self.lib.nsr_render_bwd(C.byref(a), C.byref(b), None), "bwd")
        for s in stage_slots(stage):       # back to reference layouts
            if want_grid:
                res["d_grid_" + s] = res["d_grid_" + s].transpose(3, 0, 1, 2)[None]
            if "d_flat_" + s in res:
                res.update({"dparam/" + k: v for k, v in unflat_grads(res.pop("d_flat_" + s), s).items()})
        return res
