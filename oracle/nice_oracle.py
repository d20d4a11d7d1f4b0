"""CPU oracle for the NICE-SLAM volume-rendering hot path.  TEST INFRASTRUCTURE ONLY.

This module is a from-scratch restatement (torch-on-CPU, no custom kernels) of the
arithmetic the reference executes between ``get_samples`` and the three tensors returned by
``Renderer.render_batch_ray``.  It exists so that ``tests/``, ``__graft_entry__.smoke()`` and the
``cpu_baseline`` leg of ``bench.py`` have something to check the HIP path against on machines
where ``/root/reference`` is absent.  Nothing under ``nice_slam_amd/`` may import it.

Parity status: PINNED.  ``tests/golden/make_golden.py`` imports the unmodified reference modules
from ``/root/reference`` (CPU), runs forward + backward for every stage on a small seeded scene
and stores inputs/outputs/gradients in ``tests/golden/*.npz``; ``tests/test_oracle_golden.py``
replays those fixtures through this file.

Every function cites the reference lines it follows (paths relative to ``/root/reference``).
The dtype flow (which intermediate is fp32 and which fp64) is part of the contract and is
reproduced exactly; pass ``hi=lo=torch.float64`` to get an all-double "truth" evaluation that is
used to measure the fp32 noise floor of both the reference and the HIP path.
"""
from __future__ import annotations

import math
from typing import Dict, Optional, Tuple

import numpy as np
import torch

F64 = torch.float64
F32 = torch.float32

STAGES = ("coarse", "middle", "fine", "color")


# --------------------------------------------------------------------------------------------
# scene geometry: bound + grid shapes
# --------------------------------------------------------------------------------------------
def scene_bound(bound_cfg, scale: float = 1.0, bound_divisible: float = 0.32) -> torch.Tensor:
    """src/NICE_SLAM.py:137-150 (``load_bound``).

    The upper corner is enlarged so every axis length is a multiple of ``bound_divisible``.  The
    reference does the ``(int+1)*divisible`` step on a float64 tensor, truncating with ``.int()``.
    """
    b = torch.from_numpy(np.array(bound_cfg, dtype=np.float64) * scale)
    cells = ((b[:, 1] - b[:, 0]) / bound_divisible).int() + 1
    b[:, 1] = cells * bound_divisible + b[:, 0]
    return b


def grid_shapes(bound: torch.Tensor, grid_len: Dict[str, float], coarse_enlarge: float = 2.0,
                with_coarse: bool = True) -> Dict[str, Tuple[int, int, int]]:
    """src/NICE_SLAM.py:192-250 (``grid_init``): cells per axis = int(len/grid_len), stored (Z,Y,X)."""
    xyz_len = bound[:, 1] - bound[:, 0]
    out = {}
    names = (("coarse",) if with_coarse else ()) + ("middle", "fine", "color")
    for name in names:
        ext = xyz_len * coarse_enlarge if name == "coarse" else xyz_len
        nx, ny, nz = (int(v) for v in (ext / grid_len[name]).tolist())
        out["grid_" + name] = (nz, ny, nx)
    return out


def make_grids(shapes: Dict[str, Tuple[int, int, int]], c_dim: int = 32,
               generator: Optional[torch.Generator] = None) -> Dict[str, torch.Tensor]:
    """Random grids with the reference's init statistics (NICE_SLAM.py:223,232,239,247)."""
    out = {}
    for key, zyx in shapes.items():
        std = 1e-4 if key == "grid_fine" else 1e-2
        out[key] = torch.randn((1, c_dim) + tuple(zyx), generator=generator, dtype=F32) * std
    return out


def decoder_bounds(bound: torch.Tensor, coarse_enlarge: float = 2.0) -> Dict[str, torch.Tensor]:
    """src/NICE_SLAM.py:152-157: every decoder normalises with the scene bound; the coarse one with
    both corners multiplied by ``coarse_bound_enlarge``."""
    return {"coarse": bound * coarse_enlarge, "middle": bound, "fine": bound, "color": bound}


# --------------------------------------------------------------------------------------------
# decoder parameters (same state_dict keys as src/conv_onet/models/decoder.py)
# --------------------------------------------------------------------------------------------
def _xavier(out_f, in_f, gain, g):
    a = gain * math.sqrt(6.0 / (in_f + out_f))
    return (torch.rand((out_f, in_f), generator=g, dtype=F32) * 2 - 1) * a


def _default_linear(out_f, in_f, g):
    bound = 1.0 / math.sqrt(in_f)
    w = (torch.rand((out_f, in_f), generator=g, dtype=F32) * 2 - 1) * bound
    b = (torch.rand((out_f,), generator=g, dtype=F32) * 2 - 1) * bound
    return w, b


def init_decoder_params(seed: int = 0, c_dim: int = 32, hidden: int = 32, emb: int = 93,
                        with_coarse: bool = True, bias_noise: float = 0.0) -> Dict[str, torch.Tensor]:
    """Random decoder parameters with the shapes of SURVEY Appendix A / decoder.py:124-159,235-245.

    Statistics mimic ``DenseLayer.reset_parameters`` (xavier-uniform, zero bias, decoder.py:75-79),
    ``nn.Linear`` default init for ``fc_c`` and ``randn*25`` for the Fourier matrix (decoder.py:21-22);
    the exact random stream of the reference is *not* reproduced (golden fixtures carry the
    reference's own tensors).  ``bias_noise`` adds noise to the zero-initialised biases so tests
    exercise the bias paths.
    """
    g = torch.Generator().manual_seed(seed)
    relu_gain = math.sqrt(2.0)
    p: Dict[str, torch.Tensor] = {}

    def dense(prefix, out_f, in_f, gain):
        p[prefix + ".weight"] = _xavier(out_f, in_f, gain, g)
        p[prefix + ".bias"] = torch.randn((out_f,), generator=g, dtype=F32) * bias_noise

    if with_coarse:
        for i in range(5):
            dense(f"coarse_decoder.pts_linears.{i}", hidden, hidden + (c_dim if i == 3 else 0), relu_gain)
        dense("coarse_decoder.output_linear", 1, hidden, 1.0)
    for name, cd, n_out in (("middle", c_dim, 1), ("fine", 2 * c_dim, 1), ("color", c_dim, 4)):
        pre = name + "_decoder."
        for i in range(5):
            w, b = _default_linear(hidden, cd, g)
            p[pre + f"fc_c.{i}.weight"], p[pre + f"fc_c.{i}.bias"] = w, b
        p[pre + "embedder._B"] = torch.randn((3, emb), generator=g, dtype=F32) * 25.0
        dense(pre + "pts_linears.0", hidden, emb, relu_gain)
        for i in range(1, 5):
            dense(pre + f"pts_linears.{i}", hidden, hidden + (emb if i == 3 else 0), relu_gain)
        dense(pre + "output_linear", n_out, hidden, 1.0)
    return p


# --------------------------------------------------------------------------------------------
# a1/a2: pixel index -> ray   (src/common.py:74-134)
# --------------------------------------------------------------------------------------------
def pixel_rays(indices: torch.Tensor, H0: int, H1: int, W0: int, W1: int,
               fx: float, fy: float, cx: float, cy: float, c2w: torch.Tensor,
               depth: torch.Tensor, color: torch.Tensor):
    """``get_samples`` with the random draw factored out (common.py:92-134).

    ``indices`` are flat indices into the cropped ``(H1-H0, W1-W0)`` window, row-major — exactly what
    ``torch.randint(i.shape[0], (n,))`` produces at common.py:99.  Pixel coordinates are the fp32
    ``linspace`` values of common.py:117-120, i.e. integers stored as fp32.
    """
    w = W1 - W0
    col = (indices % w + W0).to(F32)          # "i" in the reference (x / u)
    row = (indices // w + H0).to(F32)         # "j" in the reference (y / v)
    d_crop = depth[H0:H1, W0:W1].reshape(-1)
    c_crop = color[H0:H1, W0:W1].reshape(-1, 3)
    s_depth = d_crop[indices]
    s_color = c_crop[indices]
    # common.py:82-88: camera-frame direction then dirs . R^T as product-then-sum over the last axis
    dirs = torch.stack([(col - cx) / fx, -(row - cy) / fy, -torch.ones_like(col)], -1)
    rays_d = torch.sum(dirs[:, None, :] * c2w[:3, :3], -1)
    rays_o = c2w[:3, -1].expand(rays_d.shape)
    return rays_o, rays_d, s_depth, s_color


# --------------------------------------------------------------------------------------------
# a4: sample depths along each ray   (src/utils/Renderer.py:88-170)
# --------------------------------------------------------------------------------------------
def sample_depths(rays_o: torch.Tensor, rays_d: torch.Tensor, gt_depth: Optional[torch.Tensor],
                  bound: torch.Tensor, stage: str, n_samples: int = 32, n_surface: int = 16,
                  hi=F64) -> torch.Tensor:
    """Returns z (N, S) in ``hi`` precision, S = n_samples (+ n_surface when depth-guided).

    Rounding order follows SURVEY D.2: ``near*(1-t)`` and ``0.95*gt`` / ``1.05*gt`` are formed in
    the dtype of ``gt_depth`` (fp32 in the reference) before being promoted.
    """
    if stage == "coarse":                                   # Renderer.py:88-89
        gt_depth = None
    lo_dt = rays_o.dtype
    with torch.no_grad():
        o = rays_o.detach().to(hi)[:, :, None]
        d = rays_d.detach().to(hi)[:, :, None]
        t = (bound.to(device=o.device, dtype=hi)[None] - o) / d     # (N,3,2)   Renderer.py:98-103
        far_bb = t.max(dim=2)[0].min(dim=1)[0][:, None] + 0.01
        dev = rays_o.device
        tv = torch.linspace(0.0, 1.0, n_samples, dtype=lo_dt).to(dev)          # CPU linspace values, like the host tables of the kernels
        if gt_depth is None:                                # Renderer.py:90-92,110-111
            near_part = 0.01 * (1.0 - tv)                   # fp32 scalar*tensor
            far = far_bb
            z = near_part.to(hi) + far * tv.to(hi)
            return z
        g = gt_depth.detach().reshape(-1, 1)
        gmax12 = torch.max(g * 1.2)                         # Renderer.py:109 (input dtype)
        far = torch.minimum(torch.maximum(far_bb, torch.zeros((), dtype=hi, device=dev)), gmax12.to(hi))
        near = g.repeat(1, n_samples) * 0.01                # Renderer.py:94-96 (input dtype)
        z_uni = (near * (1.0 - tv)).to(hi) + far * tv.to(hi)        # Renderer.py:154-155
        if n_surface == 0:
            return z_uni
        ts = torch.linspace(0.0, 1.0, n_surface, dtype=lo_dt).to(hi).to(dev)  # Renderer.py:132-133
        lo_edge = (0.95 * g).to(hi)                         # rounded in input dtype first
        hi_edge = (1.05 * g).to(hi)
        z_hit = lo_edge * (1.0 - ts) + hi_edge * ts         # Renderer.py:135-137
        gmax = torch.max(g).to(hi)                          # Renderer.py:144
        z_miss = (0.001 * (1.0 - ts) + gmax * ts)[None].expand_as(z_hit)   # Renderer.py:143-146
        z_surf = torch.where(g > 0, z_hit, z_miss)          # Renderer.py:128-150
        z, _ = torch.sort(torch.cat([z_uni, z_surf], -1), -1)      # Renderer.py:168-170
        return z


# --------------------------------------------------------------------------------------------
# a7: trilinear feature lookup   (decoder.py:168-175 + common.py:269-284 + ATen GridSampler.h)
# --------------------------------------------------------------------------------------------
# "index": the restatement below (explicit corner gathers; documents the ATen semantics the kernels reproduce).
# "grid_sample": call ATen's grid_sampler_3d exactly like the reference does -- same numbers (tests/test_oracle_golden.py),
# and the op mix of the reference's CPU path (grid_sampler_3d_backward is 59 % of it): bench.py's cpu_baseline
# leg runs in this mode.
TRILINEAR_IMPL = "index"


def trilinear(grid: torch.Tensor, p: torch.Tensor, bound: torch.Tensor, lo=F32) -> torch.Tensor:
    """``F.grid_sample(grid, vgrid, mode='bilinear', padding_mode='border', align_corners=True)`` for
    a ``[1,C,Z,Y,X]`` grid and points ``p`` (M,3) in world coordinates, restated with index ops.

    * normalisation to [-1,1] is done in the precision of ``p`` (fp64 in the reference) and only
      then rounded to ``lo`` (common.py:281-283, decoder.py:171);
    * un-normalisation ``((g+1)/2)*(n-1)``, border clipping with zero coordinate-gradient when
      clipped, floor, and the (corner+1 - u) weight form follow
      torch/include/ATen/native/GridSampler.h:27-36,58-60,66-85;
    * x indexes the last grid dim, y the middle, z the first spatial dim.
    Returns (M, C) in ``lo``.
    """
    _, C, Z, Y, X = grid.shape
    b = bound.to(device=p.device, dtype=p.dtype)
    g = ((p - b[:, 0]) / (b[:, 1] - b[:, 0])) * 2 - 1.0
    g = g.to(lo)
    if TRILINEAR_IMPL == "grid_sample":                      # the reference's own call (decoder.py:171-175), verbatim
        vgrid = g[None, :, None, None].to(lo)
        c = torch.nn.functional.grid_sample(grid.to(lo), vgrid, padding_mode="border", align_corners=True, mode="bilinear")
        return c.squeeze(-1).squeeze(-1)[0].transpose(0, 1)
    gv = grid[0].to(lo).permute(1, 2, 3, 0)                  # (Z,Y,X,C) view

    def axis(gc, n):
        u = ((gc + 1) / 2) * (n - 1)
        u = torch.clamp(u, 0, n - 1)                         # grad 0 outside, like clip_coordinates_set_grad
        i0 = torch.floor(u).detach()
        w1 = u - i0
        w0 = (i0 + 1) - u
        i0 = i0.long()
        i1 = i0 + 1
        ok1 = (i1 <= n - 1)
        i1 = torch.where(ok1, i1, i0)
        w1 = torch.where(ok1, w1, torch.zeros_like(w1))
        return i0, i1, w0, w1

    x0, x1, wx0, wx1 = axis(g[:, 0], X)
    y0, y1, wy0, wy1 = axis(g[:, 1], Y)
    z0, z1, wz0, wz1 = axis(g[:, 2], Z)
    out = 0
    for zi, wz in ((z0, wz0), (z1, wz1)):
        for yi, wy in ((y0, wy0), (y1, wy1)):
            for xi, wx in ((x0, wx0), (x1, wx1)):
                out = out + gv[zi, yi, xi] * (wx * wy * wz)[:, None]
    return out


# --------------------------------------------------------------------------------------------
# a8-a10: decoders   (decoder.py:17-30,177-203,262-274,312-342)
# --------------------------------------------------------------------------------------------
# How the fp32 matrix products are evaluated.  "mm" = ATen's matmul (what the reference calls; its accumulation order is the host
# BLAS's).  The other modes exist for ONE purpose -- tools/reference_fp32_ambiguity.py measures how far two LEGITIMATE fp32
# evaluations of the reference's own operators are apart on a given scene (the floor under any parity gate):
#   LINEAR_IMPL  "bf16x3": every Linear product (forward and both backward products) as six cross products of three-way bf16 splits
#                with fp32 accumulation (noise study only, see _mm_bf16x3);
#                "rounded_once": every dot product of a Linear layer accumulated in fp64 and rounded to fp32 once (the most
#                accurate fp32 evaluation there is; autograd then differentiates through the casts, i.e. the backward products
#                are rounded once as well)
#   EMBED_IMPL   "fma_k" / "fma_k_rev": p @ B as an explicit fused-multiply-add chain over the three coordinates in the order
#                x, y, z (what the HIP kernels do, and bit for bit what MKL's sgemm does for this shape on the build container)
#                or z, y, x; "product_sum": three rounded products summed left to right (no fused multiply-add).  Forward
#                values only differ; the backward is autograd's through an identical graph (custom Function below).
LINEAR_IMPL = "mm"
EMBED_IMPL = "mm"


def _bf16_split3(t):
    """t = hi + mid + lo with every piece exactly representable in bfloat16 (mantissa TRUNCATION: the residuals are exact fp32
    subtractions); what is left after three pieces is below 2^-24 |t|."""
    def trunc(v):
        return (v.contiguous().view(torch.int32) & -65536).view(torch.float32)
    hi = trunc(t)
    r1 = t - hi
    mid = trunc(r1)
    lo = trunc(r1 - mid)
    return hi, mid, lo


def _mm_bf16x3(a, b):
    """a @ b as the six leading cross products of the three-way bf16 splits, accumulated in fp32 (noise study of a possible
    bf16-MFMA formulation of the decoders' products, profiles/r06_experiments.txt item 20; never used by a test's reference)."""
    ah, am, al = _bf16_split3(a)
    bh, bm, bl = _bf16_split3(b)
    return (((al @ bh + ah @ bl) + am @ bm) + (am @ bh + ah @ bm)) + ah @ bh


class _LinBf16x3(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, W, bias):
        ctx.save_for_backward(x, W)
        return _mm_bf16x3(x, W.t()) + bias

    @staticmethod
    def backward(ctx, g):
        x, W = ctx.saved_tensors
        return _mm_bf16x3(g, W), _mm_bf16x3(g.t(), x), g.sum(0)


def _lin(x, P, prefix):
    if LINEAR_IMPL == "rounded_once" and x.dtype == F32:
        return (x.to(F64) @ P[prefix + ".weight"].to(F64).t() + P[prefix + ".bias"].to(F64)).to(F32)
    if LINEAR_IMPL == "bf16x3" and x.dtype == F32:
        return _LinBf16x3.apply(x, P[prefix + ".weight"], P[prefix + ".bias"])
    return x @ P[prefix + ".weight"].to(x.dtype).t() + P[prefix + ".bias"].to(x.dtype)


def _fma32(a, b, c):
    """fl32(a * b + c) for fp32 tensors: the product of two fp32 numbers is exact in fp64; the sum is rounded to fp64 and then to
    fp32 (a double rounding that differs from a true fused multiply-add on ~2^-29 of the inputs: fine for a noise study)"""
    return (a.to(F64) * b.to(F64) + c.to(F64)).to(F32)


class _EmbedArg(torch.autograd.Function):
    """p @ B with a chosen fp32 evaluation order of the three-term dot products; backward = that of the matmul."""

    @staticmethod
    def forward(ctx, p, B, impl):
        ctx.save_for_backward(p, B)
        x, y, z = p[:, 0:1], p[:, 1:2], p[:, 2:3]
        bx, by, bz = B[0:1], B[1:2], B[2:3]
        if impl == "fma_k":
            return _fma32(z, bz, _fma32(y, by, x * bx))
        if impl == "fma_k_rev":
            return _fma32(x, bx, _fma32(y, by, z * bz))
        if impl == "product_sum":
            return (x * bx + y * by) + z * bz
        raise ValueError(impl)

    @staticmethod
    def backward(ctx, g):
        p, B = ctx.saved_tensors
        return g @ B.t(), p.t() @ g, None


def mlp_xyz(p: torch.Tensor, c: torch.Tensor, P: Dict[str, torch.Tensor], name: str, lo=F32):
    """``MLP.forward`` (decoder.py:177-203) given the already-sampled feature ``c``.
    h_i = relu(W_i h + b_i) + (U_i c + v_i); after i == 2 the embedding is concatenated in front."""
    pre = name + "_decoder."
    if EMBED_IMPL != "mm" and lo == F32:
        e = torch.sin(_EmbedArg.apply(p.to(lo), P[pre + "embedder._B"].to(lo), EMBED_IMPL))
    else:
        e = torch.sin(p.to(lo) @ P[pre + "embedder._B"].to(lo))      # decoder.py:26-30,189-191
    h = e
    for i in range(5):
        h = torch.relu(_lin(h, P, pre + f"pts_linears.{i}"))
        h = h + _lin(c, P, pre + f"fc_c.{i}")
        if i == 2:
            h = torch.cat([e, h], -1)
    return _lin(h, P, pre + "output_linear")


def mlp_no_xyz(c: torch.Tensor, P: Dict[str, torch.Tensor], name: str = "coarse"):
    """``MLP_no_xyz.forward`` (decoder.py:262-274)."""
    pre = name + "_decoder."
    h = c
    for i in range(5):
        h = torch.relu(_lin(h, P, pre + f"pts_linears.{i}"))
        if i == 2:
            h = torch.cat([c, h], -1)
    return _lin(h, P, pre + "output_linear")


def nice_decode(p: torch.Tensor, grids: Dict[str, torch.Tensor], P: Dict[str, torch.Tensor],
                bounds: Dict[str, torch.Tensor], stage: str, lo=F32) -> torch.Tensor:
    """``NICE.forward`` (decoder.py:312-342): (M,3) world points -> (M,4) [r,g,b,occ-logit]."""
    M = p.shape[0]
    raw = torch.zeros((M, 4), dtype=lo, device=p.device)
    if stage == "coarse":
        c = trilinear(grids["grid_coarse"], p, bounds["coarse"], lo)
        occ = mlp_no_xyz(c, P)[:, 0]
        return torch.cat([raw[:, :3], occ[:, None]], -1)
    c_mid = trilinear(grids["grid_middle"], p, bounds["middle"], lo)
    mid_occ = mlp_xyz(p, c_mid, P, "middle", lo)[:, 0]
    if stage == "middle":
        return torch.cat([raw[:, :3], mid_occ[:, None]], -1)
    c_fine = trilinear(grids["grid_fine"], p, bounds["fine"], lo)
    # decoder.py:182-187: the middle feature concatenated into the fine decoder is sampled under no_grad
    fine_occ = mlp_xyz(p, torch.cat([c_fine, c_mid.detach()], -1), P, "fine", lo)[:, 0]
    occ = fine_occ + mid_occ
    if stage == "fine":
        return torch.cat([raw[:, :3], occ[:, None]], -1)
    c_col = trilinear(grids["grid_color"], p, bounds["color"], lo)
    rgb = mlp_xyz(p, c_col, P, "color", lo)[:, :3]           # 4th colour output is overwritten (:341)
    return torch.cat([rgb, occ[:, None]], -1)


def eval_points(p: torch.Tensor, grids, P, bounds, bound: torch.Tensor, stage: str, lo=F32):
    """``Renderer.eval_points`` (Renderer.py:23-61): decode + force occ=100 outside the open box.
    The test uses the renderer's un-enlarged bound for every stage."""
    b = bound.to(device=p.device, dtype=p.dtype)
    inside = ((p > b[:, 0]) & (p < b[:, 1])).all(-1)
    raw = nice_decode(p, grids, P, bounds, stage, lo)
    occ = torch.where(inside, raw[:, 3], torch.full_like(raw[:, 3], 100.0))
    return torch.cat([raw[:, :3], occ[:, None]], -1)


# --------------------------------------------------------------------------------------------
# a11: compositor   (src/common.py:231-244, occupancy branch)
# --------------------------------------------------------------------------------------------
def composite(raw: torch.Tensor, z: torch.Tensor, lo=F32):
    """raw (N,S,4), z (N,S) -> depth (N,) hi, var (N,) hi, rgb (N,3) lo, weights (N,S) lo."""
    alpha = torch.sigmoid(10 * raw[..., 3])
    ones = torch.ones((alpha.shape[0], 1), dtype=lo, device=raw.device)
    trans = torch.cumprod(torch.cat([ones, (1.0 - alpha + 1e-10).to(lo)], -1), -1)[:, :-1]
    w = alpha * trans
    rgb = torch.sum(w[..., None] * raw[..., :3], -2)
    depth = torch.sum(w * z, -1)
    dz = z - depth[:, None]
    var = torch.sum(w * dz * dz, -1)
    return depth, var, rgb, w


# --------------------------------------------------------------------------------------------
# a4 end to end
# --------------------------------------------------------------------------------------------
def render_batch_ray(grids: Dict[str, torch.Tensor], P: Dict[str, torch.Tensor],
                     rays_d: torch.Tensor, rays_o: torch.Tensor, stage: str,
                     gt_depth: Optional[torch.Tensor], bound: torch.Tensor,
                     coarse_enlarge: float = 2.0, n_samples: int = 32, n_surface: int = 16,
                     hi=F64, lo=F32):
    """``Renderer.render_batch_ray`` (Renderer.py:63-198) for the NICE configuration
    (occupancy=True, perturb=0, N_importance=0, lindisp=False).  Differentiable w.r.t. grids,
    decoder parameters and rays."""
    assert stage in STAGES
    z = sample_depths(rays_o, rays_d, gt_depth, bound, stage, n_samples, n_surface, hi)
    pts = rays_o[:, None, :].to(hi) + rays_d[:, None, :].to(hi) * z[:, :, None]     # Renderer.py:172-174
    N, S = z.shape
    raw = eval_points(pts.reshape(-1, 3), grids, P, decoder_bounds(bound, coarse_enlarge), bound, stage, lo)
    depth, var, rgb, _ = composite(raw.reshape(N, S, 4), z, lo)
    return depth, var, rgb


def cast_all(d: Dict[str, torch.Tensor], dtype) -> Dict[str, torch.Tensor]:
    return {k: v.to(dtype) for k, v in d.items()}
