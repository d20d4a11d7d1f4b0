"""TEST INFRASTRUCTURE -- CPU restatement of the frustum feature selection of the mapper.

Follows ``Mapper.get_mask_from_c2w`` (reference src/Mapper.py:93-164) line by line in numpy.  Only tests/,
``__graft_entry__.smoke()`` and bench.py's checker legs may import this module; the product (nice_slam_amd) never does.

PARITY STATUS: **unpinned**.  The reference function cannot be executed in this container: it needs ``cv2.remap``
(OpenCV is not installed and not in the wheelhouse), and the reference tree holds no golden vectors
for it.  Two pieces are therefore restated from the published algorithm instead of being checked against the reference:

* ``remap_bilinear`` = ``cv2.remap(src_f32, mapx_f32, mapy_f32, INTER_LINEAR)`` with the default
  ``BORDER_CONSTANT`` / ``borderValue=0`` (OpenCV 4.x ``imgwarp.cpp``: the float maps are converted to fixed point with
  ``INTER_BITS = 5``: ``sx = cvRound(x*32)``, integer pixel ``sx >> 5``, sub-pixel ``sx & 31``; the four weights come from
  the float table ``(1 - fx/32, fx/32) x (1 - fy/32, fy/32)``; pixels outside the image contribute 0);
* the float32 ``w2c @ homo_vertices`` matmul (Mapper.py:122), whose summation order inside numpy/BLAS is not defined:
  restated as the sequential fp32 sum ((w0*x + w1*y) + w2*z) + w3 without fused multiply-add.

Everything else (voxel centres from torch.linspace in fp32, the fp64 ``K @ cam`` projection, ``+1e-5``, the fp32 cast of
uv, the zero-depth fill with the max of the *remapped* depths, the 0.5 m tolerances) is the reference's arithmetic.
"""
from __future__ import annotations

import numpy as np
import torch

INTER_BITS = 5
INTER_TAB = 1 << INTER_BITS


def remap_bilinear(src: np.ndarray, mapx: np.ndarray, mapy: np.ndarray) -> np.ndarray:
    """cv2.remap(src, mapx, mapy, cv2.INTER_LINEAR) for a single-channel float32 image; see the module header."""
    src = np.ascontiguousarray(src, dtype=np.float32)
    H, W = src.shape
    with np.errstate(invalid="ignore", over="ignore"):
        fx32 = (mapx.astype(np.float32) * np.float32(INTER_TAB)).astype(np.float32)
        fy32 = (mapy.astype(np.float32) * np.float32(INTER_TAB)).astype(np.float32)
        # cvRound: round half to even; out-of-int-range values saturate (those pixels are outside the image anyway)
        sx = np.clip(np.rint(fx32.astype(np.float64)), -2 ** 31, 2 ** 31 - 1)
        sy = np.clip(np.rint(fy32.astype(np.float64)), -2 ** 31, 2 ** 31 - 1)
    sx = np.nan_to_num(sx, nan=-2.0 ** 31).astype(np.int64)
    sy = np.nan_to_num(sy, nan=-2.0 ** 31).astype(np.int64)
    ix = np.clip(sx >> INTER_BITS, -32768, 32767)          # saturate_cast<short>
    iy = np.clip(sy >> INTER_BITS, -32768, 32767)
    ax = (sx & (INTER_TAB - 1)).astype(np.float32) / np.float32(INTER_TAB)
    ay = (sy & (INTER_TAB - 1)).astype(np.float32) / np.float32(INTER_TAB)
    one = np.float32(1.0)
    w00 = (one - ay) * (one - ax)
    w01 = (one - ay) * ax
    w10 = ay * (one - ax)
    w11 = ay * ax

    def px(yy, xx):
        ok = (yy >= 0) & (yy < H) & (xx >= 0) & (xx < W)
        v = src[np.clip(yy, 0, H - 1), np.clip(xx, 0, W - 1)]
        return np.where(ok, v, np.float32(0.0)).astype(np.float32)

    out = px(iy, ix) * w00
    out = out + px(iy, ix + 1) * w01
    out = out + px(iy + 1, ix) * w10
    out = out + px(iy + 1, ix + 1) * w11
    return out.astype(np.float32)


def voxel_axes(bound, val_shape):
    """Mapper.py:111-113: per-axis voxel-centre coordinates, fp32 torch.linspace over the (un-enlarged) scene bound.
    ``val_shape`` = grid.shape[2:] = (Zc, Yc, Xc)."""
    b = [[float(bound[a][0]), float(bound[a][1])] for a in range(3)]
    xs = torch.linspace(b[0][0], b[0][1], int(val_shape[2]))
    ys = torch.linspace(b[1][0], b[1][1], int(val_shape[1]))
    zs = torch.linspace(b[2][0], b[2][1], int(val_shape[0]))
    return xs.numpy(), ys.numpy(), zs.numpy()


def get_mask_from_c2w(c2w, key, val_shape, depth_np, bound, H, W, fx, fy, cx, cy):
    """Mapper.py:93-164.  c2w: (4,4) float32 array; depth_np: (H,W) float32; returns bool (X,Y,Z) like the reference
    (the caller permutes it to (Z,Y,X), Mapper.py:318)."""
    nz, ny, nx = int(val_shape[0]), int(val_shape[1]), int(val_shape[2])
    if key == "grid_coarse":                                             # :116-118
        return np.ones((nx, ny, nz), dtype=bool)
    xs, ys, zs = voxel_axes(bound, val_shape)
    X, Y, Z = np.meshgrid(xs, ys, zs, indexing="ij")                     # :111-115  (x slowest, z fastest)
    pts = np.stack([X, Y, Z], -1).reshape(-1, 3).astype(np.float32)
    c2w = np.asarray(c2w, dtype=np.float32)
    w2c = np.linalg.inv(c2w)                                             # :120  (float32 in, float32 out)
    f32 = np.float32
    cam = np.empty((pts.shape[0], 3), dtype=np.float32)
    for r in range(3):                                                   # :122-125, sequential fp32 (see header)
        acc = (w2c[r, 0] * pts[:, 0]).astype(f32)
        acc = (acc + (w2c[r, 1] * pts[:, 1]).astype(f32)).astype(f32)
        acc = (acc + (w2c[r, 2] * pts[:, 2]).astype(f32)).astype(f32)
        cam[:, r] = (acc + w2c[r, 3]).astype(f32)
    cam[:, 0] *= f32(-1.0)                                               # :127
    c64 = cam.astype(np.float64)
    uh = (float(fx) * c64[:, 0] + 0.0 * c64[:, 1]) + float(cx) * c64[:, 2]     # :126,128  K @ cam_cord in fp64
    vh = (0.0 * c64[:, 0] + float(fy) * c64[:, 1]) + float(cy) * c64[:, 2]
    z = c64[:, 2] + 1e-5                                                 # :129
    with np.errstate(divide="ignore", invalid="ignore", over="ignore"):
        u = (uh / z).astype(np.float32)                                  # :130-131
        v = (vh / z).astype(np.float32)
    depths = remap_bilinear(depth_np, u, v)                              # :133-140
    edge = 0
    with np.errstate(invalid="ignore"):
        mask = (u < W - edge) & (u > edge) & (v < H - edge) & (v > edge)     # :142-144
        zero = depths == 0                                               # :147-148
        depths = depths.copy()
        depths[zero] = np.max(depths)
        mask = mask & (0 <= -z) & (-z <= (depths + f32(0.5)).astype(np.float64))    # :151
    o = c2w[:3, 3].astype(f32)                                           # :155-161
    d = (pts - o[None]).astype(f32)
    d2 = (d * d).astype(f32)
    dist = ((d2[:, 0] + d2[:, 1]).astype(f32) + d2[:, 2]).astype(f32)
    mask = mask | (dist < f32(0.5 * 0.5))
    return mask.reshape(nx, ny, nz)                                      # :164
