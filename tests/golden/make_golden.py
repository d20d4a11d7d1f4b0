#!/usr/bin/env python3
"""Mint golden fixtures by executing the UNMODIFIED reference modules on CPU.

Run (in the build container only; /root/reference does not exist on the GPU box):

    PYTHONDONTWRITEBYTECODE=1 python tests/golden/make_golden.py

Writes
    tests/golden/small_scene.npz      inputs + outputs + gradients of Renderer.render_batch_ray
                                      (reference src/utils/Renderer.py:63-198) for all 4 stages,
                                      and get_samples (src/common.py:125-134) for a seeded draw
    tests/golden/scene_shapes.json    bound / grid shapes produced by NICE_SLAM.load_bound +
                                      grid_init (src/NICE_SLAM.py:137-157,192-250) for every
                                      scene config shipped with the reference

Arithmetic-neutral shims:
  * ``NICE.forward`` builds ``f'cuda:{p.get_device()}'`` (decoder.py:316) which is invalid on CPU
    for the coarse/middle/fine stages; we call the reference sub-decoders directly and assemble
    ``raw`` exactly as decoder.py:317-335 does.
  * cv2 / colorama / open3d / skimage / trimesh are absent; empty stub modules let
    ``src.NICE_SLAM`` import so that its ``load_bound`` / ``grid_init`` can be called on a bare object.
"""
import json
import os
import sys
import types

os.environ.setdefault("PYTHONDONTWRITEBYTECODE", "1")
sys.dont_write_bytecode = True
REF = "/root/reference"
sys.path.insert(0, REF)
HERE = os.path.dirname(os.path.abspath(__file__))

import numpy as np
import torch

for _m in ("cv2", "colorama", "open3d", "skimage", "trimesh", "skimage.measure"):
    if _m not in sys.modules:
        mod = types.ModuleType(_m)
        sys.modules[_m] = mod
sys.modules["colorama"].Fore = types.SimpleNamespace(GREEN="", MAGENTA="", RED="")
sys.modules["colorama"].Style = types.SimpleNamespace(RESET_ALL="")

from src import config as ref_config            # noqa: E402
from src.common import get_samples              # noqa: E402
from src.conv_onet.models import decoder as ref_decoder   # noqa: E402
from src.utils.Renderer import Renderer         # noqa: E402
from src.NICE_SLAM import NICE_SLAM             # noqa: E402


def ref_decode(decoders, p, c_grid, stage):
    """decoder.py:317-342 with the device string replaced by p.device."""
    if stage == "color":
        return ref_decoder.NICE.forward(decoders, p, c_grid, stage=stage) if p.is_cuda else _color(decoders, p, c_grid)
    if stage == "coarse":
        occ = decoders.coarse_decoder(p, c_grid).squeeze(0)
    elif stage == "middle":
        occ = decoders.middle_decoder(p, c_grid).squeeze(0)
    else:
        fine = decoders.fine_decoder(p, c_grid)
        occ = fine + decoders.middle_decoder(p, c_grid).squeeze(0)
    raw = torch.zeros(occ.shape[0], 4).float()
    raw[..., -1] = occ
    return raw


def _color(decoders, p, c_grid):
    fine = decoders.fine_decoder(p, c_grid)
    raw = decoders.color_decoder(p, c_grid)
    mid = decoders.middle_decoder(p, c_grid).squeeze(0)
    raw[..., -1] = fine + mid
    return raw


class PatchedNICE(ref_decoder.NICE):
    def forward(self, p, c_grid, stage="middle", **kw):
        return ref_decode(self, p, c_grid, stage)


def bare_slam(cfg):
    """Run the reference's own load_bound + grid_init on an attribute bag."""
    s = types.SimpleNamespace()
    s.scale = cfg["scale"]
    s.nice = True
    s.coarse = cfg["coarse"]
    s.coarse_bound_enlarge = cfg["model"]["coarse_bound_enlarge"]
    s.shared_decoders = types.SimpleNamespace(
        middle_decoder=types.SimpleNamespace(), fine_decoder=types.SimpleNamespace(),
        color_decoder=types.SimpleNamespace(), coarse_decoder=types.SimpleNamespace())
    NICE_SLAM.load_bound(s, cfg)
    NICE_SLAM.grid_init(s, cfg)
    return s


def scene_shapes():
    out = {}
    os.chdir(REF)
    for root, _, files in os.walk("configs"):
        for f in sorted(files):
            path = os.path.join(root, f)
            cfg = ref_config.load_config(path, "configs/nice_slam.yaml")
            if "bound" not in cfg.get("mapping", {}):
                continue
            torch.manual_seed(0)
            s = bare_slam(cfg)
            out[path] = {
                "bound_cfg": cfg["mapping"]["bound"], "scale": cfg["scale"],
                "grid_len": {k: cfg["grid_len"][k] for k in ("coarse", "middle", "fine", "color")},
                "bound_divisible": cfg["grid_len"]["bound_divisible"],
                "coarse_bound_enlarge": cfg["model"]["coarse_bound_enlarge"],
                "bound": s.bound.tolist(),
                "shapes": {k: list(v.shape[2:]) for k, v in s.shared_c.items()},
            }
    return out


def small_scene():
    torch.manual_seed(1234)
    cfg = ref_config.load_config(os.path.join(REF, "configs/nice_slam.yaml"))
    cfg["mapping"]["bound"] = [[-0.7, 0.8], [-0.6, 0.7], [-0.5, 0.6]]
    cfg["grid_len"]["coarse"] = 0.8
    s = bare_slam(cfg)
    bound = s.bound
    grids = {k: v.clone() for k, v in s.shared_c.items()}
    # make the fine grid non-negligible so its gradient paths are visible in fp32
    grids["grid_fine"] = grids["grid_fine"] * 100.0

    dec = PatchedNICE(dim=3, c_dim=32, coarse=True, coarse_grid_len=0.8, middle_grid_len=0.32,
                      fine_grid_len=0.16, color_grid_len=0.16, hidden_size=32,
                      pos_embedding_method="fourier")
    # zero-initialised biases would hide bias-path bugs
    with torch.no_grad():
        for n, p in dec.named_parameters():
            if n.endswith(".bias"):
                p.add_(torch.randn_like(p) * 0.1)
    dec.bound = bound
    dec.middle_decoder.bound = bound
    dec.fine_decoder.bound = bound
    dec.color_decoder.bound = bound
    dec.coarse_decoder.bound = bound * cfg["model"]["coarse_bound_enlarge"]

    H, W = 48, 64
    fx = fy = 60.0
    cx, cy = 31.5, 23.5
    slam = types.SimpleNamespace(nice=True, bound=bound, H=H, W=W, fx=fx, fy=fy, cx=cx, cy=cy)
    renderer = Renderer(cfg, None, slam)

    # camera inside the box looking down -z with a small rotation
    ang = 0.2
    R = torch.tensor([[np.cos(ang), 0, np.sin(ang)], [0, 1, 0], [-np.sin(ang), 0, np.cos(ang)]], dtype=torch.float32)
    c2w = torch.eye(4, dtype=torch.float32)
    c2w[:3, :3] = R
    c2w[:3, 3] = torch.tensor([0.1, 0.05, 0.45])
    depth_img = torch.rand(H, W) * 1.1 + 0.25
    depth_img[torch.rand(H, W) < 0.08] = 0.0
    color_img = torch.rand(H, W, 3)

    out = {"bound": bound.numpy(), "c2w": c2w.numpy(), "depth_img": depth_img.numpy(),
           "color_img": color_img.numpy(), "intr": np.array([H, W, fx, fy, cx, cy], dtype=np.float64),
           "coarse_bound_enlarge": np.float64(cfg["model"]["coarse_bound_enlarge"])}
    for k, v in grids.items():
        out["grid/" + k] = v.numpy()
    for k, v in dec.state_dict().items():
        out["param/" + k] = v.numpy()

    # ---- get_samples: replay the reference with a recorded index draw -------------------------
    n = 40
    H0, H1, W0, W1 = 4, H - 4, 6, W - 6
    torch.manual_seed(99)
    st = torch.get_rng_state()
    idx = torch.randint((H1 - H0) * (W1 - W0), (n,))
    torch.set_rng_state(st)
    ro, rd, sd, sc = get_samples(H0, H1, W0, W1, n, H, W, fx, fy, cx, cy, c2w, depth_img, color_img, "cpu")
    out.update({"gs/idx": idx.numpy(), "gs/crop": np.array([H0, H1, W0, W1]), "gs/rays_o": ro.numpy(),
                "gs/rays_d": rd.numpy(), "gs/depth": sd.numpy(), "gs/color": sc.numpy()})

    rays_o = ro.clone().contiguous()
    rays_d = rd.clone().contiguous()
    gt_depth = sd.clone()
    gt_depth[3] = 0.0                      # make sure a zero-depth ray is present
    gt_depth[7] = 3.0                      # a depth beyond the box exit -> out-of-bound samples
    out.update({"rays_o": rays_o.numpy(), "rays_d": rays_d.numpy(), "gt_depth": gt_depth.numpy()})

    g = torch.Generator().manual_seed(5)
    w_depth = torch.randn(n, generator=g, dtype=torch.float64)
    w_var = torch.randn(n, generator=g, dtype=torch.float64)
    w_rgb = torch.randn(n, 3, generator=g)
    out.update({"w_depth": w_depth.numpy(), "w_var": w_var.numpy(), "w_rgb": w_rgb.numpy()})

    for stage in ("coarse", "middle", "fine", "color"):
        c = {k: v.clone().requires_grad_(True) for k, v in grids.items()}
        dec.zero_grad()
        o = rays_o.clone().requires_grad_(True)
        d = rays_d.clone().requires_grad_(True)
        depth, var, rgb = renderer.render_batch_ray(c, dec, d, o, "cpu", stage, gt_depth=gt_depth)
        loss = (depth * w_depth).sum() + (var * w_var).sum() + (rgb * w_rgb).sum()
        loss.backward()
        pre = f"out/{stage}/"
        out[pre + "depth"] = depth.detach().numpy()
        out[pre + "var"] = var.detach().numpy()
        out[pre + "rgb"] = rgb.detach().numpy()
        out[pre + "d_rays_o"] = o.grad.numpy()
        out[pre + "d_rays_d"] = d.grad.numpy()
        for k, v in c.items():
            if v.grad is not None:
                out[pre + "d_" + k] = v.grad.numpy()
        for k, p in dec.named_parameters():
            if p.grad is not None:
                out[pre + "dparam/" + k] = p.grad.numpy().copy()
        # stage with gt_depth=None through the non-coarse decoders (coarse mapper style call)
        if stage == "middle":
            depth2, var2, rgb2 = renderer.render_batch_ray(grids, dec, rays_d, rays_o, "cpu", stage, gt_depth=None)
            out["out/middle_nodepth/depth"] = depth2.detach().numpy()
            out["out/middle_nodepth/var"] = var2.detach().numpy()
    return out


if __name__ == "__main__":
    fix = small_scene()
    np.savez_compressed(os.path.join(HERE, "small_scene.npz"), **fix)
    shapes = scene_shapes()
    with open(os.path.join(HERE, "scene_shapes.json"), "w") as f:
        json.dump(shapes, f, indent=1, sort_keys=True)
    print("wrote", len(fix), "arrays;", len(shapes), "scene configs")
    print({k: v["shapes"] for k, v in shapes.items() if "room0" in k or "apartment" in k})
