"""Pin the CPU oracle (oracle/nice_oracle.py) to fixtures minted from the real reference
(tests/golden/make_golden.py).  CPU only."""
import json
import os

import numpy as np
import pytest
import torch

from conftest import GOLDEN, golden_scene, rel_err
from oracle import nice_oracle as orc

STAGES = ("coarse", "middle", "fine", "color")
# The oracle re-states the same torch ops in the same dtypes; differences come only from
# summation order inside sgemm / index_add, so the gate is far below the 1e-4 product gate.
TOL_FWD = 2e-6
TOL_GRAD = 2e-5


def test_scene_shapes_match_reference():
    with open(os.path.join(GOLDEN, "scene_shapes.json")) as f:
        shapes = json.load(f)
    assert len(shapes) >= 40
    for path, rec in shapes.items():
        b = orc.scene_bound(rec["bound_cfg"], rec["scale"], rec["bound_divisible"])
        assert np.array_equal(b.numpy(), np.array(rec["bound"])), path
        got = orc.grid_shapes(b, rec["grid_len"], rec["coarse_bound_enlarge"])
        assert {k: list(v) for k, v in got.items()} == rec["shapes"], path
    # Replica room0 middle grid is 21 (not 22) cells along z
    assert shapes["configs/Replica/room0.yaml"]["shapes"]["grid_middle"] == [21, 28, 37]


def test_pixel_rays_bit_exact(golden):
    H, W, fx, fy, cx, cy = golden["intr"]
    H0, H1, W0, W1 = (int(v) for v in golden["gs/crop"])
    o, d, sd, sc = orc.pixel_rays(torch.from_numpy(golden["gs/idx"]), H0, H1, W0, W1, fx, fy, cx, cy,
                                  torch.from_numpy(golden["c2w"]), torch.from_numpy(golden["depth_img"]),
                                  torch.from_numpy(golden["color_img"]))
    assert np.array_equal(o.numpy(), golden["gs/rays_o"])
    assert np.array_equal(d.numpy(), golden["gs/rays_d"])
    assert np.array_equal(sd.numpy(), golden["gs/depth"])
    assert np.array_equal(sc.numpy(), golden["gs/color"])


def _run(golden, stage, gt=True):
    grids, params, bound = golden_scene(golden)
    grids = {k: v.clone().requires_grad_(True) for k, v in grids.items()}
    params = {k: v.clone().requires_grad_(True) for k, v in params.items()}
    o = torch.from_numpy(golden["rays_o"]).clone().requires_grad_(True)
    d = torch.from_numpy(golden["rays_d"]).clone().requires_grad_(True)
    gd = torch.from_numpy(golden["gt_depth"]) if gt else None
    depth, var, rgb = orc.render_batch_ray(grids, params, d, o, stage, gd, bound,
                                           float(golden["coarse_bound_enlarge"]))
    return grids, params, o, d, depth, var, rgb


@pytest.mark.parametrize("stage", STAGES)
def test_forward_matches_reference(golden, stage):
    *_, depth, var, rgb = _run(golden, stage)
    assert depth.dtype == torch.float64 and var.dtype == torch.float64 and rgb.dtype == torch.float32
    pre = f"out/{stage}/"
    assert rel_err(depth.detach(), golden[pre + "depth"]) < TOL_FWD
    assert rel_err(var.detach(), golden[pre + "var"]) < TOL_FWD
    assert rel_err(rgb.detach(), golden[pre + "rgb"]) < TOL_FWD


def test_forward_without_depth(golden):
    *_, depth, var, rgb = _run(golden, "middle", gt=False)
    assert depth.shape == (40,)
    assert rel_err(depth.detach(), golden["out/middle_nodepth/depth"]) < TOL_FWD
    assert rel_err(var.detach(), golden["out/middle_nodepth/var"]) < TOL_FWD


@pytest.mark.parametrize("stage", STAGES)
def test_backward_matches_reference(golden, stage):
    grids, params, o, d, depth, var, rgb = _run(golden, stage)
    loss = (depth * torch.from_numpy(golden["w_depth"])).sum() + (var * torch.from_numpy(golden["w_var"])).sum() \
        + (rgb * torch.from_numpy(golden["w_rgb"])).sum()
    loss.backward()
    pre = f"out/{stage}/"
    assert rel_err(o.grad, golden[pre + "d_rays_o"]) < TOL_GRAD
    assert rel_err(d.grad, golden[pre + "d_rays_d"]) < TOL_GRAD
    n_checked = 0
    for k, v in grids.items():
        key = pre + "d_" + k
        if key in golden:
            got = v.grad if v.grad is not None else torch.zeros_like(v)
            assert rel_err(got, golden[key]) < TOL_GRAD, k
            n_checked += 1
        else:
            assert v.grad is None or float(v.grad.abs().max()) == 0.0, k
    for k, v in params.items():
        key = pre + "dparam/" + k
        if key in golden:
            got = v.grad if v.grad is not None else torch.zeros_like(v)
            assert rel_err(got, golden[key]) < TOL_GRAD, k
            n_checked += 1
        else:
            assert v.grad is None or float(v.grad.abs().max()) == 0.0, k
    assert n_checked > 5


def test_oracle_grid_sample_mode_equals_index_mode():
    """oracle.TRILINEAR_IMPL = "grid_sample" calls ATen's grid_sampler_3d like the reference (decoder.py:173); the explicit
    restatement must give the same outputs and gradients (this is also what bench.py's cpu_baseline leg executes)."""
    import torch
    from oracle import nice_oracle as orc
    from scene_util import make_scene, oracle_render, rel_err
    sc = make_scene(seed=9, n_rays=40, small=True)
    a = oracle_render(sc, "color", backward=True)
    try:
        orc.TRILINEAR_IMPL = "grid_sample"
        b = oracle_render(sc, "color", backward=True)
    finally:
        orc.TRILINEAR_IMPL = "index"
    assert set(a) == set(b)
    for k in a:
        assert rel_err(b[k], a[k]) < 2e-5, (k, rel_err(b[k], a[k]))
