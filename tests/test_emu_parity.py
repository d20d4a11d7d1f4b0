"""CPU: the kernel SOURCES (nice_slam_amd/csrc, compiled for the host against the fiber shim in tests/emu)
against the reference goldens and the oracle.  This exercises every index, layout and reduction of the HIP
kernels without a GPU; the GPU run (test_hip_parity.py) then only has to confirm the hardware primitives."""
import shutil

import numpy as np
import pytest
import torch

from conftest import golden_scene, rel_err
from scene_util import make_scene, oracle_render

STAGES = ("coarse", "middle", "fine", "color")
TOL = 1e-4


@pytest.fixture(scope="module")
def emu():
    import os
    if not (os.path.exists("/opt/rocm/lib/llvm/bin/clang++") or shutil.which("clang++")):
        pytest.skip("no host clang++ for the emulator build")
    from emu_harness import emu_lib
    return emu_lib()


def _host_scene(emu, grids, params, bound, enl=2.0):
    from emu_harness import HostScene
    return HostScene(emu, grids, params, np.asarray(bound), enl)


@pytest.mark.parametrize("stage", STAGES)
def test_golden_forward_backward(emu, golden, stage):
    grids, params, bound = golden_scene(golden)
    sc = _host_scene(emu, grids, params, bound.numpy(), float(golden["coarse_bound_enlarge"]))
    fwd = sc.forward(stage, golden["rays_o"], golden["rays_d"], golden["gt_depth"])
    pre = f"out/{stage}/"
    for k in ("depth", "var", "rgb"):
        assert rel_err(fwd[k], golden[pre + k]) < TOL, (stage, k)
    res = sc.backward(stage, fwd, golden["w_depth"], golden["w_var"], golden["w_rgb"])
    checked = 0
    for k, v in res.items():
        gk = pre + k
        if gk in golden:
            assert rel_err(v, golden[gk]) < TOL, (stage, k)
            checked += 1
        else:
            assert float(np.abs(v).max()) == 0.0, (stage, k)
    assert checked >= 8


def test_golden_forward_without_depth(emu, golden):
    grids, params, bound = golden_scene(golden)
    sc = _host_scene(emu, grids, params, bound.numpy(), float(golden["coarse_bound_enlarge"]))
    fwd = sc.forward("middle", golden["rays_o"], golden["rays_d"], None)
    assert fwd["raw"].shape[1] == 32
    assert rel_err(fwd["depth"], golden["out/middle_nodepth/depth"]) < TOL
    assert rel_err(fwd["var"], golden["out/middle_nodepth/var"]) < TOL


@pytest.mark.parametrize("stage,n", [("color", 37), ("fine", 5), ("coarse", 13), ("middle", 1)])
def test_random_scene_against_oracle(emu, stage, n):
    """ragged ray counts (not a multiple of rays-per-block), persistent loop with a tiny grid cap"""
    s = make_scene(seed=100 + n, n_rays=n, small=True)
    s["gt_depth"][0] = 0.0
    sc = _host_scene(emu, s["grids"], s["params"], s["bound"].numpy())
    fwd = sc.forward(stage, s["rays_o"].numpy(), s["rays_d"].numpy(), s["gt_depth"].numpy())
    ref = oracle_render(s, stage, backward=True)
    for k in ("depth", "var", "rgb"):
        assert rel_err(fwd[k], ref[k]) < TOL, (stage, k)
    res = sc.backward(stage, fwd, s["w"]["depth"].numpy(), s["w"]["var"].numpy(), s["w"]["rgb"].numpy(), max_blocks=2)
    for k, v in ref.items():
        if k in ("depth", "var", "rgb"):
            continue
        assert rel_err(res[k], v) < TOL, (stage, k)
    # nsr_bwd_args.overwrite_dparams: same parameter gradients into a blob that was NOT zeroed (NaN-filled here)
    res2 = sc.backward(stage, fwd, s["w"]["depth"].numpy(), s["w"]["var"].numpy(), s["w"]["rgb"].numpy(), max_blocks=2,
                       overwrite_dparams=True)
    for k, v in res.items():
        if k.startswith("dparam/"):
            assert np.array_equal(res2[k], v), (stage, k)


def test_backward_flag_subsets(emu):
    """tracking (ray grads only) and grid-only requests give the same numbers as the full backward"""
    s = make_scene(seed=7, n_rays=9, small=True)
    sc = _host_scene(emu, s["grids"], s["params"], s["bound"].numpy())
    w = s["w"]
    fwd = sc.forward("color", s["rays_o"].numpy(), s["rays_d"].numpy(), s["gt_depth"].numpy())
    full = sc.backward("color", fwd, w["depth"].numpy(), w["var"].numpy(), w["rgb"].numpy())
    rays = sc.backward("color", fwd, w["depth"].numpy(), w["var"].numpy(), w["rgb"].numpy(), want_grid=False, want_params=False)
    grid = sc.backward("color", fwd, w["depth"].numpy(), w["var"].numpy(), w["rgb"].numpy(), want_params=False, want_rays=False)
    assert set(rays) == {"d_rays_o", "d_rays_d"}
    for k in rays:
        assert rel_err(rays[k], full[k]) < 1e-6
    for k in grid:
        assert rel_err(grid[k], full[k]) < 1e-6
    # backward that re-derives the sample depths instead of loading the ones the forward saved
    sc.save_z = False
    fwd2 = sc.forward("color", s["rays_o"].numpy(), s["rays_d"].numpy(), s["gt_depth"].numpy())
    again = sc.backward("color", fwd2, w["depth"].numpy(), w["var"].numpy(), w["rgb"].numpy())
    for k in full:
        assert rel_err(again[k], full[k]) < 1e-6, k
    sc.save_z = True
    # depth-only upstream gradient (d_var = d_rgb = NULL)
    dd = sc.backward("fine", sc.forward("fine", s["rays_o"].numpy(), s["rays_d"].numpy(), s["gt_depth"].numpy()),
                     w["depth"].numpy(), None, None)
    assert np.isfinite(dd["d_grid_fine"]).all()


def test_eval_points_and_get_samples(emu, golden):
    import ctypes as C
    from emu_harness import ptr
    from nice_slam_amd import _capi
    from oracle import nice_oracle as orc
    grids, params, bound = golden_scene(golden)
    sc = _host_scene(emu, grids, params, bound.numpy(), float(golden["coarse_bound_enlarge"]))
    g = torch.Generator().manual_seed(3)
    pts = (torch.rand((50, 3), generator=g, dtype=torch.float64) * 1.3 - 0.15) * (bound[:, 1] - bound[:, 0]) + bound[:, 0]
    for stage in STAGES:
        a = sc._args(stage, np.zeros((1, 3), np.float32), np.zeros((1, 3), np.float32), None, [])
        out = np.full((50, 4), np.nan, dtype=np.float32)
        p = np.ascontiguousarray(pts.numpy())
        emu.check(emu.nsr_eval_points_fwd(C.byref(a), ptr(p), 50, ptr(out), None))
        ref = orc.eval_points(pts, grids, params, orc.decoder_bounds(bound, float(golden["coarse_bound_enlarge"])), bound, stage)
        assert rel_err(out, ref) < TOL, stage
    H, W, fx, fy, cx, cy = golden["intr"]
    H0, H1, W0, W1 = (int(v) for v in golden["gs/crop"])
    n = golden["gs/idx"].shape[0]
    o, d = np.empty((n, 3), np.float32), np.empty((n, 3), np.float32)
    sd, scol = np.empty(n, np.float32), np.empty((n, 3), np.float32)
    idx = np.ascontiguousarray(golden["gs/idx"].astype(np.int64))
    c2w = np.ascontiguousarray(golden["c2w"])
    emu.check(emu.nsr_get_samples(ptr(idx), n, H0, H1, W0, W1, int(W), fx, fy, cx, cy, ptr(c2w), 4,
                                  ptr(golden["depth_img"]), ptr(golden["color_img"]), ptr(o), ptr(d), ptr(sd), ptr(scol), None))
    assert np.array_equal(o, golden["gs/rays_o"]) and np.array_equal(d, golden["gs/rays_d"])      # bit-exact
    assert np.array_equal(sd, golden["gs/depth"]) and np.array_equal(scol, golden["gs/color"])


def test_masked_adam_matches_reference_flow(emu):
    """`val[mask] = val_grad` + torch.optim.Adam on the masked leaf + write-back
    (src/Mapper.py:303-333,368-379,394-401,504,511-519) against ONE in-place nsr_masked_adam per step."""
    from emu_harness import ptr
    g = torch.Generator().manual_seed(9)
    Z, Y, X = 5, 4, 7
    val0 = torch.randn((1, 32, Z, Y, X), generator=g) * 0.01
    vmask = torch.rand((Z, Y, X), generator=g) < 0.6
    mask = vmask[None, None].expand(1, 32, Z, Y, X)
    lrs = [0.1, 0.0, 0.005, None, 0.005, 0.02]                # None: the grid got no gradient in that iteration (skipped)
    grads = [torch.randn((1, 32, Z, Y, X), generator=g) * (10.0 ** -i) for i in range(len(lrs))]
    # reference flow
    val = val0.clone()
    leaf = val[mask].clone().requires_grad_(True)
    opt = torch.optim.Adam([{"params": [leaf], "lr": 0.0}])
    for lr, gr in zip(lrs, grads):
        if lr is None:
            continue
        opt.param_groups[0]["lr"] = lr
        leaf.grad = gr[mask].clone()
        opt.step()
        val = val.detach()
        val[mask] = leaf.detach()
    # fused
    cl = lambda t: np.ascontiguousarray(t.numpy()[0].transpose(1, 2, 3, 0))          # [Z][Y][X][32]
    p = cl(val0.clone()); m = np.zeros_like(p); v = np.zeros_like(p)
    vm = np.ascontiguousarray(vmask.numpy().astype(np.uint8))
    t = 0
    for lr, gr in zip(lrs, grads):
        if lr is None:
            continue
        t += 1
        gg = cl(gr)
        emu.check(emu.nsr_masked_adam(ptr(p), ptr(gg), ptr(m), ptr(v), ptr(vm), Z * Y * X,
                                      lr / (1 - 0.9 ** t), 0.9, 0.999, 1e-8, (1 - 0.999 ** t) ** 0.5, None))
    got = torch.from_numpy(p.transpose(3, 0, 1, 2)[None])
    assert torch.equal(got[~mask], val0[~mask])                # untouched outside the mask
    assert rel_err(got, val) < 2e-5                            # fp32 rounding-order differences of the Adam arithmetic
    assert float((got - val0).abs().max()) > 1e-3              # and it did move


@pytest.mark.parametrize("ns,nsurf", [(20, 5), (16, 0), (40, 24)])
def test_other_sample_counts(emu, ns, nsurf):
    """rendering.N_samples / N_surface other than 32 + 16: S = 25 (tiles straddle rays, padded last tile), 16, 64"""
    from emu_harness import HostScene
    from oracle import nice_oracle as orc
    s = make_scene(seed=200 + ns, n_rays=7, small=True)
    sc = HostScene(emu, s["grids"], s["params"], s["bound"].numpy(), 2.0, n_samples=ns, n_surface=nsurf)
    stage = "color"
    fwd = sc.forward(stage, s["rays_o"].numpy(), s["rays_d"].numpy(), s["gt_depth"].numpy())
    grids = {k: v.clone().requires_grad_(True) for k, v in s["grids"].items()}
    params = {k: v.clone().requires_grad_(True) for k, v in s["params"].items()}
    o = s["rays_o"].clone().requires_grad_(True); d = s["rays_d"].clone().requires_grad_(True)
    depth, var, rgb = orc.render_batch_ray(grids, params, d, o, stage, s["gt_depth"], s["bound"], n_samples=ns, n_surface=nsurf)
    assert fwd["raw"].shape[1] == ns + nsurf
    for k, v in (("depth", depth), ("var", var), ("rgb", rgb)):
        assert rel_err(fwd[k], v.detach()) < TOL, k
    w = s["w"]
    ((depth * w["depth"]).sum() + (var * w["var"]).sum() + (rgb * w["rgb"]).sum()).backward()
    res = sc.backward(stage, fwd, w["depth"].numpy(), w["var"].numpy(), w["rgb"].numpy())
    assert rel_err(res["d_rays_d"], d.grad) < TOL and rel_err(res["d_rays_o"], o.grad) < TOL
    for k in ("grid_middle", "grid_fine", "grid_color"):
        assert rel_err(res["d_" + k], grids[k].grad) < TOL, k
    for k in ("color_decoder.pts_linears.3.weight", "color_decoder.embedder._B", "fine_decoder.fc_c.2.weight", "middle_decoder.pts_linears.0.bias"):
        assert rel_err(res["dparam/" + k], params[k].grad) < TOL, k


@pytest.mark.parametrize("seed", [0, 1, 2])
def test_frustum_mask(emu, seed):
    """nsr_frustum_mask (same kernel source, host-compiled) against the numpy restatement of
    Mapper.get_mask_from_c2w (oracle/frustum_oracle.py; unpinned: cv2 is not installed).  Bit-exact masks."""
    import ctypes as C
    from emu_harness import ptr
    from scene_util import frustum_case
    from oracle import frustum_oracle as fo
    fc = frustum_case(seed)
    nz, ny, nx = fc["shape"]
    ref = fo.get_mask_from_c2w(fc["c2w"], "grid_fine", fc["shape"], fc["depth"], fc["bound"], fc["H"], fc["W"],
                               fc["fx"], fc["fy"], fc["cx"], fc["cy"])
    assert ref.shape == (nx, ny, nz) and 0.02 < ref.mean() < 0.9
    xs, ys, zs = fo.voxel_axes(fc["bound"], fc["shape"])
    w2c = np.ascontiguousarray(np.linalg.inv(fc["c2w"])[:3].astype(np.float32))
    o = np.ascontiguousarray(fc["c2w"][:3, 3])
    n = nx * ny * nz
    ws = np.zeros(emu.nsr_frustum_workspace_floats(n), dtype=np.float32)
    out = np.full((nz, ny, nx), 7, dtype=np.uint8)
    fp = lambda a: a.ctypes.data_as(C.POINTER(C.c_float))
    emu.check(emu.nsr_frustum_mask(fp(w2c), fp(o), fc["fx"], fc["fy"], fc["cx"], fc["cy"], fc["H"], fc["W"], ptr(fc["depth"]),
                                   ptr(xs), ptr(ys), ptr(zs), nx, ny, nz, ptr(ws), ptr(out), None))
    assert set(np.unique(out)) <= {0, 1}
    assert np.array_equal(out.astype(bool), ref.transpose(2, 1, 0))


def test_aabb_keep_matches_reference_expression(emu):
    """the callers' bounding-box pre-filter (Mapper.py:471-481) as a byte mask + kept-ray max depth"""
    import ctypes as C
    from emu_harness import ptr
    g = torch.Generator().manual_seed(3)
    n = 1000
    bound = torch.tensor([[-2.0, 2.4], [-1.6, 1.9], [-2.2, 2.1]], dtype=torch.float64)
    o = ((torch.rand((n, 3), generator=g) - 0.5) * 3.0).float()
    d = torch.randn((n, 3), generator=g).float()
    d[::7, 1] = 0.0                                            # axis-parallel components: +-inf slabs
    d[5::21, 0] = 0.0                                          # ... and starting ON the slab plane (-2.0 is exact in fp32): 0/0 = NaN;
    o[5::21, 0] = -2.0                                         # torch.max / torch.min propagate it, `t >= depth` is False: ray dropped
    assert torch.isnan((bound.unsqueeze(0) - o.double().unsqueeze(-1)) / d.double().unsqueeze(-1)).any()
    depth = (torch.rand((n,), generator=g) * 4.0).float()
    depth[::11] = 0.0
    t = (bound.unsqueeze(0) - o.unsqueeze(-1)) / d.unsqueeze(-1)            # the reference's expression, fp64 by promotion
    t, _ = torch.min(torch.max(t, dim=2)[0], dim=1)
    ref = t >= depth
    assert 0.2 < ref.float().mean() < 0.95
    keep = np.full((n,), 7, dtype=np.uint8)
    kmax = np.zeros((1,), dtype=np.float32)
    lo, hi = (C.c_double * 3)(*bound[:, 0].tolist()), (C.c_double * 3)(*bound[:, 1].tolist())
    on, dn, zn = o.numpy().copy(), d.numpy().copy(), depth.numpy().copy()
    emu.check(emu.nsr_aabb_keep(ptr(on), ptr(dn), ptr(zn), n, lo, hi, ptr(keep), ptr(kmax), None))
    assert np.array_equal(keep.astype(bool), ref.numpy())
    assert kmax[0] == depth[ref].max().item()


def test_degenerate_shapes(emu):
    """feature grids with a single cell along every axis (tiny scenes: the coarse level often is 1 x 1 x 1 ... cells), and
    one / two samples per ray"""
    import scene_util as su
    from emu_harness import HostScene
    from oracle import nice_oracle as orc
    su.SCENES["tiny_grids"] = ([[-0.7, 0.8], [-0.6, 0.7], [-0.5, 0.6]], dict(su.GRID_LEN, coarse=2.1, middle=0.9),
                               (48, 64, 60.0, 60.0, 31.5, 23.5))
    s = make_scene(seed=3, n_rays=7, scene="tiny_grids")
    assert tuple(s["grids"]["grid_coarse"].shape[2:]) == (1, 1, 1) and tuple(s["grids"]["grid_middle"].shape[2:]) == (1, 1, 1)
    sc = _host_scene(emu, s["grids"], s["params"], s["bound"].numpy())
    for stage in ("coarse", "middle", "color"):
        fwd = sc.forward(stage, s["rays_o"].numpy(), s["rays_d"].numpy(), s["gt_depth"].numpy())
        ref = oracle_render(s, stage, backward=True)
        res = sc.backward(stage, fwd, s["w"]["depth"].numpy(), s["w"]["var"].numpy(), s["w"]["rgb"].numpy())
        for k in ("depth", "var", "rgb"):
            assert rel_err(fwd[k], ref[k]) < TOL, (stage, k)
        for k, v in ref.items():
            if k in res:
                assert rel_err(res[k], v) < TOL, (stage, k)
    s = make_scene(seed=4, n_rays=5, small=True)
    for ns, nsurf in ((1, 0), (1, 1)):
        hs = HostScene(emu, s["grids"], s["params"], s["bound"].numpy(), 2.0, n_samples=ns, n_surface=nsurf)
        fwd = hs.forward("color", s["rays_o"].numpy(), s["rays_d"].numpy(), s["gt_depth"].numpy())
        grids = {k: v.clone().requires_grad_(True) for k, v in s["grids"].items()}
        params = {k: v.clone().requires_grad_(True) for k, v in s["params"].items()}
        depth, var, rgb = orc.render_batch_ray(grids, params, s["rays_d"], s["rays_o"], "color", s["gt_depth"], s["bound"],
                                               n_samples=ns, n_surface=nsurf)
        for k, v in (("depth", depth), ("var", var), ("rgb", rgb)):
            assert rel_err(fwd[k], v.detach()) < TOL, (ns, nsurf, k)
        w = s["w"]
        ((depth * w["depth"]).sum() + (var * w["var"]).sum() + (rgb * w["rgb"]).sum()).backward()
        res = hs.backward("color", fwd, w["depth"].numpy(), w["var"].numpy(), w["rgb"].numpy())
        for k in ("grid_middle", "grid_fine", "grid_color"):
            assert rel_err(res["d_" + k], grids[k].grad) < TOL, (ns, nsurf, k)


@pytest.mark.parametrize("seed", list(range(8)))
def test_randomised_configurations(emu, seed):
    """random stage / sample counts / ray count / persistent-grid cap / missing depth, forward + full backward vs the oracle"""
    from emu_harness import HostScene
    from oracle import nice_oracle as orc
    rng = np.random.RandomState(1000 + seed)
    stage = ["coarse", "middle", "fine", "color"][rng.randint(4)]
    ns = int(rng.randint(1, 41))
    nsurf = int(rng.randint(0, min(24, 64 - ns) + 1))
    n = int(rng.randint(1, 12))
    with_depth = bool(rng.rand() < 0.8)
    cap = int(rng.randint(1, 4))
    s = make_scene(seed=300 + seed, n_rays=n, small=True)
    if n > 2:
        s["gt_depth"][1] = 0.0
    gd = s["gt_depth"] if with_depth else None
    hs = HostScene(emu, s["grids"], s["params"], s["bound"].numpy(), 2.0, n_samples=ns, n_surface=nsurf)
    fwd = hs.forward(stage, s["rays_o"].numpy(), s["rays_d"].numpy(), None if gd is None else gd.numpy())
    grids = {k: v.clone().requires_grad_(True) for k, v in s["grids"].items()}
    params = {k: v.clone().requires_grad_(True) for k, v in s["params"].items()}
    o = s["rays_o"].clone().requires_grad_(True); d = s["rays_d"].clone().requires_grad_(True)
    depth, var, rgb = orc.render_batch_ray(grids, params, d, o, stage, gd, s["bound"], n_samples=ns, n_surface=nsurf)
    cfg = (stage, ns, nsurf, n, with_depth, cap)
    for k, v in (("depth", depth), ("var", var), ("rgb", rgb)):
        assert rel_err(fwd[k], v.detach()) < TOL, (cfg, k)
    w = s["w"]
    ((depth * w["depth"]).sum() + (var * w["var"]).sum() + (rgb * w["rgb"]).sum()).backward()
    res = hs.backward(stage, fwd, w["depth"].numpy(), w["var"].numpy(), w["rgb"].numpy(), max_blocks=cap,
                      overwrite_dparams=bool(seed & 1))
    assert rel_err(res["d_rays_o"], o.grad) < TOL and rel_err(res["d_rays_d"], d.grad) < TOL, cfg
    for k, v in grids.items():
        if v.grad is not None:
            assert rel_err(res["d_" + k], v.grad) < TOL, (cfg, k)
    blob = {}
    for k, v in params.items():
        if v.grad is not None and ("dparam/" + k) in res:
            blob.setdefault(k.split(".")[0], []).append((res["dparam/" + k].ravel(), v.grad.numpy().ravel()))
    for dec, parts in blob.items():        # whole-decoder blob in the max norm (bias gradients are cancellation-limited)
        a = np.concatenate([p[0] for p in parts]); b = np.concatenate([p[1] for p in parts])
        assert rel_err(a, b) < TOL, (cfg, dec)
