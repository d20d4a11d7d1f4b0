"""CPU: the kernel SOURCES (nice_slam_amd/csrc, compiled for the host against the fiber shim in tests/emu)
against the reference goldens and the oracle.  This exercises every index, layout and reduction of the HIP
kernels without a GPU; the GPU run (test_hip_parity.py) then only has to confirm the hardware primitives."""
import os
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
    # a forward that saved no sample depths (zvals = NULL) cannot be differentiated: loud error, no silent recomputation
    sc.save_z = False
    fwd2 = sc.forward("color", s["rays_o"].numpy(), s["rays_d"].numpy(), s["gt_depth"].numpy())
    with pytest.raises(Exception, match="activation buffer"):
        sc.backward("color", fwd2, w["depth"].numpy(), w["var"].numpy(), w["rgb"].numpy())
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


@pytest.mark.parametrize("pre", ["map/", "ba/"])
def test_frustum_mask_against_the_real_mappers_masks(emu, pre):
    """Half a pin for SURVEY §8(f) rank 3: the masks in caller_steps.npz were produced by the REAL Mapper.get_mask_from_c2w
    (src/Mapper.py:93-164) -- its numpy ``w2c @ p``, projection, depth test and near-camera sphere -- with only ``cv2.remap``
    replaced by the restatement (OpenCV is absent).  nsr_frustum_mask must reproduce them bit for bit."""
    import ctypes as C
    from emu_harness import ptr
    from oracle import frustum_oracle as fo
    import caller_replay as cr
    gold = cr.load()
    H, W, fx, fy, cx, cy = (float(v) for v in gold["intr"])
    c2w = gold[pre + "cur_c2w"].astype(np.float64)
    depth = np.ascontiguousarray(gold["frame/0/depth"], dtype=np.float32)
    for key in ("grid_middle", "grid_fine", "grid_color"):
        ref = gold[f"{pre}mask/{key}"].astype(bool)                      # [Z,Y,X]
        nz, ny, nx = ref.shape
        xs, ys, zs = fo.voxel_axes(gold["bound"], (nz, ny, nx))
        w2c = np.ascontiguousarray(np.linalg.inv(c2w)[:3].astype(np.float32))
        o = np.ascontiguousarray(c2w[:3, 3].astype(np.float32))
        ws = np.zeros(emu.nsr_frustum_workspace_floats(nx * ny * nz), dtype=np.float32)
        out = np.full((nz, ny, nx), 7, dtype=np.uint8)
        fp = lambda a: a.ctypes.data_as(C.POINTER(C.c_float))
        emu.check(emu.nsr_frustum_mask(fp(w2c), fp(o), fx, fy, cx, cy, int(H), int(W), ptr(depth), ptr(xs), ptr(ys), ptr(zs),
                                       nx, ny, nz, ptr(ws), ptr(out), None))
        assert 0 < ref.sum() < ref.size and np.array_equal(out.astype(bool), ref), (pre, key)


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


# ----------------------------------------------------------------------------------------------------------------------
# C ABI v2 entry points of the fused mapping iteration, on the emulator
# ----------------------------------------------------------------------------------------------------------------------
def _window_case(seed=5, K=3, n=37):
    import scene_util as su
    sc = su.make_scene(seed=seed, n_rays=8, small=True)
    H, W, fx, fy, cx, cy = sc["intr"]
    g = torch.Generator().manual_seed(seed)
    frames = []
    for k in range(K):
        c2w = sc["c2w"].clone()
        c2w[:3, 3] += 0.03 * k
        frames.append((c2w[:3].contiguous() if k % 2 else c2w, sc["depth_img"] * (1 + 0.1 * k), torch.rand((H, W, 3), generator=g)))
    idx = torch.randint((H - 8) * (W - 10), (K * n,), generator=g)
    return sc, frames, idx, (4, H - 4, 5, W - 5)


def test_get_samples_window_and_pose_grad(emu):
    """nsr_get_samples_window = the per-frame get_samples loop + torch.cat (Mapper.py:437-468) + the bounding-box mask (:471-481);
    nsr_pose_grad = autograd of common.py:74-88 w.r.t. the poses"""
    import ctypes as C
    from emu_harness import ptr
    from nice_slam_amd import _capi
    from oracle import nice_oracle as orc
    sc, frames, idx, (H0, H1, W0, W1) = _window_case()
    H, W, fx, fy, cx, cy = sc["intr"]
    K, n = len(frames), idx.numel() // len(frames)
    N = K * n
    fr = (_capi.NsrFrame * K)()
    hold = []
    for k, (c2w, d, col) in enumerate(frames):
        arrs = [np.ascontiguousarray(d.numpy(), dtype=np.float32), np.ascontiguousarray(col.numpy(), dtype=np.float32), np.ascontiguousarray(c2w.numpy(), dtype=np.float32)]
        hold += arrs
        fr[k].depth, fr[k].color, fr[k].c2w, fr[k].c2w_stride = arrs[0].ctypes.data, arrs[1].ctypes.data, arrs[2].ctypes.data, 4
    ro, rd = np.full((N, 3), np.nan, np.float32), np.full((N, 3), np.nan, np.float32)
    gd, gc = np.full((N,), np.nan, np.float32), np.full((N, 3), np.nan, np.float32)
    keep, kmax = np.full((N,), 9, np.uint8), np.zeros((1,), np.float32)
    b = sc["bound"]
    lo, hi = (C.c_double * 3)(*b[:, 0].tolist()), (C.c_double * 3)(*b[:, 1].tolist())
    idn = idx.numpy().copy()
    emu.check(emu.nsr_get_samples_window(ptr(idn), K, n, H0, H1, W0, W1, W, fx, fy, cx, cy, fr, ptr(ro), ptr(rd), ptr(gd), ptr(gc),
                                         lo, hi, ptr(keep), ptr(kmax), None))
    c2ws = [f[0].clone().requires_grad_(True) for f in frames]
    parts = [orc.pixel_rays(idx[k * n:(k + 1) * n], H0, H1, W0, W1, fx, fy, cx, cy, c2ws[k], frames[k][1], frames[k][2]) for k in range(K)]
    o_r, d_r, gd_r, gc_r = (torch.cat([p[i] for p in parts]) for i in range(4))
    assert np.array_equal(ro, o_r.detach().numpy()) and np.array_equal(rd, d_r.detach().numpy())
    assert np.array_equal(gd, gd_r.numpy()) and np.array_equal(gc, gc_r.numpy())
    t = (b.unsqueeze(0) - o_r.detach().unsqueeze(-1)) / d_r.detach().unsqueeze(-1)
    t, _ = torch.min(torch.max(t, dim=2)[0], dim=1)
    ref_keep = t >= gd_r
    assert np.array_equal(keep.astype(bool), ref_keep.numpy()) and 0.05 < ref_keep.float().mean() < 0.99
    assert kmax[0] == gd_r[ref_keep].max().item()
    # pose gradients
    g = torch.Generator().manual_seed(1)
    wo, wd = torch.randn((N, 3), generator=g), torch.randn((N, 3), generator=g)
    ((o_r * wo).sum() + (d_r * wd).sum()).backward()
    out = np.zeros((K, 4, 4), np.float32)
    won, wdn = wo.numpy().copy(), wd.numpy().copy()
    emu.check(emu.nsr_pose_grad(ptr(idn), K, n, H0, H1, W0, W1, fx, fy, cx, cy, ptr(won), ptr(wdn), ptr(out), 16, None))
    for k in range(K):
        ref = c2ws[k].grad.numpy()
        assert rel_err(out[k, :3], ref[:3]) < 1e-5, k
        assert np.all(out[k, 3] == 0)


def _philox_word(c, key):
    """philox4x32-10, first output word, restated from Salmon et al. (SC'11) -- the test's own copy, vectorised over counters
    c [n, 4] uint32; key (k0, k1)."""
    c = c.astype(np.uint64).copy()
    k0, k1 = np.uint64(key[0]), np.uint64(key[1])
    M0, M1, m32 = np.uint64(0xD2511F53), np.uint64(0xCD9E8D57), np.uint64(0xFFFFFFFF)
    for _ in range(10):
        p0, p1 = M0 * c[:, 0], M1 * c[:, 2]
        n0 = ((p1 >> np.uint64(32)) ^ c[:, 1] ^ k0) & m32
        n2 = ((p0 >> np.uint64(32)) ^ c[:, 3] ^ k1) & m32
        c = np.stack([n0, p1 & m32, n2, p0 & m32], 1)
        k0, k1 = (k0 + np.uint64(0x9E3779B9)) & m32, (k1 + np.uint64(0xBB67AE85)) & m32
    return c[:, 0]


def test_window_kernel_draws_its_own_pixels(emu):
    """nsr_get_samples_window_draw: the index draw of common.py:99 inside the window kernel -- philox4x32-10(counter = (ray,
    call), key = seed) mapped to [0, crop pixels) -- equals the test's own philox, the rays equal those of the explicit-index
    entry point on the drawn indices, and the kernel advances the call counter (a second call draws other pixels)."""
    import ctypes as C
    from emu_harness import ptr
    from nice_slam_amd import _capi
    sc, frames, idx, (H0, H1, W0, W1) = _window_case()
    H, W, fx, fy, cx, cy = sc["intr"]
    K, n = len(frames), 300                                  # two blocks of 256 per frame: the last-block hand-off is exercised
    N = K * n
    fr = (_capi.NsrFrame * K)()
    hold = []
    for k, (c2w, d, col) in enumerate(frames):
        arrs = [np.ascontiguousarray(d.numpy(), dtype=np.float32), np.ascontiguousarray(col.numpy(), dtype=np.float32), np.ascontiguousarray(c2w.numpy(), dtype=np.float32)]
        hold += arrs
        fr[k].depth, fr[k].color, fr[k].c2w, fr[k].c2w_stride = arrs[0].ctypes.data, arrs[1].ctypes.data, arrs[2].ctypes.data, 4
    b = sc["bound"]
    lo, hi = (C.c_double * 3)(*b[:, 0].tolist()), (C.c_double * 3)(*b[:, 1].tolist())
    seed = 0x1234567811223344
    state = np.array([seed, 7, 0, 0], dtype=np.uint64)
    draws = []
    for call in (7, 8):
        ind = np.full((N,), -1, np.int64)
        ro, rd = np.full((N, 3), np.nan, np.float32), np.full((N, 3), np.nan, np.float32)
        gd, gc = np.full((N,), np.nan, np.float32), np.full((N, 3), np.nan, np.float32)
        keep, kmax = np.full((N,), 9, np.uint8), np.zeros((1,), np.float32)
        emu.check(emu.nsr_get_samples_window_draw(ptr(ind), ptr(state), K, n, H0, H1, W0, W1, W, fx, fy, cx, cy, fr, ptr(ro), ptr(rd),
                                                  ptr(gd), ptr(gc), lo, hi, ptr(keep), ptr(kmax), None))
        assert state.tolist() == [seed, call + 1, 0, 0]
        ctr = np.zeros((N, 4), np.uint32)
        ctr[:, 0], ctr[:, 2] = np.arange(N), call
        r = _philox_word(ctr, (seed & 0xFFFFFFFF, seed >> 32))
        crop = (H1 - H0) * (W1 - W0)
        assert np.array_equal(ind, ((r * np.uint64(crop)) >> np.uint64(32)).astype(np.int64))
        assert ind.min() >= 0 and ind.max() < crop
        ro2, rd2 = np.full((N, 3), np.nan, np.float32), np.full((N, 3), np.nan, np.float32)
        gd2, gc2 = np.full((N,), np.nan, np.float32), np.full((N, 3), np.nan, np.float32)
        keep2, kmax2 = np.full((N,), 9, np.uint8), np.zeros((1,), np.float32)
        emu.check(emu.nsr_get_samples_window(ptr(ind), K, n, H0, H1, W0, W1, W, fx, fy, cx, cy, fr, ptr(ro2), ptr(rd2), ptr(gd2), ptr(gc2),
                                             lo, hi, ptr(keep2), ptr(kmax2), None))
        for a_, b_ in ((ro, ro2), (rd, rd2), (gd, gd2), (gc, gc2), (keep, keep2), (kmax, kmax2)):
            assert np.array_equal(a_, b_)
        draws.append(ind)
    assert (draws[0] != draws[1]).mean() > 0.9


def test_window_kernel_as_first_launch_of_a_fused_iteration(emu):
    """nsr_get_samples_window_fused (ABI 7): the same rays / mask as the plain entry points (explicit indices and drawn ones), the
    span zero-filled to the float (lengths that are not multiples of four, lengths spread over many fill blocks, an empty span),
    the header {loss = 0, kept max, 0} written by the launch itself over whatever was there, the hand-off words of the state zero
    again afterwards and the call counter advanced only when the kernel drew."""
    import ctypes as C
    from emu_harness import ptr
    from nice_slam_amd import _capi
    sc, frames, idx, (H0, H1, W0, W1) = _window_case()
    H, W, fx, fy, cx, cy = sc["intr"]
    K, n = len(frames), 300
    N = K * n
    fr = (_capi.NsrFrame * K)()
    hold = []
    for k, (c2w, d, col) in enumerate(frames):
        arrs = [np.ascontiguousarray(d.numpy(), dtype=np.float32), np.ascontiguousarray(col.numpy(), dtype=np.float32), np.ascontiguousarray(c2w.numpy(), dtype=np.float32)]
        hold += arrs
        fr[k].depth, fr[k].color, fr[k].c2w, fr[k].c2w_stride = arrs[0].ctypes.data, arrs[1].ctypes.data, arrs[2].ctypes.data, 4
    b = sc["bound"]
    lo, hi = (C.c_double * 3)(*b[:, 0].tolist()), (C.c_double * 3)(*b[:, 1].tolist())
    seed = 0x0BADC0DE12345678
    state = np.array([seed, 3, 0, 0], dtype=np.uint64)
    g = np.random.default_rng(0)
    given = g.integers(0, (H1 - H0) * (W1 - W0), size=N).astype(np.int64)
    for zero_n, draw in ((0, False), (5, True), (4 * 8192 * 3 + 2, False), (70001, True)):
        calls = int(state[1])
        ind = np.full((N,), -1, np.int64) if draw else given.copy()
        ro, rd = np.full((N, 3), np.nan, np.float32), np.full((N, 3), np.nan, np.float32)
        gd, gc = np.full((N,), np.nan, np.float32), np.full((N, 3), np.nan, np.float32)
        keep = np.full((N,), 9, np.uint8)
        Z = np.full((4 + zero_n + 4,), np.nan, np.float32)              # header | span | guard
        Z[-4:] = 7.0
        emu.check(emu.nsr_get_samples_window_fused(None if draw else ptr(ind), ptr(ind) if draw else None, ptr(state), K, n, H0, H1, W0, W1, W,
                                                   fx, fy, cx, cy, fr, ptr(ro), ptr(rd), ptr(gd), ptr(gc), lo, hi, ptr(keep),
                                                   ptr(Z), ptr(Z[4:]) if zero_n else None, zero_n, None))
        assert state.tolist() == [seed, calls + (1 if draw else 0), 0, 0]
        ro2, rd2 = np.full((N, 3), np.nan, np.float32), np.full((N, 3), np.nan, np.float32)
        gd2, gc2 = np.full((N,), np.nan, np.float32), np.full((N, 3), np.nan, np.float32)
        keep2, kmax2 = np.full((N,), 9, np.uint8), np.zeros((1,), np.float32)
        emu.check(emu.nsr_get_samples_window(ptr(ind), K, n, H0, H1, W0, W1, W, fx, fy, cx, cy, fr, ptr(ro2), ptr(rd2), ptr(gd2), ptr(gc2),
                                             lo, hi, ptr(keep2), ptr(kmax2), None))
        for a_, b_ in ((ro, ro2), (rd, rd2), (gd, gd2), (gc, gc2), (keep, keep2)):
            assert np.array_equal(a_, b_)
        assert Z[0] == 0 and Z[1] == 0 and Z[3] == 0 and Z[2] == kmax2[0] and kmax2[0] > 0
        assert np.all(Z[4:4 + zero_n] == 0) and np.all(Z[-4:] == 7.0)
        if draw:
            assert ind.min() >= 0 and not np.array_equal(ind, given)
    # a span that is not 16-byte aligned is refused
    Z = np.zeros((64,), np.float32)
    assert emu.nsr_get_samples_window_fused(ptr(given), None, ptr(state), K, n, H0, H1, W0, W1, W, fx, fy, cx, cy, fr, ptr(ro), ptr(rd), ptr(gd),
                                            ptr(gc), lo, hi, ptr(keep), ptr(Z), ptr(Z[5:]), 8, None) != 0


def test_fused_mapping_loss_in_the_forward(emu):
    """render forward with the mapping loss (Mapper.py:487-493): the loss value and d loss / d outputs it writes equal torch's
    on the forward's own outputs (masked by the bounding-box mask, depth term on gt > 0 only, colour term in the colour stage)"""
    import scene_util as su
    from emu_harness import HostScene, ptr
    import ctypes as C
    sc = su.make_scene(seed=14, n_rays=50, small=True)
    hs = HostScene(emu, sc["grids"], sc["params"], sc["bound"])
    g = torch.Generator().manual_seed(2)
    keep = (torch.rand((50,), generator=g) < 0.8).numpy().astype(np.uint8)
    gcol = sc["gt_color"].numpy().astype(np.float32).copy()
    for stage in ("middle", "color", "coarse"):
        ro, rd, gt = (np.ascontiguousarray(sc[k].numpy(), dtype=np.float32) for k in ("rays_o", "rays_d", "gt_depth"))
        kp = []
        a = hs._args(stage, ro, rd, gt, kp)
        n = 50
        S = 32 if stage == "coarse" else 48
        out = {"depth": np.full(n, np.nan), "var": np.full(n, np.nan), "rgb": np.full((n, 3), np.nan, np.float32)}
        a.depth, a.var, a.rgb = ptr(out["depth"]), ptr(out["var"]), ptr(out["rgb"])
        loss, dld, dlr = np.zeros(1), np.full(n, np.nan), np.full((n, 3), np.nan, np.float32)
        a.gt_color, a.keep, a.loss, a.w_color, a.dl_depth, a.dl_rgb = ptr(gcol), ptr(keep), ptr(loss), 0.2, ptr(dld), ptr(dlr)
        emu.check(emu.nsr_render_fwd(C.byref(a), None))
        depth = torch.from_numpy(out["depth"]).requires_grad_(True)
        rgb = torch.from_numpy(out["rgb"]).requires_grad_(True)
        k_t, gd_t = torch.from_numpy(keep.astype(bool)), sc["gt_depth"]
        ref = (torch.abs(gd_t - depth) * (k_t & (gd_t > 0))).sum()
        if stage == "color":
            ref = ref + (0.2 * (torch.abs(torch.from_numpy(gcol) - rgb) * k_t[:, None])).sum()
        ref.backward()
        assert abs(loss[0] - float(ref)) < 1e-6 * abs(float(ref)), (stage, loss[0], float(ref))
        assert np.array_equal(dld, depth.grad.numpy()), stage
        want = rgb.grad.numpy() if stage == "color" else np.zeros((n, 3), np.float32)
        assert np.allclose(dlr, want, rtol=0, atol=1e-7), stage


def test_masked_adam_multi_and_pack_rows(emu):
    import ctypes as C
    from emu_harness import ptr
    from nice_slam_amd import _capi
    rng = np.random.RandomState(4)
    shapes = [(3, 4, 5), (2, 6, 3)]
    P = [rng.randn(int(np.prod(s)), 32).astype(np.float32) for s in shapes]
    Pm = [p.copy() for p in P]
    M = [np.zeros_like(p) for p in P]; V = [np.zeros_like(p) for p in P]
    M1 = [np.zeros_like(p) for p in P]; V1 = [np.zeros_like(p) for p in P]
    masks = [(rng.rand(p.shape[0]) < 0.6).astype(np.uint8) for p in P]
    steps = np.zeros(2, np.int32)
    scratch = np.zeros(8, np.float32)
    lrs = [0.1, 0.005]
    for t in range(1, 4):
        G = [(rng.randn(*p.shape) * 1e-2).astype(np.float32) for p in P]
        Gm = [g.copy() for g in G]
        arr = (_capi.NsrAdamGrid * 2)()
        for i in range(2):
            arr[i].p, arr[i].g, arr[i].m, arr[i].v = Pm[i].ctypes.data, Gm[i].ctypes.data, M[i].ctypes.data, V[i].ctypes.data
            arr[i].voxel_mask, arr[i].n_voxels, arr[i].step, arr[i].lr = masks[i].ctypes.data, P[i].shape[0], steps.ctypes.data + 4 * i, lrs[i]
        emu.check(emu.nsr_masked_adam_multi(arr, 2, 0.9, 0.999, 1e-8, 1, ptr(scratch), None))
        for i in range(2):
            emu.check(emu.nsr_masked_adam(ptr(P[i]), ptr(G[i]), ptr(M1[i]), ptr(V1[i]), ptr(masks[i]), P[i].shape[0],
                                          lrs[i] / (1 - 0.9 ** t), 0.9, 0.999, 1e-8, (1 - 0.999 ** t) ** 0.5, None))
            sel = masks[i].astype(bool)
            assert np.all(Gm[i][sel] == 0) and np.array_equal(Gm[i][~sel], G[i][~sel])          # zero_grad on the consumed voxels only
    assert steps.tolist() == [3, 3]
    for i in range(2):
        assert np.abs(P[i] - Pm[i]).max() <= 1e-6 * np.abs(P[i]).max()
    # row packing: gather, (all-reduce stand-in: x2), scatter
    rows = [np.flatnonzero(m).astype(np.int64) for m in masks]
    flat, pose = rng.randn(101).astype(np.float32), rng.randn(48).astype(np.float32)
    ra = (_capi.NsrRows * 2)()
    for i in range(2):
        ra[i].grid, ra[i].rows, ra[i].n_rows = P[i].ctypes.data, rows[i].ctypes.data, rows[i].size
    sa = (_capi.NsrSpan * 2)()
    sa[0].ptr, sa[0].n, sa[1].ptr, sa[1].n = flat.ctypes.data, flat.size, pose.ctypes.data, pose.size
    total = sum(r.size for r in rows) * 32 + flat.size + pose.size
    buf = np.full(total, np.nan, np.float32)
    emu.check(emu.nsr_pack_rows(ra, 2, sa, 2, ptr(buf), 0, None))
    want = np.concatenate([P[0][rows[0]].ravel(), P[1][rows[1]].ravel(), flat, pose])
    assert np.array_equal(buf, want)
    before = [p.copy() for p in P]
    buf *= 2
    emu.check(emu.nsr_pack_rows(ra, 2, sa, 2, ptr(buf), 1, None))
    for i in range(2):
        sel = masks[i].astype(bool)
        assert np.array_equal(P[i][sel], 2 * before[i][sel]) and np.array_equal(P[i][~sel], before[i][~sel])
    assert np.array_equal(flat * 0.5 * 2, flat) and np.array_equal(pose, want[-48:] * 2)


@pytest.mark.parametrize("n,dyn,col", [(203, True, True), (1, True, True), (2, True, False), (255, True, False), (256, True, True), (300, True, True),
                                       (1024, True, True), (1500, True, True), (5000, True, True), (64, False, True),
                                       (300, False, False)])       # <= block size: one ray per thread, rank by counting; <= 4096: radix select on cached keys; above: recomputed
def test_tracking_loss_matches_the_reference_expression(emu, n, dyn, col):
    """nsr_tracking_loss vs Tracker.optimize_cam_in_batch's loss (src/Tracker.py:108-124) on the COMPACTED batch, with autograd
    for d loss / d depth and d loss / d rgb: the median of `tmp` (lower middle element) is found without a sort."""
    from emu_harness import ptr
    g = torch.Generator().manual_seed(100 + n)
    gd = (torch.rand((n,), generator=g) * 4 + 0.5).float()
    gd[torch.rand((n,), generator=g) < 0.1] = 0.0
    depth = (gd.double() + torch.randn((n,), generator=g).double() * 0.3)
    depth[torch.rand((n,), generator=g) < 0.05] *= 3.0                    # outliers: the dynamic-object mask must cut some rays
    if n > 10:
        depth[5] = gd[5].double()                                          # |0|: zero gradient
    var = torch.rand((n,), generator=g).double() * 0.2
    rgb, gc = torch.rand((n, 3), generator=g), torch.rand((n, 3), generator=g)
    if n > 10:
        rgb[7, 1] = gc[7, 1]
    keep = torch.rand((n,), generator=g) < 0.85
    if n <= 2:
        keep[:] = True
    w = 0.5
    # reference, on the compacted batch
    d_l, r_l = depth[keep].clone().requires_grad_(True), rgb[keep].clone().requires_grad_(True)
    g_k, c_k, v_k = gd[keep], gc[keep], var[keep]
    tmp = torch.abs(g_k - d_l) / torch.sqrt(v_k + 1e-10)
    mask = ((tmp < 10 * tmp.median()) & (g_k > 0)) if dyn else (g_k > 0)
    ref = tmp[mask].sum()
    if col:
        ref = ref + w * torch.abs(c_k - r_l)[mask].sum()
    ref.backward()
    want_d = torch.zeros(n, dtype=torch.float64); want_d[keep] = d_l.grad
    want_r = torch.zeros((n, 3)); want_r[keep] = r_l.grad if r_l.grad is not None else torch.zeros_like(r_l)
    if dyn and n > 100:
        assert int(mask.sum()) < int((g_k > 0).sum())                      # the median test really removes rays in this case
    loss = np.zeros(1)
    dld, dlr = np.full(n, np.nan), np.full((n, 3), np.nan, np.float32)
    a = [np.ascontiguousarray(x) for x in (gd.numpy(), gc.numpy(), keep.numpy().astype(np.uint8), depth.numpy(), var.numpy(), rgb.numpy())]
    emu.check(emu.nsr_tracking_loss(n, ptr(a[0]), ptr(a[1]), ptr(a[2]), ptr(a[3]), ptr(a[4]), ptr(a[5]), int(dyn), int(col), w,
                                    ptr(loss), ptr(dld), ptr(dlr) if col else None, None))
    assert abs(loss[0] - float(ref.detach())) <= 1e-6 * abs(float(ref.detach())) + 1e-12, (loss[0], float(ref.detach()))
    # (torch's CPU sqrt for doubles is not correctly rounded -- 1 ulp off IEEE on ~1 % of the inputs -- so 1/sqrt agrees to 1e-15, not bitwise)
    assert np.array_equal(dld == 0, want_d.numpy() == 0) and np.allclose(dld, want_d.numpy(), rtol=1e-14, atol=0)
    if col:
        assert np.array_equal(dlr, want_r.numpy())
    # keep = NULL: every ray counts
    loss2, dld2 = np.zeros(1), np.full(n, np.nan)
    emu.check(emu.nsr_tracking_loss(n, ptr(a[0]), ptr(a[1]), None, ptr(a[3]), ptr(a[4]), ptr(a[5]), int(dyn), 0, w, ptr(loss2), ptr(dld2), None, None))
    tmp = torch.abs(gd - depth) / torch.sqrt(var + 1e-10)
    mask = ((tmp < 10 * tmp.median()) & (gd > 0)) if dyn else (gd > 0)
    assert abs(loss2[0] - float(tmp[mask].sum())) <= 1e-9 * float(tmp[mask].sum()) + 1e-12
    assert np.array_equal(dld2 != 0, (mask & (gd.double() != depth)).numpy())


def test_camera_from_tensor_forward_and_backward(emu):
    """nsr_camera_from_tensor vs quad2rotation / get_camera_from_tensor (src/common.py:137-176, restated with torch ops) and
    autograd through them; quaternions are NOT normalised (the optimiser moves all four components)."""
    from emu_harness import ptr
    g = torch.Generator().manual_seed(9)
    B = 5
    cam = torch.randn((B, 7), generator=g)
    cam[:, :4] += torch.tensor([1.0, 0.0, 0.0, 0.0])
    cam[0, :4] = torch.tensor([1.0, 0.0, 0.0, 0.0])                      # identity rotation
    cam[1, :4] *= 1.7                                                    # far from unit norm

    def ref(t):
        qr, qi, qj, qk = t[:, 0], t[:, 1], t[:, 2], t[:, 3]
        two_s = 2.0 / (t[:, :4] * t[:, :4]).sum(-1)
        R = torch.stack([1 - two_s * (qj ** 2 + qk ** 2), two_s * (qi * qj - qk * qr), two_s * (qi * qk + qj * qr),
                         two_s * (qi * qj + qk * qr), 1 - two_s * (qi ** 2 + qk ** 2), two_s * (qj * qk - qi * qr),
                         two_s * (qi * qk - qj * qr), two_s * (qj * qk + qi * qr), 1 - two_s * (qi ** 2 + qj ** 2)], -1).reshape(-1, 3, 3)
        return torch.cat([R, t[:, 4:, None]], 2)

    t = cam.clone().requires_grad_(True)
    want = ref(t)
    w = torch.randn((B, 3, 4), generator=g)
    (want * w).sum().backward()
    c_np = np.ascontiguousarray(cam.numpy())
    rt = np.full((B, 3, 4), np.nan, np.float32)
    emu.check(emu.nsr_camera_from_tensor(ptr(c_np), B, ptr(rt), None, None, None))
    assert np.allclose(rt, want.detach().numpy(), rtol=0, atol=2e-6)
    assert np.array_equal(rt[0, :, :3], np.eye(3, dtype=np.float32))
    d = np.full((B, 7), np.nan, np.float32)
    w_np = np.ascontiguousarray(w.numpy())
    emu.check(emu.nsr_camera_from_tensor(ptr(c_np), B, None, ptr(w_np), ptr(d), None))
    assert np.allclose(d, t.grad.numpy(), rtol=1e-5, atol=1e-5 * float(t.grad.abs().max()))


def test_full_size_forward_blocks_in_a_subprocess():
    """nsr_render_fwd shrinks its blocks for small batches (one block per CU); NSR_FWD_SMALL=0 keeps the 12-tile blocks the
    large batches use, so that the small CPU scenes cover that shape too (the switch is read once per process)."""
    import subprocess, sys
    if os.environ.get("NSR_FWD_SMALL") == "0":
        pytest.skip("already the inner run")
    env = dict(os.environ, NSR_FWD_SMALL="0", NSR_EMU_SAVE_ACTS="0")      # ... through the one-launch forward kernel (no activation buffer)
    r = subprocess.run([sys.executable, "-m", "pytest", os.path.abspath(__file__), "-x", "-q", "-k",
                        "golden_forward or random_scene or fused_mapping_loss or other_sample_counts"],
                       env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    assert " passed" in r.stdout


@pytest.mark.parametrize("cells,slots", [("0", None), ("64", None), ("64", "16")])
def test_hot_voxel_table_extremes_in_a_subprocess(cells, slots):
    """The dX kernel's hot-voxel table (LDS rows for the samples next to their ray's origin, claimed with a compare-and-swap,
    flushed once per block; since round 5 as many slots as the block's LDS has left): NSR_DX_HOT_CELLS=0 switches it off, 64 sends
    EVERY sample of these small scenes through it, and with NSR_DX_HOT_SLOTS=16 most of them find their slot taken by another voxel
    (the fall-back to memory atomics).  All against the oracle, like the default of six cells in the rest of this file (the
    switches are read once per process)."""
    import subprocess, sys
    if os.environ.get("NSR_DX_HOT_CELLS") is not None:
        pytest.skip("already the inner run")
    env = dict(os.environ, NSR_DX_HOT_CELLS=cells)
    if slots is not None:
        env["NSR_DX_HOT_SLOTS"] = slots
    r = subprocess.run([sys.executable, "-m", "pytest", os.path.abspath(__file__), "-x", "-q", "-k",
                        "golden_forward or random_scene or saved_activations_equal"],
                       env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    assert " passed" in r.stdout


@pytest.mark.parametrize("stage", ["coarse", "middle", "fine", "color"])
def test_saved_activations_equal_the_forward_rerun(emu, stage):
    """nsr_render_args.acts: the three-launch forward (nsr_fwd2.h) writes every decoder's hidden states, relu masks and grid
    features for the split backward (nsr_bwd2.h) -- its outputs equal the one-launch forward kernel's (no activation buffer)
    BIT FOR BIT, every slot of every point is written, and the backward's specialisations (no parameter gradients; relu masks
    only) agree with the full one.  Ragged ray count (a partial last tile), small persistent grid."""
    s = make_scene(seed=120, n_rays=29, small=True)
    res = {}
    for mode in (True, False):
        sc = _host_scene(emu, s["grids"], s["params"], s["bound"].numpy())
        sc.save_acts = mode
        fwd = sc.forward(stage, s["rays_o"].numpy(), s["rays_d"].numpy(), s["gt_depth"].numpy())
        assert ("acts" in fwd) == mode
        if mode:
            npts = fwd["raw"].shape[0] * fwd["raw"].shape[1]
            npad = (npts + 15) // 16 * 16
            passes = max(1, ["coarse", "middle", "fine", "color"].index(stage))
            slots = fwd["acts"][:passes * 13 * npad * 16].reshape(passes, npad // 16, 13, 16, 16)   # [pass][tile][slot][point][lane group, 4]
            slots = slots.transpose(0, 2, 1, 3, 4).reshape(passes, 13, npad, 16)
            assert not np.isnan(slots[:, :12, :npts]).any()     # every hidden-state / feature slot of every point was written (slot 12: mask bits)
        res[mode] = (fwd, sc.backward(stage, fwd, s["w"]["depth"].numpy(), s["w"]["var"].numpy(), s["w"]["rgb"].numpy(), max_blocks=2))
    for k in ("depth", "var", "rgb", "raw"):
        assert np.array_equal(res[True][0][k], res[False][0][k]), k
    for k, v in res[False][1].items():                              # (mode False: backward() ran the saving forward itself)
        assert rel_err(res[True][1][k], v) < 1e-5, (stage, k)      # (emulated blocks run on several OS threads: atomics order)
    # without parameter gradients (tracking): the specialisation without accumulators takes the same path
    sc = _host_scene(emu, s["grids"], s["params"], s["bound"].numpy())
    fwd = sc.forward(stage, s["rays_o"].numpy(), s["rays_d"].numpy(), s["gt_depth"].numpy())
    r2 = sc.backward(stage, fwd, s["w"]["depth"].numpy(), s["w"]["var"].numpy(), s["w"]["rgb"].numpy(), want_params=False, want_grid=False)
    for k in ("d_rays_o", "d_rays_d"):
        assert rel_err(r2[k], res[False][1][k]) < 1e-5, k
    # ... and needs the relu masks only (nsr_render_args.acts_masks_only: 1 of the 13 KB per tile and decoder is written)
    sc = _host_scene(emu, s["grids"], s["params"], s["bound"].numpy())
    sc.acts_masks_only = True
    fwd = sc.forward(stage, s["rays_o"].numpy(), s["rays_d"].numpy(), s["gt_depth"].numpy())
    r3 = sc.backward(stage, fwd, s["w"]["depth"].numpy(), s["w"]["var"].numpy(), s["w"]["rgb"].numpy(), want_params=False, want_grid=True)
    for k, v in r3.items():
        assert rel_err(v, res[False][1][k]) < 1e-5, k
    with pytest.raises(Exception, match="masks only"):
        sc.backward(stage, fwd, s["w"]["depth"].numpy(), s["w"]["var"].numpy(), s["w"]["rgb"].numpy(), want_params=True)
    # per decoder (ABI 8: bits 1..3 of acts_masks_only): only the LAST decoder of the stage will be stepped (the mapper steps the colour
    # decoder, Mapper.py:335-341) -- the others save their relu masks alone, the backward still yields every grid / ray gradient and
    # that decoder's parameter gradients, and refuses parameter gradients of a decoder whose activations were not saved
    from emu_harness import stage_slots
    slots = stage_slots(stage)
    if len(slots) > 1:
        sc = _host_scene(emu, s["grids"], s["params"], s["bound"].numpy())
        sc.acts_masks_only = sum(2 << i for i in range(len(slots) - 1))
        fwd = sc.forward(stage, s["rays_o"].numpy(), s["rays_d"].numpy(), s["gt_depth"].numpy())
        npts = s["rays_o"].shape[0] * 48
        npad = (npts + 15) // 16 * 16
        sl = fwd["acts"][:len(slots) * 13 * npad * 16].reshape(len(slots), npad // 16, 13, 16, 16)
        assert np.isnan(sl[:-1, :, :12]).all() and not np.isnan(sl[-1, :, :12].reshape(-1, 16)[:npts]).any()     # hidden states of the last decoder only
        if stage == "fine":        # the fine decoder's dW reads the middle pass's features ([c_fine | c_mid]): refused, not silently wrong
            with pytest.raises(Exception, match="masks only for the middle decoder"):
                sc.backward(stage, fwd, s["w"]["depth"].numpy(), s["w"]["var"].numpy(), s["w"]["rgb"].numpy(), want_params=(slots[-1],))
        else:
            r4 = sc.backward(stage, fwd, s["w"]["depth"].numpy(), s["w"]["var"].numpy(), s["w"]["rgb"].numpy(), want_params=(slots[-1],))
            assert any(k.startswith("dparam/" + slots[-1]) for k in r4) and not any(k.startswith("dparam/" + slots[0]) for k in r4)
            for k, v in r4.items():
                assert rel_err(v, res[False][1][k]) < 1e-5, k
        with pytest.raises(Exception, match="masks only"):
            sc.backward(stage, fwd, s["w"]["depth"].numpy(), s["w"]["var"].numpy(), s["w"]["rgb"].numpy(), want_params=(slots[0],))


@pytest.mark.parametrize("stage,acts", [("color", True), ("fine", False), ("coarse", True)])
def test_grad_scale_multiplies_every_gradient(emu, stage, acts):
    """nsr_bwd_args.grad_scale (the incoming gradient of a fused loss node, a device scalar): every gradient of the backward
    is the unscaled one times the scalar."""
    s = make_scene(seed=121, n_rays=13, small=True)
    res = {}
    for sc_ in (None, -2.5):
        sc = _host_scene(emu, s["grids"], s["params"], s["bound"].numpy())
        sc.save_acts = acts
        fwd = sc.forward(stage, s["rays_o"].numpy(), s["rays_d"].numpy(), s["gt_depth"].numpy())
        res[sc_] = sc.backward(stage, fwd, s["w"]["depth"].numpy(), s["w"]["var"].numpy(), s["w"]["rgb"].numpy(), grad_scale=sc_)
    for k, v in res[None].items():
        assert rel_err(res[-2.5][k], -2.5 * v) < 1e-5, (stage, k)


@pytest.mark.parametrize("stage", ["coarse", "fine", "color"])
def test_fused_forward_leaves_d_raw_for_the_split_backward(emu, stage):
    """Fused mapping loss + activation buffer: the forward's loss epilogue writes d raw / positions itself and nsr_render_bwd,
    handed back the very derivative arrays of that forward, skips its compositor-backward kernel -- same gradients as the
    backward that is given copies of those arrays (and therefore runs comp_bwd_kernel), also under a grad_scale."""
    s = make_scene(seed=122, n_rays=21, small=True)
    g = torch.Generator().manual_seed(5)
    keep = (torch.rand((21,), generator=g) < 0.8).numpy().astype(np.uint8)
    fl = {"gt_color": s["gt_color"].numpy(), "keep": keep, "w_color": 0.2}
    res = {}
    for mode in ("forward", "comp_bwd"):
        sc = _host_scene(emu, s["grids"], s["params"], s["bound"].numpy())
        fwd = sc.forward(stage, s["rays_o"].numpy(), s["rays_d"].numpy(), s["gt_depth"].numpy(), fused_loss=fl)
        assert np.isfinite(fwd["dl_depth"]).all() and fwd["loss"][0] > 0
        if mode == "forward":
            res[mode] = sc.backward(stage, fwd, None, None, None, from_forward=True, grad_scale=1.75)
        else:
            res[mode] = sc.backward(stage, fwd, fwd["dl_depth"].copy(), None, fwd["dl_rgb"].copy() if stage == "color" else None, grad_scale=1.75)
    assert set(res["forward"]) == set(res["comp_bwd"])
    for k, v in res["comp_bwd"].items():
        assert rel_err(res["forward"][k], v) < 1e-5, (stage, k)
    # a C-API caller that edits dl_* IN PLACE (masking, re-weighting) and hands the same pointers back without the opt-in
    # flag: the edit must be honoured (ABI 6: nsr_bwd_args.loss_grads_from_forward; pointer equality alone decided before)
    sc = _host_scene(emu, s["grids"], s["params"], s["bound"].numpy())
    fwd = sc.forward(stage, s["rays_o"].numpy(), s["rays_d"].numpy(), s["gt_depth"].numpy(), fused_loss=fl)
    fwd["dl_depth"] *= -3.0
    fwd["dl_rgb"] *= -3.0
    edited = sc.backward(stage, fwd, None, None, None, in_place=True, grad_scale=1.75)
    for k, v in res["comp_bwd"].items():
        assert rel_err(edited[k], -3.0 * v) < 1e-5, (stage, k)


@pytest.mark.parametrize("stage,n_surface,n_samples", [("coarse", 16, 32), ("middle", 16, 32), ("color", 16, 32), ("color", 5, 32),
                                                       ("middle", 0, 1), ("color", 1, 1), ("fine", 3, 60)])
def test_masked_rays_are_removed_from_the_batch(emu, stage, n_surface, n_samples):
    """nsr_render_args.skip_masked: the rays the bounding-box pre-filter rejects (keep == 0) are not rendered -- what the
    reference's compaction does (Mapper.py:471-481) -- in the forward passes, the compositor, dX and dW alike: the loss and every
    gradient equal those of the run that renders them and masks the loss (their terms are exact zeros there), the kept rays'
    outputs are bit-identical, the removed rays' outputs are 0.  n_surface = 5: 37 samples per ray, tiles that straddle two rays
    (a tile is skipped only if BOTH are removed); long runs of removed rays: whole tiles, whole dW ring slots stay empty.
    One, two and 63 samples per ray: the ends of the range of the kernels' point-index -> ray division (ray_of_point)."""
    s = make_scene(seed=123, n_rays=53, small=True)
    g = torch.Generator().manual_seed(9)
    keep = (torch.rand((53,), generator=g) < 0.7).numpy().astype(np.uint8)
    keep[10:25] = 0                                    # a run of removed rays
    keep[40] = 1
    res = {}
    for skip in (False, True):
        sc = _host_scene(emu, s["grids"], s["params"], s["bound"].numpy())
        sc.n_surface, sc.n_samples = n_surface, n_samples
        fl = {"gt_color": s["gt_color"].numpy(), "keep": keep, "w_color": 0.2, "skip_masked": skip}
        fwd = sc.forward(stage, s["rays_o"].numpy(), s["rays_d"].numpy(), s["gt_depth"].numpy(), fused_loss=fl)
        bwd = sc.backward(stage, fwd, None, None, None, from_forward=True, grad_scale=0.5, max_blocks=3)
        res[skip] = (fwd, bwd)
    f0, f1 = res[False][0], res[True][0]
    k = keep.astype(bool)
    assert abs(f0["loss"][0] - f1["loss"][0]) <= 1e-12 * abs(f0["loss"][0]) and f0["loss"][0] > 0
    for key in ("depth", "var", "rgb"):
        assert np.array_equal(f0[key][k], f1[key][k]), key
        assert not np.any(f1[key][~k]), key
    assert set(res[False][1]) == set(res[True][1])
    for key, v in res[False][1].items():
        assert rel_err(res[True][1][key], v) < 1e-5, (stage, key)


def test_flat_adam_is_torch_adam(emu):
    """nsr_flat_adam (decoder blobs / pose tensors: the dense rest of the callers' optimiser) against torch.optim.Adam on the CPU:
    three spans of odd lengths, five steps, one span at lr = 0 (moments still move), device-side step counts, zero_grad."""
    from emu_harness import ptr
    from nice_slam_amd import _capi
    g = torch.Generator().manual_seed(11)
    ns, lrs = [7, 15899, 35], [1e-3, 5e-3, 0.0]
    ref = [torch.randn(n, generator=g).requires_grad_(True) for n in ns]
    opt = torch.optim.Adam([{"params": [r], "lr": lr} for r, lr in zip(ref, lrs)])
    P = [r.detach().numpy().copy() for r in ref]
    M = [np.zeros_like(p) for p in P]; V = [np.zeros_like(p) for p in P]
    steps, scratch = np.zeros(3, np.int32), np.zeros(8, np.float32)
    for t in range(5):
        G = [(torch.randn(n, generator=g) * (10.0 ** (t - 2))).numpy() for n in ns]
        for r, gg in zip(ref, G):
            r.grad = torch.from_numpy(gg.copy())
        opt.step()
        Gk = [gg.copy() for gg in G]
        arr = (_capi.NsrAdamSpan * 3)()
        for i in range(3):
            arr[i].p, arr[i].g, arr[i].m, arr[i].v = P[i].ctypes.data, Gk[i].ctypes.data, M[i].ctypes.data, V[i].ctypes.data
            arr[i].n, arr[i].step, arr[i].lr = ns[i], steps.ctypes.data + 4 * i, lrs[i]
        emu.check(emu.nsr_flat_adam(arr, 3, 0.9, 0.999, 1e-8, 1 if t == 4 else 0, ptr(scratch), None))
        for i in range(3):
            assert np.array_equal(Gk[i], G[i]) if t < 4 else not Gk[i].any()
    assert steps.tolist() == [5, 5, 5]
    for i in range(3):
        assert rel_err(P[i], ref[i].detach().numpy()) < 2e-6, i
        assert rel_err(M[i], opt.state[ref[i]]["exp_avg"].numpy()) < 2e-6 and rel_err(V[i], opt.state[ref[i]]["exp_avg_sq"].numpy()) < 2e-6
    assert np.array_equal(P[2], ref[2].detach().numpy())            # lr = 0: parameters untouched, moments moved



@pytest.mark.parametrize("stage", ["coarse", "middle", "color"])
def test_consumed_gradient_voxel_masks(emu, stage):
    """nsr_render_args.grad_voxel_mask (ABI 8, opt-in): with `frustum_feature_selection` the mapper's optimiser holds only
    `val[mask]` (src/Mapper.py:315-333,394-401).  Given the same byte masks the scatter skips every other voxel: gradients of voxels
    inside a mask equal the dense run's (the emulator's adds happen in one order: exactly, whenever the run structure is the same;
    1e-6 otherwise -- a masked-out voxel between two samples of a run splits the run sum), voxels outside stay exactly zero, every
    other gradient is untouched."""
    s = make_scene(seed=131, n_rays=21, small=True)
    sc = _host_scene(emu, s["grids"], s["params"], s["bound"].numpy())
    w = s["w"]
    args = (w["depth"].numpy(), w["var"].numpy(), w["rgb"].numpy())
    fwd = sc.forward(stage, s["rays_o"].numpy(), s["rays_d"].numpy(), s["gt_depth"].numpy())
    dense = sc.backward(stage, fwd, *args)
    rng = np.random.default_rng(5)
    masks = {k: (rng.random(g.shape[:3]) < 0.6).astype(np.uint8) for k, g in sc.grids.items()}
    masks["middle"][...] = 1                                   # one grid fully selected, one partly, (colour stage) one without a mask
    use = {k: v for k, v in masks.items() if k != "color"}
    fwd = sc.forward(stage, s["rays_o"].numpy(), s["rays_d"].numpy(), s["gt_depth"].numpy())
    got = sc.backward(stage, fwd, *args, grad_voxel_masks=use)
    from emu_harness import stage_slots
    for slot in stage_slots(stage):
        k = "d_grid_" + slot
        a, b = got[k][0], dense[k][0]                          # [32, Z, Y, X]
        m = use.get(slot)
        if m is None:
            assert rel_err(a, b) < 1e-6, k
            continue
        inside = m.astype(bool)[None]
        assert np.all(a[:, ~inside[0]] == 0.0), k              # nothing was scattered outside the mask
        assert m.all() or np.abs(b[:, ~inside[0]]).max() > 0, k    # (and the dense run does put gradient there: the test has teeth)
        den = np.abs(b).max()
        assert np.abs(np.where(inside, a - b, 0.0)).max() <= 1e-6 * den, k
    for k in dense:
        if not k.startswith("d_grid_"):
            assert rel_err(got[k], dense[k]) < 1e-6, k
    # an all-zero mask: no grid gradient at all, everything else unchanged
    fwd = sc.forward(stage, s["rays_o"].numpy(), s["rays_d"].numpy(), s["gt_depth"].numpy())
    none = sc.backward(stage, fwd, *args, grad_voxel_masks={k: np.zeros_like(v) for k, v in masks.items()})
    for k in dense:
        if k.startswith("d_grid_"):
            assert not none[k].any(), k
        else:
            assert rel_err(none[k], dense[k]) < 1e-6, k
