"""GPU: the two unchanged callers of the hot path, restated in miniature, through the drop-in surface.

* mapping (src/Mapper.py:303-333,394-401,457-519): masked leaf -> ``val[mask] = val_grad`` -> get_samples ->
  render_batch_ray -> L1 losses -> backward -> Adam, a few iterations, staged middle -> fine -> color;
* tracking (src/Tracker.py:71-128): 7-vector pose -> c2w -> get_samples -> render_batch_ray(color) ->
  uncertainty-weighted loss with the median outlier mask -> gradient of the pose.
The same loops run on the CPU with the oracle; losses / gradients must agree."""
import numpy as np
import pytest
import torch

from scene_util import build_product, make_scene, rel_err
from oracle import nice_oracle as orc

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def quad2rotation(q):          # src/common.py:137-160 (restated)
    qr, qi, qj, qk = q[0], q[1], q[2], q[3]
    two_s = 2.0 / (q * q).sum()
    return torch.stack([
        torch.stack([1 - two_s * (qj ** 2 + qk ** 2), two_s * (qi * qj - qk * qr), two_s * (qi * qk + qj * qr)]),
        torch.stack([two_s * (qi * qj + qk * qr), 1 - two_s * (qi ** 2 + qk ** 2), two_s * (qj * qk - qi * qr)]),
        torch.stack([two_s * (qi * qk - qj * qr), two_s * (qj * qk + qi * qr), 1 - two_s * (qi ** 2 + qj ** 2)])])


def cam_to_c2w(cam):           # src/common.py:163-176
    return torch.cat([quad2rotation(cam[:4]), cam[4:, None]], 1)


def test_mapping_style_iterations():
    """A staged mapping loop (masked leaves, ``val[mask] = val_grad``, Adam; Mapper.py:303-333,394-401,457-519) on the CPU
    oracle defines the trajectory; the HIP path is evaluated TEACHER-FORCED on that trajectory (same state, same pixels at
    every iteration), so its loss and every gradient the optimiser would consume are compared in MAX-norm (Adam would turn
    1e-6 noise on near-zero gradient components into lr-sized parameter differences in a free-running comparison)."""
    import nice_slam_amd as nsa
    sc = make_scene(seed=31, n_rays=8, small=True)
    H, W, fx, fy, cx, cy = sc["intr"]
    renderer, dec, grids_dev = build_product(sc, DEV)
    g = torch.Generator().manual_seed(2)
    masks = {k: (torch.rand(v.shape[2:], generator=g) < 0.7)[None, None].expand_as(v).clone() for k, v in sc["grids"].items()}
    keys = ("grid_middle", "grid_fine", "grid_color")
    n_pix, iters = 300, 6
    idx = [torch.randint(H * W, (n_pix,), generator=g) for _ in range(iters)]
    stages = ["middle", "middle", "fine", "fine", "color", "color"]
    lr = {"middle": {"grid_middle": 0.1}, "fine": {"grid_middle": 0.005, "grid_fine": 0.005},
          "color": {"grid_middle": 0.005, "grid_fine": 0.005, "grid_color": 0.005, "dec": 0.005}}
    col_names = [k for k in sc["params"] if k.startswith("color_decoder.")]

    # ---- reference trajectory on the CPU oracle
    c = {k: v.clone() for k, v in sc["grids"].items()}
    P = {k: v.clone().requires_grad_(True) for k, v in sc["params"].items()}
    leaves = {k: c[k][masks[k]].clone().requires_grad_(True) for k in keys}
    opt = torch.optim.Adam([{"params": [P[k] for k in col_names], "lr": 0.0}] + [{"params": [leaves[k]], "lr": 0.0} for k in keys])
    traj = []
    for it in range(iters):
        stage = stages[it]
        state = {k: v.detach().clone() for k, v in leaves.items()}
        state.update({k: P[k].detach().clone() for k in col_names})
        for k in keys:                                         # Mapper.py:394-401
            val = c[k]; val[masks[k]] = leaves[k]; c[k] = val
        opt.param_groups[0]["lr"] = lr[stage].get("dec", 0.0)
        for gi, k in enumerate(keys):
            opt.param_groups[gi + 1]["lr"] = lr[stage].get(k, 0.0)
        opt.zero_grad()
        o, d, gd, gc = orc.pixel_rays(idx[it], 0, H, 0, W, fx, fy, cx, cy, sc["c2w"], sc["depth_img"], sc["color_img"])
        depth, unc, col = orc.render_batch_ray(c, P, d, o, stage, gd, sc["bound"])
        m = gd > 0
        loss = torch.abs(gd[m] - depth[m]).sum()                # Mapper.py:487-493
        if stage == "color":
            loss = loss + 0.2 * torch.abs(gc - col).sum()
        loss.backward()
        grads = {k: v.grad.detach().clone() for k, v in leaves.items() if v.grad is not None}
        if stage == "color":
            grads["color_blob"] = torch.cat([P[k].grad.reshape(-1) for k in col_names])
        traj.append((state, float(loss.detach()), grads))
        opt.step()
        for k in keys:                                         # Mapper.py:511-519
            val = c[k].detach(); val[masks[k]] = leaves[k].clone().detach(); c[k] = val
    assert float((traj[-1][0]["grid_middle"] - traj[0][0]["grid_middle"]).abs().max()) > 0.05       # the trajectory moves

    # ---- the HIP path on the same trajectory
    cd = {k: v.detach().clone(memory_format=torch.preserve_format) for k, v in grids_dev.items()}
    for p in dec.parameters():
        p.requires_grad_(True)
    col_params = dict(dec.color_decoder.named_parameters())
    c2w, depth_img, color_img = sc["c2w"].to(DEV), sc["depth_img"].to(DEV), sc["color_img"].to(DEV)
    mdev = {k: masks[k].to(DEV) for k in keys}
    n_cmp = 0
    for it in range(iters):
        stage = stages[it]
        state, ref_loss, ref_grads = traj[it]
        lv = {k: state[k].to(DEV).requires_grad_(True) for k in keys}
        with torch.no_grad():
            for k in col_names:
                col_params[k[len("color_decoder."):]].copy_(state[k])
        for p in dec.parameters():
            p.grad = None
        for k in keys:
            val = cd[k]; val[mdev[k]] = lv[k]; cd[k] = val
        o, d, gd, gc = nsa.common.samples_from_indices(idx[it].to(DEV), 0, H, 0, W, fx, fy, cx, cy, c2w, depth_img, color_img)
        depth, unc, col = renderer.render_batch_ray(cd, dec, d, o, DEV, stage, gt_depth=gd)
        m = gd > 0
        loss = torch.abs(gd[m] - depth[m]).sum()
        if stage == "color":
            loss = loss + 0.2 * torch.abs(gc - col).sum()
        loss.backward()
        assert abs(float(loss.detach()) - ref_loss) < 1e-5 * abs(ref_loss), (it, float(loss.detach()), ref_loss)
        for k, gr in ref_grads.items():
            got = torch.cat([col_params[n[len("color_decoder."):]].grad.reshape(-1) for n in col_names]) if k == "color_blob" else lv[k].grad
            assert rel_err(got, gr) < 1e-4, (it, stage, k, rel_err(got, gr))
            n_cmp += 1
        for k in keys:
            val = cd[k].detach(); val[mdev[k]] = lv[k].detach().clone(); cd[k] = val
    assert n_cmp == 2 * 1 + 2 * 2 + 2 * 4


def test_tracking_style_pose_gradient():
    import nice_slam_amd as nsa
    sc = make_scene(seed=32, n_rays=8, small=True)
    H, W, fx, fy, cx, cy = sc["intr"]
    renderer, dec, grids = build_product(sc, DEV)
    for p in dec.parameters():
        p.requires_grad_(False)
    g = torch.Generator().manual_seed(4)
    idx = torch.randint((H - 8) * (W - 12), (200,), generator=g)
    cam0 = torch.tensor([0.98, 0.02, 0.15, -0.03, *sc["c2w"][:3, 3].tolist()], dtype=torch.float32)

    def loss_fn(depth, unc, col, gd, gc):                       # Tracker.py:110-123
        unc = unc.detach()
        tmp = torch.abs(gd - depth) / torch.sqrt(unc + 1e-10)
        mask = (tmp < 10 * tmp.median()) & (gd > 0)
        return (torch.abs(gd - depth) / torch.sqrt(unc + 1e-10))[mask].sum() + 0.5 * torch.abs(gc - col)[mask].sum()

    cam = cam0.clone().to(DEV).requires_grad_(True)
    o, d, gd, gc = nsa.common.samples_from_indices(idx.to(DEV), 4, H - 4, 6, W - 6, fx, fy, cx, cy, cam_to_c2w(cam),
                                                   sc["depth_img"].to(DEV), sc["color_img"].to(DEV))
    depth, unc, col = renderer.render_batch_ray(grids, dec, d, o, DEV, "color", gt_depth=gd)
    l_hip = loss_fn(depth, unc, col, gd, gc)
    l_hip.backward()

    cam_r = cam0.clone().requires_grad_(True)
    o, d, gd, gc = orc.pixel_rays(idx, 4, H - 4, 6, W - 6, fx, fy, cx, cy, cam_to_c2w(cam_r), sc["depth_img"], sc["color_img"])
    depth, unc, col = orc.render_batch_ray(sc["grids"], sc["params"], d, o, "color", gd, sc["bound"])
    l_ref = loss_fn(depth, unc, col, gd, gc)
    l_ref.backward()
    assert abs(float(l_hip) - float(l_ref)) / abs(float(l_ref)) < 2e-4
    assert rel_err(cam.grad, cam_r.grad) < 2e-3, (cam.grad, cam_r.grad)
    assert all(p.grad is None for p in dec.parameters())       # tracking: no decoder gradients were produced


def test_ncdhw_grids_are_accepted():
    """a caller that keeps the reference's own grid_init (NCDHW) still gets correct results and gradients"""
    sc = make_scene(seed=33, n_rays=64, small=True)
    renderer, dec, _ = build_product(sc, DEV)
    from scene_util import oracle_render
    grids = {k: v.to(DEV).contiguous().requires_grad_(True) for k, v in sc["grids"].items()}      # plain NCDHW leaves
    for p in dec.parameters():
        p.requires_grad_(False)
    o, d, gd = sc["rays_o"].to(DEV), sc["rays_d"].to(DEV), sc["gt_depth"].to(DEV)
    depth, unc, col = renderer.render_batch_ray(grids, dec, d, o, DEV, "color", gt_depth=gd)
    w = sc["w"]
    ((depth * w["depth"].to(DEV)).sum() + (unc * w["var"].to(DEV)).sum() + (col * w["rgb"].to(DEV)).sum()).backward()
    ref = oracle_render(sc, "color", backward=True)
    assert rel_err(depth, ref["depth"]) < 1e-4 and rel_err(col, ref["rgb"]) < 1e-4
    for k in ("grid_middle", "grid_fine", "grid_color"):
        assert grids[k].grad.shape == grids[k].shape
        assert rel_err(grids[k].grad, ref["d_" + k]) < 1e-4, k


def test_masked_grid_adam_replaces_masked_leaf_flow():
    """§8(f) rank 1 on the GPU: nice_slam_amd.MaskedGridAdam (one in-place kernel per grid) against the reference's
    masked-leaf + torch.optim.Adam + write-back flow, driven by real render gradients for a few mapping iterations."""
    import nice_slam_amd as nsa
    sc = make_scene(seed=41, n_rays=300, small=True)
    renderer, dec, grids_dev = build_product(sc, DEV)
    for p in dec.parameters():
        p.requires_grad_(False)
    g = torch.Generator().manual_seed(6)
    vmasks = {k: (torch.rand(v.shape[2:], generator=g) < 0.7) for k, v in sc["grids"].items()}
    keys = ("grid_middle", "grid_fine", "grid_color")
    stages = ["middle", "fine", "color", "color"]
    lr = {"middle": {"grid_middle": 0.1}, "fine": {"grid_middle": 0.005, "grid_fine": 0.005},
          "color": {"grid_middle": 0.005, "grid_fine": 0.005, "grid_color": 0.005}}
    o, d, gd, gc = (sc[k].to(DEV) for k in ("rays_o", "rays_d", "gt_depth", "gt_color"))

    def loss_of(c, stage):
        depth, unc, col = renderer.render_batch_ray(c, dec, d, o, DEV, stage, gt_depth=gd)
        loss = (torch.abs(gd - depth) * (gd > 0)).sum()
        return loss + 0.2 * torch.abs(gc - col).sum() if stage == "color" else loss

    # Both flows are stepped with the SAME gradient tensor (rendered once per iteration from flow B's grids, which stay
    # equal to flow A's), so the comparison isolates the optimiser arithmetic and can be held in max-norm.
    # (A) reference flow: masked 1-D leaves + torch Adam + index_put write-back (Mapper.py:303-333,394-401,504,511-519)
    cA = {k: v.detach().clone(memory_format=torch.preserve_format) for k, v in grids_dev.items()}
    full = {k: vmasks[k][None, None].expand_as(cA[k]).to(DEV) for k in keys}
    leaves = {k: cA[k][full[k]].clone().requires_grad_(True) for k in keys}
    opt = torch.optim.Adam([{"params": [leaves[k]], "lr": 0.0} for k in keys])
    # (B) fused: dense grids are the parameters
    cB = {k: v.detach().clone(memory_format=torch.preserve_format).requires_grad_(k in keys) for k, v in grids_dev.items()}
    fused = nsa.MaskedGridAdam({k: cB[k] for k in keys}, {k: vmasks[k] for k in keys})
    for stage in stages:
        for k in keys:
            cB[k].grad = None
        loss_of(cB, stage).backward()
        grads = {k: cB[k].grad for k in keys if cB[k].grad is not None}
        for k in keys:
            val = cA[k]; val[full[k]] = leaves[k].detach(); cA[k] = val      # Mapper.py:394-401
        for gi, k in enumerate(keys):
            opt.param_groups[gi]["lr"] = lr[stage].get(k, 0.0)
            leaves[k].grad = grads[k][full[k]].clone() if k in grads else None
        opt.step()
        for k in keys:
            val = cA[k].detach(); val[full[k]] = leaves[k].detach().clone(); cA[k] = val     # Mapper.py:511-519
        with torch.no_grad():
            fused.step({k: lr[stage].get(k, 0.0) for k in keys})
    for k in keys:
        a, b = cA[k].detach().cpu(), cB[k].detach().cpu()
        assert torch.equal(b[~full[k].cpu()], sc["grids"][k][~full[k].cpu()]), k          # unmasked voxels untouched
        # max-norm; torch's (foreach) Adam on the GPU and the fused kernel round a few operations differently: after four
        # steps of up to lr = 0.1 the largest observed difference is 5 ulp of the largest parameter
        assert float((a - b).abs().max()) <= 1e-5 * float(a.abs().max()), (k, float((a - b).abs().max()), float(a.abs().max()))
        assert float((b - sc["grids"][k]).abs().max()) > 1e-4, k


def test_aabb_mask_flow_equals_compaction():
    """Mapper.py:471-489 two ways through the HIP renderer: the reference's boolean-mask compaction of the ray batch, and
    the sync-free form (nice_slam_amd.aabb_keep: full batch, kept-ray max depth, loss weighted by the mask).  Same loss,
    same grid / decoder gradients."""
    import nice_slam_amd as nsa
    sc = make_scene(seed=51, n_rays=400, small=True)
    renderer, dec, grids_dev = build_product(sc, DEV)
    o = sc["rays_o"].to(DEV) * 1.0
    d = sc["rays_d"].to(DEV)
    gd = sc["gt_depth"].to(DEV)                              # ~43 % of these rays end inside the bound
    gc = sc["gt_color"].to(DEV)
    bound = sc["bound"]

    def run(masked):
        c = {k: v.detach().clone(memory_format=torch.preserve_format).requires_grad_(True) for k, v in grids_dev.items()}
        for p in dec.parameters():
            p.grad = None; p.requires_grad_(True)
        if masked:
            keep, kmax = nsa.aabb_keep(o, d, gd, bound)
            depth, _, color = renderer.render_batch_ray(c, dec, d, o, DEV, "color", gt_depth=gd, gt_max=kmax)
            w = keep & (gd > 0)
            loss = (torch.abs(gd - depth) * w).sum() + 0.2 * (torch.abs(gc - color) * keep[:, None]).sum()
        else:
            t = (bound.unsqueeze(0).to(DEV) - o.unsqueeze(-1)) / d.unsqueeze(-1)
            t, _ = torch.min(torch.max(t, dim=2)[0], dim=1)
            keep = t >= gd
            o2, d2, gd2, gc2 = o[keep], d[keep], gd[keep], gc[keep]
            depth, _, color = renderer.render_batch_ray(c, dec, d2, o2, DEV, "color", gt_depth=gd2)
            m = gd2 > 0
            loss = torch.abs(gd2[m] - depth[m]).sum() + 0.2 * torch.abs(gc2 - color).sum()
        loss.backward()
        return (float(loss), keep.clone(), {k: v.grad.clone() for k, v in c.items() if v.grad is not None},
                {k: p.grad.clone() for k, p in dec.named_parameters() if p.grad is not None})

    l0, k0, g0, p0 = run(False)
    l1, k1, g1, p1 = run(True)
    assert torch.equal(k0, k1) and 0.1 < k0.float().mean() < 0.95
    assert abs(l0 - l1) <= 1e-6 * abs(l0)
    for k in g0:
        assert rel_err(g1[k], g0[k]) < 1e-5, k
    for k in p0:
        assert rel_err(p1[k], p0[k]) < 2e-5, k


def test_captured_iteration_replays_like_eager():
    """nice_slam_amd.graphs.CapturedStep: a mapping-style iteration (get_samples -> aabb_keep -> render -> masked loss ->
    backward) captured once and replayed with new inputs written into its static tensors equals the eager iteration."""
    import nice_slam_amd as nsa
    sc = make_scene(seed=61, n_rays=8, small=True)
    H, W, fx, fy, cx, cy = sc["intr"]
    renderer, dec, grids_dev = build_product(sc, DEV)
    c = {k: v.detach().clone(memory_format=torch.preserve_format).requires_grad_(True) for k, v in grids_dev.items()}
    for p in dec.parameters():
        p.requires_grad_(True)
    depth_img, color_img, c2w = sc["depth_img"].to(DEV), sc["color_img"].to(DEV), sc["c2w"].to(DEV)
    g = torch.Generator().manual_seed(4)
    idx_batches = [torch.randint(H * W, (256,), generator=g).to(DEV) for _ in range(3)]
    idx = idx_batches[0].clone()                                   # the static input of the captured iteration

    def iteration():
        for t in c.values():
            t.grad = None
        for p in dec.parameters():
            p.grad = None
        o, d, gd, gc = nsa.common.samples_from_indices(idx, 0, H, 0, W, fx, fy, cx, cy, c2w, depth_img, color_img)
        keep, kmax = nsa.aabb_keep(o, d, gd, sc["bound"])
        depth, _, color = renderer.render_batch_ray(c, dec, d, o, DEV, "color", gt_depth=gd, gt_max=kmax)
        loss = (torch.abs(gd - depth) * (keep & (gd > 0))).sum() + 0.2 * (torch.abs(gc - color) * keep[:, None]).sum()
        loss.backward()
        return loss

    def snapshot(loss):
        return (loss.detach().clone(), {k: v.grad.clone() for k, v in c.items() if v.grad is not None},
                {k: p.grad.clone() for k, p in dec.named_parameters() if p.grad is not None})

    eager = []
    for b in idx_batches:
        idx.copy_(b)
        eager.append(snapshot(iteration()))
    step = nsa.graphs.CapturedStep(iteration)
    for b, (l0, g0, p0) in zip(idx_batches, eager):
        idx.copy_(b)
        l1, g1, p1 = snapshot(step())
        assert abs(float(l1) - float(l0)) <= 1e-6 * abs(float(l0))
        assert set(g1) == set(g0) and set(p1) == set(p0)
        for k in g0:
            assert rel_err(g1[k], g0[k]) < 1e-5, k
        for k in p0:
            assert rel_err(p1[k], p0[k]) < 2e-5, k


def test_flat_adam_steps_decoder_and_poses_like_torch_adam():
    """nice_slam_amd.FlatAdam (nsr_flat_adam: one launch pair for the decoder blob + the pose tensors, step counts on the device)
    against torch.optim.Adam on copies (Mapper.py:368-387,504; Tracker.py:214-222,127): six colour-stage iterations with real
    render gradients, eager and replayed from a hipGraph; the decoder's packed operand streams follow the stepped parameters
    (mark_dirty -> re-pack into the same buffer), so the NEXT render agrees too."""
    import copy
    import nice_slam_amd as nsa
    sc = make_scene(seed=12, n_rays=256, small=True)
    renderer, decA, grids = build_product(sc, DEV)
    decB = copy.deepcopy(decA)
    H, W, fx, fy, cx, cy = sc["intr"]
    g = torch.Generator().manual_seed(4)
    camA = torch.tensor([[1.0, 0.02, -0.01, 0.03, 0.1, -0.2, 0.05], [0.98, -0.03, 0.02, 0.01, -0.1, 0.1, 0.0]], device=DEV).requires_grad_(True)
    camB = camA.detach().clone().requires_grad_(True)
    rays_o = sc["rays_o"].to(DEV); rays_d = sc["rays_d"].to(DEV); gt = sc["gt_depth"].to(DEV)
    tgt = torch.rand(rays_o.shape[0], 3, generator=g).to(DEV)

    def loss_of(dec, cam):
        # the poses enter through the ray origins (a translation of the camera centre): gradients for both entries
        o = rays_o + nsa.get_camera_from_tensor(cam)[0, :3, 3] * 1e-2 + nsa.get_camera_from_tensor(cam)[1, :3, 3] * 1e-2
        d, u, c = renderer.render_batch_ray(grids, dec, rays_d, o, DEV, "color", gt_depth=gt)
        return (d - gt).abs().sum() + 0.2 * (c - tgt).abs().sum()

    lrs = [5e-3, 1e-3]
    flat = nsa.FlatAdam([decA.color_decoder, camA], lr=lrs)
    ref = torch.optim.Adam([{"params": list(decB.color_decoder.parameters()), "lr": lrs[0]}, {"params": [camB], "lr": lrs[1]}])

    def iter_flat():
        flat.zero_grad(set_to_none=True)
        loss = loss_of(decA, camA)
        loss.backward()
        flat.step()
        return loss

    def iter_ref():
        ref.zero_grad(set_to_none=True)
        lb = loss_of(decB, camB)
        lb.backward()
        ref.step()
        return float(lb)

    for it in range(2):                                  # eager
        la, lb = float(iter_flat()), iter_ref()
        assert abs(la - lb) < 2e-4 * abs(lb), it
    # torch's capture recipe (warm-up on a side stream: an AccumulateGrad node of the pose tensor that survives from an eager
    # iteration on the default stream would pull that stream into the capture); the warm-up EXECUTES one iteration
    step = nsa.graphs.CapturedStep(iter_flat, warmup=1)
    iter_ref()
    for it in range(3):                                  # recording executed nothing: three replays
        la, lb = float(step()), iter_ref()
        assert abs(la - lb) < 5e-4 * abs(lb), it
    assert flat._steps.tolist() == [6, 6]
    fa, fb = decA.color_decoder.flat_params(), torch.cat([p.detach().reshape(-1) for p in decB.color_decoder.parameters()])
    # Adam turns rounding-level gradient differences of near-zero components into lr-sized steps: the bulk must agree, and every
    # element to a few steps' worth
    diff = (fa - fb).abs()
    assert float(diff.max()) < 6 * lrs[0] and float((diff > 1e-4).float().mean()) < 0.02
    assert float((camA - camB).abs().max()) < 6 * lrs[1]
    with torch.no_grad():                                # the re-pack follows the kernel-written blob
        decB.color_decoder.load_state_dict(decA.color_decoder.state_dict())
        la, lb = loss_of(decA, camA), loss_of(decB, camA)
    assert abs(float(la) - float(lb)) < 1e-5 * abs(float(lb))
    # exactness on identical gradients: both optimisers fed the same tensors
    x = torch.randn(1001, device=DEV).requires_grad_(True); y = x.detach().clone().requires_grad_(True)
    fo, to = nsa.FlatAdam([x], lr=3e-3), torch.optim.Adam([y], lr=3e-3)
    for it in range(4):
        gr = torch.randn(1001, device=DEV) * 10.0 ** (it - 2)
        x.grad, y.grad = gr.clone(), gr.clone()
        fo.step(zero_grad=(it == 3)); to.step()
    assert rel_err(x.detach().cpu().numpy(), y.detach().cpu().numpy()) < 2e-6
    assert rel_err(fo.state[0]["exp_avg_sq"].cpu().numpy(), to.state[y]["exp_avg_sq"].cpu().numpy()) < 2e-6
    assert not x.grad.any()
    fo.reset_state()
    assert int(fo._steps[0]) == 0 and not fo.state[0]["exp_avg"].any()
    from nice_slam_amd._capi import NsrError
    # loud failures: wrong number of learning rates, a parameter whose storage was replaced, CPU tensors, more than four entries
    with pytest.raises(NsrError):
        fo.step(lr=[1e-3, 1e-3])
    x.data = x.data.clone()
    x.grad = torch.ones_like(x)
    with pytest.raises(NsrError):
        fo.step()
    with pytest.raises(NsrError):
        nsa.FlatAdam([torch.zeros(3, requires_grad=True)])
    with pytest.raises(NsrError):
        nsa.FlatAdam([torch.zeros(3, device=DEV) for _ in range(5)])


@pytest.mark.parametrize("n,dyn,col", [(1, True, True), (2, True, False), (200, True, True), (255, True, False), (256, True, True), (300, True, True),
                                       (1024, True, True), (1500, True, True), (5000, True, True), (300, False, True)])
def test_tracking_loss_kernel_against_the_reference_expression(n, dyn, col):
    """nsr_tracking_loss on the device (through the C ABI) vs Tracker.optimize_cam_in_batch's loss on the COMPACTED batch
    (src/Tracker.py:92-124) and autograd's d loss / d depth, d loss / d rgb, over the kernel's three median paths: one ray per thread
    (n <= block size), radix select on cached keys (<= 4096), recomputed keys (above)."""
    import ctypes as C
    from nice_slam_amd import _capi
    lib = _capi.get_lib()
    g = torch.Generator().manual_seed(100 + n)
    gd = (torch.rand((n,), generator=g) * 4 + 0.5).float()
    gd[torch.rand((n,), generator=g) < 0.1] = 0.0
    depth = gd.double() + torch.randn((n,), generator=g).double() * 0.3
    depth[torch.rand((n,), generator=g) < 0.05] *= 3.0
    if n > 10:
        depth[5] = gd[5].double()
    var = torch.rand((n,), generator=g).double() * 0.2
    rgb, gc = torch.rand((n, 3), generator=g), torch.rand((n, 3), generator=g)
    keep = torch.rand((n,), generator=g) < 0.85
    if n <= 2:
        keep[:] = True
    w = 0.5
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
    dev = torch.device(DEV)
    t = [x.to(dev).contiguous() for x in (gd, gc, keep.to(torch.uint8), depth, var, rgb)]
    loss = torch.zeros(1, dtype=torch.float64, device=dev)
    dld = torch.full((n,), float("nan"), dtype=torch.float64, device=dev)
    dlr = torch.full((n, 3), float("nan"), dtype=torch.float32, device=dev)
    with _capi.on_device(dev):
        lib.check(lib.nsr_tracking_loss(n, t[0].data_ptr(), t[1].data_ptr(), t[2].data_ptr(), t[3].data_ptr(), t[4].data_ptr(), t[5].data_ptr(),
                                        int(dyn), int(col), w, loss.data_ptr(), dld.data_ptr(), dlr.data_ptr() if col else None,
                                        C.c_void_p(torch.cuda.current_stream(dev).cuda_stream)), "nsr_tracking_loss")
    torch.cuda.synchronize()
    # (the reference's colour term is an fp32 sum; the kernel accumulates both terms in fp64)
    assert abs(float(loss) - float(ref.detach())) <= (1e-6 if col else 1e-9) * abs(float(ref.detach())) + 1e-12
    got_d = dld.cpu()
    assert torch.equal(got_d == 0, want_d == 0) and torch.allclose(got_d, want_d, rtol=1e-14, atol=0)
    if col:
        assert torch.equal(dlr.cpu(), want_r)
