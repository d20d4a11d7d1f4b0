"""GPU: the fused mapping iteration (nice_slam_amd/mapping.py, optim.MaskedGridAdam(capturable=True)) against the unfused
drop-in path it replaces -- which tests/test_hip_parity.py / test_hip_real_callers.py hold to the oracle and to the real
Mapper.  Window sampling is bit-exact; the fused loss + backward agree with render_batch_ray + torch loss + autograd to the
noise of the gradient atomics."""
import numpy as np
import pytest
import torch

from scene_util import build_product, make_scene, rel_err

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def _frames(sc, K, dev, grad=False):
    g = torch.Generator().manual_seed(17)
    H, W = sc["intr"][:2]
    out = []
    for k in range(K):
        c2w = sc["c2w"].clone()
        c2w[:3, 3] += 0.02 * k
        depth = (sc["depth_img"] * (1.0 + 0.03 * k)).to(dev)
        color = torch.rand((H, W, 3), generator=g).to(dev)
        c2w = c2w[:3].contiguous().to(dev) if k % 2 else c2w.to(dev)        # 3x4 and 4x4 poses mixed, like the reference
        out.append((c2w.requires_grad_(grad), depth, color))
    return out


def test_window_sampling_is_the_per_frame_loop():
    import nice_slam_amd as nsa
    from nice_slam_amd.common import samples_from_indices
    sc = make_scene(seed=81, n_rays=8, small=True)
    H, W, fx, fy, cx, cy = sc["intr"]
    K, n = 3, 77
    frames = _frames(sc, K, DEV, grad=True)
    idx = torch.randint((H - 8) * (W - 10), (K * n,), generator=torch.Generator().manual_seed(3))
    w = nsa.get_samples_window(4, H - 4, 5, W - 5, n, H, W, fx, fy, cx, cy, [f[0] for f in frames], [f[1] for f in frames],
                               [f[2] for f in frames], sc["bound"], DEV, indices=idx)
    parts = [samples_from_indices(idx[k * n:(k + 1) * n].to(DEV), 4, H - 4, 5, W - 5, fx, fy, cx, cy, *frames[k]) for k in range(K)]
    ro, rd, gd, gc = (torch.cat([p[i] for p in parts]) for i in range(4))
    for a, b in ((w.rays_o, ro), (w.rays_d, rd), (w.gt_depth, gd), (w.gt_color, gc)):
        assert torch.equal(a.detach(), b.detach())
    keep, kmax = nsa.aabb_keep(ro.detach(), rd.detach(), gd, sc["bound"])
    assert torch.equal(w.keep, keep) and torch.equal(w.kept_max, kmax) and 0.05 < keep.float().mean() < 0.98
    # pose gradients (local BA): one kernel vs autograd through the per-frame path
    gen = torch.Generator().manual_seed(4)
    wo, wd = torch.randn((K * n, 3), generator=gen).to(DEV), torch.randn((K * n, 3), generator=gen).to(DEV)
    ((w.rays_o * wo).sum() + (w.rays_d * wd).sum()).backward()
    got = [f[0].grad.clone() for f in frames]
    for f in frames:
        f[0].grad = None
    ((ro * wo).sum() + (rd * wd).sum()).backward()
    for k in range(K):
        assert got[k].shape == frames[k][0].shape
        assert rel_err(got[k], frames[k][0].grad) < 1e-5, k


def test_window_kernel_zero_fills_the_iterations_buffer():
    """nsr_get_samples_window_fused (ABI 7) on the GPU: the rays / mask / kept maximum of the plain window entry point, the span
    zero-filled to the last float whatever was there (48 MB-sized and tiny, lengths that are not multiples of four), the header
    {loss = 0, kept max, 0} written by the launch itself, the hand-off words of the state zero again after every call, the call
    counter advanced only when the kernel drew -- and the same under back-to-back launches on one stream."""
    import ctypes as C
    import nice_slam_amd as nsa
    from nice_slam_amd import _capi, mapping
    from nice_slam_amd.common import _stream
    lib = _capi.get_lib()
    sc = make_scene(seed=83, n_rays=8, small=True)
    H, W, fx, fy, cx, cy = sc["intr"]
    K, n = 3, 1500
    N = K * n
    frames = _frames(sc, K, DEV)
    dev = torch.device(DEV)
    fr, hold = mapping._frames_block([f[0] for f in frames], [f[1] for f in frames], [f[2] for f in frames], dev)
    lo, hi = mapping._bound_arrays(sc["bound"])
    crop = (4, H - 4, 5, W - 5)
    given = torch.randint((H - 8) * (W - 10), (N,), generator=torch.Generator().manual_seed(5)).to(DEV)
    ref = nsa.get_samples_window(*crop, n, H, W, fx, fy, cx, cy, [f[0] for f in frames], [f[1] for f in frames], [f[2] for f in frames],
                                 sc["bound"], DEV, indices=given)
    state = torch.tensor([99, 4, 0, 0], dtype=torch.int64, device=DEV)
    for zero_n, draw in ((0, False), (7, False), (12 * 1024 * 1024 + 3, False), (70001, True), (12 * 1024 * 1024 + 3, True)):
        calls = int(state[1])
        Z = torch.full((4 + zero_n + 8,), float("nan"), dtype=torch.float32, device=DEV)
        Z[-8:] = 7.0
        sbuf = torch.full((10 * N + (N + 3) // 4,), float("nan"), dtype=torch.float32, device=DEV)
        keep = sbuf[10 * N:].view(torch.uint8)[:N]
        ind = torch.full((N,), -1, dtype=torch.int64, device=DEV) if draw else given.clone()
        for rep in range(2 if not draw else 1):                # twice in a row: the second launch finds the state as the first left it
            lib.check(lib.nsr_get_samples_window_fused(None if draw else ind.data_ptr(), ind.data_ptr() if draw else None, state.data_ptr(), K, n,
                                                       crop[0], crop[1], crop[2], crop[3], W, fx, fy, cx, cy, fr, sbuf.data_ptr(), sbuf.data_ptr() + 12 * N,
                                                       sbuf.data_ptr() + 24 * N, sbuf.data_ptr() + 28 * N, lo, hi, keep.data_ptr(), Z.data_ptr(),
                                                       Z.data_ptr() + 16 if zero_n else None, zero_n, _stream(dev)), "fused")
        torch.cuda.synchronize()
        assert state.tolist() == [99, calls + (1 if draw else 0), 0, 0]
        assert torch.all(Z[4:4 + zero_n] == 0) and torch.all(Z[-8:] == 7.0) and Z[0] == 0 and Z[1] == 0 and Z[3] == 0
        if draw:
            assert int(ind.min()) >= 0 and int(ind.max()) < (H - 8) * (W - 10)
            chk = nsa.get_samples_window(*crop, n, H, W, fx, fy, cx, cy, [f[0] for f in frames], [f[1] for f in frames],
                                         [f[2] for f in frames], sc["bound"], DEV, indices=ind)
        else:
            chk = ref
        assert torch.equal(sbuf[:3 * N].view(N, 3), chk.rays_o) and torch.equal(sbuf[3 * N:6 * N].view(N, 3), chk.rays_d)
        assert torch.equal(sbuf[6 * N:7 * N], chk.gt_depth) and torch.equal(sbuf[7 * N:10 * N].view(N, 3), chk.gt_color)
        assert torch.equal(keep.bool(), chk.keep) and float(Z[2]) == float(chk.kept_max) and float(Z[2]) > 0


def test_window_kernel_draws_its_own_pixels():
    """Default pixel draw of the fused entry points (mapping.PIXEL_DRAW = "kernel"): indices in range and close to uniform, the
    window's rays are those of the explicit-index path on the drawn indices, every call (and every replay of a captured graph)
    draws afresh, seed_pixel_draws restarts the sequence."""
    import nice_slam_amd as nsa
    from nice_slam_amd import mapping
    assert mapping.PIXEL_DRAW == "kernel"
    sc = make_scene(seed=82, n_rays=8, small=True)
    H, W, fx, fy, cx, cy = sc["intr"]
    K, n = 3, 4000
    frames = _frames(sc, K, DEV)
    args = (4, H - 4, 5, W - 5, n, H, W, fx, fy, cx, cy, [f[0] for f in frames], [f[1] for f in frames], [f[2] for f in frames], sc["bound"], DEV)
    nsa.seed_pixel_draws(123, DEV)
    w1 = nsa.get_samples_window(*args)
    w2 = nsa.get_samples_window(*args)
    crop = (H - 8) * (W - 10)
    for w in (w1, w2):
        assert w.indices.dtype == torch.int64 and int(w.indices.min()) >= 0 and int(w.indices.max()) < crop
        hist = torch.bincount((w.indices * 16 // crop), minlength=16).double()
        assert float(((hist - K * n / 16) ** 2 / (K * n / 16)).sum()) < 60.0           # chi-square, 15 degrees of freedom
        ref = nsa.get_samples_window(*args, indices=w.indices)
        for a, b in ((w.rays_o, ref.rays_o), (w.rays_d, ref.rays_d), (w.gt_depth, ref.gt_depth), (w.gt_color, ref.gt_color), (w.keep, ref.keep)):
            assert torch.equal(a, b)
    assert float((w1.indices != w2.indices).double().mean()) > 0.99
    nsa.seed_pixel_draws(123, DEV)
    assert torch.equal(nsa.get_samples_window(*args).indices, w1.indices)
    # under graph capture: the call counter lives on the device, every replay draws other pixels
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    nsa.get_samples_window(*args)
    with torch.cuda.graph(g):
        wg = nsa.get_samples_window(*args)
    seen = []
    for _ in range(3):
        g.replay()
        seen.append(wg.indices.clone())
    assert float((seen[0] != seen[1]).double().mean()) > 0.99 and float((seen[1] != seen[2]).double().mean()) > 0.99
    # Re-seeding AFTER the capture: the state tensor is updated in place (the graph has its address baked in), so the replays
    # restart the sequence -- and nothing the allocator may have placed at a freed address is written to.  Both ways of
    # re-seeding: seed_pixel_draws and a new torch.manual_seed value.
    st = mapping._draw_state(torch.device(DEV))
    ptr = st.data_ptr()
    nsa.seed_pixel_draws(123, DEV)
    g.replay()
    assert torch.equal(wg.indices, w1.indices)
    assert mapping._draw_state(torch.device(DEV)).data_ptr() == ptr
    canary = [torch.full((4,), 0x5a5a5a5a, dtype=torch.int64, device=DEV) for _ in range(64)]     # small blocks the allocator reuses
    old_seed = torch.initial_seed()
    torch.manual_seed(old_seed + 1)
    wn = nsa.get_samples_window(*args)                          # eager call: notices the new torch seed, re-seeds in place
    assert mapping._draw_state(torch.device(DEV)).data_ptr() == ptr
    nsa.seed_pixel_draws(int(mapping._DRAW_STATE[("cuda", 0)][1][0]), DEV)
    g.replay()
    assert torch.equal(wg.indices, wn.indices)                  # the captured graph follows the new seed
    assert all(bool((c == 0x5a5a5a5a).all()) for c in canary)
    torch.manual_seed(old_seed)


@pytest.mark.parametrize("stage", ["coarse", "middle", "fine", "color"])
def test_fused_mapping_loss_equals_unfused_path(stage):
    import nice_slam_amd as nsa
    sc = make_scene(seed=82, n_rays=8, small=True)
    H, W, fx, fy, cx, cy = sc["intr"]
    renderer, dec, grids_dev = build_product(sc, DEV)
    K, n = 4, 150
    idx = torch.randint(H * W, (K * n,), generator=torch.Generator().manual_seed(5))

    def run(fused):
        frames = _frames(sc, K, DEV, grad=True)
        c = {k: v.detach().clone(memory_format=torch.preserve_format).requires_grad_(True) for k, v in grids_dev.items()}
        for p in dec.parameters():
            p.requires_grad_(True); p.grad = None
        if fused:
            loss = nsa.mapping_loss(renderer, c, dec, frames, n, stage, w_color=0.2, indices=idx, coarse_mapper=stage == "coarse")
        else:
            w = nsa.get_samples_window(0, H, 0, W, n, H, W, fx, fy, cx, cy, [f[0] for f in frames], [f[1] for f in frames],
                                       [f[2] for f in frames], sc["bound"], DEV, indices=idx)
            depth, _, color = renderer.render_batch_ray(c, dec, w.rays_d, w.rays_o, DEV, stage, gt_max=w.kept_max,
                                                        gt_depth=None if stage == "coarse" else w.gt_depth)      # Mapper.py:484
            loss = (torch.abs(w.gt_depth - depth) * (w.keep & (w.gt_depth > 0))).sum()          # Mapper.py:487-493, mask form
            if stage == "color":
                loss = loss + 0.2 * (torch.abs(w.gt_color - color) * w.keep[:, None]).sum()
        loss.backward()
        return (float(loss.detach()), {k: v.grad.clone() for k, v in c.items() if v.grad is not None},
                {k: p.grad.clone() for k, p in dec.named_parameters() if p.grad is not None}, [f[0].grad.clone() for f in frames])

    l0, g0, p0, c0 = run(False)
    l1, g1, p1, c1 = run(True)
    # (the colour term of the unfused path is an fp32 torch.sum, Mapper.py:491: 1e-7 of summation noise)
    assert abs(l0 - l1) <= (1e-6 if stage == "color" else 1e-9) * abs(l0), (l0, l1)
    assert set(g0) == set(g1) and set(p0) == set(p1) and len(g0) == {"coarse": 1, "middle": 1, "fine": 2, "color": 3}[stage]
    for k in g0:
        assert rel_err(g1[k], g0[k]) < 1e-5, k
    for k in p0:
        assert rel_err(p1[k], p0[k]) < 2e-5, k
    for k in range(K):
        assert rel_err(c1[k], c0[k]) < 1e-4, (k, c1[k], c0[k])


@pytest.mark.parametrize("dyn,col,dec_grads", [(True, True, False), (True, True, True), (False, True, False), (True, False, False)])
def test_fused_tracking_loss_equals_unfused_path(dyn, col, dec_grads):
    """nice_slam_amd.tracking_loss (window kernel + render + nsr_tracking_loss + backward to the pose) against the same
    iteration spelled with get_samples_window / render_batch_ray / the torch loss of Tracker.py:108-124 on the compacted batch."""
    import nice_slam_amd as nsa
    sc = make_scene(seed=85, n_rays=8, small=True)
    H, W, fx, fy, cx, cy = sc["intr"]
    renderer, dec, grids_dev = build_product(sc, DEV)
    n, eh, ew = 300, 3, 5
    idx = torch.randint((H - 2 * eh) * (W - 2 * ew), (n,), generator=torch.Generator().manual_seed(6))
    depth_img = sc["depth_img"].clone()
    depth_img[::7, ::5] *= 3.0                                   # "dynamic objects": depth the map does not explain
    depth_img, color_img = depth_img.to(DEV), torch.rand((H, W, 3), generator=torch.Generator().manual_seed(7)).to(DEV)
    for p in dec.parameters():
        p.requires_grad_(dec_grads)

    def run(fused):
        c2w = sc["c2w"].clone().to(DEV).requires_grad_(True)
        for p in dec.parameters():
            p.grad = None
        out = {}
        if fused:
            loss = nsa.tracking_loss(renderer, grids_dev, dec, c2w, depth_img, color_img, n, eh, ew, w_color=0.5, handle_dynamic=dyn,
                                     use_color=col, indices=idx, out=out)
        else:
            w = nsa.get_samples_window(eh, H - eh, ew, W - ew, n, H, W, fx, fy, cx, cy, [c2w], [depth_img], [color_img], sc["bound"], DEV,
                                       indices=idx)
            depth, unc, color = renderer.render_batch_ray(grids_dev, dec, w.rays_d, w.rays_o, DEV, "color", gt_depth=w.gt_depth,
                                                          gt_max=w.kept_max)
            k = w.keep
            gd, dep, unc, colr, gc = w.gt_depth[k], depth[k], unc[k].detach(), color[k], w.gt_color[k]
            tmp = torch.abs(gd - dep) / torch.sqrt(unc + 1e-10)
            mask = ((tmp < 10 * tmp.median()) & (gd > 0)) if dyn else (gd > 0)
            loss = tmp[mask].sum()
            if col:
                loss = loss + 0.5 * torch.abs(gc - colr)[mask].sum()
            out["n_mask"], out["n_pos"] = int(mask.sum()), int((gd > 0).sum())
        loss.backward()
        return float(loss.detach()), c2w.grad.clone(), {k_: p.grad.clone() for k_, p in dec.named_parameters() if p.grad is not None}, out

    l0, g0, p0, o0 = run(False)
    l1, g1, p1, o1 = run(True)
    # (whether the median test removes rays depends on the random-init map; tests/test_emu_parity.py pins a case where it does)
    assert 0.3 < float(o1["keep"].float().mean()) < 1.0
    assert abs(l0 - l1) <= 1e-6 * abs(l0), (l0, l1)
    assert rel_err(g1, g0) < 1e-5
    assert set(p0) == set(p1) and (len(p0) > 0) == dec_grads
    for k_ in p0:
        assert rel_err(p1[k_], p0[k_]) < 2e-5, k_
    assert all(v.grad is None for v in grids_dev.values())


def test_get_camera_from_tensor_drop_in():
    """nice_slam_amd.get_camera_from_tensor vs src/common.py:137-176 restated with torch ops, (7,) and (B,7), with gradients."""
    import nice_slam_amd as nsa
    g = torch.Generator().manual_seed(11)
    cam = torch.randn((4, 7), generator=g)
    cam[:, 0] += 1.5

    def ref(t):
        qr, qi, qj, qk = t[:, 0], t[:, 1], t[:, 2], t[:, 3]
        two_s = 2.0 / (t[:, :4] * t[:, :4]).sum(-1)
        R = torch.stack([1 - two_s * (qj ** 2 + qk ** 2), two_s * (qi * qj - qk * qr), two_s * (qi * qk + qj * qr),
                         two_s * (qi * qj + qk * qr), 1 - two_s * (qi ** 2 + qk ** 2), two_s * (qj * qk - qi * qr),
                         two_s * (qi * qk - qj * qr), two_s * (qj * qk + qi * qr), 1 - two_s * (qi ** 2 + qj ** 2)], -1).reshape(-1, 3, 3)
        return torch.cat([R, t[:, 4:, None]], 2)

    w = torch.randn((4, 3, 4), generator=g)
    t0 = cam.clone().requires_grad_(True)
    (ref(t0) * w).sum().backward()
    t1 = cam.clone().to(DEV).requires_grad_(True)
    got = nsa.get_camera_from_tensor(t1)
    (got * w.to(DEV)).sum().backward()
    assert got.shape == (4, 3, 4) and rel_err(got, ref(cam)) < 1e-6 and rel_err(t1.grad, t0.grad) < 1e-5
    t2 = cam[2].clone().to(DEV).requires_grad_(True)
    one = nsa.get_camera_from_tensor(t2)
    (one * w[2].to(DEV)).sum().backward()
    assert one.shape == (3, 4) and torch.equal(one, got[2].detach()) and rel_err(t2.grad, t0.grad[2]) < 1e-5


def test_masked_adam_multi_equals_per_grid_steps():
    import nice_slam_amd as nsa
    sc = make_scene(seed=83, n_rays=8, small=True)
    _, _, grids_dev = build_product(sc, DEV)
    keys = ("grid_middle", "grid_fine", "grid_color")
    g = torch.Generator().manual_seed(6)
    masks = {k: (torch.rand(grids_dev[k].shape[2:], generator=g) < 0.6) for k in keys}
    A = {k: grids_dev[k].detach().clone(memory_format=torch.preserve_format) for k in keys}
    B = {k: grids_dev[k].detach().clone(memory_format=torch.preserve_format) for k in keys}
    oa, ob = nsa.MaskedGridAdam(A, masks), nsa.MaskedGridAdam(B, masks, capturable=True)
    lrs = [{"grid_middle": 0.1}, {"grid_middle": 0.005, "grid_fine": 0.005}, {k: 0.005 for k in keys}, {k: 0.005 for k in keys}]
    for it, lr in enumerate(lrs):
        grads = {k: nsa.to_channels_last(torch.randn(A[k].shape, generator=g).to(DEV) * 1e-3) for k in lr}
        gb = {k: v.clone(memory_format=torch.preserve_format) for k, v in grads.items()}
        oa.step(lr, grads=grads)
        ob.step(lr, grads=gb, zero_grad=(it % 2 == 1))
        for k in lr:
            full = masks[k][None, None].expand_as(gb[k]).to(DEV)
            if it % 2 == 1:
                assert float(gb[k][full].abs().max()) == 0.0 and torch.equal(gb[k][~full], grads[k][~full])
            else:
                assert torch.equal(gb[k], grads[k])
    assert ob._dev_steps.tolist() == [4, 3, 2]
    for k in keys:
        # (step size: fp64 on the device from the fp32 lr vs python floats on the host -- a few ulp after four steps)
        assert float((A[k] - B[k]).abs().max()) <= 1e-5 * float(A[k].abs().max()), k
        assert float((A[k] - grids_dev[k]).abs().max()) > 1e-4


def test_captured_fused_iteration_replays_like_eager():
    """A whole colour-stage mapping iteration -- index draw, window sampling, render, fused loss, backward, capturable grid
    Adam (device step counts) and a capturable torch Adam on the colour decoder -- captured once in a hipGraph; replays
    must walk the same trajectory as the eager loop."""
    import nice_slam_amd as nsa
    sc = make_scene(seed=84, n_rays=8, small=True)
    renderer, dec0, grids_dev = build_product(sc, DEV)
    keys = ("grid_middle", "grid_fine", "grid_color")
    K, n, iters = 3, 120, 4
    frames = _frames(sc, K, DEV)
    H, W = sc["intr"][:2]
    idx_all = [torch.randint(H * W, (K * n,), generator=torch.Generator().manual_seed(50 + i)).to(DEV) for i in range(iters)]

    def build():
        import copy
        dec = copy.deepcopy(dec0)
        for p in dec.parameters():
            p.requires_grad_(True)
        c = {k: v.detach().clone(memory_format=torch.preserve_format).requires_grad_(k in keys) for k, v in grids_dev.items()}
        gopt = nsa.MaskedGridAdam({k: c[k] for k in keys}, capturable=True)
        dopt = torch.optim.Adam(list(dec.color_decoder.parameters()), lr=torch.tensor(0.005, device=DEV), capturable=True, foreach=True)
        idx = idx_all[0].clone()
        losses = torch.zeros(1, dtype=torch.float64, device=DEV)

        def it():
            for t in c.values():
                t.grad = None
            dopt.zero_grad(set_to_none=True)
            for p in dec.parameters():
                p.grad = None
            loss = nsa.mapping_loss(renderer, c, dec, frames, n, "color", indices=idx)
            loss.backward()
            dopt.step()
            with torch.no_grad():
                gopt.step({k: 0.005 for k in keys})
            losses.copy_(loss.detach().reshape(1))
        return c, dec, idx, losses, it

    # the captured loop: CapturedStep EXECUTES the iteration once (warm-up) before it records it (recording executes
    # nothing) -- that run is part of the trajectory, so the eager twin takes the same iteration on the same pixels first
    c_g, dec_g, idx_g, loss_g, it_g = build()
    idx_g.copy_(idx_all[0])
    step = nsa.graphs.CapturedStep(it_g, warmup=1)
    c_e, dec_e, idx_e, loss_e, it_e = build()
    idx_e.copy_(idx_all[0]); it_e()
    for i in range(1, iters):
        idx_g.copy_(idx_all[i]); step()
        idx_e.copy_(idx_all[i]); it_e()
        a, b = float(loss_g), float(loss_e)
        assert abs(a - b) <= 1e-6 * abs(b), (i, a, b)
    for k in keys:
        assert rel_err(c_g[k], c_e[k]) < 1e-5, k
        assert float((c_g[k] - grids_dev[k]).abs().max()) > 1e-3, k
    for (n_, p), q in zip(dec_g.color_decoder.named_parameters(), dec_e.color_decoder.parameters()):
        assert rel_err(p, q) < 1e-4, n_


def test_captured_tracking_iteration_follows_repacked_decoders():
    """The tracker's iteration (get_camera_from_tensor + tracking_loss + capturable Adam on the 7-vector) recorded once; the
    tracker-side decoder copy is then refreshed like Tracker.update_para_from_mapping (src/Tracker.py:130-142) -- flat copy +
    NICE.repack(), in place -- and the SAME graph must render with the new weights: replay == eager on the new weights."""
    import copy
    import nice_slam_amd as nsa
    sc = make_scene(seed=86, n_rays=8, small=True)
    H, W, fx, fy, cx, cy = sc["intr"]
    renderer, dec_map, grids_dev = build_product(sc, DEV)
    depth_img, color_img = sc["depth_img"].to(DEV), torch.rand((H, W, 3), generator=torch.Generator().manual_seed(9)).to(DEV)
    cam0 = torch.tensor([1.0, 0.02, -0.01, 0.03, 0.0, 0.0, 0.0]) + torch.cat([torch.zeros(4), sc["c2w"][:3, 3]])
    idx = torch.randint((H - 4) * (W - 4), (150,), generator=torch.Generator().manual_seed(10)).to(DEV)
    new_flat = {s: dec_map.sub(s).flat_params().clone() * 1.03 for s in ("coarse", "middle", "fine", "color")}

    def build():
        dec = copy.deepcopy(dec_map)
        for p in dec.parameters():
            p.requires_grad_(False)
        cam = cam0.clone().to(DEV).requires_grad_(True)
        opt = torch.optim.Adam([cam], lr=torch.tensor(1e-3, device=DEV), capturable=True)
        loss_out = torch.zeros(1, dtype=torch.float64, device=DEV)

        def it():
            opt.zero_grad(set_to_none=True)
            loss = nsa.tracking_loss(renderer, grids_dev, dec, nsa.get_camera_from_tensor(cam), depth_img, color_img, 150, 2, 2,
                                     indices=idx)
            loss.backward()
            opt.step()
            loss_out.copy_(loss.detach().reshape(1))
        return dec, cam, loss_out, it

    def refresh(dec):
        with torch.no_grad():
            for s, f in new_flat.items():
                dec.sub(s).flat_params().copy_(f)
        dec.repack()

    dec_g, cam_g, loss_g, it_g = build()
    step = nsa.graphs.CapturedStep(it_g, warmup=1)              # executes one iteration, then records one (recording runs nothing)
    dec_e, cam_e, loss_e, it_e = build()
    it_e()
    step(); it_e()
    assert rel_err(cam_g, cam_e) < 1e-6 and abs(float(loss_g) - float(loss_e)) < 1e-7 * abs(float(loss_e))
    before = float(loss_g)
    refresh(dec_g); refresh(dec_e)
    step(); it_e()
    assert abs(float(loss_g) - float(loss_e)) < 1e-7 * abs(float(loss_e)), (float(loss_g), float(loss_e))
    assert abs(float(loss_g) - before) > 1e-6 * abs(before)     # ... and the new weights really changed the render
    assert rel_err(cam_g, cam_e) < 1e-6


@pytest.mark.parametrize("stage", ["coarse", "middle", "fine", "color"])
def test_fused_mapping_loss_against_the_oracle_at_replica_shape(stage):
    """The timed entry point itself, not a path equal to it: ``mapping_loss`` at the Replica room0 shape with 5 x 200 rays
    (BASELINE configs[1]) against the CPU oracle running the reference's iteration -- per-frame ``get_samples``, ``torch.cat``,
    bounding-box compaction, ``render_batch_ray``, the L1 losses, autograd (src/Mapper.py:437-503) -- on the same pixel
    draws: the loss, every grid gradient, every decoder parameter gradient and the pose gradients at 1e-4 of the tensor's
    maximum (a parameter tensor that misses gets the usual second chance against the fp64 evaluation of the same graph)."""
    import nice_slam_amd as nsa
    from oracle import nice_oracle as orc
    sc = make_scene(seed=91, n_rays=8, scene="replica_room0", fine_scale=1.0)
    H, W, fx, fy, cx, cy = sc["intr"]
    renderer, dec, grids_dev = build_product(sc, DEV)
    K, n = 5, 200
    idx = torch.randint(H * W, (K * n,), generator=torch.Generator().manual_seed(7))
    frames = _frames(sc, K, DEV, grad=True)
    c = {k: v.detach().clone(memory_format=torch.preserve_format).requires_grad_(True) for k, v in grids_dev.items()}
    for p in dec.parameters():
        p.requires_grad_(True); p.grad = None
    loss = nsa.mapping_loss(renderer, c, dec, frames, n, stage, w_color=0.2, indices=idx, coarse_mapper=stage == "coarse")
    loss.backward()
    got = {"grid/" + k: v.grad for k, v in c.items() if v.grad is not None}
    got.update({"param/" + k: p.grad for k, p in dec.named_parameters() if p.grad is not None})
    got.update({f"pose/{k}": f[0].grad for k, f in enumerate(frames)})

    def oracle(lo):
        grids = {k: v.detach().clone().to(lo).requires_grad_(True) for k, v in sc["grids"].items()}
        params = {k: v.detach().clone().to(lo).requires_grad_(True) for k, v in sc["params"].items()}
        poses = [f[0].detach().cpu().clone().requires_grad_(True) for f in frames]
        parts = [orc.pixel_rays(idx[k * n:(k + 1) * n], 0, H, 0, W, fx, fy, cx, cy, poses[k], frames[k][1].cpu(), frames[k][2].cpu())
                 for k in range(K)]
        ro, rd, gd, gc = (torch.cat([p[i] for p in parts]) for i in range(4))
        with torch.no_grad():                                   # Mapper.py:471-481
            t = (sc["bound"].unsqueeze(0) - ro.detach().unsqueeze(-1)) / rd.detach().unsqueeze(-1)
            inside = torch.min(torch.max(t, dim=2)[0], dim=1)[0] >= gd
        ro, rd, gd, gc = ro[inside], rd[inside], gd[inside], gc[inside]
        depth, _, color = orc.render_batch_ray(grids, params, rd, ro, stage, None if stage == "coarse" else gd, sc["bound"], lo=lo)
        dm = gd > 0
        ls = torch.abs(gd[dm] - depth[dm]).sum()
        if stage == "color":
            ls = ls + 0.2 * torch.abs(gc - color).sum()
        ls.backward()
        res = {"grid/" + k: v.grad for k, v in grids.items() if v.grad is not None}
        res.update({"param/" + k: v.grad for k, v in params.items() if v.grad is not None})
        res.update({f"pose/{k}": p.grad for k, p in enumerate(poses)})
        return float(ls.detach()), res

    l_ref, ref = oracle(torch.float32)
    assert abs(float(loss.detach()) - l_ref) < 1e-5 * abs(l_ref), (float(loss.detach()), l_ref)
    assert set(ref) <= set(got), sorted(set(ref) - set(got))
    truth = None
    for k, v in ref.items():
        e = rel_err(got[k], v)
        if e < 1e-4:
            continue
        assert k.startswith("param/"), (stage, k, e)            # only cancelling parameter sums may need the fp64 evaluation
        if truth is None:
            truth = oracle(torch.float64)[1]
        e_t, e_r = rel_err(got[k], truth[k]), rel_err(v, truth[k])
        assert e_t <= max(2.0 * e_r, 1e-4), (stage, k, e, e_t, e_r)


def test_fused_losses_are_ordinary_autograd_nodes():
    """(loss * w).backward(), loss / n, a loss that is one term of a larger sum: the incoming gradient of the fused node
    multiplies every gradient it produces (device scalar, no sync) -- the reference's loss is an ordinary autograd scalar."""
    import nice_slam_amd as nsa
    sc = make_scene(seed=83, n_rays=8, small=True)
    H, W = sc["intr"][:2]
    renderer, dec, grids_dev = build_product(sc, DEV)
    K, n = 3, 120
    idx = torch.randint(H * W, (K * n,), generator=torch.Generator().manual_seed(6))

    def run(w):
        frames = _frames(sc, K, DEV, grad=True)
        c = {k: v.detach().clone(memory_format=torch.preserve_format).requires_grad_(True) for k, v in grids_dev.items()}
        for p in dec.parameters():
            p.requires_grad_(True); p.grad = None
        loss = nsa.mapping_loss(renderer, c, dec, frames, n, "color", w_color=0.2, indices=idx)
        if w == "helper":
            nsa.backward(loss)                       # loss.backward() without autograd's ones_like fill
        else:
            (loss if w is None else (loss * w + 1.0) / 2.0).backward()
        g = {k: v.grad.clone() for k, v in c.items() if v.grad is not None}
        g.update({k: p.grad.clone() for k, p in dec.named_parameters() if p.grad is not None})
        g.update({f"pose{k}": f[0].grad.clone() for k, f in enumerate(frames)})
        assert len(g) >= 3 + 30 + K
        return g

    g1, g3, gh = run(None), run(-3.0), run("helper")
    for k in g1:
        assert rel_err(g3[k], -1.5 * g1[k]) < 2e-5, k
        assert rel_err(gh[k], g1[k]) < 2e-5, k
    c2w = sc["c2w"][:3].clone().to(DEV).requires_grad_(True)
    lt = nsa.tracking_loss(renderer, grids_dev, dec, c2w, sc["depth_img"].to(DEV), sc["color_img"].to(DEV), 150, 4, 4, indices=idx[:150] % ((H - 8) * (W - 8)))
    (0.25 * lt).backward()
    ga = c2w.grad.clone()
    c2w.grad = None
    nsa.tracking_loss(renderer, grids_dev, dec, c2w, sc["depth_img"].to(DEV), sc["color_img"].to(DEV), 150, 4, 4, indices=idx[:150] % ((H - 8) * (W - 8))).backward()
    assert rel_err(ga, 0.25 * c2w.grad) < 2e-5


@pytest.mark.parametrize("stage,fused", [("middle", True), ("color", True), ("color", False)])
def test_consumed_gradient_voxel_masks(stage, fused):
    """Renderer.grad_voxel_masks (nsr_render_args.grad_voxel_mask, ABI 8; opt-in): with `frustum_feature_selection` the mapper's
    optimiser holds only `val[mask]` (src/Mapper.py:315-333,394-401), the rest of the reference's dense grid gradient is thrown
    away.  With the frustum masks handed to the renderer the backward's scatter skips those voxels: gradients inside a mask equal
    the dense run's up to the order of the atomic adds, gradients outside are EXACTLY zero, decoder and pose gradients are the
    dense run's.  Through the fused iteration and through render_batch_ray + autograd."""
    import nice_slam_amd as nsa
    sc = make_scene(seed=83, n_rays=8, small=True)
    H, W, fx, fy, cx, cy = sc["intr"]
    renderer, dec, grids_dev = build_product(sc, DEV)
    K, n = 3, 200
    idx = torch.randint(H * W, (K * n,), generator=torch.Generator().manual_seed(6))
    sel = nsa.FrustumSelector(sc["bound"], H, W, fx, fy, cx, cy)
    frames0 = _frames(sc, K, DEV)
    pose = torch.eye(4)
    pose[:3] = frames0[-1][0][:3].detach().cpu()
    masks = {k: sel.voxel_mask(pose, k, v.shape[2:], frames0[-1][1]) for k, v in grids_dev.items() if k != "grid_coarse"}
    assert all(0.02 < float(m.float().mean()) < 0.98 for m in masks.values()), {k: float(m.float().mean()) for k, m in masks.items()}

    def run(use_masks):
        renderer.grad_voxel_masks = masks if use_masks else None
        frames = _frames(sc, K, DEV, grad=True)
        c = {k: v.detach().clone(memory_format=torch.preserve_format).requires_grad_(True) for k, v in grids_dev.items()}
        for p in dec.parameters():
            p.requires_grad_(True); p.grad = None
        try:
            if fused:
                loss = nsa.mapping_loss(renderer, c, dec, frames, n, stage, w_color=0.2, indices=idx)
            else:
                w = nsa.get_samples_window(0, H, 0, W, n, H, W, fx, fy, cx, cy, [f[0] for f in frames], [f[1] for f in frames],
                                           [f[2] for f in frames], sc["bound"], DEV, indices=idx)
                depth, _, color = renderer.render_batch_ray(c, dec, w.rays_d, w.rays_o, DEV, stage, gt_max=w.kept_max, gt_depth=w.gt_depth)
                loss = (torch.abs(w.gt_depth - depth) * (w.keep & (w.gt_depth > 0))).sum() + 0.2 * (torch.abs(w.gt_color - color) * w.keep[:, None]).sum()
            loss.backward()
        finally:
            renderer.grad_voxel_masks = None
        return ({k: v.grad.clone() for k, v in c.items() if v.grad is not None},
                {k: p.grad.clone() for k, p in dec.named_parameters() if p.grad is not None}, [f[0].grad.clone() for f in frames])

    g0, p0, c0 = run(False)
    g1, p1, c1 = run(True)
    assert set(g0) == set(g1) and len(g0) == {"middle": 1, "color": 3}[stage]
    for k in g0:
        m = masks[k].bool()[None, None]                                     # [1, 1, Z, Y, X]
        dense, got = g0[k], g1[k]
        assert float(dense.masked_fill(m, 0).abs().max()) > 0, k           # the dense run does scatter outside the mask
        assert float(got.masked_fill(m, 0).abs().max()) == 0.0, k          # the masked run does not: exact zeros
        assert rel_err(got.masked_fill(~m, 0), dense.masked_fill(~m, 0)) < 1e-5, k
    for k in p0:
        assert rel_err(p1[k], p0[k]) < 2e-5, k
    for k in range(K):
        assert rel_err(c1[k], c0[k]) < 1e-5, k
    with pytest.raises(Exception, match="grad_voxel_masks"):
        renderer.grad_voxel_masks = {"grid_middle": masks["grid_middle"].float()}
        try:
            run_c = {k: v.detach().clone(memory_format=torch.preserve_format).requires_grad_(True) for k, v in grids_dev.items()}
            nsa.mapping_loss(renderer, run_c, dec, _frames(sc, K, DEV), n, stage, w_color=0.2, indices=idx).backward()
        finally:
            renderer.grad_voxel_masks = None


@pytest.mark.parametrize("stepped,fused", [(("color",), True), (("color",), False), (("fine", "color"), True)])
def test_stepped_decoders_only(stepped, fused):
    """Renderer.decoder_grads (opt-in): parameter gradients only for the decoders the optimiser steps (src/Mapper.py:335-341: the colour
    decoder, + the fine one when fix_fine is off).  The other decoders of the stage get no `.grad`, the stepped ones' and every grid /
    pose gradient equal the all-decoders run's (up to atomic order) -- and, since ABI 8, the forward saves only the relu masks of a
    decoder nobody differentiates (nsr_render_args.acts_masks_only bits 1..3; the middle pass stays complete when the fine decoder is
    stepped: its dW reads the middle features)."""
    import nice_slam_amd as nsa
    sc = make_scene(seed=86, n_rays=8, small=True)
    H, W, fx, fy, cx, cy = sc["intr"]
    renderer, dec, grids_dev = build_product(sc, DEV)
    K, n = 3, 150
    idx = torch.randint(H * W, (K * n,), generator=torch.Generator().manual_seed(12))

    def run(which):
        renderer.decoder_grads = which
        frames = _frames(sc, K, DEV, grad=True)
        c = {k: v.detach().clone(memory_format=torch.preserve_format).requires_grad_(True) for k, v in grids_dev.items()}
        for p in dec.parameters():
            p.requires_grad_(True); p.grad = None
        try:
            if fused:
                loss = nsa.mapping_loss(renderer, c, dec, frames, n, "color", w_color=0.2, indices=idx)
            else:
                w = nsa.get_samples_window(0, H, 0, W, n, H, W, fx, fy, cx, cy, [f[0] for f in frames], [f[1] for f in frames],
                                           [f[2] for f in frames], sc["bound"], DEV, indices=idx)
                depth, _, color = renderer.render_batch_ray(c, dec, w.rays_d, w.rays_o, DEV, "color", gt_max=w.kept_max, gt_depth=w.gt_depth)
                loss = (torch.abs(w.gt_depth - depth) * (w.keep & (w.gt_depth > 0))).sum() + 0.2 * (torch.abs(w.gt_color - color) * w.keep[:, None]).sum()
            loss.backward()
        finally:
            renderer.decoder_grads = None
        return ({k: v.grad.clone() for k, v in c.items() if v.grad is not None},
                {k: p.grad.clone() for k, p in dec.named_parameters() if p.grad is not None}, [f[0].grad.clone() for f in frames])

    g0, p0, c0 = run(None)
    g1, p1, c1 = run(stepped)
    assert p1 and all(any(k.startswith(s + "_decoder.") for s in stepped) for k in p1), sorted(p1)[:4]
    assert {k for k in p0 if any(k.startswith(s + "_decoder.") for s in stepped)} == set(p1)
    for k in p1:
        assert rel_err(p1[k], p0[k]) < 2e-5, k
    for k in g0:
        assert rel_err(g1[k], g0[k]) < 1e-5, k
    for a, b in zip(c0, c1):
        assert rel_err(b, a) < 1e-5
