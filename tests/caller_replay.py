"""TEST INFRASTRUCTURE: replay of the caller steps minted by tests/golden/make_golden_callers.py (the UNMODIFIED reference
``Mapper.optimize_map`` / ``Tracker.optimize_cam_in_batch`` on the CPU) through an `ops` binding -- the product on a GPU
(``ProductOps``) or the CPU oracle (``OracleOps``) -- in a loop of the SAME SHAPE as the reference's:

* masked 1-D leaves ``val_grad = val[mask]`` written into the dense grids with ``val[mask] = val_grad`` before every
  iteration (src/Mapper.py:315-333,394-401), 7-vector camera tensors -> ``get_camera_from_tensor`` -> ``get_samples`` per
  frame (:437-468), bounding-box pre-filter by boolean-mask compaction (:471-481), ``render_batch_ray``, the mapping loss
  (:487-493), ``loss.backward()``;
* tracking: camera tensor -> crop ``get_samples`` -> pre-filter -> colour-stage render -> uncertainty-weighted loss with the
  median outlier mask (src/Tracker.py:87-125).

The replay is TEACHER-FORCED: before iteration k every optimised tensor is set to the value the reference held at that
point (fixture ``after`` arrays of iteration k-1), so each iteration's loss and gradients can be compared in max-norm against
what the reference's autograd produced for the same state and the same pixel draws (Adam would otherwise turn 1e-6
gradient noise on near-zero components into lr-sized differences).
"""
import os

import numpy as np
import torch

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "caller_steps.npz")
GRID_KEYS = ("grid_coarse", "grid_middle", "grid_fine", "grid_color")


def load():
    z = np.load(GOLDEN)
    return {k: z[k] for k in z.files}


def quad2rotation(quad):                                   # src/common.py:137-160
    qr, qi, qj, qk = quad[0], quad[1], quad[2], quad[3]
    two_s = 2.0 / (quad * quad).sum(-1)
    return torch.stack([
        torch.stack([1 - two_s * (qj ** 2 + qk ** 2), two_s * (qi * qj - qk * qr), two_s * (qi * qk + qj * qr)]),
        torch.stack([two_s * (qi * qj + qk * qr), 1 - two_s * (qi ** 2 + qk ** 2), two_s * (qj * qk - qi * qr)]),
        torch.stack([two_s * (qi * qk - qj * qr), two_s * (qj * qk + qi * qr), 1 - two_s * (qi ** 2 + qj ** 2)])])


def get_camera_from_tensor(cam):                           # src/common.py:163-176
    return torch.cat([quad2rotation(cam[:4]), cam[4:, None]], 1)


def stage_of(it, n, coarse):                               # src/Mapper.py:403-410
    if coarse:
        return "coarse"
    return "middle" if it <= int(n * 0.4) else ("fine" if it <= int(n * 0.6) else "color")


# --------------------------------------------------------------------------------------------------------------------
# ops bindings
# --------------------------------------------------------------------------------------------------------------------
class OracleOps:
    """oracle/nice_oracle.py on the CPU; ``lo=torch.float64`` evaluates decoders + compositor in double (the 'truth')."""

    def __init__(self, gold, lo=torch.float32):
        from oracle import nice_oracle as orc
        self.orc, self.lo, self.dev = orc, lo, torch.device("cpu")
        self.bound = torch.from_numpy(gold["bound"])
        self.intr = tuple(float(v) for v in gold["intr"])
        self.grids0 = {k[5:]: torch.from_numpy(v).to(lo) for k, v in gold.items() if k.startswith("grid/")}
        self.P = {k[6:]: torch.from_numpy(v).to(lo).requires_grad_(True) for k, v in gold.items() if k.startswith("param/")}

    def fresh_grids(self):
        return {k: v.clone() for k, v in self.grids0.items()}

    def decoder_tensors(self, sub):
        return {k: v for k, v in self.P.items() if k.startswith(sub + "_decoder.")}

    def zero_decoder_grads(self):
        for v in self.P.values():
            v.grad = None

    def samples(self, idx, H0, H1, W0, W1, c2w, depth, color):
        H, W, fx, fy, cx, cy = self.intr
        return self.orc.pixel_rays(idx, H0, H1, W0, W1, fx, fy, cx, cy, c2w, depth, color)

    def render(self, c, stage, rays_d, rays_o, gt_depth):
        return self.orc.render_batch_ray(c, self.P, rays_d, rays_o, stage, gt_depth, self.bound, lo=self.lo)


class ProductOps:
    """nice_slam_amd on the GPU (HIP kernels through the C ABI)."""

    def __init__(self, gold, device="cuda:0"):
        import types
        import nice_slam_amd as nsa
        from nice_slam_amd.common import set_decoder_bounds
        self.nsa, self.dev = nsa, torch.device(device)
        self.bound = torch.from_numpy(gold["bound"])
        self.intr = tuple(float(v) for v in gold["intr"])
        H, W, fx, fy, cx, cy = self.intr
        cfg = {"rendering": {"lindisp": False, "perturb": 0.0, "N_samples": 32, "N_surface": 16, "N_importance": 0},
               "scale": 1, "occupancy": True}
        slam = types.SimpleNamespace(nice=True, bound=self.bound, H=int(H), W=int(W), fx=fx, fy=fy, cx=cx, cy=cy)
        self.renderer = nsa.Renderer(cfg, None, slam)
        self.dec = nsa.NICE(coarse=True)
        self.dec.load_state_dict({k[6:]: torch.from_numpy(v) for k, v in gold.items() if k.startswith("param/")})
        self.dec = self.dec.to(self.dev)
        set_decoder_bounds(self.dec, self.bound, 2.0)
        for p in self.dec.parameters():                          # the reference never freezes decoder parameters
            p.requires_grad_(True)
        self.grids0 = {k[5:]: nsa.to_channels_last(torch.from_numpy(v).to(self.dev)) for k, v in gold.items() if k.startswith("grid/")}

    def fresh_grids(self):
        return {k: v.detach().clone(memory_format=torch.preserve_format) for k, v in self.grids0.items()}

    def decoder_tensors(self, sub):
        return {f"{sub}_decoder." + n: p for n, p in self.dec.sub(sub).named_parameters()}

    def zero_decoder_grads(self):
        for p in self.dec.parameters():
            p.grad = None

    def samples(self, idx, H0, H1, W0, W1, c2w, depth, color):
        H, W, fx, fy, cx, cy = self.intr
        return self.nsa.common.samples_from_indices(idx.to(self.dev), H0, H1, W0, W1, fx, fy, cx, cy, c2w, depth, color)

    def render(self, c, stage, rays_d, rays_o, gt_depth):
        return self.renderer.render_batch_ray(c, self.dec, rays_d, rays_o, self.dev, stage, gt_depth=gt_depth)


# --------------------------------------------------------------------------------------------------------------------
# the loops
# --------------------------------------------------------------------------------------------------------------------
def _t(a, dev, dtype=None):
    t = torch.from_numpy(np.ascontiguousarray(a)).to(dev)
    return t if dtype is None else t.to(dtype)


def replay_mapper(gold, pre, ops):
    """-> list (one per iteration) of {"stage", "loss", "grads": {name: tensor}}; names as in the fixture."""
    dev = ops.dev
    coarse, ba = pre == "coarse/", pre == "ba/"
    n_iters = int(gold[pre + "n_iters"])
    H, W = int(ops.intr[0]), int(ops.intr[1])
    frames = {}
    i = 0
    while f"frame/{i}/depth" in gold:
        frames[i] = (_t(gold[f"frame/{i}/depth"], dev), _t(gold[f"frame/{i}/color"], dev))
        i += 1
    order = [int(v) for v in gold[pre + "draw_frames"]]
    n_sel = 0 if coarse else 1                                   # the overlap keyframe selection draws once first (Mapper.py:185)
    per_iter = (len(order) - n_sel) // n_iters
    it_frames = order[n_sel:n_sel + per_iter]                    # e.g. [kf .., last kf, current]
    # estimated poses: keyframes (frame id -> est_c2w as handed to optimize_map), current frame = id 0
    est = {0: _t(gold[pre + "cur_c2w"], dev)}
    j = 0
    while f"{pre}kf/{j}/frame" in gold:
        est[int(gold[f"{pre}kf/{j}/frame"])] = _t(gold[f"{pre}kf/{j}/est_c2w_in"], dev)
        j += 1
    # BA: every frame of the window except the oldest keyframe gets a camera tensor, in window order (Mapper.py:346-363)
    cam_of = {}
    if ba:
        kf_ids = [f for f in it_frames if f != 0]
        oldest = min(kf_ids)                                     # keyframe index order == frame id order in the fixture
        cam_of = {f: n for n, f in enumerate([f for f in it_frames if f != oldest])}
    keys = ("grid_coarse",) if coarse else ("grid_middle", "grid_fine", "grid_color")
    c = ops.fresh_grids()
    masks = {k: _t(gold[f"{pre}mask/{k}"], dev).bool()[None, None].expand(c[k].shape) for k in keys}      # Mapper.py:315-318
    state = {k: c[k][masks[k]].clone() for k in keys}            # val_grad = val[mask].clone()   (:321)
    for n_, f in enumerate(sorted(cam_of, key=cam_of.get)):
        state[f"cam{n_}"] = _t(gold[f"{pre}init/cam{n_}"], dev)
    dec_col = ops.decoder_tensors("color")
    for k, p in dec_col.items():
        state[k] = p.detach().clone()
    draw = n_sel
    out = []
    for it in range(n_iters):
        stage = stage_of(it, n_iters, coarse)
        leaves = {}
        for k in state:                                          # teacher forcing: the reference's state at this point
            if it > 0 and f"{pre}it{it - 1}/after/{k}" in gold:
                state[k] = _t(gold[f"{pre}it{it - 1}/after/{k}"], dev, state[k].dtype)
            if k in dec_col:
                with torch.no_grad():
                    dec_col[k].copy_(state[k])
            else:
                leaves[k] = state[k].clone().requires_grad_(True)
        for k in keys:                                           # Mapper.py:394-401
            val = c[k]
            val[masks[k]] = leaves[k]
            c[k] = val
        ops.zero_decoder_grads()
        ro, rd, gd, gc = [], [], [], []
        for f in it_frames:                                      # Mapper.py:437-468
            depth, color = frames[f]
            c2w = get_camera_from_tensor(leaves[f"cam{cam_of[f]}"]) if f in cam_of else est[f]
            idx = _t(gold[f"{pre}draw/{draw}"], "cpu")
            draw += 1
            o, d, dep, col = ops.samples(idx, 0, H, 0, W, c2w, depth, color)
            ro.append(o.float()); rd.append(d.float()); gd.append(dep.float()); gc.append(col.float())
        rays_o, rays_d, gt_depth, gt_color = torch.cat(ro), torch.cat(rd), torch.cat(gd), torch.cat(gc)
        with torch.no_grad():                                    # Mapper.py:471-481
            t = (ops.bound.unsqueeze(0).to(dev) - rays_o.detach().unsqueeze(-1)) / rays_d.detach().unsqueeze(-1)
            t, _ = torch.min(torch.max(t, dim=2)[0], dim=1)
            inside = t >= gt_depth
        rays_d, rays_o, gt_depth, gt_color = rays_d[inside], rays_o[inside], gt_depth[inside], gt_color[inside]
        depth, unc, color = ops.render(c, stage, rays_d, rays_o, None if coarse else gt_depth)
        dm = gt_depth > 0                                        # Mapper.py:487-493
        loss = torch.abs(gt_depth[dm] - depth[dm]).sum()
        if stage == "color":
            loss = loss + 0.2 * torch.abs(gt_color - color).sum()
        loss.backward()
        grads = {k: v.grad.detach().clone() for k, v in leaves.items() if v.grad is not None}
        for sub in ("color", "middle", "fine", "coarse"):
            for k, p in ops.decoder_tensors(sub).items():
                if p.grad is not None:
                    grads[k] = p.grad.detach().clone()
        out.append({"stage": stage, "loss": float(loss.detach()), "grads": grads, "n_rays": int(rays_o.shape[0])})
        for k in keys:                                           # Mapper.py:511-519
            val = c[k].detach()
            val[masks[k]] = leaves[k].clone().detach()
            c[k] = val
    return out


def replay_mapper_fused(gold, pre, ops):
    """The same iterations through the entry point the bench times: ``nice_slam_amd.mapping_loss`` (window kernel + render +
    loss as ONE autograd node, the bounding-box pre-filter as a mask) fed with the RECORDED pixel draws -- compared with what
    the real Mapper.optimize_map produced (src/Mapper.py:437-503)."""
    nsa, dev = ops.nsa, ops.dev
    coarse, ba = pre == "coarse/", pre == "ba/"
    n_iters = int(gold[pre + "n_iters"])
    frames = {}
    i = 0
    while f"frame/{i}/depth" in gold:
        frames[i] = (_t(gold[f"frame/{i}/depth"], dev), _t(gold[f"frame/{i}/color"], dev))
        i += 1
    order = [int(v) for v in gold[pre + "draw_frames"]]
    n_sel = 0 if coarse else 1
    per_iter = (len(order) - n_sel) // n_iters
    it_frames = order[n_sel:n_sel + per_iter]
    est = {0: _t(gold[pre + "cur_c2w"], dev)}
    j = 0
    while f"{pre}kf/{j}/frame" in gold:
        est[int(gold[f"{pre}kf/{j}/frame"])] = _t(gold[f"{pre}kf/{j}/est_c2w_in"], dev)
        j += 1
    cam_of = {}
    if ba:
        kf_ids = [f for f in it_frames if f != 0]
        oldest = min(kf_ids)
        cam_of = {f: n for n, f in enumerate([f for f in it_frames if f != oldest])}
    keys = ("grid_coarse",) if coarse else ("grid_middle", "grid_fine", "grid_color")
    c = ops.fresh_grids()
    masks = {k: _t(gold[f"{pre}mask/{k}"], dev).bool()[None, None].expand(c[k].shape) for k in keys}
    state = {k: c[k][masks[k]].clone() for k in keys}
    for n_, f in enumerate(sorted(cam_of, key=cam_of.get)):
        state[f"cam{n_}"] = _t(gold[f"{pre}init/cam{n_}"], dev)
    dec_col = ops.decoder_tensors("color")
    for k, p in dec_col.items():
        state[k] = p.detach().clone()
    draw = n_sel
    out = []
    for it in range(n_iters):
        stage = stage_of(it, n_iters, coarse)
        leaves = {}
        for k in state:
            if it > 0 and f"{pre}it{it - 1}/after/{k}" in gold:
                state[k] = _t(gold[f"{pre}it{it - 1}/after/{k}"], dev, state[k].dtype)
            if k in dec_col:
                with torch.no_grad():
                    dec_col[k].copy_(state[k])
            else:
                leaves[k] = state[k].clone().requires_grad_(True)
        for k in keys:
            val = c[k]
            val[masks[k]] = leaves[k]
            c[k] = val
        ops.zero_decoder_grads()
        fr, idx = [], []
        for f in it_frames:
            depth, color = frames[f]
            c2w = get_camera_from_tensor(leaves[f"cam{cam_of[f]}"]) if f in cam_of else est[f]
            fr.append((c2w, depth, color))
            idx.append(_t(gold[f"{pre}draw/{draw}"], dev))
            draw += 1
        info = {}
        loss = nsa.mapping_loss(ops.renderer, c, ops.dec, fr, int(idx[0].numel()), stage, w_color=0.2, indices=torch.cat(idx),
                                coarse_mapper=coarse, out=info)
        loss.backward()
        grads = {k: v.grad.detach().clone() for k, v in leaves.items() if v.grad is not None}
        for sub in ("color", "middle", "fine", "coarse"):
            for k, p in ops.decoder_tensors(sub).items():
                if p.grad is not None:
                    grads[k] = p.grad.detach().clone()
        out.append({"stage": stage, "loss": float(loss.detach()), "grads": grads, "n_rays": int(info["keep"].sum())})
        for k in keys:
            val = c[k].detach()
            val[masks[k]] = leaves[k].clone().detach()
            c[k] = val
    return out


def replay_tracker_fused(gold, ops):
    """Tracker.optimize_cam_in_batch through ``nice_slam_amd.tracking_loss`` (one autograd node) with the recorded draws."""
    nsa, dev = ops.nsa, ops.dev
    pre = "track/"
    H0, H1, W0, W1 = (int(v) for v in gold[pre + "crop"])
    H, W = int(ops.intr[0]), int(ops.intr[1])
    assert H1 == H - H0 and W1 == W - W0
    depth_img, color_img = _t(gold["frame/0/depth"], dev), _t(gold["frame/0/color"], dev)
    c = ops.fresh_grids()
    out = []
    for it in range(int(gold[pre + "n_iters"])):
        cam0 = gold[pre + "init/cam"] if it == 0 else gold[f"{pre}it{it - 1}/after/cam"]
        cam = _t(cam0, dev).requires_grad_(True)
        ops.zero_decoder_grads()
        idx = _t(gold[f"{pre}draw/{it}"], dev)
        info = {}
        loss = nsa.tracking_loss(ops.renderer, c, ops.dec, get_camera_from_tensor(cam), depth_img, color_img, int(idx.numel()), H0, W0,
                                 w_color=0.5, handle_dynamic=True, use_color=True, indices=idx, out=info)
        loss.backward()
        out.append({"stage": "color", "loss": float(loss.detach()), "grads": {"cam": cam.grad.detach().clone()}, "n_rays": int(info["keep"].sum())})
    return out


def replay_tracker(gold, ops):
    """Tracker.optimize_cam_in_batch (src/Tracker.py:87-125), teacher-forced on the camera tensor."""
    dev = ops.dev
    pre = "track/"
    H0, H1, W0, W1 = (int(v) for v in gold[pre + "crop"])
    depth_img, color_img = _t(gold["frame/0/depth"], dev), _t(gold["frame/0/color"], dev)
    c = ops.fresh_grids()
    out = []
    for it in range(int(gold[pre + "n_iters"])):
        cam0 = gold[pre + "init/cam"] if it == 0 else gold[f"{pre}it{it - 1}/after/cam"]
        cam = _t(cam0, dev).requires_grad_(True)
        ops.zero_decoder_grads()
        c2w = get_camera_from_tensor(cam)
        o, d, gd, gc = ops.samples(_t(gold[f"{pre}draw/{it}"], "cpu"), H0, H1, W0, W1, c2w, depth_img, color_img)
        with torch.no_grad():                                    # Tracker.py:95-104
            t = (ops.bound.unsqueeze(0).to(dev) - o.detach().unsqueeze(-1)) / d.detach().unsqueeze(-1)
            t, _ = torch.min(torch.max(t, dim=2)[0], dim=1)
            inside = t >= gd
        d, o, gd, gc = d[inside], o[inside], gd[inside], gc[inside]
        depth, unc, color = ops.render(c, "color", d, o, gd)
        unc = unc.detach()                                       # Tracker.py:110-123 (handle_dynamic, colour term 0.5)
        tmp = torch.abs(gd - depth) / torch.sqrt(unc + 1e-10)
        mask = (tmp < 10 * tmp.median()) & (gd > 0)
        loss = (torch.abs(gd - depth) / torch.sqrt(unc + 1e-10))[mask].sum()
        loss = loss + 0.5 * torch.abs(gc - color)[mask].sum()
        loss.backward()
        out.append({"stage": "color", "loss": float(loss.detach()), "grads": {"cam": cam.grad.detach().clone()}, "n_rays": int(o.shape[0])})
    return out


def rel_err(a, b):
    a = np.asarray(a.detach().cpu() if torch.is_tensor(a) else a, dtype=np.float64)
    b = np.asarray(b.detach().cpu() if torch.is_tensor(b) else b, dtype=np.float64)
    den = np.abs(b).max()
    return float(np.abs(a - b).max() / den) if den > 0 else float(np.abs(a).max())


def compare(gold, pre, got, tol, truth=None, loss_tol=None):
    """Every gradient the fixture holds for this case, per iteration, in max-norm.  Parameter-gradient tensors that miss
    `tol` get the same second chance as in scene_util.parity_failures: the tensor may be no further from the fp64 `truth`
    than twice the reference's own value is."""
    bad = []
    n_checked = 0
    for it, res in enumerate(got):
        ref_loss = float(gold[pre + "losses"][it])
        if abs(res["loss"] - ref_loss) > (loss_tol or tol) * abs(ref_loss):
            bad.append((it, "loss", res["loss"], ref_loss))
        names = [k[len(f"{pre}it{it}/grad/"):] for k in gold if k.startswith(f"{pre}it{it}/grad/")]
        assert names, (pre, it)
        for nm in names:
            ref = gold[f"{pre}it{it}/grad/{nm}"]
            n_checked += 1
            if nm not in res["grads"]:
                bad.append((it, nm, "missing"))
                continue
            e = rel_err(res["grads"][nm], ref)
            if e < tol:
                continue
            if "_decoder." in nm and truth is not None:
                blob = sorted(q for q in names if q.split(".")[0] == nm.split(".")[0])
                bg = np.concatenate([res["grads"][q].detach().cpu().numpy().reshape(-1) for q in blob])
                br = np.concatenate([gold[f"{pre}it{it}/grad/{q}"].reshape(-1) for q in blob])
                tr = truth[it]["grads"][nm]
                e_t, e_r = rel_err(res["grads"][nm], tr), rel_err(ref, tr)
                if e_t <= max(2.0 * e_r, tol):
                    continue
                bad.append((it, nm, e, rel_err(bg, br), e_t, e_r))
            else:
                bad.append((it, nm, e))
    return bad, n_checked
