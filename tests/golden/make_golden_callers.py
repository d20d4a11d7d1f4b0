#!/usr/bin/env python3
"""Mint golden fixtures of the two CALLERS of the hot path by executing the UNMODIFIED reference
``src/Mapper.py`` (``Mapper.optimize_map``, :230-540) and ``src/Tracker.py`` (``Tracker.optimize_cam_in_batch``,
:71-128) on the CPU with the reference's own Renderer / decoders / get_samples.

Run in the build container only (``/root/reference`` does not exist on the GPU box):

    PYTHONDONTWRITEBYTECODE=1 python tests/golden/make_golden_callers.py

Writes ``tests/golden/caller_steps.npz``:  for each of four cases
    map/      mapper, 2 keyframes + current frame, no BA, 5 iterations (middle, middle, middle, fine, color)
    ba/       mapper with local BA (camera tensors optimised in the colour stage), 3 keyframes, 5 iterations
    coarse/   coarse mapper (stage 'coarse', gt_depth=None, keyframe selection 'global'), 3 iterations
    track/    tracker, 3 iterations of optimize_cam_in_batch (Replica-style crop, median outlier mask)
the inputs (grids, decoder state_dict, frames, poses, frustum masks as the reference computed them), every index
draw of ``torch.randint`` in call order, and per iteration: the loss, the gradient of every tensor the optimiser
holds, and the value of those tensors after the step.  ``tests/test_hip_real_callers.py`` replays the same steps
through ``nice_slam_amd`` in a loop of the same shape and compares state by state.

Arithmetic-neutral shims; none of them touches the arithmetic of the path under test:
  * stub modules for colorama / open3d / skimage / trimesh / ``src.utils.datasets`` / ``src.utils.Visualizer``;
  * ``cv2.remap`` -> oracle/frustum_oracle.remap_bilinear (OpenCV is absent; the resulting frustum masks are
    recorded as INPUTS of the fixture, so nothing downstream depends on that restatement);
  * ``mathutils.Matrix.to_quaternion`` -> scipy Rotation, (w,x,y,z) order (src/common.py:190-193); the resulting
    camera tensors are recorded;
  * ``NICE.forward`` builds ``f'cuda:{p.get_device()}'`` (decoder.py:316) and ``quad2rotation`` does
    ``.to(quad.get_device())`` (common.py:150): both invalid on CPU, replaced by the same expressions on ``.device``.
"""
import os
import sys
import types

os.environ.setdefault("PYTHONDONTWRITEBYTECODE", "1")
sys.dont_write_bytecode = True
REF = "/root/reference"
HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, REF)
sys.path.insert(1, ROOT)

import numpy as np
import torch

from oracle.frustum_oracle import remap_bilinear          # noqa: E402  (stands in for the absent cv2.remap only)

# ------------------------------------------------------------------------------------------------
# stub modules
# ------------------------------------------------------------------------------------------------
for _m in ("cv2", "colorama", "open3d", "skimage", "skimage.measure", "trimesh", "mathutils"):
    sys.modules.setdefault(_m, types.ModuleType(_m))
sys.modules["colorama"].Fore = types.SimpleNamespace(GREEN="", MAGENTA="", RED="")
sys.modules["colorama"].Style = types.SimpleNamespace(RESET_ALL="")
sys.modules["cv2"].INTER_LINEAR = 1


def _remap(src, mapx, mapy, interpolation=1):
    return remap_bilinear(src, np.asarray(mapx).reshape(-1), np.asarray(mapy).reshape(-1)).reshape(-1, 1)


sys.modules["cv2"].remap = _remap


class _Matrix:
    def __init__(self, R):
        self.R = np.asarray(R, dtype=np.float64)

    def to_quaternion(self):
        from scipy.spatial.transform import Rotation
        x, y, z, w = Rotation.from_matrix(self.R).as_quat()
        return np.array([w, x, y, z])


sys.modules["mathutils"].Matrix = _Matrix

_ds = types.ModuleType("src.utils.datasets")
_ds.get_dataset = lambda cfg, args, scale, device="cpu": [None] * 8
sys.modules["src.utils.datasets"] = _ds
_vis = types.ModuleType("src.utils.Visualizer")


class _NoVis:
    def __init__(self, *a, **k):
        pass

    def vis(self, *a, **k):
        pass


_vis.Visualizer = _NoVis
sys.modules["src.utils.Visualizer"] = _vis

import src.common as ref_common                            # noqa: E402
from src import config as ref_config                       # noqa: E402
from src.conv_onet.models import decoder as ref_decoder    # noqa: E402
from src.utils.Renderer import Renderer                    # noqa: E402
from src.Mapper import Mapper                              # noqa: E402
from src.Tracker import Tracker                            # noqa: E402


def _quad2rotation(quad):                                   # common.py:137-160 with .device instead of get_device()
    bs = quad.shape[0]
    qr, qi, qj, qk = quad[:, 0], quad[:, 1], quad[:, 2], quad[:, 3]
    two_s = 2.0 / (quad * quad).sum(-1)
    rot_mat = torch.zeros(bs, 3, 3).to(quad.device)
    rot_mat[:, 0, 0] = 1 - two_s * (qj ** 2 + qk ** 2)
    rot_mat[:, 0, 1] = two_s * (qi * qj - qk * qr)
    rot_mat[:, 0, 2] = two_s * (qi * qk + qj * qr)
    rot_mat[:, 1, 0] = two_s * (qi * qj + qk * qr)
    rot_mat[:, 1, 1] = 1 - two_s * (qi ** 2 + qk ** 2)
    rot_mat[:, 1, 2] = two_s * (qj * qk - qi * qr)
    rot_mat[:, 2, 0] = two_s * (qi * qk - qj * qr)
    rot_mat[:, 2, 1] = two_s * (qj * qk + qi * qr)
    rot_mat[:, 2, 2] = 1 - two_s * (qi ** 2 + qj ** 2)
    return rot_mat


ref_common.quad2rotation = _quad2rotation


class PatchedNICE(ref_decoder.NICE):
    """decoder.py:317-342 with the device string replaced by p.device"""

    def forward(self, p, c_grid, stage="middle", **kw):
        if stage == "coarse":
            occ = self.coarse_decoder(p, c_grid).squeeze(0)
        elif stage == "middle":
            occ = self.middle_decoder(p, c_grid).squeeze(0)
        elif stage == "fine":
            occ = self.fine_decoder(p, c_grid) + self.middle_decoder(p, c_grid).squeeze(0)
        else:
            fine = self.fine_decoder(p, c_grid)
            raw = self.color_decoder(p, c_grid)
            raw[..., -1] = fine + self.middle_decoder(p, c_grid).squeeze(0)
            return raw
        raw = torch.zeros(occ.shape[0], 4).float()
        raw[..., -1] = occ
        return raw


# ------------------------------------------------------------------------------------------------
# recording hooks
# ------------------------------------------------------------------------------------------------
REC = {"draws": [], "steps": [], "losses": [], "init": None, "frames": []}
_randint = torch.randint


def _rec_randint(*a, **k):
    out = _randint(*a, **k)
    REC["draws"].append(out.detach().cpu().numpy().copy())
    return out


class RecAdam(torch.optim.Adam):
    def __init__(self, params, *a, **k):
        super().__init__(params, *a, **k)
        REC["init"] = [[p.detach().clone() for p in g["params"]] for g in self.param_groups]

    def step(self, closure=None):
        grads = [[None if p.grad is None else p.grad.detach().clone() for p in g["params"]] for g in self.param_groups]
        r = super().step(closure)
        after = [[p.detach().clone() for p in g["params"]] for g in self.param_groups]
        REC["steps"].append({"grads": grads, "after": after, "lrs": [g["lr"] for g in self.param_groups]})
        return r


_backward = torch.Tensor.backward


def _rec_backward(self, *a, **k):
    if self.dim() == 0:
        REC["losses"].append(float(self.detach()))
    return _backward(self, *a, **k)


def reset_rec():
    REC["draws"].clear(); REC["steps"].clear(); REC["losses"].clear(); REC["frames"].clear(); REC["init"] = None


# ------------------------------------------------------------------------------------------------
# scene
# ------------------------------------------------------------------------------------------------
H, W, FX, FY, CX, CY = 48, 64, 60.0, 60.0, 31.5, 23.5


def pose(ang, t):
    c2w = torch.eye(4, dtype=torch.float32)
    c2w[:3, :3] = torch.tensor([[np.cos(ang), 0, np.sin(ang)], [0, 1, 0], [-np.sin(ang), 0, np.cos(ang)]], dtype=torch.float32)
    c2w[:3, 3] = torch.tensor(t)
    return c2w


def frame(g, lo=0.25, hi=1.2):
    depth = torch.rand((H, W), generator=g) * (hi - lo) + lo
    depth[torch.rand((H, W), generator=g) < 0.06] = 0.0
    return depth, torch.rand((H, W, 3), generator=g)


def build():
    cfg = ref_config.load_config(os.path.join(REF, "configs/nice_slam.yaml"))
    cfg["mapping"]["bound"] = [[-0.7, 0.8], [-0.6, 0.7], [-0.5, 0.6]]
    cfg["grid_len"].update({"coarse": 0.8, "middle": 0.4, "fine": 0.25, "color": 0.25})
    cfg["mapping"].update({"device": "cpu", "pixels": 300})
    cfg["tracking"].update({"device": "cpu", "pixels": 120, "ignore_edge_W": 6, "ignore_edge_H": 4})
    cfg["data"] = {"output": "output/golden", "dim": 3}
    # NICE_SLAM.load_bound (src/NICE_SLAM.py:137-150)
    bound = torch.from_numpy(np.array(cfg["mapping"]["bound"]) * cfg["scale"])
    div = cfg["grid_len"]["bound_divisible"]
    bound[:, 1] = (((bound[:, 1] - bound[:, 0]) / div).int() + 1) * div + bound[:, 0]
    g = torch.Generator().manual_seed(77)
    xyz = bound[:, 1] - bound[:, 0]
    grids = {}
    for name, std in (("coarse", 0.01), ("middle", 0.01), ("fine", 0.01), ("color", 0.01)):   # grid_init, :192-250
        ext = xyz * cfg["model"]["coarse_bound_enlarge"] if name == "coarse" else xyz
        shape = list(map(int, (ext / cfg["grid_len"][name]).tolist()))
        shape[0], shape[2] = shape[2], shape[0]
        grids["grid_" + name] = torch.zeros([1, 32, *shape]).normal_(mean=0, std=std, generator=g)
    torch.manual_seed(4321)
    dec = PatchedNICE(dim=3, c_dim=32, coarse=True, coarse_grid_len=0.8, middle_grid_len=0.4, fine_grid_len=0.25,
                      color_grid_len=0.25, hidden_size=32, pos_embedding_method="fourier")
    with torch.no_grad():
        for n, p in dec.named_parameters():
            if n.endswith(".bias"):
                p.add_(torch.randn_like(p) * 0.1)
    dec.bound = bound
    dec.middle_decoder.bound = dec.fine_decoder.bound = dec.color_decoder.bound = bound
    dec.coarse_decoder.bound = bound * cfg["model"]["coarse_bound_enlarge"]
    return cfg, bound, grids, dec, g


def make_slam(cfg, bound, grids, dec):
    s = types.SimpleNamespace()
    s.idx = torch.zeros(1).int(); s.nice = True; s.shared_c = grids; s.bound = bound
    s.logger = None; s.mesher = None; s.output = "output/golden"; s.verbose = False; s.low_gpu_mem = False
    s.mapping_idx = torch.zeros(1).int(); s.mapping_cnt = torch.zeros(1).int(); s.shared_decoders = dec
    s.estimate_c2w_list = torch.zeros((8, 4, 4)); s.gt_c2w_list = torch.zeros((8, 4, 4))
    s.mapping_first_frame = torch.zeros(1).int()
    s.H, s.W, s.fx, s.fy, s.cx, s.cy = H, W, FX, FY, CX, CY
    s.renderer = Renderer(cfg, None, s)
    return s


def save_steps(out, pre, names):
    """names: per param group, the key under which each tensor is stored"""
    out[pre + "n_iters"] = np.array(len(REC["steps"]))
    out[pre + "losses"] = np.array(REC["losses"], dtype=np.float64)
    for i, d in enumerate(REC["draws"]):
        out[f"{pre}draw/{i}"] = d
    out[pre + "n_draws"] = np.array(len(REC["draws"]))
    for gi, group in enumerate(names):
        for pi, nm in enumerate(group):
            if nm.startswith("cam"):
                out[f"{pre}init/{nm}"] = REC["init"][gi][pi].numpy()
    if REC["frames"]:
        out[pre + "draw_frames"] = np.array(REC["frames"])
    for it, st in enumerate(REC["steps"]):
        out[f"{pre}it{it}/lrs"] = np.array(st["lrs"], dtype=np.float64)
        for gi, group in enumerate(names):
            for pi, nm in enumerate(group):
                g = st["grads"][gi][pi]
                if g is not None and (st["lrs"][gi] > 0 or float(g.abs().max()) > 0):
                    out[f"{pre}it{it}/grad/{nm}"] = g.numpy()
                if st["lrs"][gi] > 0:
                    out[f"{pre}it{it}/after/{nm}"] = st["after"][gi][pi].numpy()


def main():
    torch.randint = _rec_randint
    torch.optim.Adam = RecAdam
    torch.Tensor.backward = _rec_backward
    cfg, bound, grids0, dec0, g = build()
    out = {"bound": bound.numpy(), "intr": np.array([H, W, FX, FY, CX, CY])}
    for k, v in grids0.items():
        out["grid/" + k] = v.numpy()
    for k, v in dec0.state_dict().items():
        out["param/" + k] = v.numpy()
    poses = [pose(0.20, [0.10, 0.05, 0.45]), pose(0.05, [0.0, 0.0, 0.40]), pose(0.35, [0.2, 0.1, 0.40]),
             pose(-0.1, [-0.1, 0.05, 0.42])]
    frames = [frame(g) for _ in poses]
    for i, (p, (d, c)) in enumerate(zip(poses, frames)):
        out[f"frame/{i}/c2w"] = p.numpy(); out[f"frame/{i}/depth"] = d.numpy(); out[f"frame/{i}/color"] = c.numpy()
    dec_names = lambda sub: [f"{sub}_decoder." + n for n, _ in getattr(dec0, sub + "_decoder").named_parameters()]

    def fresh():
        import copy
        return {k: v.clone() for k, v in grids0.items()}, copy.deepcopy(dec0)

    def kf(i, est_noise=0.0):
        est = poses[i].clone()
        est[:3, 3] += est_noise
        return {"gt_c2w": poses[i].clone(), "idx": 10 * i, "color": frames[i][1].clone(), "depth": frames[i][0].clone(), "est_c2w": est}

    def run_mapper(pre, n_kf, ba, coarse, iters):
        grids, dec = fresh()
        slam = make_slam(cfg, bound, grids, dec)
        m = Mapper(cfg, None, slam, coarse_mapper=coarse)
        m.BA = ba
        kfd = [kf(i + 1, 0.01 * (i + 1)) for i in range(n_kf)]
        kfl = [d["idx"] for d in kfd]
        m.keyframe_dict, m.keyframe_list = kfd, kfl
        cur = poses[0].clone(); cur[:3, 3] += 0.005
        torch.manual_seed(11); np.random.seed(11)
        reset_rec()
        masks = {}
        _gm = m.get_mask_from_c2w

        def rec_mask(c2w, key, val_shape, depth_np):
            r = _gm(c2w, key, val_shape, depth_np)
            masks[key] = np.asarray(r).copy()
            return r

        m.get_mask_from_c2w = rec_mask
        import src.Mapper as ref_mapper_mod
        _gs = ref_common.get_samples

        def rec_get_samples(H0, H1, W0, W1, n, HH, WW, fx, fy, cx, cy, c2w, depth, color, device):
            fid = [i for i, (d, _) in enumerate(frames) if d.shape == depth.shape and torch.equal(d, depth.cpu())]
            REC["frames"].append(fid[0])
            return _gs(H0, H1, W0, W1, n, HH, WW, fx, fy, cx, cy, c2w, depth, color, device)

        ref_mapper_mod.get_samples = rec_get_samples
        ret = m.optimize_map(iters, 1.0 if not coarse else 1.0, 40, frames[0][1], frames[0][0], poses[0], kfd, kfl, cur_c2w=cur)
        for k, v in masks.items():
            out[f"{pre}mask/{k}"] = np.ascontiguousarray(np.transpose(v, (2, 1, 0)))       # (Z,Y,X) like Mapper.py:317
        out[pre + "cur_c2w"] = cur.numpy()
        for i, d in enumerate(kfd):
            out[f"{pre}kf/{i}/frame"] = np.array(i + 1); out[f"{pre}kf/{i}/est_c2w_in"] = (poses[i + 1].numpy() + 0)
            out[f"{pre}kf/{i}/est_c2w_in"][:3, 3] += 0.01 * (i + 1)
            out[f"{pre}kf/{i}/est_c2w_out"] = d["est_c2w"].numpy()
        names = [dec_names("color"), ["grid_coarse"], ["grid_middle"], ["grid_fine"], ["grid_color"]]
        if ba:
            names.append([f"cam{i}" for i in range(len(REC["steps"][0]["grads"][5]))])
            out[pre + "cur_c2w_out"] = ret.numpy()
        save_steps(out, pre, names)
        ref_mapper_mod.get_samples = _gs
        for k, v in m.c.items():
            out[f"{pre}final/{k}"] = v.detach().numpy()
        for k, v in dec.state_dict().items():
            if k.startswith("color_decoder."):
                out[f"{pre}final/param/{k}"] = v.numpy()
        print(pre, "losses", REC["losses"], "draws", len(REC["draws"]), "stages ok")

    run_mapper("map/", 2, False, False, 5)
    run_mapper("ba/", 3, True, False, 5)
    run_mapper("coarse/", 2, False, True, 3)

    # ---- tracker --------------------------------------------------------------------------------
    grids, dec = fresh()
    slam = make_slam(cfg, bound, grids, dec)
    t = Tracker(cfg, None, slam)
    t.c, t.decoders = grids, dec
    cam0 = ref_common.get_tensor_from_camera(poses[0])
    cam0 = cam0 + torch.tensor([0.0, 0.004, -0.003, 0.002, 0.01, -0.008, 0.006])
    cam = torch.autograd.Variable(cam0.clone(), requires_grad=True)
    reset_rec()
    opt = torch.optim.Adam([cam], lr=cfg["tracking"]["lr"])
    torch.manual_seed(21)
    ret_losses = [t.optimize_cam_in_batch(cam, frames[0][1], frames[0][0], t.tracking_pixels, opt) for _ in range(3)]
    out["track/cam0"] = cam0.numpy()
    out["track/ret_losses"] = np.array(ret_losses, dtype=np.float64)
    out["track/crop"] = np.array([t.ignore_edge_H, H - t.ignore_edge_H, t.ignore_edge_W, W - t.ignore_edge_W])
    save_steps(out, "track/", [["cam"]])
    print("track/ losses", REC["losses"])

    path = os.path.join(HERE, "caller_steps.npz")
    np.savez_compressed(path, **out)
    print("wrote", path, os.path.getsize(path) // 1024, "KiB,", len(out), "arrays")


if __name__ == "__main__":
    main()
