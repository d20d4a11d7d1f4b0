"""TEST INFRASTRUCTURE: the `ops` binding of tools/slam_synthetic.MiniSLAM on the CPU oracle (oracle/nice_oracle.py,
oracle/frustum_oracle.py).  Lets tests/test_slam_synthetic.py run the tracker + mapper loops without a GPU; the product
binding is tools/slam_synthetic.ProductOps."""
import numpy as np
import torch

from oracle import frustum_oracle as fo
from oracle import nice_oracle as orc

GRID_LEN = {"coarse": 2.0, "middle": 0.32, "fine": 0.16, "color": 0.16}


class TorchMaskedAdam:
    """nice_slam_amd.MaskedGridAdam in plain torch (same single-tensor Adam formulas, masked voxels only)."""

    def __init__(self, grids, masks, betas=(0.9, 0.999), eps=1e-8):
        self.grids, self.masks, self.betas, self.eps = grids, masks, betas, eps
        self.state = {k: {"t": 0, "m": torch.zeros_like(g), "v": torch.zeros_like(g)} for k, g in grids.items()}

    @torch.no_grad()
    def step(self, lrs):
        b1, b2 = self.betas
        for k, g in self.grids.items():
            if g.grad is None:
                continue
            st = self.state[k]
            st["t"] += 1
            t = st["t"]
            mask = self.masks[k][None, None].to(g.dtype)
            st["m"].lerp_(g.grad, 1 - b1)
            st["v"].mul_(b2).addcmul_(g.grad, g.grad, value=1 - b2)
            denom = st["v"].sqrt() / ((1 - b2 ** t) ** 0.5) + self.eps
            g.sub_(mask * (lrs.get(k, 0.0) / (1 - b1 ** t)) * st["m"] / denom)


class OracleOps:
    def __init__(self, seq, seed=0, device="cpu", grids=None, params=None):
        """``device="cuda:0"``: the same oracle functions on the GPU, i.e. the reference's operator sequence on stock ATen /
        rocBLAS kernels (tests/perf/ate_compare.py); ``grids`` / ``params``: start from given values (e.g. the product's)."""
        self.device = torch.device(device)
        g = torch.Generator().manual_seed(seed)
        self.bound = orc.scene_bound(seq.bound_cfg, 1.0, 0.32)
        self.bound_dev = self.bound.to(self.device)
        shapes = orc.grid_shapes(self.bound, GRID_LEN, 2.0)
        grids = grids if grids is not None else orc.make_grids(shapes, generator=g)
        params = params if params is not None else orc.init_decoder_params(seed=seed + 1)
        self.c = {k: v.detach().to(self.device).contiguous().clone().requires_grad_(True) for k, v in grids.items()}
        self.P = {k: v.detach().to(self.device).clone().requires_grad_(True) for k, v in params.items()}
        self.seq = seq

    def get_samples(self, H0, H1, W0, W1, n, c2w, depth, color):
        s = self.seq
        idx = torch.randint((H1 - H0) * (W1 - W0), (n,), device=self.device)
        return orc.pixel_rays(idx, H0, H1, W0, W1, s.fx, s.fy, s.cx, s.cy, c2w, depth, color)

    def render(self, stage, rays_d, rays_o, gt_depth, gt_max=None):
        if gt_max is not None:
            # the oracle takes max(gt_depth) itself: clamp the rejected rays' depths to the kept rays' maximum (their
            # outputs are masked out of every loss term, so only the batch-global scalar matters)
            gt_depth = torch.minimum(gt_depth, gt_max.to(gt_depth.dtype).reshape(()))
        return orc.render_batch_ray(self.c, self.P, rays_d, rays_o, stage, gt_depth, self.bound)

    def keep_mask(self, rays_o, rays_d, gt_depth):
        with torch.no_grad():                       # Mapper.py:471-481 / Tracker.py:95-104, fp64 by promotion
            t = (self.bound_dev.unsqueeze(0) - rays_o.detach().unsqueeze(-1)) / rays_d.detach().unsqueeze(-1)
            t, _ = torch.min(torch.max(t, dim=2)[0], dim=1)
            keep = t >= gt_depth
            kmax = torch.where(keep, gt_depth, torch.zeros_like(gt_depth)).max().reshape(1)
        return keep, kmax

    def color_decoder_params(self):
        return [v for k, v in self.P.items() if k.startswith("color_decoder.")]

    def frustum_masks(self, c2w, depth):
        s = self.seq
        c = c2w.detach().cpu().numpy().astype(np.float32)
        if c.shape[0] == 3:
            c = np.concatenate([c, np.array([[0, 0, 0, 1]], dtype=np.float32)], 0)
        out = {}
        for k, v in self.c.items():
            if k != "grid_coarse":
                m = fo.get_mask_from_c2w(c, k, tuple(v.shape[2:]), depth.cpu().numpy(), self.bound.numpy(), s.H, s.W, s.fx, s.fy, s.cx, s.cy)
                out[k] = torch.from_numpy(np.ascontiguousarray(m.transpose(2, 1, 0))).to(self.device)
        return out

    def grid_optimizer(self, masks):
        return TorchMaskedAdam({k: self.c[k] for k in ("grid_middle", "grid_fine", "grid_color")}, masks)

    def zero_grads(self):
        for t in list(self.c.values()) + list(self.P.values()):
            t.grad = None
