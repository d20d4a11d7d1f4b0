#!/usr/bin/env python3
"""TEST INFRASTRUCTURE (measurement script; builds its inputs with tests/scene_util.py, i.e. with oracle helpers).
Secondary baseline: the reference's op sequence -- restated by the oracle -- executed by stock
PyTorch-ROCm ATen kernels on the same MI355X (no nsr kernels involved).  Prints rays/s per stage and the mapping mix."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
from scene_util import make_scene
from oracle import nice_oracle as orc
dev = "cuda:0"
n = int(sys.argv[1]) if len(sys.argv) > 1 else 1000
sc = make_scene(seed=0, n_rays=n, scene="replica_room0", fine_scale=1.0, zero_frac=0.01, depth_range=(1.0, 4.0))
grids = {k: v.to(dev).requires_grad_(True) for k, v in sc["grids"].items()}
params = {k: v.to(dev).requires_grad_(True) for k, v in sc["params"].items()}
o, d, gd, gc = (sc[k].to(dev) for k in ("rays_o", "rays_d", "gt_depth", "gt_color"))
t = {}
for stage in ("middle", "fine", "color"):
    for i in range(8):
        if i == 3:
            torch.cuda.synchronize(); t0 = time.perf_counter()
        for v in list(grids.values()) + list(params.values()): v.grad = None
        depth, var, rgb = orc.render_batch_ray(grids, params, d, o, stage, gd, sc["bound"])
        loss = (torch.abs(gd - depth) * (gd > 0)).sum() + (0.2 * torch.abs(gc - rgb).sum() if stage == "color" else 0)
        loss.backward()
    torch.cuda.synchronize()
    t[stage] = (time.perf_counter() - t0) / 5
    print(f"stock ATen on MI355X, {stage}: {t[stage]*1e3:.2f} ms / iteration  -> {n/t[stage]:.0f} rays/s")
mix = (25 * t["middle"] + 12 * t["fine"] + 23 * t["color"]) / 60
print(f"mapping mix (25/12/23): {mix*1e3:.2f} ms / iteration -> {n/mix:.0f} rays/s")
