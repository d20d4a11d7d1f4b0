#!/usr/bin/env python3
"""TEST INFRASTRUCTURE (measurement script; builds its inputs with tests/scene_util.py, i.e. with oracle helpers).
Forward-only, large-batch entry points of the path: Renderer.eval_points on a Mesher-sized
point cloud (256^3 = 16.8 M points, src/utils/Mesher.py:382-433) and Renderer.render_img on a full 680x1200 frame
(816 k rays, src/utils/Renderer.py:200-255).  Prints throughput and the fraction of the fp32 MFMA roofline."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
from scene_util import make_scene, build_product
dev = torch.device("cuda", 0)
sc = make_scene(seed=0, n_rays=16, scene="replica_room0", fine_scale=1.0)
renderer, dec, grids = build_product(sc, dev)
MAC = {"coarse": 6176, "middle": 15479, "fine": 36078, "color": 51653}
b = sc["bound"]
n = 256 ** 3
g = torch.Generator(device=dev).manual_seed(0)
pts = (torch.rand((n, 3), generator=g, device=dev, dtype=torch.float64)) * (b[:, 1] - b[:, 0]).to(dev) + b[:, 0].to(dev)
for stage in ("coarse", "middle", "fine", "color"):
    renderer.eval_points(pts[:4096], dec, grids, stage, dev); torch.cuda.synchronize()
    t0 = time.perf_counter(); out = renderer.eval_points(pts, dec, grids, stage, dev); torch.cuda.synchronize(); dt = time.perf_counter() - t0
    tf = n * MAC[stage] * 2 / dt / 1e12
    print(f"eval_points {stage:6s}: {n/1e6:.1f} M points in {dt*1e3:7.1f} ms -> {n/dt/1e6:7.1f} M points/s, {tf:6.1f} TFLOP/s = {tf/157.3*100:4.1f}% of fp32 MFMA peak")
H, W = renderer.H, renderer.W
gt = sc["depth_img"].to(dev)
for stage in ("middle", "color"):
    renderer.render_img(grids, dec, sc["c2w"].to(dev), dev, stage, gt_depth=gt); torch.cuda.synchronize()
    t0 = time.perf_counter(); d, u, c = renderer.render_img(grids, dec, sc["c2w"].to(dev), dev, stage, gt_depth=gt); torch.cuda.synchronize(); dt = time.perf_counter() - t0
    tf = H * W * 48 * MAC[stage] * 2 / dt / 1e12
    print(f"render_img  {stage:6s}: {H}x{W} = {H*W/1e3:.0f} k rays in {dt*1e3:7.1f} ms -> {H*W/dt/1e6:6.2f} M rays/s, {tf:6.1f} TFLOP/s = {tf/157.3*100:4.1f}% of fp32 MFMA peak")
