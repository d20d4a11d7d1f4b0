"""TEST INFRASTRUCTURE (measurement script; builds its inputs with tests/scene_util.py, i.e. with oracle helpers).
Host enqueue time of each part of the eager mapping iteration (no hipGraph): where a Python caller's time goes.
    python tests/perf/eager_sections.py  (on the GPU box)"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
import nice_slam_amd as nsa
from scene_util import make_scene, build_product

dev = torch.device("cuda", 0)
sc = make_scene(seed=0, n_rays=1000, scene="replica_room0", fine_scale=1.0, zero_frac=0.01, depth_range=(1.0, 4.0))
renderer, dec, grids = build_product(sc, dev)
grids = {k: v.requires_grad_(True) for k, v in grids.items()}
for p in dec.parameters(): p.requires_grad_(True)
H, W, fx, fy, cx, cy = sc["intr"]
depth_img, color_img, c2w = sc["depth_img"].to(dev), sc["color_img"].to(dev), sc["c2w"].to(dev)
params = list(dec.parameters())
T = {k: 0.0 for k in ("get_samples x5", "cat x4", "zero grads", "render fwd", "loss", "backward", "sync wait")}
N = 300
for it in range(N + 20):
    if it == 20:
        torch.cuda.synchronize(); T = {k: 0.0 for k in T}; t_all = time.perf_counter()
    t0 = time.perf_counter()
    ro, rd, gd, gc = [], [], [], []
    for _ in range(5):
        o, d, dep, col = nsa.get_samples(0, H, 0, W, 200, H, W, fx, fy, cx, cy, c2w, depth_img, color_img, dev)
        ro.append(o); rd.append(d); gd.append(dep); gc.append(col)
    t1 = time.perf_counter()
    rays_o, rays_d, gt_depth, gt_color = torch.cat(ro), torch.cat(rd), torch.cat(gd), torch.cat(gc)
    t2 = time.perf_counter()
    for g in grids.values(): g.grad = None
    for p in params: p.grad = None
    t3 = time.perf_counter()
    depth, unc, color = renderer.render_batch_ray(grids, dec, rays_d, rays_o, dev, "color", gt_depth=gt_depth)
    t4 = time.perf_counter()
    loss = (torch.abs(gt_depth - depth) * (gt_depth > 0)).sum() + 0.2 * torch.abs(gt_color - color).sum()
    t5 = time.perf_counter()
    loss.backward()
    t6 = time.perf_counter()
    if it % 10 == 9: torch.cuda.synchronize()
    t7 = time.perf_counter()
    for k, v in zip(T, (t1 - t0, t2 - t1, t3 - t2, t4 - t3, t5 - t4, t6 - t5, t7 - t6)): T[k] += v
torch.cuda.synchronize()
tot = time.perf_counter() - t_all
print(f"colour-stage eager iteration: {tot / N * 1e3:.3f} ms wall per iteration")
for k, v in T.items(): print(f"   {k:16s} {v / N * 1e6:8.1f} us host")
