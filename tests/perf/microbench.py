#!/usr/bin/env python3
"""TEST INFRASTRUCTURE (measurement script; builds its inputs with tests/scene_util.py, i.e. with oracle helpers).
Ablation timing of the backward kernel: which gradient outputs cost what (HIP events inside nsr_render_bwd)."""
import os, sys, ctypes, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
from scene_util import make_scene, build_product
sys.path.insert(0, ROOT)
from bench import HipEvents

def run(stage, n, grid, params, rays, reps=5, scene="replica_room0"):
    dev = torch.device("cuda", 0)
    sc = make_scene(seed=0, n_rays=n, scene=scene, fine_scale=1.0, zero_frac=0.01, depth_range=(1.0, 4.0))
    renderer, dec, grids = build_product(sc, dev)
    grids = {k: v.requires_grad_(grid) for k, v in grids.items()}
    for p in dec.parameters():
        p.requires_grad_(params)
    ev = HipEvents(); renderer.profile_events = ev.pair_for
    o = sc["rays_o"].to(dev).requires_grad_(rays); d = sc["rays_d"].to(dev).requires_grad_(rays)
    gd = sc["gt_depth"].to(dev); gc = sc["gt_color"].to(dev)
    tf = []
    for i in range(reps + 3):
        if i == 3:
            ev.pairs.clear()                 # drop warm-up launches
        for g in grids.values(): g.grad = None
        for p in dec.parameters(): p.grad = None
        e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
        e0.record()
        depth, unc, col = renderer.render_batch_ray(grids, dec, d, o, dev, stage, gt_depth=gd)
        e1.record()
        loss = torch.abs(gd - depth).sum() + 0.2 * torch.abs(gc - col).sum()
        loss.backward()
        torch.cuda.synchronize()
        tf.append(e0.elapsed_time(e1))
    ks = ev.summary()
    return min(tf[3:]), ks[stage][0]

if __name__ == "__main__":
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 1000
    for stage in ("color", "middle", "coarse"):
        for (g, p, r) in ((True, True, False), (False, True, False), (True, False, False), (False, False, True), (True, True, True)):
            f, b = run(stage, n, g, p, r)
            print(f"{stage:7s} N={n} grid={int(g)} params={int(p)} rays={int(r)}: fwd {f*1e3:8.1f} us   bwd-kernel {b*1e3:8.1f} us", flush=True)
