#!/usr/bin/env python3
"""Measurement tooling (build container only: needs /root/reference).  bench.py's `cpu_baseline` leg times the CPU ORACLE
(kind "port": oracle/nice_oracle.py in grid_sample mode) because the reference tree does not exist on the GPU box.  This script
times the reference's OWN modules (src/utils/Renderer.Renderer.render_batch_ray + src/conv_onet NICE decoders, with the
arithmetic-neutral CPU stubs of tests/golden/make_golden_callers.py) and the port on the same inputs -- Replica room0 shapes,
colour-stage forward + backward, the mapper's L1 loss -- so that the port-vs-reference ratio is known.

    PYTHONDONTWRITEBYTECODE=1 python tools/calibrate_cpu_baseline.py [n_rays] [threads]
"""
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
import make_golden_callers as mk                                # noqa: E402  (imports the reference with its CPU stubs)
from oracle import nice_oracle as orc                           # noqa: E402
from scene_util import make_scene                               # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 1000
threads = int(sys.argv[2]) if len(sys.argv) > 2 else min(16, os.cpu_count() or 1)
torch.set_num_threads(threads)
sc = make_scene(seed=3, n_rays=n, scene="replica_room0", fine_scale=1.0)
H, W, fx, fy, cx, cy = sc["intr"]
cfg, _, _, _, _ = mk.build()
cfg["rendering"].update({"N_samples": 32, "N_surface": 16, "N_importance": 0, "lindisp": False, "perturb": 0.0})
dec = mk.PatchedNICE(dim=3, c_dim=32, coarse=True, coarse_grid_len=2.0, middle_grid_len=0.32, fine_grid_len=0.16, color_grid_len=0.16,
                     hidden_size=32, pos_embedding_method="fourier")
dec.load_state_dict(sc["params"])
bound = sc["bound"]
dec.bound = bound
dec.middle_decoder.bound = dec.fine_decoder.bound = dec.color_decoder.bound = bound
dec.coarse_decoder.bound = bound * 2.0
import types
slam = types.SimpleNamespace(nice=True, bound=bound, H=H, W=W, fx=fx, fy=fy, cx=cx, cy=cy)
ref_renderer = mk.Renderer(cfg, None, slam)
o, d, gd, gc = sc["rays_o"], sc["rays_d"], sc["gt_depth"], sc["gt_color"] if "gt_color" in sc else torch.rand((n, 3))


def loss_of(depth, color, stage):
    ls = torch.abs(gd[gd > 0] - depth[gd > 0]).sum()
    return ls + 0.2 * torch.abs(gc - color).sum() if stage == "color" else ls


def ref_once(stage):
    G = {k: v.clone().requires_grad_(True) for k, v in sc["grids"].items()}
    for p in dec.parameters():
        p.grad = None
    depth, _, color = ref_renderer.render_batch_ray(G, dec, d, o, "cpu", stage, gt_depth=gd)
    loss_of(depth, color, stage).backward()


def port_once(stage):
    G = {k: v.clone().requires_grad_(True) for k, v in sc["grids"].items()}
    P = {k: v.clone().requires_grad_(True) for k, v in sc["params"].items()}
    depth, _, color = orc.render_batch_ray(G, P, d, o, stage, gd, bound)
    loss_of(depth, color, stage).backward()


orc.TRILINEAR_IMPL = "grid_sample"
print(f"Replica room0 shapes, {n} rays, fwd + bwd, {threads} threads, median of 3 (ms): reference modules vs oracle port")
for stage in ("middle", "fine", "color"):
    res = {}
    for name, fn in (("reference", ref_once), ("port", port_once)):
        fn(stage)
        ts = []
        for _ in range(3):
            t0 = time.perf_counter()
            fn(stage)
            ts.append(time.perf_counter() - t0)
        res[name] = sorted(ts)[1] * 1e3
    print(f"  {stage:7s} reference {res['reference']:8.1f}   port {res['port']:8.1f}   port / reference = {res['port'] / res['reference']:.3f}")
