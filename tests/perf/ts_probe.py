"""TEST INFRASTRUCTURE (measurement script; builds its inputs with tests/scene_util.py, i.e. with oracle helpers).
Per-phase cycle breakdown of the backward kernel from s_memtime stamps.

Needs an INSTRUMENTED build of libnsr (not part of the product): a scratch copy of nice_slam_amd/csrc with a `long long *dbg`
field in RenderParams (set from the environment variable NSR_DBG_PTR in nsr_render_bwd) and TS(O, slot) stamps at the
phase boundaries of bwd_pass / mlp_xyz_bwd / XyzBwd::layer (slot map below); point NSR_LIB_PATH at it.
    NSR_LIB_PATH=$PWD/nice_slam_amd/_ab/libnsr_ts.so python tests/perf/ts_probe.py [n_rays]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch, numpy as np
from scene_util import make_scene, build_product
dev = torch.device("cuda", 0)
n_rays = int(sys.argv[1]) if len(sys.argv) > 1 else 1000
sc = make_scene(seed=0, n_rays=n_rays, scene="replica_room0", fine_scale=1.0, zero_frac=0.01, depth_range=(1.0, 4.0))
renderer, dec, grids = build_product(sc, dev)
grids = {k: v.requires_grad_(True) for k, v in grids.items()}
for p in dec.parameters(): p.requires_grad_(True)
o = sc["rays_o"].to(dev); d = sc["rays_d"].to(dev); gd = sc["gt_depth"].to(dev); gc = sc["gt_color"].to(dev)
NBX, NSLOT = 256, 48
buf = torch.zeros((3 * NBX * 8 * NSLOT,), dtype=torch.int64, device=dev)
for it in range(3):
    if it == 2: os.environ["NSR_DBG_PTR"] = hex(buf.data_ptr())
    for g in grids.values(): g.grad = None
    depth, unc, col = renderer.render_batch_ray(grids, dec, d, o, dev, "color", gt_depth=gd)
    ((gd - depth).abs().sum() + 0.2 * (gc - col).abs().sum()).backward()
    torch.cuda.synchronize()
t = buf.cpu().numpy().reshape(3, NBX, 8, NSLOT)
names = {0: "group start", 1: "z loaded", 2: "gather issue + compositor + barrier", 3: "forward re-run", 4: "output layer"}
for I in range(4, -1, -1):
    b = 5 + (4 - I) * 5
    names.update({b: f"L{I} dX(fc_c) + staging", b + 1: f"L{I} barrier 1 wait", b + 2: f"L{I} owner tasks", b + 3: f"L{I} barrier 2 wait", b + 4: f"L{I} dX(hidden)"})
names.update({30: "embedding stage", 31: "dB barrier 1", 32: "dB owners", 33: "dB barrier 2", 34: "coord grad", 35: "grid scatter", 36: "tail (out-layer image, barrier)"})
order = list(range(0, 37))
for p_, nm in ((0, "middle"), (1, "fine"), (2, "color")):
    blk = t[p_][:, :6, :]
    ok = (blk[:, :, 0] > 0) & (blk[:, :, 36] > 0)
    print(f"pass {nm}: waves with data {ok.sum()}")
    tot = (blk[:, :, 36] - blk[:, :, 0])[ok]
    groups = {}
    prev = None
    for s_ in order:
        v = blk[:, :, s_][ok]
        if prev is not None:
            dlt = v - pv
            print("   %-38s mean %8.0f  p10 %8.0f  p90 %8.0f   %5.1f %%" % (names[s_], dlt.mean(), np.percentile(dlt, 10), np.percentile(dlt, 90), 100 * dlt.mean() / tot.mean()))
            key = "barrier waits" if "barrier" in names[s_] and "wait" in names[s_] or names[s_].startswith("dB barrier") else ("owner tasks" if "owner" in names[s_] else ("staging + dX" if names[s_].startswith("L") else names[s_]))
            groups[key] = groups.get(key, 0.0) + dlt.mean()
        prev = s_; pv = v
    print("   total per ray group %8.0f cycles (p90 %8.0f)" % (tot.mean(), np.percentile(tot, 90)))
    print("   summary: " + ", ".join(f"{k} {100 * v / tot.mean():.1f} %" for k, v in sorted(groups.items(), key=lambda kv: -kv[1])))
