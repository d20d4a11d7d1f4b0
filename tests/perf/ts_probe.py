"""TEST INFRASTRUCTURE (measurement script; builds its inputs with tests/scene_util.py, i.e. with oracle helpers).
Per-phase cycle breakdown of the backward kernel from s_memtime stamps (Dbg::stamp in nsr_bwd.h).

Needs the instrumented build (tools/build_ts.sh -> nice_slam_amd/_ab/libnsr_ts.so, not part of the product):
    sh tools/build_ts.sh && NSR_LIB_PATH=$PWD/nice_slam_amd/_ab/libnsr_ts.so python tests/perf/ts_probe.py [n_rays] [stage] [stepped]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch, numpy as np
from scene_util import make_scene, build_product
dev = torch.device("cuda", 0)
n_rays = int(sys.argv[1]) if len(sys.argv) > 1 else 1000
stage = sys.argv[2] if len(sys.argv) > 2 else "color"
stepped = "stepped" in sys.argv[3:]
sc = make_scene(seed=0, n_rays=n_rays, scene="replica_room0", fine_scale=1.0, zero_frac=0.01, depth_range=(1.0, 4.0))
renderer, dec, grids = build_product(sc, dev)
if stepped:
    renderer.decoder_grads = ("color",)
grids = {k: v.requires_grad_(True) for k, v in grids.items()}
for p in dec.parameters(): p.requires_grad_(True)
o = sc["rays_o"].to(dev); d = sc["rays_d"].to(dev); gd = sc["gt_depth"].to(dev); gc = sc["gt_color"].to(dev)
NBX, NW, NSLOT = 256, 4, 64
buf = torch.zeros((3 * NBX * NW * NSLOT,), dtype=torch.int64, device=dev)
fused = "fused" in sys.argv[3:]
import nice_slam_amd as nsa
frames = [(sc["c2w"].to(dev), sc["depth_img"].to(dev), sc["color_img"].to(dev)) for _ in range(5)]
for it in range(3):
    if it == 2: os.environ["NSR_DBG_PTR"] = hex(buf.data_ptr())
    for g in grids.values(): g.grad = None
    for p in dec.parameters(): p.grad = None
    if fused:        # the bench's path: one autograd node, per-ray inputs in two buffers
        nsa.mapping_loss(renderer, grids, dec, frames, n_rays // 5, stage).backward()
    else:
        depth, unc, col = renderer.render_batch_ray(grids, dec, d, o, dev, stage, gt_depth=gd)
        ((gd - depth).abs().sum() + 0.2 * (gc - col).abs().sum()).backward()
    torch.cuda.synchronize()
t_all = buf.cpu().numpy().reshape(3 * NBX, NW, NSLOT)            # one row per block of the launch; slot 61 = decoder pass + 1, 62 = its block count
_ok = (t_all[:, :, 0] > 0) & (t_all[:, :, 7] > 0)
print(f"kernel span (first entry -> last exit over all blocks): {t_all[:, :, 7][_ok].max() - t_all[:, :, 0][_ok].min()} cycles (s_memtime, 100 MHz ticks x ?: see block lifetimes)")
blockn = {0: "entry", 1: "aux + packed weights copy issued", 2: "z loaded (barrier)", 3: "tile-0 setup + compositor", 4: "barrier wait (d raw)",
          5: "all tiles", 6: "barrier wait (tiles)", 7: "ray reduce + flush"}
tilen = ["start (setup / gather wait)", "forward re-run", "output layer", "layer 4", "layer 3", "layer 2", "layer 1", "layer 0 (+W0/W3e)",
         "embedding stage", "dB", "coord grad + scatter"]
for p_, nm in enumerate(("middle", "fine", "color")[:{"middle": 1, "fine": 2, "color": 3}[stage]]):
    blk = t_all[t_all[:, 0, 61] == p_ + 2]
    ok = (blk[:, :, 0] > 0) & (blk[:, :, 7] > 0)
    if not ok.any():
        continue
    tot = (blk[:, :, 7] - blk[:, :, 0])[ok]
    print(f"pass {nm}: {blk.shape[0]} blocks (partition: {int(blk[0, 0, 62])}), block lifetime mean {tot.mean():.0f} cycles (p90 {np.percentile(tot, 90):.0f}, "
          f"max {tot.max():.0f}); per-group stamps below are those of each block's LAST ray group")
    for s_ in range(1, 8):
        dlt = (blk[:, :, s_] - blk[:, :, s_ - 1])[ok]
        print("   %-36s mean %8.0f  p10 %8.0f  p90 %8.0f   %5.1f %%" % (blockn[s_], dlt.mean(), np.percentile(dlt, 10), np.percentile(dlt, 90), 100 * dlt.mean() / tot.mean()))
    if (blk[:, :, 60][ok] > 0).all():
        dlt = (blk[:, :, 60] - blk[:, :, 0])[ok]
        print("      %-32s mean %8.0f  p10 %8.0f  p90 %8.0f" % ("early loads landed", dlt.mean(), np.percentile(dlt, 10), np.percentile(dlt, 90)))
    for a_, b_, nm_ in ((56, 6, "flush: LDS-tile reload"), (57, 56, "flush: barrier"), (58, 57, "flush: vectors + v"), (59, 58, "flush: barrier"), (7, 59, "flush: tile rounds")):
        if (blk[:, :, a_][ok] > 0).all():
            dlt = (blk[:, :, a_] - blk[:, :, b_])[ok]
            print("      %-32s mean %8.0f  p10 %8.0f  p90 %8.0f" % (nm_, dlt.mean(), np.percentile(dlt, 10), np.percentile(dlt, 90)))
    for k in range(3):
        b = 8 + 12 * k
        okk = ok & (blk[:, :, b] > 0) & (blk[:, :, b + 10] > 0)
        if not okk.any():
            continue
        tt = (blk[:, :, b + 10] - blk[:, :, b])[okk]
        print(f"   tile {k}: {tt.mean():.0f} cycles")
        prev = 4 if k == 0 else 8 + 12 * (k - 1) + 10
        for j in range(0, 11):
            cur = b + j
            if not (blk[:, :, cur][okk] > 0).all():
                continue
            dlt = (blk[:, :, cur] - blk[:, :, prev])[okk]
            print("      %-32s mean %8.0f  p10 %8.0f  p90 %8.0f" % (tilen[j], dlt.mean(), np.percentile(dlt, 10), np.percentile(dlt, 90)))
            prev = cur
