"""TEST INFRASTRUCTURE (measurement script; builds its inputs with tests/scene_util.py, i.e. with oracle helpers).
Per-phase breakdown of the split backward's dX kernel from s_memtime stamps (Dbg::stamp in nsr_bwd2.h, 100 MHz ticks).

Needs the instrumented build (tools/build_ts.sh -> nice_slam_amd/_ab/libnsr_ts.so, not part of the product):
    sh tools/build_ts.sh && NSR_LIB_PATH=$PWD/nice_slam_amd/_ab/libnsr_ts.so python tests/perf/ts_dx.py [n_rays] [stage]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch, numpy as np
from scene_util import make_scene, build_product
import nice_slam_amd as nsa
dev = torch.device("cuda", 0)
n_rays = int(sys.argv[1]) if len(sys.argv) > 1 else 1000
stage = sys.argv[2] if len(sys.argv) > 2 else "color"
sc = make_scene(seed=0, n_rays=n_rays, scene="replica_room0", fine_scale=1.0, zero_frac=0.01, depth_range=(1.0, 4.0))
renderer, dec, grids = build_product(sc, dev)
grids = {k: v.requires_grad_(True) for k, v in grids.items()}
for p in dec.parameters(): p.requires_grad_(True)
NB, NW, NS = 3 * 256, 12, 64
buf = torch.zeros((NB * NW * NS,), dtype=torch.int64, device=dev)
frames = [(sc["c2w"].to(dev), sc["depth_img"].to(dev), sc["color_img"].to(dev)) for _ in range(5)]
for it in range(3):
    if it == 2: os.environ["NSR_DBG_PTR"] = hex(buf.data_ptr())
    for g in grids.values(): g.grad = None
    for p in dec.parameters(): p.grad = None
    nsa.mapping_loss(renderer, grids, dec, frames, n_rays // 5, stage).backward()
    torch.cuda.synchronize()
t = buf.cpu().numpy().reshape(NB, NW, NS).astype(np.float64) / 2100.0           # microseconds (s_memtime ticks at the ~2.1 GHz the stamps were calibrated to against rocprofv3 kernel times)
ok = (t[:, :, 0] > 0) & (t[:, :, 9] > 0)
t0 = t[:, :, 0][ok].min()
print(f"{n_rays} rays, stage {stage}: {int(ok.any(1).sum())} blocks, {int(ok.sum())} waves; kernel span (first entry -> last exit) "
      f"{t[:, :, 9][ok].max() - t0:.1f} us")
names = {1: "entry -> operands staged (barrier)", 2: "-> last tile: inputs of the next requested", 3: "   output layer + 5 layers (+dY stores)",
         4: "   embedding backward", 5: "   level + coord grad + wait for next inputs", 6: "   grid scatter issued", 7: "   ray gradients",
         8: "-> loop exit", 9: "-> d _B partials + exit"}
print("   %-48s %8s %8s %8s" % ("wave entry after kernel start", f"{(t[:, :, 0][ok] - t0).mean():.1f}", f"{np.percentile(t[:, :, 0][ok] - t0, 10):.1f}",
                                f"{np.percentile(t[:, :, 0][ok] - t0, 90):.1f}"))
for s in range(1, 10):
    d = (t[:, :, s] - t[:, :, s - 1])[ok & (t[:, :, s] > 0) & (t[:, :, s - 1] > 0)]
    if d.size:
        print("   %-48s %8.2f %8.2f %8.2f   (mean / p10 / p90 us)" % (names[s], d.mean(), np.percentile(d, 10), np.percentile(d, 90)))
print("   absolute times after the first wave's entry (mean / p10 / p90 / max us):")
for s, nm in ((0, "wave entry"), (1, "operands staged"), (2, "last tile: requests issued"), (3, "last tile: layers done"), (4, "last tile: embedding done"),
              (5, "last tile: level done, next inputs landed"), (6, "last tile: scatter issued"), (7, "last tile: ray gradients"), (8, "loop exit (incl. deferred walk)"),
              (10, "block barrier before the hot-table flush"), (11, "hot table flushed"), (9, "d _B partials written, exit")):
    m = ok & (t[:, :, s] > 0)
    if m.any():
        d = t[:, :, s][m] - t0
        print("      %-46s %8.2f %8.2f %8.2f %8.2f" % (nm, d.mean(), np.percentile(d, 10), np.percentile(d, 90), d.max()))
npass = {"coarse": 1, "middle": 1, "fine": 2, "color": 3}[stage]
per = int(ok.any(1).sum()) // npass
for p_ in range(npass):                                      # blocks [p * per, (p + 1) * per): decoder pass p (grid.y)
    sl = slice(p_ * per, (p_ + 1) * per)
    o2 = ok[sl]
    lf = (t[sl, :, 9] - t[sl, :, 0])[o2]
    print(f"   pass {p_}: {per} blocks, wave lifetime mean {lf.mean():.1f} max {lf.max():.1f} us; last exit after kernel start {(t[sl, :, 9][o2] - t0).max():.1f} us")
life = (t[:, :, 9] - t[:, :, 0])[ok]
print(f"   wave lifetime mean {life.mean():.1f} us, max {life.max():.1f}; exit after kernel start mean {(t[:, :, 9][ok] - t0).mean():.1f}, max {(t[:, :, 9][ok] - t0).max():.1f}")
# per-phase totals of a wave over ALL its tiles (Dbg::stamp accumulates slot + 16, counts in slot + 32), per decoder pass
print("   per-phase totals per wave over all its tiles (mean us | mean per passage | passages per wave):")
for p_ in range(npass):
    sl = slice(p_ * per, (p_ + 1) * per)
    o2 = ok[sl]
    print(f"   pass {p_}:")
    for s in range(2, 8):
        tot = (t[sl, :, 16 + s])[o2]
        cnt = (buf.cpu().numpy().reshape(NB, NW, NS)[sl, :, 32 + s])[o2].astype(np.float64)
        if cnt.sum() > 0:
            print("      %-46s %8.2f %8.2f %6.2f" % (names[s], tot.mean(), tot.sum() / cnt.sum(), cnt.mean()))
