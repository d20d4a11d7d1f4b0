"""TEST INFRASTRUCTURE (measurement script; builds its inputs with tests/scene_util.py, i.e. with oracle helpers).
Per-phase breakdown of the forward kernel from s_memtime stamps (Dbg::stamp in nsr_kernels.h).

Needs the instrumented build (tools/build_ts.sh -> nice_slam_amd/_ab/libnsr_ts.so, not part of the product):
    sh tools/build_ts.sh && NSR_LIB_PATH=$PWD/nice_slam_amd/_ab/libnsr_ts.so python tests/perf/ts_fwd.py [n_rays] [stage]"""
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
NB, NW, NS = 1024, 12, 64
SPLIT = os.environ.get("NSR_FWD_SPLIT", "1") != "0"         # the three-launch forward (nsr_fwd2.h): stamps of its pass kernel
buf = torch.zeros((NB * NW * NS,), dtype=torch.int64, device=dev)
frames = [(sc["c2w"].to(dev), sc["depth_img"].to(dev), sc["color_img"].to(dev)) for _ in range(5)]
for it in range(3):
    if it == 2: os.environ["NSR_DBG_FWD_PTR"] = hex(buf.data_ptr())
    for g in grids.values(): g.grad = None
    for p in dec.parameters(): p.grad = None
    nsa.mapping_loss(renderer, grids, dec, frames, n_rays // 5, stage).backward()
    torch.cuda.synchronize()
t = buf.cpu().numpy().reshape(NB, NW, NS).astype(np.float64) / 2100.0           # microseconds at ~2.1 GHz
ok = (t[:, :, 0] > 0) & (t[:, :, 9] > 0)
t0 = t[:, :, 0][ok].min()
print(f"{n_rays} rays, stage {stage}: {int(ok.any(1).sum())} blocks, {int(ok.sum())} waves; kernel span {t[:, :, 9][ok].max() - t0:.1f} us; "
      f"wave entry after kernel start mean {(t[:, :, 0][ok] - t0).mean():.1f} us (p90 {np.percentile(t[:, :, 0][ok] - t0, 90):.1f})")
if SPLIT:
    for a, b, nm in ((0, 1, "entry -> aux + stream staged (barrier)"), (1, 2, "-> last tile: position loaded"), (2, 3, "   gather + decoder + stores"), (3, 9, "-> exit")):
        m = ok & (t[:, :, a] > 0) & (t[:, :, b] > 0)
        d = (t[:, :, b] - t[:, :, a])[m]
        print("   %-52s %7.2f %7.2f %7.2f   (mean / p10 / p90 us)" % (nm, d.mean(), np.percentile(d, 10), np.percentile(d, 90)))
    life = (t[:, :, 9] - t[:, :, 0])[ok]
    print(f"   wave lifetime mean {life.mean():.1f} us, max {life.max():.1f}")
    nblk = int(ok.any(1).sum())
    npass = {"coarse": 1, "middle": 1, "fine": 2, "color": 3}[stage]
    per = nblk // npass
    for p_ in range(npass):                                  # blocks [p * per, (p + 1) * per) belong to decoder pass p
        sl = slice(p_ * per, (p_ + 1) * per)
        o2 = ok[sl]
        lf = (t[sl, :, 9] - t[sl, :, 0])[o2]
        tl = (t[sl, :, 3] - t[sl, :, 2])[o2 & (t[sl, :, 3] > 0)]
        print(f"   pass {p_}: {per} blocks, wave lifetime mean {lf.mean():.1f} max {lf.max():.1f} us; last tile (gather + decoder + stores) mean {tl.mean():.1f} p90 {np.percentile(tl, 90):.1f} us; "
              f"exit after kernel start max {(t[sl, :, 9][o2] - t0).max():.1f} us")
    sys.exit(0)
seq = [(0, 1, "entry -> aux / stream staging issued"), (1, 2, "sample placement (compute_z, barrier)"), (2, 10, "positions, feature gathers issued"),
       (10, 3, "middle decoder"), (3, 4, "fine: gather + barrier + stream staging + barrier"), (4, 5, "fine decoder"),
       (5, 11, "colour: gather + barrier + stream staging + barrier"), (11, 6, "colour decoder"), (3, 6, "(middle stage: -> decode done)"),
       (6, 7, "raw stores + barrier"), (7, 8, "compositor + loss epilogue (d raw, positions)"), (8, 9, "barrier + exit")]
for a, b, nm in seq:
    m = ok & (t[:, :, a] > 0) & (t[:, :, b] > 0)
    if stage != "middle" and nm.startswith("(middle"):
        continue
    if m.any():
        d = (t[:, :, b] - t[:, :, a])[m]
        print("   %-52s %7.2f %7.2f %7.2f   (mean / p10 / p90 us)" % (nm, d.mean(), np.percentile(d, 10), np.percentile(d, 90)))
life = (t[:, :, 9] - t[:, :, 0])[ok]
print(f"   wave lifetime mean {life.mean():.1f} us, max {life.max():.1f}")
