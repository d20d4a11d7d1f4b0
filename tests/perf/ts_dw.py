"""TEST INFRASTRUCTURE (measurement script; builds its inputs with tests/scene_util.py, i.e. with oracle helpers).
Where the waves of the split backward's dW kernel spend their time, from s_memtime stamps (Dbg::stamp in dw_compute / dw_loader,
nsr_bwd2.h): compute waves -- waiting for a tile to land / LDS reads + MFMAs / publishing; loader waves -- waiting for a free ring
slot / issuing the DMA pieces / waiting for the previous tile's pieces.

Needs the instrumented build (tools/build_ts.sh -> nice_slam_amd/_ab/libnsr_ts.so, not part of the product):
    sh tools/build_ts.sh && NSR_LIB_PATH=$PWD/nice_slam_amd/_ab/libnsr_ts.so python tests/perf/ts_dw.py [n_rays] [stage]"""
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
buf = torch.zeros((2 * NB * NW * NS,), dtype=torch.int64, device=dev)       # dX kernel's slots | dW kernel's slots
frames = [(sc["c2w"].to(dev), sc["depth_img"].to(dev), sc["color_img"].to(dev)) for _ in range(5)]
for it in range(3):
    if it == 2: os.environ["NSR_DBG_PTR"] = hex(buf.data_ptr())
    for g in grids.values(): g.grad = None
    for p in dec.parameters(): p.grad = None
    nsa.mapping_loss(renderer, grids, dec, frames, n_rays // 5, stage).backward()
    torch.cuda.synchronize()
raw = buf.cpu().numpy()[NB * NW * NS:].reshape(NB, NW, NS)
t = raw.astype(np.float64) / 2100.0          # microseconds (see ts_dx.py)
comp, load = t[:, :8], t[:, 8:10]
okc = (comp[:, :, 0] > 0) & (comp[:, :, 8] > 0)
t0 = comp[:, :, 0][okc].min()
print(f"{n_rays} rays, stage {stage}: {int(okc.any(1).sum())} dW blocks; kernel span (first entry -> last exit) {comp[:, :, 8][okc].max() - t0:.1f} us")
cn = {1: "flag store -> next loop top", 2: "waiting for the tile to land", 3: "LDS reads + sines + MFMAs", 4: "publish (flag store)"}
print("   compute waves, totals per wave over all its tiles (mean us | mean per passage | passages):")
for s in range(1, 5):
    tot = comp[:, :, 16 + s][okc]
    cnt = raw[:, :8, 32 + s][okc].astype(np.float64)
    if cnt.sum() > 0:
        print("      %-40s %8.2f %8.3f %6.1f" % (cn[s], tot.mean(), tot.sum() / cnt.sum(), cnt.mean()))
tot9, cnt9 = comp[:, :, 16 + 9][okc], raw[:, :8, 32 + 9][okc].astype(np.float64)
if cnt9.sum() > 0:
    print("      %-40s %8.2f %8.3f %6.1f   (of the line above; slot 3 then holds sines + MFMAs only)" % ("   operand reads (LDS) until they have landed", tot9.mean(), tot9.sum() / cnt9.sum(), cnt9.mean()))
for a, b, nm in ((0, 1, "entry -> first loop top"), (5, 6, "loop exit -> every wave done (barrier)"), (6, 7, "W^T products + barrier"), (7, 8, "image stores")):
    d = (comp[:, :, b] - comp[:, :, a])[okc]
    print("      %-40s %8.2f   (p10 %.2f p90 %.2f)" % (nm, d.mean(), np.percentile(d, 10), np.percentile(d, 90)))
d = (comp[:, :, 5] - comp[:, :, 0])[okc]
print("      %-40s %8.2f   (p10 %.2f p90 %.2f)" % ("entry -> loop exit", d.mean(), np.percentile(d, 10), np.percentile(d, 90)))
# per role (wave index): who is the slowest
for w in range(8):
    o = okc[:, w]
    if o.any():
        print("      wave %d: wait %.2f work %.2f per tile" % (w, comp[:, w, 16 + 2][o].sum() / max(1, raw[:, w, 32 + 2][o].sum()),
                                                              comp[:, w, 16 + 3][o].sum() / max(1, raw[:, w, 32 + 3][o].sum())))
okl = (load[:, :, 0] > 0) & (load[:, :, 5] > 0)
ln = {1: "loop top", 2: "waiting for a free ring slot", 3: "issuing the DMA pieces", 4: "waiting for the previous tile's pieces + flag"}
print("   loader waves, totals per wave (mean us | mean per passage | passages):")
for s in range(1, 5):
    tot = load[:, :, 16 + s][okl]
    cnt = raw[:, 8:10, 32 + s][okl].astype(np.float64)
    if cnt.sum() > 0:
        print("      %-40s %8.2f %8.3f %6.1f" % (ln[s], tot.mean(), tot.sum() / cnt.sum(), cnt.mean()))
