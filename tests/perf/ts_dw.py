"""TEST INFRASTRUCTURE (measurement script).  Per-phase cycle totals of the dW kernel's waves (nsr_bwd2.h, -DNSR_TS build):
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
frames = [(sc["c2w"].to(dev), sc["depth_img"].to(dev), sc["color_img"].to(dev)) for _ in range(5)]
NB, NW = 3 * 512, 16
buf = torch.zeros((NB * NW * 8,), dtype=torch.int64, device=dev)
for it in range(3):
    if it == 2: os.environ["NSR_DBG_PTR"] = hex(buf.data_ptr())
    for g in grids.values(): g.grad = None
    for p in dec.parameters(): p.grad = None
    nsa.mapping_loss(renderer, grids, dec, frames, n_rays // 5, stage).backward()
    torch.cuda.synchronize()
t = buf.cpu().numpy().reshape(NB, NW, 8)
ok = t[:, 0, 7] > 0
t = t[ok]
print(f"{t.shape[0]} blocks; cycles per wave over the kernel (mean over blocks); columns: dma_wait, barrier, issue, fetch, mma, tail-wait, exchange+flush, total")
for w in range(NW):
    m = t[:, w, :].mean(0)
    print("wave %2d (role %d half %d): " % (w, w & 7, w >> 3) + " ".join("%8.0f" % v for v in m))
print("all waves: " + " ".join("%8.0f" % v for v in t.mean((0, 1))))
