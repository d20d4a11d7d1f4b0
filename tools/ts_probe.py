import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch, numpy as np
from scene_util import make_scene, build_product
dev = torch.device("cuda", 0)
sc = make_scene(seed=0, n_rays=1000, scene="replica_room0", fine_scale=1.0, zero_frac=0.01, depth_range=(1.0, 4.0))
renderer, dec, grids = build_product(sc, dev)
grids = {k: v.requires_grad_(True) for k, v in grids.items()}
for p in dec.parameters(): p.requires_grad_(True)
o = sc["rays_o"].to(dev); d = sc["rays_d"].to(dev); gd = sc["gt_depth"].to(dev); gc = sc["gt_color"].to(dev)
NB = 3 * 512
buf = torch.zeros((NB * 8 * 32,), dtype=torch.int64, device=dev)
for it in range(3):
    if it == 2: os.environ["NSR_DBG_PTR"] = hex(buf.data_ptr())
    for g in grids.values(): g.grad = None
    depth, unc, col = renderer.render_batch_ray(grids, dec, d, o, dev, "color", gt_depth=gd)
    ((gd - depth).abs().sum() + 0.2 * (gc - col).abs().sum()).backward()
    torch.cuda.synchronize()
t = buf.cpu().numpy().reshape(NB, 8, 32)
names = {0: "start", 1: "z done", 2: "setup+compositor+sync", 3: "draw read", 4: "fwd recompute", 5: "out layer(+own_out)", 6: "L4", 7: "L3", 8: "L2", 9: "L1", 10: "L0", 11: "E-stage", 12: "dB owners", 20: "mlp done", 21: "scatter", 22: "end sync"}
order = [0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11, 12, 20, 21, 22]
nblk = 256
for p_, nm in ((0, "middle"), (1, "fine"), (2, "color")):
    blk = t[p_ * nblk:(p_ + 1) * nblk, :6, :]
    ok = blk[:, :, 0] > 0
    print("pass", nm, "waves with data", ok.sum())
    prev = None
    for s in order:
        v = blk[:, :, s][ok]
        if prev is not None:
            dlt = (v - pv)
            print("   %-22s mean %8.0f  p90 %8.0f cycles" % (names[s], dlt.mean(), np.percentile(dlt, 90)))
        prev = s; pv = v
    tot = (blk[:, :, 22][ok] - blk[:, :, 0][ok])
    print("   total per group %8.0f cycles (p90 %8.0f)" % (tot.mean(), np.percentile(tot, 90)))
    l3 = blk[:, :, 13:17][ok]
    print("   L3 detail: stage->bar %6.0f | owners %6.0f | 2nd bar wait %6.0f" % ((l3[:,1]-l3[:,0]).mean(), (l3[:,2]-l3[:,1]).mean(), (l3[:,3]-l3[:,2]).mean()))
    l2 = blk[:, :, 23:27][ok]
    print("   L2 detail: stage->bar %6.0f | owners %6.0f (min %6.0f max %6.0f) | 2nd bar wait %6.0f" % ((l2[:,1]-l2[:,0]).mean(), (l2[:,2]-l2[:,1]).mean(), (l2[:,2]-l2[:,1]).min(), (l2[:,2]-l2[:,1]).max(), (l2[:,3]-l2[:,2]).mean()))
