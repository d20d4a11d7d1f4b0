#!/usr/bin/env python3
"""TEST INFRASTRUCTURE (measurement script; builds its inputs with tests/scene_util.py, i.e. with oracle helpers).
SURVEY §8(f) rank 1 measurement: the per-iteration grid update of Mapper.optimize_map on Replica room0 shapes.
(A) reference flow: val[mask] = val_grad (x2 per iteration) + torch.optim.Adam on the masked leaves
(B) nice_slam_amd.MaskedGridAdam: one in-place kernel per grid.  Reports time per iteration and, for (B), achieved
HBM GB/s against the algorithmic bytes (896 B per updated voxel + 1 B mask per voxel)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
import nice_slam_amd as nsa
from scene_util import make_scene
dev = torch.device("cuda", 0)
sc = make_scene(seed=0, n_rays=8, scene="replica_room0", fine_scale=1.0)
keys = ("grid_middle", "grid_fine", "grid_color")
g = torch.Generator().manual_seed(0)
frac = 0.5
vm = {k: (torch.rand(sc["grids"][k].shape[2:], generator=g) < frac) for k in keys}
grids = {k: nsa.to_channels_last(sc["grids"][k].to(dev)) for k in keys}
grad = {k: torch.randn_like(grids[k]) * 1e-3 for k in keys}
lrs = {k: 0.005 for k in keys}

def timeit(fn, n=20):
    for _ in range(3): fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n

# (A)
cA = {k: v.clone(memory_format=torch.preserve_format) for k, v in grids.items()}
full = {k: vm[k][None, None].expand_as(cA[k]).to(dev) for k in keys}
leaves = {k: cA[k][full[k]].clone().requires_grad_(True) for k in keys}
opt = torch.optim.Adam([{"params": [leaves[k]], "lr": 0.005} for k in keys])
def ref_iter():
    for k in keys:
        val = cA[k]; val[full[k]] = leaves[k].detach(); cA[k] = val            # Mapper.py:394-401
    for k in keys:
        leaves[k].grad = grad[k][full[k]]                                     # what autograd's index_put backward produces
    opt.step()
    for k in keys:
        val = cA[k]; val[full[k]] = leaves[k].detach().clone(); cA[k] = val    # Mapper.py:511-519
tA = timeit(ref_iter)
# (B)
cB = {k: v.clone(memory_format=torch.preserve_format) for k, v in grids.items()}
fused = nsa.MaskedGridAdam(cB, {k: vm[k] for k in keys})
tB = timeit(lambda: fused.step(lrs, grad))
nvox = sum(int(vm[k].sum()) for k in keys); tot = sum(vm[k].numel() for k in keys)
bytes_alg = nvox * 896 + tot
print(f"reference flow (index_put x2 + torch Adam on masked leaves), 3 grids: {tA*1e3:.3f} ms / iteration")
print(f"nsr MaskedGridAdam, 3 grids ({nvox} of {tot} voxels masked in):          {tB*1e3:.3f} ms / iteration  ({tA/tB:.1f}x)")
print(f"fused kernel: {bytes_alg/1e6:.1f} MB algorithmic per iteration -> {bytes_alg/tB/1e9:.0f} GB/s = {bytes_alg/tB/8e12*100:.1f}% of 8 TB/s HBM peak ({bytes_alg/tB/6.3e12*100:.1f}% of the 6.3 TB/s achievable)")
