"""GPU: the reference's process architecture on the drop-in objects.  NICE_SLAM builds ONE set of feature grids, decoders
and a Renderer, calls ``share_memory_()`` / ``share_memory()`` on them and hands them to three spawned processes (tracker,
mapper, coarse mapper: src/NICE_SLAM.py:70-91,288-305); the mapper optimises grids and decoders IN PLACE and the tracker
picks the new values up with ``copy.deepcopy(self.shared_decoders).to(device)`` and ``val.clone()``
(src/Tracker.py:130-142).  Here: a spawned "mapper" process takes a mapping step on the pickled objects (CUDA IPC), a spawned
"tracker" process then copies and renders; the parent's own handles must show the same updated state."""
import copy
import os
import sys

import numpy as np
import pytest
import torch
import torch.multiprocessing as mp

from conftest import ROOT

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def _setup_path():
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))


def _mapper(renderer, dec, c, rays, flag):
    """one colour-stage mapping iteration on the SHARED objects: grids through MaskedGridAdam, colour decoder through Adam"""
    _setup_path()
    import nice_slam_amd as nsa
    o, d, gd, gc = rays
    keys = ("grid_middle", "grid_fine", "grid_color")
    for k in keys:
        c[k].requires_grad_(True)
    for p in dec.parameters():
        p.requires_grad_(True)
    opt = torch.optim.Adam(list(dec.color_decoder.parameters()), lr=0.005)
    fused = nsa.MaskedGridAdam({k: c[k] for k in keys})
    depth, _, col = renderer.render_batch_ray(c, dec, d, o, DEV, "color", gt_depth=gd)
    loss = (torch.abs(gd - depth) * (gd > 0)).sum() + 0.2 * torch.abs(gc - col).sum()
    loss.backward()
    opt.step()
    with torch.no_grad():
        fused.step({k: 0.005 for k in keys})
    torch.cuda.synchronize()
    flag[0] = 1


def _tracker(renderer, shared_dec, shared_c, rays, flag, q):
    """Tracker.update_para_from_mapping (src/Tracker.py:130-142) + a render with the private copies"""
    _setup_path()
    import time
    t0 = time.time()
    while int(flag[0]) != 1:
        time.sleep(0.05)
        assert time.time() - t0 < 240
    dec = copy.deepcopy(shared_dec).to(DEV)
    c = {k: v.clone().to(DEV) for k, v in shared_c.items()}
    o, d, gd, _ = rays
    with torch.no_grad():
        depth, unc, col = renderer.render_batch_ray(c, dec, d, o, DEV, "color", gt_depth=gd)
    torch.cuda.synchronize()
    q.put({"depth": depth.cpu().numpy(), "var": unc.cpu().numpy(), "rgb": col.cpu().numpy(),
           "flat_color": dec.color_decoder.flat_params().cpu().numpy()})


def test_spawned_processes_share_grids_and_decoders():
    from scene_util import build_product, make_scene
    sc = make_scene(seed=71, n_rays=128, small=True)
    renderer, dec, c = build_product(sc, DEV)
    for v in c.values():                                        # src/NICE_SLAM.py:82-87
        v.share_memory_()
    dec.share_memory()
    rays = tuple(sc[k].to(DEV) for k in ("rays_o", "rays_d", "gt_depth", "gt_color"))
    with torch.no_grad():
        before = [t.clone() for t in renderer.render_batch_ray(c, dec, rays[1], rays[0], DEV, "color", gt_depth=rays[2])]
        grid_before = c["grid_color"].clone()
        flat_before = dec.color_decoder.flat_params().clone()
    torch.cuda.synchronize()
    flag = torch.zeros(1, dtype=torch.int32).share_memory_()    # like NICE_SLAM.mapping_idx (:76-81)
    ctx = mp.get_context("spawn")                               # src/NICE_SLAM.py:63-66
    q = ctx.Queue()
    pm = ctx.Process(target=_mapper, args=(renderer, dec, c, rays, flag))
    pt = ctx.Process(target=_tracker, args=(renderer, dec, c, rays, flag, q))
    pm.start(); pt.start()
    got = q.get(timeout=300)
    pm.join(timeout=120); pt.join(timeout=120)
    assert pm.exitcode == 0 and pt.exitcode == 0
    # the parent's handles see the mapper's in-place updates ...
    assert float((c["grid_color"] - grid_before).abs().max()) > 1e-4
    flat_now = dec.color_decoder.flat_params()
    assert float((flat_now - flat_before).abs().max()) > 1e-4
    assert all(p.data_ptr() == flat_now.data_ptr() + 4 * off for p, off in zip(dec.color_decoder._views, dec.color_decoder._offsets))
    # ... and renders the same image as the tracker process did from its deep copies
    with torch.no_grad():
        after = renderer.render_batch_ray(c, dec, rays[1], rays[0], DEV, "color", gt_depth=rays[2])
    assert np.array_equal(flat_now.cpu().numpy(), got["flat_color"])
    for a, k in zip(after, ("depth", "var", "rgb")):
        assert np.array_equal(a.cpu().numpy(), got[k]), k
    assert float((after[0] - before[0]).abs().max()) > 0
