"""GPU: frustum feature selection and its use by the masked multi-GPU gradient exchange.

nsr_frustum_mask against oracle/frustum_oracle.py (the numpy restatement of Mapper.get_mask_from_c2w, Mapper.py:93-164;
unpinned -- cv2 is absent) must agree bit for bit; the masked exchange of ShardedRenderer runs here through the real HIP
renderer on a single-rank RCCL group (the 2-rank logic is covered on CPU by tests/test_dist_gloo.py)."""
import numpy as np
import pytest
import torch

from scene_util import build_product, frustum_case, hip_render, make_scene, rel_err
from oracle import frustum_oracle as fo

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def _selector(fc):
    import nice_slam_amd as nsa
    return nsa.FrustumSelector(fc["bound"], fc["H"], fc["W"], fc["fx"], fc["fy"], fc["cx"], fc["cy"])


def _oracle(fc, key="grid_fine"):
    return fo.get_mask_from_c2w(fc["c2w"], key, fc["shape"], fc["depth"], fc["bound"], fc["H"], fc["W"],
                                fc["fx"], fc["fy"], fc["cx"], fc["cy"])


@pytest.mark.parametrize("seed", [0, 1, 2, 3])
def test_frustum_mask_small(seed):
    fc = frustum_case(seed)
    sel = _selector(fc)
    depth = torch.from_numpy(fc["depth"]).to(DEV)
    got = sel.get_mask_from_c2w(torch.from_numpy(fc["c2w"]), "grid_middle", fc["shape"], depth)
    ref = _oracle(fc)
    assert got.dtype == torch.bool and tuple(got.shape) == ref.shape
    assert np.array_equal(got.cpu().numpy(), ref)
    vm = sel.voxel_mask(fc["c2w"], "grid_middle", fc["shape"], depth)
    assert vm.dtype == torch.uint8 and vm.is_contiguous() and np.array_equal(vm.cpu().numpy().astype(bool), ref.transpose(2, 1, 0))
    coarse = sel.get_mask_from_c2w(fc["c2w"], "grid_coarse", (3, 4, 5), depth)        # Mapper.py:116-118
    assert tuple(coarse.shape) == (5, 4, 3) and bool(coarse.all())


def test_frustum_mask_replica_fine_grid():
    """Replica room0: 680x1200 frame, fine grid 43x56x74 (tests/golden/scene_shapes.json), Replica intrinsics."""
    import json, os
    shapes = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "scene_shapes.json")))
    r0 = shapes["configs/Replica/room0.yaml"]
    bound = np.array(r0["bound"], dtype=np.float64)
    shape = tuple(r0["shapes"]["grid_fine"])
    assert shape == (43, 56, 74)
    fc = frustum_case(11, H=680, W=1200, shape=shape, bound=bound)
    fc.update(fx=600.0, fy=600.0, cx=599.5, cy=339.5)
    sel = _selector(fc)
    got = sel.get_mask_from_c2w(fc["c2w"], "grid_fine", shape, torch.from_numpy(fc["depth"]).to(DEV))
    ref = _oracle(fc)
    assert 0.002 < ref.mean() < 0.9 and ref.sum() > 200          # a ~2 m deep frustum in a 714 m^3 bound: ~1 % of the voxels
    assert np.array_equal(got.cpu().numpy(), ref)


@pytest.mark.parametrize("pre", ["map/", "ba/"])
def test_frustum_mask_against_the_real_mappers_masks(pre):
    """The masks recorded from the REAL Mapper.get_mask_from_c2w (src/Mapper.py:93-164; only ``cv2.remap`` stubbed by the
    restatement when the fixture was minted): FrustumSelector must return the same voxels.  Pins numpy's ``w2c @ p``, the
    projection, the depth test and the near-camera sphere; the row stays "partial" for cv2.remap itself."""
    import caller_replay as cr
    import nice_slam_amd as nsa
    gold = cr.load()
    H, W, fx, fy, cx, cy = (float(v) for v in gold["intr"])
    sel = nsa.FrustumSelector(gold["bound"], int(H), int(W), fx, fy, cx, cy)
    depth = torch.from_numpy(np.ascontiguousarray(gold["frame/0/depth"], dtype=np.float32)).to(DEV)
    for key in ("grid_middle", "grid_fine", "grid_color"):
        ref = gold[f"{pre}mask/{key}"].astype(bool)                      # [Z,Y,X]
        vm = sel.voxel_mask(gold[pre + "cur_c2w"], key, ref.shape, depth)
        assert np.array_equal(vm.cpu().numpy().astype(bool), ref), (pre, key)


def test_frustum_mask_errors():
    from nice_slam_amd._capi import NsrError
    fc = frustum_case(0)
    sel = _selector(fc)
    with pytest.raises(NsrError):
        sel.voxel_mask(fc["c2w"], "grid_fine", fc["shape"], torch.from_numpy(fc["depth"]))                 # CPU tensor
    with pytest.raises(NsrError):
        sel.voxel_mask(fc["c2w"], "grid_fine", fc["shape"], torch.zeros((3, 3), device=DEV))               # wrong frame size
    with pytest.raises(NsrError):
        sel.voxel_mask(fc["c2w"][:3], "grid_fine", fc["shape"], torch.from_numpy(fc["depth"]).to(DEV))     # 3x4 pose


def test_masked_exchange_single_rank_group():
    """ShardedRenderer + set_voxel_masks on a 1-rank RCCL group: the packed exchange (compaction, the decoder-gradient
    blob riding along, scatter back) must leave every gradient exactly as the plain renderer produced it."""
    import socket
    import torch.distributed as dist
    from nice_slam_amd.parallel import ShardedRenderer
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    dist.init_process_group("nccl", init_method=f"tcp://127.0.0.1:{port}", rank=0, world_size=1,
                            device_id=torch.device(DEV))
    try:
        sc = make_scene(seed=41, n_rays=64, small=True)
        product = build_product(sc, DEV)
        ref = hip_render(sc, "color", DEV, backward=True, product=product)
        renderer, dec, grids = product
        g = torch.Generator().manual_seed(3)
        masks = {k: (torch.rand(tuple(v.shape[2:]), generator=g) < 0.35).to(DEV) for k, v in grids.items()}
        masks["grid_middle"] = None                                            # this grid is exchanged densely
        sh = ShardedRenderer(renderer)
        sh.set_voxel_masks(masks)
        got = hip_render(sc, "color", DEV, backward=True, product=(sh, dec, grids))
        n_rows = int(masks["grid_fine"].sum()) + int(masks["grid_color"].sum())
        n_par = sum(p.numel() for d in (dec.middle_decoder, dec.fine_decoder, dec.color_decoder) for p in d.parameters())
        assert sh.last_exchange_floats == n_rows * 32 + n_par + grids["grid_middle"].numel()
        assert set(got) == set(ref)
        for k in ref:
            assert torch.equal(got[k], ref[k]) or rel_err(got[k], ref[k]) < 1e-5, k   # atomics: summation order may differ
        sh.set_voxel_masks(None)
        got = hip_render(sc, "fine", DEV, backward=True, product=(sh, dec, grids))
        ref = hip_render(sc, "fine", DEV, backward=True, product=product)
        for k in ref:
            assert rel_err(got[k], ref[k]) < 1e-5, k          # two GPU runs: fp32 atomics add in a different order
    finally:
        dist.destroy_process_group()
