"""SURVEY §8(f) rank 4 on the CPU: the ATE evaluation against the reference's own alignment (golden), the camera
parametrisation, the synthetic sequence generator, and a miniature tracker + mapper run (tools/slam_synthetic.MiniSLAM
with the oracle as the renderer) whose tracking must beat the motion model alone."""
import copy
import os
import sys

import numpy as np
import pytest
import torch

from conftest import GOLDEN, ROOT

sys.path.insert(0, os.path.join(ROOT, "tools"))
import ate  # noqa: E402
import slam_synthetic as ss  # noqa: E402


def test_ate_alignment_matches_reference_golden():
    """tools/ate.align vs src/tools/eval_ate.py:44-78 run by tests/golden/make_golden_ate.py (incl. the det<0 branch)."""
    z = np.load(os.path.join(GOLDEN, "ate_golden.npz"))
    for c in range(4):
        rot, trans, err = ate.align(z[f"c{c}/est"], z[f"c{c}/gt"])
        assert np.allclose(rot, z[f"c{c}/rot"], atol=1e-12) and np.allclose(trans, z[f"c{c}/trans"], atol=1e-12)
        assert np.allclose(err, z[f"c{c}/err"], atol=1e-12)
    est = [np.eye(4) for _ in range(5)]
    gt = [np.eye(4) for _ in range(5)]
    for i in range(5):
        gt[i][:3, 3] = [0.1 * i, 0.02 * i * i, 0.0]
        est[i][:3, 3] = np.array([0.02 * i * i, -0.1 * i, 0.0]) + 3.0         # rotated by 90 deg and shifted: ATE 0
    assert ate.ate_rmse(est, gt)["rmse"] < 1e-12


def test_camera_tensor_round_trip():
    from scipy.spatial.transform import Rotation
    rng = np.random.RandomState(0)
    for _ in range(20):
        c2w = torch.eye(4)
        c2w[:3, :3] = torch.tensor(Rotation.from_rotvec(rng.randn(3)).as_matrix(), dtype=torch.float32)
        c2w[:3, 3] = torch.tensor(rng.randn(3), dtype=torch.float32)
        t = ss.get_tensor_from_camera(c2w)
        assert t.shape == (7,) and abs(float(t[:4].norm()) - 1.0) < 1e-6
        back = ss.get_camera_from_tensor(t)
        assert torch.allclose(back, c2w[:3], atol=2e-6)
        assert torch.allclose(ss.get_camera_from_tensor(t * torch.tensor([2.0] * 4 + [1.0] * 3)), c2w[:3], atol=2e-6)   # common.py:149: scale-free


def test_synthetic_sequence_is_consistent():
    """Depth is z-depth along the reference's ray convention: back-projected pixels land on the room / cuboid surfaces,
    and consecutive frames move by about a centimetre."""
    seq = ss.SyntheticSequence(6, 48, 64)
    color, depth, c2w = seq.frame(3)
    assert color.shape == (48, 64, 3) and depth.shape == (48, 64) and float(depth.min()) > 0.2 and float(depth.max()) < 6.0
    assert 0.0 <= float(color.min()) and float(color.max()) <= 1.0
    p = c2w[:3, 3] + (seq.dirs @ c2w[:3, :3].T) * depth[..., None]
    on_room = ((p - seq.room[:, 0]).abs().min(-1)[0] < 1e-4) | ((p - seq.room[:, 1]).abs().min(-1)[0] < 1e-4)
    on_box = torch.zeros_like(on_room)
    for b in seq.boxes:
        inside = ((p > b[:, 0] - 1e-4) & (p < b[:, 1] + 1e-4)).all(-1)
        face = ((p - b[:, 0]).abs().min(-1)[0] < 1e-4) | ((p - b[:, 1]).abs().min(-1)[0] < 1e-4)
        on_box |= inside & face
    assert bool((on_room | on_box).all()) and bool(on_box.any()) and bool(on_room.any())
    step = [float((seq.poses[i + 1][:3, 3] - seq.poses[i][:3, 3]).norm()) for i in range(5)]
    assert 0.003 < min(step) and max(step) < 0.03


@pytest.mark.timeout(600)
def test_miniature_slam_run_tracks():
    from slam_oracle_ops import OracleOps
    torch.set_num_threads(1)            # bit-reproducible reductions: the run is a chaotic feedback loop, thread-count dependent otherwise
    cfg = copy.deepcopy(ss.DEFAULT_CFG)
    cfg["tracking"].update(ignore_edge_W=4, ignore_edge_H=4, pixels=100, iters=8)
    cfg["mapping"].update(pixels=200, iters_first=150, iters=20, every_frame=4, keyframe_every=4)
    out = {}
    for name, iters in (("tracked", 8), ("motion_model_only", 0)):
        cfg["tracking"]["iters"] = iters
        torch.manual_seed(0)
        seq = ss.SyntheticSequence(13, 48, 64)
        slam = ss.MiniSLAM(OracleOps(seq), seq, cfg)
        out[name] = slam.run()
        assert slam.keyframe_list == [0, 4, 8, 12]
    tr, mm = out["tracked"], out["motion_model_only"]
    assert tr["tracking_iters"] == 12 * 8 and tr["mapping_iters"] == 150 + 3 * 20 and mm["tracking_iters"] == 0
    assert np.isfinite(tr["ate"]["rmse"]) and tr["ate"]["rmse"] < 0.025                  # 1 cm in the pilot run
    assert tr["ate"]["rmse"] < 0.7 * mm["ate"]["rmse"]                                   # pilot: 0.97 cm vs 3.5 cm
    assert tr["raw_translation_error_cm"]["mean"] < mm["raw_translation_error_cm"]["mean"]


def test_host_quaternion_is_scipys():
    """tools/slam_synthetic._cam_np (the fused loop's host-side get_tensor_from_camera, common.py:179-201) against scipy over
    rotations that visit all four branches; _pose_np inverts it."""
    from scipy.spatial.transform import Rotation
    import slam_synthetic as ss
    rng = np.random.RandomState(5)
    seen = set()
    for k in range(400):
        R = Rotation.from_rotvec(rng.randn(3) * (0.05 if k % 2 else 2.5)).as_matrix()
        t = R[0, 0] + R[1, 1] + R[2, 2]
        seen.add(0 if t > 0 else 1 + int(np.argmax(np.diag(R))))
        m = np.eye(4); m[:3, :3] = R; m[:3, 3] = rng.randn(3)
        cam = ss._cam_np(m).numpy().astype(np.float64)
        x, y, z, w = Rotation.from_matrix(R).as_quat()
        ref = np.array([w, x, y, z])
        assert min(np.abs(cam[:4] - ref).max(), np.abs(cam[:4] + ref).max()) < 1e-6
        assert np.abs(ss._pose_np(cam).numpy() - m).max() < 1e-5
        assert np.abs(ss._inv44(torch.from_numpy(m)).numpy() @ m - np.eye(4)).max() < 1e-5
    assert seen == {0, 1, 2, 3}
