"""Absolute trajectory error after a closed-form rigid alignment.

Restates ``align`` / the RMSE of ``evaluate_ate`` of the reference's evaluation script (src/tools/eval_ate.py:44-78,
147-223; Horn's method via the SVD of the 3x3 correlation matrix) on plain ndarrays.  Pinned against the reference's
own ``align`` by tests/golden/ate_golden.npz (tests/golden/make_golden_ate.py, tests/test_slam_synthetic.py)."""
from __future__ import annotations

import numpy as np


def align(model: np.ndarray, data: np.ndarray):
    """model, data: (3, n) trajectories.  Returns rot (3,3), trans (3,1), per-point translational error (n,) of
    ``rot @ model + trans`` against ``data`` (eval_ate.py:44-78)."""
    model = np.asarray(model, dtype=np.float64)
    data = np.asarray(data, dtype=np.float64)
    mz = model - model.mean(1, keepdims=True)
    dz = data - data.mean(1, keepdims=True)
    W = mz @ dz.T                                      # sum of outer(model_i, data_i)
    U, _, Vh = np.linalg.svd(W.T)
    S = np.eye(3)
    if np.linalg.det(U) * np.linalg.det(Vh) < 0:
        S[2, 2] = -1.0
    rot = U @ S @ Vh
    trans = data.mean(1, keepdims=True) - rot @ model.mean(1, keepdims=True)
    err = rot @ model + trans - data
    return rot, trans, np.sqrt((err * err).sum(0))


def ate_rmse(est_c2w, gt_c2w) -> dict:
    """est_c2w, gt_c2w: sequences of (>=3, 4) camera-to-world matrices with matching indices.  Mirrors evaluate_ate with
    identical timestamps: the estimate is aligned to the ground truth (eval_ate.py:166)."""
    est = np.stack([np.asarray(c, dtype=np.float64)[:3, 3] for c in est_c2w], 1)
    gt = np.stack([np.asarray(c, dtype=np.float64)[:3, 3] for c in gt_c2w], 1)
    _, _, e = align(est, gt)
    return {"compared_pose_pairs": int(e.shape[0]), "rmse": float(np.sqrt(np.dot(e, e) / len(e))), "mean": float(e.mean()),
            "median": float(np.median(e)), "std": float(e.std()), "min": float(e.min()), "max": float(e.max())}
