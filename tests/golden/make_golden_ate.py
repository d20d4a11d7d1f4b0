"""Mint tests/golden/ate_golden.npz by running the reference's own trajectory alignment (src/tools/eval_ate.py:44-78)
in this container.  Usage (reference tree required, read-only):  PYTHONDONTWRITEBYTECODE=1 python tests/golden/make_golden_ate.py"""
import os
import sys
import warnings

import numpy as np
from scipy.spatial.transform import Rotation

warnings.simplefilter("ignore")
sys.path.insert(0, "/root/reference")
from src.tools import eval_ate  # noqa: E402

out = {}
rng = np.random.RandomState(7)
for case, (n, noise) in enumerate([(40, 0.01), (200, 0.05), (5, 0.0), (60, 0.3)]):
    gt = np.cumsum(rng.randn(3, n) * 0.05, 1)
    R = Rotation.from_rotvec(rng.randn(3) * (0.5 + case)).as_matrix()
    est = R @ gt + rng.randn(3, 1) + rng.randn(3, n) * noise
    if case == 3:
        est[2] *= -1.0                      # a reflection: exercises the det < 0 branch
    rot, trans, err = eval_ate.align(np.matrix(est), np.matrix(gt))
    out[f"c{case}/est"], out[f"c{case}/gt"] = est, gt
    out[f"c{case}/rot"], out[f"c{case}/trans"], out[f"c{case}/err"] = np.asarray(rot), np.asarray(trans), np.asarray(err)
np.savez_compressed(os.path.join(os.path.dirname(os.path.abspath(__file__)), "ate_golden.npz"), **out)
print("wrote", len(out), "arrays")
