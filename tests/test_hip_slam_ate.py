"""GPU: ATE of a short synthetic RGB-D sequence, product vs the reference's operator sequence (BASELINE.json north_star:
"ATE within 0.5 cm of reference").  tools/slam_synthetic.MiniSLAM (the reference's tracker + mapper loops, strict sync) runs
twice from identical grids / decoders: on nice_slam_amd (fused mapping iterations replayed from hipGraphs, HIP kernels) and on
the oracle functions executed on the same GPU (stock ATen / rocBLAS kernels, the reference's ops).  ATE = the reference's
Horn alignment + translational RMSE (tools/ate.py, pinned to src/tools/eval_ate.py by tests/golden/ate_golden.npz)."""
import copy
import os
import sys
import types

import pytest
import torch

from conftest import ROOT

pytestmark = pytest.mark.gpu


def test_ate_product_within_half_a_cm_of_the_reference_ops():
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    sys.path.insert(0, os.path.join(ROOT, "tests", "perf"))
    import slam_synthetic as ss
    import ate_compare as ac
    dev = torch.device("cuda", 0)
    cfg = copy.deepcopy(ss.DEFAULT_CFG)
    cfg["mapping"].update({"iters": 100, "every_frame": 2, "iters_first": 400, "keyframe_every": 4})
    seq = ss.SyntheticSequence(14, 120, 160, device=dev, seed=0)
    torch.manual_seed(0)
    p0 = ss.ProductOps(seq, dev, seed=0)
    init = {"grids": {k: v.detach().cpu().contiguous().clone() for k, v in p0.c.items()},
            "params": {k: v.detach().cpu().clone() for k, v in p0.decoders.state_dict().items()}}
    del p0
    # The loop is chaotic in the small: the two paths draw different pixels (the fused path draws inside its window kernel, the
    # reference path once per frame with torch.randint), the HIP path's gradient atomics are unordered, and a different early
    # pose estimate changes every later keyframe.  Single runs of the product on this tiny sequence land between 0.3 and 1.4 cm
    # whatever the pixel source (eight seeds each, torch.randint: median 1.0, in-kernel draw: median 0.8), the reference path
    # between 0.3 and 0.85, so the comparison is between MEDIANS: 31 seeds of the product (a run takes under a second) against
    # five of the reference path (13 s each), all starting from the same map; the means are printed next to them.
    # (Measured over 15 seeds each at the end of round 3: product median 0.54 ... 0.86 cm over six configurations of pixel
    # source / tracker loop / kernels, reference path 0.45 cm.)
    ate = {"fused": [], "aten": []}
    for k, seeds in (("fused", range(31)), ("aten", range(5))):
        for sd in seeds:
            r = ac.run(k, types.SimpleNamespace(seed=sd), seq, cfg, init)
            ate[k].append(r["ate"]["rmse"] * 100)
            assert r["mapping_iters"] == 1100 and r["tracking_iters"] == 130, (k, sd, r["mapping_iters"], r["tracking_iters"])
    mean = {k: sum(v) / len(v) for k, v in ate.items()}
    med = {k: sorted(v)[len(v) // 2] for k, v in ate.items()}
    print("ATE [cm] per seed:", {k: [round(x, 2) for x in v] for k, v in ate.items()}, "medians:", med, "means:", mean)
    assert med["aten"] < 3.0, ate                       # the reference path itself holds the trajectory on this sequence
    assert abs(med["fused"] - med["aten"]) < 0.5, (ate, med, mean)
