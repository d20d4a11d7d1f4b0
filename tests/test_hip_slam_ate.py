"""GPU: ATE of a short synthetic RGB-D sequence, product vs the reference's operator sequence (BASELINE.json north_star:
"ATE within 0.5 cm of reference").  tools/slam_synthetic.MiniSLAM (the reference's tracker + mapper loops, strict sync) runs
twice from identical grids / decoders: on nice_slam_amd (fused mapping iterations replayed from hipGraphs, HIP kernels) and on
the oracle functions executed on the same GPU (stock ATen / rocBLAS kernels, the reference's ops).  ATE = the reference's
Horn alignment + translational RMSE (tools/ate.py, pinned to src/tools/eval_ate.py by tests/golden/ate_golden.npz)."""
import os
import sys
import types

import pytest
import torch

from conftest import ROOT

pytestmark = pytest.mark.gpu


def test_ate_product_distribution_matches_the_reference_ops():
    """The ATE of a run of this tiny sequence is a draw from a distribution (profiles/r04_ate.txt: ONE seed of the product run
    thirty times spreads as widely, sd 0.41 cm, as thirty different seeds do -- unordered gradient atomics in a chaotic loop), so
    the comparison is between distributions: 41 seeds of the product (0.4 s each) against the recorded 30-seed distribution of the
    reference's operators on stock ATen kernels (tests/golden/ate_reference_ops.json; that loop is deterministic per seed, and two
    of its seeds are re-run here as the pin -- if they do not reproduce, fifteen fresh reference runs replace the recording).
    Gates: |median difference| < 0.25 cm, |mean difference| < 0.30 cm, 90th percentile below the reference's + 0.60 cm.
    (Round 4, 100 vs 30 seeds: means 0.649 vs 0.581 cm, difference +0.068 with 95 % CI [-0.042, +0.177]; medians 0.586 vs 0.503.)"""
    import json
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    sys.path.insert(0, os.path.join(ROOT, "tests", "perf"))
    import ate_compare as ac
    import ate_study
    dev = torch.device("cuda", 0)
    seq, cfg, init = ate_study.setup(dev)
    rec = json.load(open(os.path.join(ROOT, "tests", "golden", "ate_reference_ops.json")))
    assert rec["seeds"] == [0, 30] and len(rec["ate_cm"]) == 30

    def run(kind, sd):
        r = ac.run(kind, types.SimpleNamespace(seed=sd), seq, cfg, init)
        assert r["mapping_iters"] == 1100 and r["tracking_iters"] == 130, (kind, sd, r["mapping_iters"], r["tracking_iters"])
        return r["ate"]["rmse"] * 100

    ref = list(rec["ate_cm"])
    pin = {sd: run("aten", sd) for sd in (3, 17)}
    pinned = all(abs(v - ref[sd]) < 2e-2 for sd, v in pin.items())
    if not pinned:                                       # another ATen / ROCm build: measure the reference distribution afresh
        ref = [pin.get(sd) if sd in pin else run("aten", sd) for sd in range(15)]
    fused = [run("fused", sd) for sd in range(41)]
    st_f, st_r = ate_study.stats(fused), ate_study.stats(ref)
    print("ATE [cm]: product", {k: round(v, 3) for k, v in st_f.items()}, "reference ops", {k: round(v, 3) for k, v in st_r.items()},
          "pin", {k: round(v, 3) for k, v in pin.items()}, "recorded distribution used" if pinned else "fresh reference runs (pin moved)")
    assert st_r["median"] < 3.0, ref                     # the reference path itself holds the trajectory on this sequence
    assert abs(st_f["median"] - st_r["median"]) < 0.25, (st_f, st_r)
    assert abs(st_f["mean"] - st_r["mean"]) < 0.30, (st_f, st_r)
    assert st_f["p90"] < st_r["p90"] + 0.60, (st_f, st_r)
