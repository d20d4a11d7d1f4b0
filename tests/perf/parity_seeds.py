#!/usr/bin/env python3
"""TEST INFRASTRUCTURE (GPU box): how general is the parity of the large-bound scenes?  The committed GPU tests hold ONE seed per
BASELINE configuration; this sweep renders the ScanNet / Apartment configurations (5000 rays, every stage, forward + every
gradient) for several seeds on the MI355X and on the CPU oracle and counts, per case, the tensors whose max|a-b| / max|b| reaches
1e-4 (north_star's tolerance) -- no secondary gate, no budget, just the primary comparison.

    python tests/perf/parity_seeds.py [--seeds 22 101 102 103] [--rays 5000] [--out gpurun_out/parity_seeds.json]
"""
import argparse
import json
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
TESTS = os.path.dirname(HERE)
ROOT = os.path.dirname(TESTS)
sys.path[:0] = [ROOT, TESTS]

import torch  # noqa: E402

from scene_util import make_scene, oracle_render, hip_render, rel_err  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--seeds", type=int, nargs="+", default=[22, 101, 102, 103])
    ap.add_argument("--rays", type=int, default=5000)
    ap.add_argument("--scenes", nargs="+", default=["scannet_0000", "apartment"])
    ap.add_argument("--out", default=os.path.join(ROOT, "gpurun_out", "parity_seeds.json"))
    a = ap.parse_args()
    torch.set_num_threads(min(16, os.cpu_count() or 1))
    out = {"tolerance": 1e-4, "rays": a.rays, "cases": {}}
    for scene in a.scenes:
        for seed in a.seeds:
            sc = make_scene(seed=seed, n_rays=a.rays, scene=scene, fine_scale=1.0)
            for stage in ("middle", "fine", "color"):
                got = hip_render(sc, stage, backward=True)
                ref = oracle_render(sc, stage, backward=True)
                errs = {k: rel_err(got[k], v) for k, v in ref.items()}
                miss = {k: float("%.3g" % e) for k, e in errs.items() if e >= 1e-4}
                worst = max(errs.items(), key=lambda kv: kv[1])
                rec = {"tensors": len(errs), "missing_1e-4": len(miss), "max_rel_err": worst[1], "worst_tensor": worst[0], "missing": miss}
                out["cases"]["%s/%s/seed%d" % (scene, stage, seed)] = rec
                print("%-14s %-7s seed %4d: %2d of %2d tensors >= 1e-4, max %.2e (%s)" % (scene, stage, seed, len(miss), len(errs), worst[1], worst[0]), flush=True)
    tot = sum(v["missing_1e-4"] for v in out["cases"].values())
    out["total_missing"] = tot
    out["total_tensors"] = sum(v["tensors"] for v in out["cases"].values())
    print("total: %d of %d tensors beyond 1e-4" % (tot, out["total_tensors"]))
    os.makedirs(os.path.dirname(a.out), exist_ok=True)
    json.dump(out, open(a.out, "w"), indent=1)


if __name__ == "__main__":
    main()
