#!/usr/bin/env python3
"""Measurement tooling (CPU only, test infrastructure): how far apart are LEGITIMATE fp32 evaluations of the reference's own
operators on the scenes whose parity cases take the secondary gate (tests/scene_util.py::parity_failures)?

The round-4 verdict asked for the claimed cause of the ScanNet fine-stage misses -- "the K-order of the fp32 product p @ B in
front of sines of 1e3 rad arguments" -- to be SHOWN.  This tool evaluates the oracle (oracle/nice_oracle.py, pinned to the
reference) on the test scene in several modes and writes the distance of every output / gradient tensor to the default fp32
evaluation (max|a-b| / max|b|) to profiles/r05_reference_self_disagreement.json:

  embed:fma_k         p @ B as an x,y,z fused-multiply-add chain (the HIP kernels' order)       -> is ATen's mm bit-equal to it?
  embed:fma_k_rev     the chain in z,y,x order                                                   -> what a different K order costs
  embed:product_sum   three rounded products, summed (no fma)
  linear:rounded_once every Linear dot product accumulated in fp64, rounded once                 -> a different (better) summation order
  linear:bf16x3       every Linear product as six cross products of three-way bf16 splits, fp32 accumulation -> what a bf16-MFMA
                      formulation of the decoders would do to parity (round 6, profiles/r06_experiments.txt item 20)
  ulp_grids           the feature grids moved by one fp32 ulp                                    -> input noise
  fp64                the all-double evaluation                                                  -> the reference's distance to the truth

  python tools/reference_fp32_ambiguity.py [--rays 5000] [--scene scannet_0000] [--stage fine] [--out profiles/...json]
"""
import argparse
import json
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

from oracle import nice_oracle as orc  # noqa: E402
from scene_util import make_scene, oracle_render, rel_err, ulp_perturbed  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--rays", type=int, default=5000)
    ap.add_argument("--scene", default="scannet_0000")
    ap.add_argument("--stages", default="fine,color")
    ap.add_argument("--seed", type=int, default=22)
    ap.add_argument("--out", default=os.path.join(ROOT, "profiles", "r05_reference_self_disagreement.json"))
    args = ap.parse_args()
    sc = make_scene(seed=args.seed, n_rays=args.rays, scene=args.scene, fine_scale=1.0)      # the scene of tests/test_hip_parity.py::test_scannet_config
    # is ATen's p @ B the x,y,z fma chain on this host?
    p = (sc["rays_o"][:, None, :].double() + sc["rays_d"][:, None, :].double() * torch.linspace(0.1, 6.0, 48, dtype=torch.float64)[None, :, None]).reshape(-1, 3).float()
    B = sc["params"]["fine_decoder.embedder._B"]
    mm = p @ B
    chain = orc._EmbedArg.apply(p, B, "fma_k")
    rev = orc._EmbedArg.apply(p, B, "fma_k_rev")
    report = {"what": __doc__.split("\n\n")[1].replace("\n", " "),
              "host": {"torch": torch.__version__, "threads": torch.get_num_threads(),
                       "blas": [ln.strip() for ln in torch.__config__.show().splitlines() if "Math Kernel" in ln or "BLAS_INFO" in ln][:2]},
              "scene": {"name": args.scene, "rays": args.rays, "seed": args.seed, "max_abs_embedding_argument": float(mm.abs().max())},
              "aten_mm_vs_fma_chain_xyz": {"mismatching_elements": int((mm != chain).sum()), "of": mm.numel(), "max_abs_diff": float((mm - chain).abs().max())},
              "aten_mm_vs_fma_chain_zyx": {"mismatching_elements": int((mm != rev).sum()), "of": mm.numel(), "max_abs_diff": float((mm - rev).abs().max()),
                                           "max_abs_diff_of_the_sines": float((torch.sin(mm) - torch.sin(rev)).abs().max())},
              "cases": {}}
    for stage in args.stages.split(","):
        base = oracle_render(sc, stage, backward=True)
        modes = {}

        def run(name, **kw):
            old = (orc.EMBED_IMPL, orc.LINEAR_IMPL)
            try:
                orc.EMBED_IMPL = kw.get("embed", "mm")
                orc.LINEAR_IMPL = kw.get("linear", "mm")
                r = oracle_render(kw.get("scene", sc), stage, backward=True, lo=kw.get("lo", torch.float32))
            finally:
                orc.EMBED_IMPL, orc.LINEAR_IMPL = old
            modes[name] = {k: rel_err(r[k], base[k]) for k in base}

        run("embed:fma_k", embed="fma_k")
        run("embed:fma_k_rev", embed="fma_k_rev")
        run("embed:product_sum", embed="product_sum")
        run("linear:rounded_once", linear="rounded_once")
        run("linear:bf16x3", linear="bf16x3")
        run("ulp_grids", scene=ulp_perturbed(sc, 1234))
        run("fp64", lo=torch.float64)
        rows = {}
        for k in base:
            rows[k] = {m: modes[m][k] for m in modes}
        summ = {m: {"max_over_gradient_tensors": max(v for k, v in modes[m].items() if k not in ("depth", "var", "rgb")),
                    "gradient_tensors_at_or_above_1e-4": sum(1 for k, v in modes[m].items() if k not in ("depth", "var", "rgb") and v >= 1e-4),
                    "of": sum(1 for k in modes[m] if k not in ("depth", "var", "rgb")),
                    "max_over_outputs": max(modes[m][k] for k in ("depth", "var", "rgb"))} for m in modes}
        report["cases"][f"{args.scene}/{stage}"] = {"summary": summ, "tensors": rows}
        print(stage, json.dumps(summ, indent=1))
    json.dump(report, open(args.out, "w"), indent=1)
    print("wrote", args.out)


if __name__ == "__main__":
    main()
