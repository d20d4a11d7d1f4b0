#!/usr/bin/env python3
"""TEST INFRASTRUCTURE (CPU only, build container): what makes the ScanNet-bound fine stage miss 1e-4 against the fp32 oracle?

The kernel sources run under the CPU emulator (tests/emu: same sources, MFMA modelled as the k = 0..3 fma chain) twice:
  A  as shipped until round 5 (packed-fp32 Cody-Waite sine everywhere, abs err ~1.5e-7; -DNSR_X_FWD_SIN_F32)
  P  the product of round 6 (the forward's embedding sine in fp64, rounded once: nsr_kernels.h sin_f64; no define)
  B  -DNSR_X_LIBM_SIN: every sine / cosine correctly rounded (double libm, rounded once to fp32)
and each result is compared, tensor by tensor, against
  O1 the fp32 oracle as the parity tests use it (ATen mm / MKL order inside the Linear layers),
  O2 the fp32 oracle with every Linear dot product rounded once (tools/reference_fp32_ambiguity.py's second legitimate evaluation),
  O3 the oracle with the Linear layers summed in the kernels' K order (LINEAR_IMPL "mfma_k": bias last, k ascending in one fma chain).
Counts of gradient tensors with max|a-b|/max|b| >= 1e-4 and the worst tensor of each pairing go to profiles/r06_parity_causes.json.

    python tests/perf/parity_causes.py [--rays 1500] [--seed 22] [--stage fine] [--out profiles/r06_parity_causes.json]
"""
import argparse
import json
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
TESTS = os.path.dirname(HERE)
ROOT = os.path.dirname(TESTS)
sys.path[:0] = [ROOT, TESTS]

import numpy as np  # noqa: E402
import torch  # noqa: E402

from scene_util import make_scene, oracle_render, rel_err  # noqa: E402
from oracle import nice_oracle as orc  # noqa: E402
from nice_slam_amd import _capi  # noqa: E402
import emu_harness  # noqa: E402


def build_variant(name, defs):
    out = os.path.join(TESTS, "emu", f"libnsr_emu_{name}.so")
    env = dict(os.environ, NSR_EMU_OUT=out, NSR_EMU_BUILD=os.path.join(TESTS, "emu", f"_build_{name}"), NSR_EMU_DEFS=defs)
    subprocess.run([os.path.join(TESTS, "emu", "build_emu.sh")], check=True, capture_output=True, env=env)
    return out


def run_emu(lib_path, s, stage):
    lib = _capi.Lib(lib_path)
    sc = emu_harness.HostScene(lib, s["grids"], s["params"], s["bound"].numpy())
    fwd = sc.forward(stage, s["rays_o"].numpy(), s["rays_d"].numpy(), s["gt_depth"].numpy())
    res = sc.backward(stage, fwd, s["w"]["depth"].numpy(), s["w"]["var"].numpy(), s["w"]["rgb"].numpy())
    res.update({k: fwd[k] for k in ("depth", "var", "rgb")})
    return res


def table(got, ref):
    rows = {k: rel_err(got[k], v) for k, v in ref.items() if k in got}
    grads = {k: e for k, e in rows.items() if k not in ("depth", "var", "rgb")}
    miss = {k: e for k, e in grads.items() if e >= 1e-4}
    worst = max(grads.items(), key=lambda kv: kv[1])
    return {"gradient_tensors": len(grads), "missing_1e-4": len(miss), "max_rel_err": worst[1], "worst_tensor": worst[0],
            "outputs_max_rel_err": max(rows[k] for k in ("depth", "var", "rgb")),
            "missing": {k: float("%.3g" % e) for k, e in sorted(miss.items(), key=lambda kv: -kv[1])}}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--rays", type=int, default=1500)
    ap.add_argument("--seed", type=int, default=22)
    ap.add_argument("--stage", default="fine")
    ap.add_argument("--scene", default="scannet_0000")
    ap.add_argument("--split", action="store_true", help="also: the correctly rounded sine in ONE of its three places at a time")
    ap.add_argument("--out", default=os.path.join(ROOT, "profiles", "r06_parity_causes.json"))
    a = ap.parse_args()
    torch.set_num_threads(8)
    s = make_scene(seed=a.seed, n_rays=a.rays, scene=a.scene, fine_scale=1.0)
    libs = {"A_shipped_sine": build_variant("a", "-DNSR_X_FWD_SIN_F32"),      # rounds 1-5: the packed fp32 sine everywhere
            "P_product_fp64_forward_sine": build_variant("p", ""),            # round 6: what libnsr.so is built from
            "B_correctly_rounded_sine": build_variant("b", "-DNSR_X_LIBM_SIN=1")}
    for tag, v in (("B2_correctly_rounded_forward_embedding_only", 2), ("B3_correctly_rounded_dW_reevaluation_only", 3), ("B4_correctly_rounded_dX_cosines_only", 4)):
        if a.split:
            libs[tag] = build_variant("b%d" % v, "-DNSR_X_LIBM_SIN=%d" % v)
    got = {k: run_emu(p, s, a.stage) for k, p in libs.items()}
    refs = {}
    for name, (emb, lin) in {"O1_oracle_mm": ("mm", "mm"), "O2_oracle_linear_rounded_once": ("mm", "rounded_once"),
                             "O3_oracle_linear_in_kernel_k_order": ("mm", "mfma_k")}.items():
        if lin not in ("mm", "rounded_once") + tuple(getattr(orc, "EXTRA_LINEAR_IMPLS", ())):
            continue
        try:
            orc.EMBED_IMPL, orc.LINEAR_IMPL = emb, lin
            refs[name] = oracle_render(s, a.stage, backward=True)
        finally:
            orc.EMBED_IMPL, orc.LINEAR_IMPL = "mm", "mm"
    truth = oracle_render(s, a.stage, backward=True, lo=torch.float64)
    out = {"scene": a.scene, "stage": a.stage, "rays": a.rays, "seed": a.seed,
           "method": __doc__.split("\n\n")[1].replace("\n", " "),
           "product_vs_oracle": {g: {r: table(got[g], refs[r]) for r in refs} for g in got},
           "oracle_vs_oracle": {r: table(refs[r], refs["O1_oracle_mm"]) for r in refs if r != "O1_oracle_mm"},
           "A_vs_B": table(got["A_shipped_sine"], got["B_correctly_rounded_sine"]),
           "distance_to_fp64_truth": {**{g: table(got[g], truth) for g in got}, **{r: table(refs[r], truth) for r in refs}}}
    for sect in ("product_vs_oracle", "distance_to_fp64_truth"):
        for k, v in out[sect].items():
            print(sect, k, json.dumps({kk: (vv if kk != "missing" else len(vv)) for kk, vv in (v.items() if "gradient_tensors" in v else
                                                                                               {r: (t["missing_1e-4"], "%.3g" % t["max_rel_err"]) for r, t in v.items()}.items())}))
    json.dump(out, open(a.out, "w"), indent=1)
    print("wrote", a.out)


if __name__ == "__main__":
    main()
