#!/usr/bin/env python3
"""TEST INFRASTRUCTURE (CPU only): root cause of the ONE output row of one Linear layer that misses 1e-4 at BASELINE configs[4]
(100 000 rays, tests/test_hip_parity.py::test_synthetic_stress_full_size_vs_oracle: row 22 of color_decoder.pts_linears.1, weight
and bias, 1.33e-4, the same row since round 2).

Hypothesis of the test's comment: a relu whose pre-activation is within rounding of zero at ONE sample point switches dY_1[22] of
that point on or off, which moves row 22 of dW_1 / db_1 by that point's whole contribution.  This script finds the point and shows it:

  1. the oracle's operators (oracle/nice_oracle.py) evaluate the pre-activation z = W_1 h_0 + b_1 of the colour decoder's layer 1 at
     all 4.8 M sample points of the batch, in fp32 and in fp64; the candidates are the points with the smallest |z[22]|;
  2. for each candidate's ray the kernel sources run under the CPU emulator (tests/emu, same sources as libnsr.so) and the oracle runs
     in fp32 and fp64 on that ray alone (+ the ray that carries the batch-global max depth, zero loss weight): row 22 of dW_1, db_1
     from the three evaluations, and the kernel's saved relu mask bit of (point, layer 1, unit 22);
  3. the ray whose kernel-vs-oracle difference of that row is a whole point's contribution is the cause; its size relative to
     max|dW_1| of the full batch (--full: the 100 000-ray oracle, ~10 min of CPU) is the number the GPU test prints.

    python tests/perf/relu_kink_cause.py [--row 22] [--layer 1] [--top 6] [--full] [--out profiles/r06_relu_kink_cause.json]
"""
import argparse
import json
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
TESTS = os.path.dirname(HERE)
ROOT = os.path.dirname(TESTS)
sys.path[:0] = [ROOT, TESTS]

import numpy as np  # noqa: E402
import torch  # noqa: E402

from scene_util import make_scene, oracle_render, oracle_render_chunked  # noqa: E402
from oracle import nice_oracle as orc  # noqa: E402
from nice_slam_amd import _capi  # noqa: E402
import emu_harness  # noqa: E402


def layer_preact(sc, rays, layer, lo, dec="color"):
    """z = W_layer h + b_layer of decoder `dec` (colour by default) at every sample point of `rays` (oracle operators, decoder.py:177-203)"""
    P = {k: v.to(lo) for k, v in sc["params"].items()}
    o, d, gd = sc["rays_o"][rays], sc["rays_d"][rays], sc["gt_depth"][rays]
    # the oracle takes the batch-global max depth from the batch itself: append the maximum-depth ray (Renderer.py:109,144)
    j = int(torch.argmax(sc["gt_depth"]))
    z = orc.sample_depths(torch.cat([o, sc["rays_o"][j:j + 1]]), torch.cat([d, sc["rays_d"][j:j + 1]]), torch.cat([gd, sc["gt_depth"][j:j + 1]]),
                          sc["bound"], "color", 32, 16, torch.float64)[:-1]
    pts = (o[:, None, :].to(torch.float64) + d[:, None, :].to(torch.float64) * z[:, :, None]).reshape(-1, 3)
    bounds = orc.decoder_bounds(sc["bound"], 2.0)
    if dec == "fine":          # decoder.py:182-187: [c_fine | c_mid]
        c = torch.cat([orc.trilinear(sc["grids"]["grid_fine"].to(lo), pts, bounds["fine"], lo),
                       orc.trilinear(sc["grids"]["grid_middle"].to(lo), pts, bounds["middle"], lo)], -1)
    else:
        c = orc.trilinear(sc["grids"]["grid_" + dec].to(lo), pts, bounds[dec], lo)
    pre = dec + "_decoder."
    e = torch.sin(pts.to(lo) @ P[pre + "embedder._B"])
    h = e
    for i in range(5):
        zi = orc._lin(h, P, pre + f"pts_linears.{i}")
        if i == layer:
            return zi
        h = torch.relu(zi) + orc._lin(c, P, pre + f"fc_c.{i}")
        if i == 2:
            h = torch.cat([e, h], -1)
    raise ValueError(layer)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--row", type=int, default=22)
    ap.add_argument("--layer", type=int, default=1)
    ap.add_argument("--top", type=int, default=6)
    ap.add_argument("--rays", type=int, default=100_000)
    ap.add_argument("--scene", default="synthetic")
    ap.add_argument("--seed", type=int, default=26)
    ap.add_argument("--decoder", default="color", choices=("middle", "fine", "color"))
    ap.add_argument("--full", action="store_true", help="also evaluate the full-batch oracle (max|dW| of the layer: the test's denominator)")
    ap.add_argument("--out", default=os.path.join(ROOT, "profiles", "r06_relu_kink_cause.json"))
    args = ap.parse_args()
    torch.set_num_threads(8)
    sc = make_scene(seed=args.seed, n_rays=args.rays, scene=args.scene, fine_scale=1.0)
    n, S = args.rays, 48
    key_w, key_b = f"dparam/{args.decoder}_decoder.pts_linears.{args.layer}.weight", f"dparam/{args.decoder}_decoder.pts_linears.{args.layer}.bias"
    # 1. candidates
    best = []
    with torch.no_grad():
        for lo_i in range(0, n, 10_000):
            sl = slice(lo_i, min(n, lo_i + 10_000))
            z32 = layer_preact(sc, sl, args.layer, torch.float32, args.decoder)
            z64 = layer_preact(sc, sl, args.layer, torch.float64, args.decoder)
            if args.row >= 0:
                z32, z64 = z32[:, args.row:args.row + 1], z64[:, args.row:args.row + 1]
            a = z32.abs().double().reshape(-1)
            k = torch.topk(-a, min(args.top, a.numel())).indices
            nu = z32.shape[1]
            for i in k.tolist():
                best.append((float(a[i]), lo_i * S + i // nu, float(z32.reshape(-1)[i]), float(z64.reshape(-1)[i]), (args.row if args.row >= 0 else i % nu)))
    best.sort()
    best = best[:args.top]
    print("candidates (|z| smallest):")
    for a, gp, z32, z64, unit in best:
        print(f"   unit {unit} point {gp} = ray {gp // S} sample {gp % S}:  z32 = {z32:+.3e}   z64 = {z64:+.3e}   sign flip between fp32 and fp64: {(z32 > 0) != (z64 > 0)}")
    # 2. emulator / oracle fp32 / oracle fp64 on each candidate's ray
    lib = _capi.Lib(os.path.join(TESTS, "emu", "libnsr_emu.so"))
    hs = emu_harness.HostScene(lib, sc["grids"], sc["params"], sc["bound"].numpy())
    hs.save_acts = True
    j = int(torch.argmax(sc["gt_depth"]))
    rows = []
    for a, gp, z32, z64, unit in best:
        r = gp // S
        idx = torch.tensor([r, j])
        sub = {k: sc[k] for k in ("grids", "params", "bound", "intr")}
        for k in ("rays_o", "rays_d", "gt_depth"):
            sub[k] = sc[k][idx].clone()
        sub["w"] = {k: torch.cat([v[r:r + 1], torch.zeros_like(v[:1])]) for k, v in sc["w"].items()}
        o32 = oracle_render(sub, "color", backward=True)
        o64 = oracle_render(sub, "color", backward=True, lo=torch.float64)
        fwd = hs.forward("color", sub["rays_o"].numpy(), sub["rays_d"].numpy(), sub["gt_depth"].numpy())
        emu = hs.backward("color", fwd, sub["w"]["depth"].numpy(), sub["w"]["var"].numpy(), sub["w"]["rgb"].numpy())
        row = lambda d_, k_: np.asarray(d_[k_].detach().cpu() if torch.is_tensor(d_[k_]) else d_[k_], dtype=np.float64)[unit]
        w_e, w_32, w_64 = row(emu, key_w), row(o32, key_w), row(o64, key_w)
        b_e, b_32, b_64 = float(row(emu, key_b)), float(row(o32, key_b)), float(row(o64, key_b))
        # the whole contribution of the point to the row: the oracle with unit `row` of that point's relu forced the other way
        rec = {"unit": unit, "point": gp, "ray": r, "sample": gp % S, "z_oracle_fp32": z32, "z_oracle_fp64": z64,
               "db_row": {"kernel_sources_emulated": b_e, "oracle_fp32": b_32, "oracle_fp64": b_64},
               "max_abs_dW_row": {"kernel - oracle_fp32": float(np.abs(w_e - w_32).max()), "oracle_fp64 - oracle_fp32": float(np.abs(w_64 - w_32).max()),
                                  "kernel - oracle_fp64": float(np.abs(w_e - w_64).max()), "oracle_fp32 row": float(np.abs(w_32).max())},
               "kernel_agrees_with": "fp64" if np.abs(w_e - w_64).max() < np.abs(w_e - w_32).max() else "fp32"}
        # a whole point's contribution is missing / extra in the kernel's row (vs. rounding-level differences elsewhere): the relu of
        # (this point, layer, unit) is in the other state in the kernel -- its fp32 pre-activation rounded to the other side of zero
        rec["relu_state_differs_in_kernel"] = bool(np.abs(w_e - w_32).max() > 1e3 * max(np.abs(w_64 - w_32).max(), 1e-30)
                                                   and np.abs(w_e - w_32).max() > 1e-3 * np.abs(w_32).max())
        rows.append(rec)
        print(json.dumps(rec))
    out = {"case": "%s seed %d, %d rays" % (args.scene, args.seed, args.rays), "tensor": key_w, "row": args.row,
           "method": __doc__.split("\n\n")[2].strip(), "candidates": rows}
    if args.full:
        ref = oracle_render_chunked(sc, "color")
        den_w, den_b = float(ref[key_w].abs().max()), float(ref[key_b].abs().max())
        out["full_batch_max_abs"] = {"dW": den_w, "db": den_b}
        for rec in rows:
            rec["relative_to_full_batch"] = {"dW_row_kernel_minus_oracle32": rec["max_abs_dW_row"]["kernel - oracle_fp32"] / den_w,
                                             "db_row_kernel_minus_oracle32": abs(rec["db_row"]["kernel_sources_emulated"] - rec["db_row"]["oracle_fp32"]) / den_b}
            print(rec["point"], rec["relative_to_full_batch"])
    with open(args.out, "w") as f:
        json.dump(out, f, indent=1)
    print("wrote", args.out)


if __name__ == "__main__":
    main()
