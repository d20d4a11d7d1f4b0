"""TEST INFRASTRUCTURE (measurement script): ATE of the SAME synthetic RGB-D sequence tracked + mapped by
  (a) the product (nice_slam_amd: HIP kernels; fused mapping iterations replayed from hipGraphs),
  (b) the product through the drop-in call sequence (get_samples / render_batch_ray / torch losses, eager),
  (c) the reference's operator sequence on stock ATen / rocBLAS kernels of the same GPU (the oracle functions on cuda tensors),
all starting from identical grids / decoder parameters, with the reference's schedules (tools/slam_synthetic.MiniSLAM).
BASELINE.json north_star: ATE within 0.5 cm of the reference path.

    python tests/perf/ate_compare.py [--frames 60] [--map-iters 60] [--every-frame 5] [--iters-first 1500]
"""
import argparse
import copy
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, os.path.join(ROOT, "tools"))
import torch  # noqa: E402

import slam_synthetic as ss  # noqa: E402
from slam_oracle_ops import OracleOps  # noqa: E402


def run(kind, args, seq, cfg, init):
    torch.manual_seed(args.seed)
    dev = torch.device("cuda", 0)
    if kind == "aten":
        ops = OracleOps(seq, seed=args.seed, device=dev, grids=init["grids"], params=init["params"])
    else:
        ops = ss.ProductOps(seq, dev, seed=args.seed, fused=(kind == "fused"))
        with torch.no_grad():
            for k, v in init["grids"].items():
                ops.c[k].copy_(v.to(dev))
            ops.decoders.load_state_dict({k: v.to(dev) for k, v in init["params"].items()})
    slam = ss.MiniSLAM(ops, seq, cfg, seed=args.seed)
    t0 = time.perf_counter()
    res = slam.run()
    torch.cuda.synchronize()
    res["wall_s"] = time.perf_counter() - t0
    return res


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--frames", type=int, default=60)
    ap.add_argument("--height", type=int, default=240)
    ap.add_argument("--width", type=int, default=320)
    ap.add_argument("--map-iters", type=int, default=60)
    ap.add_argument("--every-frame", type=int, default=5)
    ap.add_argument("--iters-first", type=int, default=1500)
    ap.add_argument("--keyframe-every", type=int, default=10)
    ap.add_argument("--seed", type=int, default=0)
    ap.add_argument("--kinds", default="fused,product,aten")
    args = ap.parse_args()
    dev = torch.device("cuda", 0)
    cfg = copy.deepcopy(ss.DEFAULT_CFG)
    cfg["mapping"].update({"iters": args.map_iters, "every_frame": args.every_frame, "iters_first": args.iters_first,
                           "keyframe_every": args.keyframe_every})
    seq = ss.SyntheticSequence(args.frames, args.height, args.width, device=dev, seed=args.seed)
    torch.manual_seed(args.seed)
    p0 = ss.ProductOps(seq, dev, seed=args.seed)                    # the common starting point: the product's own initialisation
    init = {"grids": {k: v.detach().cpu().contiguous().clone() for k, v in p0.c.items()},
            "params": {k: v.detach().cpu().clone() for k, v in p0.decoders.state_dict().items()}}
    del p0
    out = {}
    for kind in args.kinds.split(","):
        r = run(kind, args, seq, cfg, init)
        out[kind] = {"ate_rmse_cm": r["ate"]["rmse"] * 100, "final_err_cm": r["raw_translation_error_cm"]["final"],
                     "mapping_ms_per_iter": 1e3 * r["mapping_s"] / max(1, r["mapping_iters"]),
                     "tracking_ms_per_iter": 1e3 * r["tracking_s"] / max(1, r["tracking_iters"]),
                     "mapping_iters": r["mapping_iters"], "tracking_iters": r["tracking_iters"], "wall_s": r["wall_s"]}
        print(kind, json.dumps(out[kind]), file=sys.stderr)
    res = {"sequence": {"frames": args.frames, "image": [args.height, args.width]},
           "schedule": {"map_iters": args.map_iters, "every_frame": args.every_frame, "iters_first": args.iters_first,
                        "keyframe_every": args.keyframe_every, "tracking_iters": cfg["tracking"]["iters"]},
           "runs": out}
    if "aten" in out:
        for k in out:
            if k != "aten":
                res[f"ate_{k}_minus_aten_cm"] = out[k]["ate_rmse_cm"] - out["aten"]["ate_rmse_cm"]
    print(json.dumps(res))


if __name__ == "__main__":
    main()
