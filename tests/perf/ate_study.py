"""TEST INFRASTRUCTURE (measurement script): the ATE distribution of the tiny synthetic SLAM run of tests/test_hip_slam_ate.py over
many seeds, one process per variant (the measurement switches of libnsr.so are read once per process).

    python tests/perf/ate_study.py --kind fused --seeds 0:100 --out gpurun_out/ate/fused.json
    NSR_PIXEL_DRAW=torch python tests/perf/ate_study.py --kind fused --seeds 0:60 --out gpurun_out/ate/fused_torchdraw.json
    python tests/perf/ate_study.py --kind aten --seeds 0:30 --out gpurun_out/ate/aten.json
    python tests/perf/ate_study.py --kind fused --seeds 0:1 --repeat 30 --out ...      # run-to-run spread of ONE seed
    python tests/perf/ate_study.py --summarize gpurun_out/ate/*.json                     # table + CI of the mean differences

kinds: fused (product, fused entry points replayed from hipGraphs), product (product kernels through the drop-in call sequence,
the same eager loop as `aten`), aten (the oracle functions on stock ATen kernels of the same GPU = the reference's operators).
Every run starts from the same grids / decoder parameters (the product's initialisation under seed 0)."""
import argparse
import copy
import json
import math
import os
import sys
import types

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, os.path.join(ROOT, "tools"))
sys.path.insert(0, os.path.join(ROOT, "tests", "perf"))

# the schedule of tests/test_hip_slam_ate.py
FRAMES, HEIGHT, WIDTH = 14, 120, 160
SCHEDULE = {"iters": 100, "every_frame": 2, "iters_first": 400, "keyframe_every": 4}


def setup(dev):
    import torch
    import slam_synthetic as ss
    cfg = copy.deepcopy(ss.DEFAULT_CFG)
    cfg["mapping"].update(SCHEDULE)
    seq = ss.SyntheticSequence(FRAMES, HEIGHT, WIDTH, device=dev, seed=0)
    torch.manual_seed(0)
    p0 = ss.ProductOps(seq, dev, seed=0)
    init = {"grids": {k: v.detach().cpu().contiguous().clone() for k, v in p0.c.items()},
            "params": {k: v.detach().cpu().clone() for k, v in p0.decoders.state_dict().items()}}
    del p0
    return seq, cfg, init


def stats(v):
    v = sorted(v)
    n = len(v)
    mean = sum(v) / n
    var = sum((x - mean) ** 2 for x in v) / max(1, n - 1)
    return {"n": n, "median": v[n // 2], "mean": mean, "sd": math.sqrt(var), "se": math.sqrt(var / n), "p90": v[min(n - 1, int(0.9 * n))],
            "max": v[-1], "min": v[0]}


def summarize(paths):
    runs = {}
    for p in paths:
        d = json.load(open(p))
        if d["label"] in runs:                         # one variant measured by several processes (disjoint seed ranges)
            runs[d["label"]]["ate_cm"] = runs[d["label"]]["ate_cm"] + d["ate_cm"]
        else:
            runs[d["label"]] = d
    ref = runs.get("aten")
    print("%-34s %4s %7s %7s %6s %6s %6s %6s   %s" % ("variant", "n", "median", "mean", "sd", "se", "p90", "max", "mean - aten [cm], 95 % CI"))
    for lab, d in sorted(runs.items()):
        s = stats(d["ate_cm"])
        extra = ""
        if ref is not None and lab != "aten":
            r = stats(ref["ate_cm"])
            diff, se = s["mean"] - r["mean"], math.sqrt(s["se"] ** 2 + r["se"] ** 2)
            extra = "%+.3f  [%+.3f, %+.3f]   median diff %+.3f" % (diff, diff - 1.96 * se, diff + 1.96 * se, s["median"] - r["median"])
        print("%-34s %4d %7.3f %7.3f %6.3f %6.3f %6.3f %6.3f   %s" % (lab, s["n"], s["median"], s["mean"], s["sd"], s["se"], s["p90"], s["max"], extra))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--kind", default="fused")
    ap.add_argument("--seeds", default="0:10", help="a:b")
    ap.add_argument("--repeat", type=int, default=1, help="run every seed this many times (run-to-run spread)")
    ap.add_argument("--label", default=None)
    ap.add_argument("--out", default=None)
    ap.add_argument("--summarize", nargs="*", default=None)
    args = ap.parse_args()
    if args.summarize is not None:
        summarize(args.summarize)
        return
    import time
    import torch
    import ate_compare as ac
    dev = torch.device("cuda", 0)
    seq, cfg, init = setup(dev)
    a, b = (int(x) for x in args.seeds.split(":"))
    ate, wall = [], []
    env = {k: v for k, v in os.environ.items() if k.startswith("NSR_")}
    if args.out:
        os.makedirs(os.path.dirname(os.path.abspath(args.out)), exist_ok=True)
    res = {}
    for sd in range(a, b):
        for _ in range(args.repeat):
            t0 = time.perf_counter()
            r = ac.run(args.kind, types.SimpleNamespace(seed=sd), seq, cfg, init)
            wall.append(time.perf_counter() - t0)
            ate.append(r["ate"]["rmse"] * 100)
            res = {"label": args.label or args.kind, "kind": args.kind, "seeds": [a, sd + 1], "repeat": args.repeat, "env": env, "ate_cm": ate,
                   "wall_s_per_run": sum(wall) / len(wall), "sequence": [FRAMES, HEIGHT, WIDTH], "schedule": SCHEDULE, "stats": stats(ate)}
            if args.out:                               # after every run: a call that hits its time limit keeps what it has
                open(args.out + ".tmp", "w").write(json.dumps(res))
                os.replace(args.out + ".tmp", args.out)
    print(json.dumps({k: res[k] for k in ("label", "stats", "wall_s_per_run", "env")}))


if __name__ == "__main__":
    main()
