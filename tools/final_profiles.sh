#!/bin/sh
# Measurement tooling: the round's final set -- every BASELINE configuration (bench line + rocprofv3 kernel statistics), the driver's
# command line, the strong-scaling proxy -- in ONE gpurun call (one box).   sh tools/final_profiles.sh <round tag, e.g. r06z>
TAG="${1:-r06z}"
ROOT="$(cd "$(dirname "$0")/.." && pwd)"
cd "$ROOT"
mkdir -p gpurun_out/$TAG
sh tools/profile_bench.sh ${TAG}_c1 --no-strong-record --no-consumed-record > gpurun_out/$TAG/c1.log 2>&1
python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/$TAG/${TAG}_default_bench.json 2> gpurun_out/$TAG/default.err
sh tools/profile_bench.sh ${TAG}_c1_stepped --stepped-grads-only --no-strong-record --no-consumed-record > gpurun_out/$TAG/c1s.log 2>&1
sh tools/profile_bench.sh ${TAG}_c1_consumed --consumed-grads-only --no-strong-record --no-consumed-record > gpurun_out/$TAG/c1c.log 2>&1
sh tools/profile_bench.sh ${TAG}_c0 --config 0 > gpurun_out/$TAG/c0.log 2>&1
sh tools/profile_bench.sh ${TAG}_c2 --config 2 --no-cpu-baseline > gpurun_out/$TAG/c2.log 2>&1
sh tools/profile_bench.sh ${TAG}_c3 --config 3 --no-cpu-baseline > gpurun_out/$TAG/c3.log 2>&1
sh tools/profile_bench.sh ${TAG}_c4 --config 4 --steps 10 --warmup 3 --no-cpu-baseline > gpurun_out/$TAG/c4.log 2>&1
sh tools/profile_bench.sh ${TAG}_ctracking --config tracking > gpurun_out/$TAG/ct.log 2>&1
NSR_BENCH_WINDOW_GRAPH=0 python bench.py --no-cpu-baseline --no-strong-record --no-consumed-record > gpurun_out/$TAG/${TAG}_c1_per_step_replays_bench.json 2> gpurun_out/$TAG/c1r.err
for R in 5000 2500 1250 625; do
  python bench.py --config 3 --no-strong-record --no-cpu-baseline --rays $R > gpurun_out/$TAG/strong_$R.json 2> gpurun_out/$TAG/strong_$R.err
done
python - "$TAG" <<'PY'
import json, sys
tag = sys.argv[1]
def load(p):
    return json.loads(open(p).read().strip().split("\n")[-1])
rows = {}
out = ["Strong-scaling proxy on ONE MI355X (no N > 1 RCCL run exists): BASELINE configs[3] (Apartment, 5000 rays per iteration) at the per-GPU",
       "batch of 1 / 2 / 4 / 8 GPUs -- python bench.py --config 3 --no-strong-record --no-cpu-baseline --rays R, the 60 timed steps replayed as one",
       "hipGraph, median of 3 windows, one box, final code of the round (tools/final_profiles.sh).", "",
       "    rays    ms / iter  rendered rays/s   backward per stage (ms, HIP events)"]
for R in (5000, 2500, 1250, 625):
    d = load(f"gpurun_out/{tag}/strong_{R}.json")
    rows[R] = d["ms_per_step"]
    out.append("   %5d    %9.4f   %14d   %s" % (R, d["ms_per_step"], d["value"], json.dumps(d.get("kernel_ms"))))
out += ["", "T(5000) / (8 T(625)) = %.3f   T(5000) / (4 T(1250)) = %.3f   T(5000) / (2 T(2500)) = %.3f" %
        (rows[5000] / (8 * rows[625]), rows[5000] / (4 * rows[1250]), rows[5000] / (2 * rows[2500]))]
open(f"gpurun_out/{tag}/{tag}_strong_proxy.txt", "w").write("\n".join(out) + "\n")
print("\n".join(out))
for t in ("c1", "c1_stepped", "c1_consumed", "c0", "c2", "c3", "c4", "ctracking"):
    try:
        d = load(f"gpurun_out/{tag}_{t}/{tag}_{t}_bench.json")
        k = d.get("roofline", {}).get("kernels", {})
        print("%-12s ms/step %.4f  value %.3f M  frac %.4f  %s" % (t, d["ms_per_step"], d["value"] / 1e6, d.get("roofline", {}).get("frac", 0),
                                                                      {a: round(b.get("ms", 0) * 1e3, 1) for a, b in k.items()}))
    except Exception as e:
        print(t, "ERR", e)
d = load(f"gpurun_out/{tag}/{tag}_default_bench.json")
print("default (driver's command): ms/step %.4f value %.3f M frac %.4f strong %.3f M (%.4f ms) consumed %.3f M" % (
    d["ms_per_step"], d["value"] / 1e6, d["roofline"]["frac"], d["strong"]["value"] / 1e6, d["strong"]["ms_per_step"], d["consumed_grads_only"]["value"] / 1e6))
d = load(f"gpurun_out/{tag}/{tag}_c1_per_step_replays_bench.json")
print("one replay per step: ms/step %.4f value %.3f M" % (d["ms_per_step"], d["value"] / 1e6))
PY
