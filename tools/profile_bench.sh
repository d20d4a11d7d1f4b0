#!/bin/sh
# Measurement tooling: bench line + rocprofv3 kernel statistics of the same command, written under gpurun_out/<tag>/ on
# the GPU box; copy <tag>_bench.json / <tag>_kernel_stats.csv / <tag>_kernel_stats_top.txt into profiles/ afterwards.
#   sh tools/profile_bench.sh <tag> [bench.py flags]
set -e
TAG="$1"; shift
ROOT="$(cd "$(dirname "$0")/.." && pwd)"
OUT="$ROOT/gpurun_out/$TAG"
mkdir -p "$OUT"
cd "$ROOT"
python bench.py "$@" > "$OUT/${TAG}_bench.json" 2> "$OUT/bench.err"
export TMPDIR=/tmp
( cd /tmp && rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/prof" -- python "$ROOT/bench.py" --no-cpu-baseline "$@" > "$OUT/prof_bench.json" 2> "$OUT/prof.err" ) || true
STATS="$(find "$OUT/prof" -name '*kernel_stats.csv' | head -1)"
cp "$STATS" "$OUT/${TAG}_kernel_stats.csv"
python - "$OUT/${TAG}_kernel_stats.csv" "$OUT/${TAG}_kernel_stats_top.txt" "$*" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: -float(r["Percentage"]))
with open(sys.argv[2], "w") as f:
    f.write(f"rocprofv3 --kernel-trace --stats --output-format csv -- python bench.py --no-cpu-baseline {sys.argv[3]}   (MI355X)\n")
    f.write("%-72s %6s %12s %10s\n" % ("kernel", "calls", "avg_us", "pct"))
    nsr = 0.0
    for r in rows[:24]:
        f.write("%-72s %6d %12.1f %10.2f\n" % (r["Name"][:72], int(r["Calls"]), float(r["AverageNs"]) / 1e3, float(r["Percentage"])))
    for r in rows:
        if "nsr::" in r["Name"]:
            nsr += float(r["Percentage"])
    f.write("nsr kernels: %.1f %% of GPU time; launches per timed+warm-up run: %d\n" % (nsr, sum(int(r["Calls"]) for r in rows)))
print(open(sys.argv[2]).read())
PY
cat "$OUT/${TAG}_bench.json" | cut -c1-400
