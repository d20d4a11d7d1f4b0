#!/bin/sh
# Measurement tooling: per-kernel average times (rocprofv3 --kernel-trace --stats) of one bench configuration under a list of
# NSR_X measurement switches / environment settings.   sh tools/xsweep.sh <tag> "<bench flags>" "ENV1=.. ENV2=.." "ENV.." ...
TAG="$1"; FLAGS="$2"; shift 2
ROOT="$(cd "$(dirname "$0")/.." && pwd)"
OUT="$ROOT/gpurun_out/$TAG"
mkdir -p "$OUT"
export TMPDIR=/tmp
i=0
for ENVS in "$@"; do
  i=$((i+1))
  D="$OUT/run$i"
  ( cd /tmp && env $ENVS rocprofv3 --kernel-trace --stats --output-format csv -d "$D" -- python "$ROOT/bench.py" --no-cpu-baseline $FLAGS > "$D.json" 2> "$D.err" )
  STATS="$(find "$D" -name '*kernel_stats.csv' | head -1)"
  echo "== $ENVS  ($FLAGS)" >> "$OUT/summary.txt"
  python - "$STATS" >> "$OUT/summary.txt" <<'PY'
import csv, sys
rows = [r for r in csv.DictReader(open(sys.argv[1])) if "nsr::" in r["Name"]]
rows.sort(key=lambda r: -float(r["Percentage"]))
for r in rows[:14]:
    print("   %-60s %5d %9.1f us %6.2f %%" % (r["Name"][:60], int(r["Calls"]), float(r["AverageNs"]) / 1e3, float(r["Percentage"])))
PY
  python -c "import json,sys; d=json.load(open('$D.json')); print('   value', round(d['value']), 'ms_per_step', round(d['ms_per_step'],4), d.get('kernel_ms'))" >> "$OUT/summary.txt" 2>&1
  cp "$STATS" "$OUT/stats$i.csv" 2>/dev/null; rm -rf "$D"
done
cat "$OUT/summary.txt"
