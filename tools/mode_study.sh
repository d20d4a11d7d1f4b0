#!/bin/sh
# Measurement tooling: the dX kernel's two modes.  N fresh processes of the same short bench command; per
# process the event-bracketed kernel times of the colour stage and where the iteration's buffers landed (NSR_DEBUG_PTRS=1).
#   sh tools/mode_study.sh <tag> <N> ["ENV=.. ENV=.."]   ->  gpurun_out/<tag>/modes.txt
TAG="$1"; N="${2:-12}"; ENVS="$3"
ROOT="$(cd "$(dirname "$0")/.." && pwd)"
OUT="$ROOT/gpurun_out/$TAG"
mkdir -p "$OUT"
: > "$OUT/modes.txt"
i=0
while [ $i -lt $N ]; do
  i=$((i+1))
  env NSR_DEBUG_PTRS=1 $ENVS python "$ROOT/bench.py" --config 1 --stage color --steps 20 --warmup 4 --windows 1 --no-cpu-baseline \
      --no-strong-record --no-consumed-record > "$OUT/run$i.json" 2> "$OUT/run$i.err"
  python - "$OUT/run$i.json" $i >> "$OUT/modes.txt" <<'PY'
import json, sys
try:
    d = json.loads(open(sys.argv[1]).read().strip().split("\n")[-1])
    k = d["roofline"]["kernels"]
    p = d.get("buffer_ptrs", {}).get("color", {})
    print("run %2s  ms/step %.4f  dx %.1f us  dw %.1f  fwd pass %.1f  bwd %.1f   Z %s (%s B) acts %s FS %s grids %s" % (
        sys.argv[2], d["ms_per_step"], k["dx"]["ms"] * 1e3, k["dw"]["ms"] * 1e3, k["forward_pass"]["ms"] * 1e3,
        d["roofline"]["avg_kernel_ms"] * 1e3, p.get("Z"), p.get("Z_bytes"), p.get("acts"), p.get("FS"), p.get("grids")))
except Exception as e:
    print("run", sys.argv[2], "FAILED", e)
PY
done
cat "$OUT/modes.txt"
