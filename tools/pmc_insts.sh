#!/bin/sh
# Measurement tooling: dynamic instruction counts per nsr kernel (rocprofv3 --pmc, two passes; no trace domain next to --pmc).
#   sh tools/pmc_insts.sh <tag> [bench.py flags]   ->  gpurun_out/<tag>/<tag>_insts.txt
TAG="$1"; shift
ROOT="$(cd "$(dirname "$0")/.." && pwd)"
OUT="$ROOT/gpurun_out/$TAG"
mkdir -p "$OUT"
export TMPDIR=/tmp
: > "$OUT/${TAG}_insts.txt"
i=0
for grp in "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SMEM SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_WAVES" \
           "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_BRANCH"; do
  i=$((i+1))
  ( cd /tmp && timeout 600 rocprofv3 --pmc $grp --output-format csv -d "$OUT/pmc$i" -- python "$ROOT/bench.py" --no-cpu-baseline --windows 1 --steps 30 --warmup 5 "$@" > "$OUT/pmc$i.json" 2> "$OUT/pmc$i.err" ) || echo "pass $i failed" >> "$OUT/${TAG}_insts.txt"
  python "$ROOT/tools/pmc_summary.py" "$OUT/pmc$i" "$OUT/${TAG}_insts.txt" > /dev/null || true
  rm -rf "$OUT/pmc$i"
done
cat "$OUT/${TAG}_insts.txt"
