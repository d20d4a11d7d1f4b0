#!/bin/sh
# Measurement tooling: rocprofv3 --pmc passes over `bench.py --no-cpu-baseline --windows 1 <flags>` (one counter group per run:
# FETCH_SIZE and WRITE_SIZE do not fit one pass; no trace domains next to --pmc), summarised per nsr kernel.
#   sh tools/pmc_bench.sh <tag> [bench.py flags]   ->  gpurun_out/<tag>/<tag>_pmc_bench_kernels.txt, <tag>_traffic.json
set -e
TAG="$1"; shift
ROOT="$(cd "$(dirname "$0")/.." && pwd)"
OUT="$ROOT/gpurun_out/$TAG"
mkdir -p "$OUT"
export TMPDIR=/tmp
: > "$OUT/${TAG}_pmc_bench_kernels.txt"
i=0
for grp in "FETCH_SIZE" "WRITE_SIZE" "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_LDS" "SQ_INSTS_VALU SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAIT_INST_LDS"; do
  i=$((i+1))
  ( cd /tmp && timeout 600 rocprofv3 --pmc $grp --output-format csv -d "$OUT/pmc$i" -- python "$ROOT/bench.py" --no-cpu-baseline --windows 1 --steps 30 --warmup 5 "$@" > "$OUT/pmc$i.json" 2> "$OUT/pmc$i.err" ) || echo "pass $i ($grp) failed" >> "$OUT/${TAG}_pmc_bench_kernels.txt"
  python "$ROOT/tools/pmc_summary.py" "$OUT/pmc$i" "$OUT/${TAG}_pmc_bench_kernels.txt" > /dev/null || true
done
python - "$OUT/${TAG}_pmc_bench_kernels.txt" "$OUT/${TAG}_traffic.json" "$*" <<'PY'
import json, re, sys
cur, vals = None, {}
for ln in open(sys.argv[1]):
    if not ln.startswith(" "):
        cur = ln.strip()
    else:
        m = re.match(r"\s+(\S+)\s+([0-9.e+\-]+)", ln)
        if m and cur:
            vals.setdefault(cur, {})[m.group(1)] = float(m.group(2))
note = ("rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE in separate passes over `bench.py --no-cpu-baseline --windows 1 --steps 30 --warmup 5 %s` (KB units); "
        "FETCH_SIZE doubled per MI355X_MICROARCH.md (gfx950 reports half the bytes of wide reads); WRITE_SIZE uncalibrated" % sys.argv[3])
out = {}
for k, v in vals.items():
    if "FETCH_SIZE" in v and "WRITE_SIZE" in v:
        out[k] = {"FETCH_SIZE_KB": v["FETCH_SIZE"], "WRITE_SIZE_KB": v["WRITE_SIZE"],
                  "hbm_bytes_per_launch": (2 * v["FETCH_SIZE"] + v["WRITE_SIZE"]) * 1024, "note": note}
json.dump(out, open(sys.argv[2], "w"), indent=1)
print(json.dumps({k: round(v["hbm_bytes_per_launch"] / 1e6, 1) for k, v in out.items()}))
PY
