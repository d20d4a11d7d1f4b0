#!/bin/sh
# Measurement tooling: which unit is busy while the dX kernel's matrix pipe idles?  rocprofv3 --pmc passes (one counter group per run, no
# trace domain next to --pmc) over ONE bench configuration under a list of environment settings (NSR_X measurement switches, NSR_LIB_PATH
# of an A/B build), summarised per nsr kernel and variant.
#   sh tools/pmc_dx.sh <tag> "<bench flags>" "ENV1=.. ENV2=.." "ENV.." ...   ->  gpurun_out/<tag>/<tag>_pmc_table.txt
# (rocprofv3 --att, the thread trace the round-5 verdict asked for first, needs librocprof-trace-decoder, which this image does not ship.)
TAG="$1"; FLAGS="$2"; shift 2
ROOT="$(cd "$(dirname "$0")/.." && pwd)"
OUT="$ROOT/gpurun_out/$TAG"
mkdir -p "$OUT"
export TMPDIR=/tmp
TABLE="$OUT/${TAG}_pmc_table.txt"
: > "$TABLE"
v=0
for ENVS in "$@"; do
  v=$((v+1))
  echo "##### variant $v: $ENVS   (bench.py --no-cpu-baseline --windows 1 --steps 30 --warmup 5 $FLAGS)" >> "$TABLE"
  i=0
  for grp in \
    "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VMEM" \
    "SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INST_LEVEL_VMEM SQ_VMEM_TA_ADDR_FIFO_FULL SQ_VMEM_TA_CMD_FIFO_FULL SQ_VMEM_WR_TA_DATA_FIFO_FULL SQ_VALU_MFMA_COEXEC_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F32" \
    "SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_INSTS_SALU SQ_INSTS_VALU SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_LDS SQ_INST_CYCLES_VMEM_WR" \
    "TA_TA_BUSY TA_ADDR_STALLED_BY_TC_CYCLES TA_DATA_STALLED_BY_TC_CYCLES TA_FLAT_ATOMIC_WAVEFRONTS" \
    "TCP_PENDING_STALL_CYCLES TCP_ATOMIC_TAGCONFLICT_STALL_CYCLES TCP_TCC_ATOMIC_WITHOUT_RET_REQ TCP_TCR_TCP_STALL_CYCLES" \
    "TCC_ATOMIC TCC_EA0_ATOMIC TCC_EA0_ATOMIC_LEVEL TCC_BUSY" \
    "TCC_EA0_WRREQ_STALL TCC_TAG_STALL TCC_TOO_MANY_EA_WRREQS_STALL TCC_CYCLE" \
    "GRBM_GUI_ACTIVE TCC_EA0_WRREQ TCC_EA0_WRREQ_64B TCC_REQ"; do
    i=$((i+1))
    D="$OUT/v${v}p$i"
    ( cd /tmp && env $ENVS timeout 600 rocprofv3 --pmc $grp --output-format csv -d "$D" -- python "$ROOT/bench.py" --no-cpu-baseline --windows 1 --steps 30 --warmup 5 $FLAGS > "$D.json" 2> "$D.err" ) \
      || echo "   pass $i ($grp) FAILED: $(tail -1 "$D.err" | cut -c1-200)" >> "$TABLE"
    python "$ROOT/tools/pmc_summary.py" "$D" 2>/dev/null | python -c "
import sys
keep = False
for ln in sys.stdin:
    if not ln.startswith(' '):
        keep = ('dx_kernel' in ln) or ('dw_kernel' in ln) or ('fwd_pass' in ln)
    if keep: sys.stdout.write(ln)
" >> "$TABLE"
    rm -rf "$D"
  done
done
cat "$TABLE"
