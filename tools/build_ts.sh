#!/bin/sh
# Measurement tooling, not part of the product: libnsr with s_memtime stamps at the phase boundaries of the backward
# kernel (-DNSR_TS, see Dbg in nsr_dev.h).  Used by tests/perf/ts_dx.py / ts_fwd.py through NSR_LIB_PATH.
#   sh tools/build_ts.sh [name [extra -D flags]]     -> nice_slam_amd/_ab/libnsr_<name>.so   (default name: ts)
set -e
ROOT="$(cd "$(dirname "$0")/.." && pwd)"
NAME="${1:-ts}"
[ $# -gt 0 ] && shift
mkdir -p "$ROOT/nice_slam_amd/_ab"
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -munsafe-fp-atomics -fPIC -shared -x hip ${NSR_TS_DEF--DNSR_TS} "$@" \
    "$ROOT/nice_slam_amd/csrc/nsr_api.cpp" -o "$ROOT/nice_slam_amd/_ab/libnsr_$NAME.so"
echo "built $ROOT/nice_slam_amd/_ab/libnsr_$NAME.so"
