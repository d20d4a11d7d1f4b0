#!/bin/sh
# TEST INFRASTRUCTURE: builds tests/emu/libnsr_emu.so -- the kernel sources of nice_slam_amd/csrc compiled
# for the host against the fiber shim in this directory (same C ABI, host pointers).
set -e
HERE="$(cd "$(dirname "$0")" && pwd)"
ROOT="$(cd "$HERE/../.." && pwd)"
CXX="${NSR_EMU_CXX:-/opt/rocm/lib/llvm/bin/clang++}"
[ -x "$CXX" ] || CXX=clang++
BUILD="${NSR_EMU_BUILD:-$HERE/_build}"
OUT="${NSR_EMU_OUT:-$HERE/libnsr_emu.so}"      # (A/B builds of the emulator: tests/perf/parity_causes.py)
mkdir -p "$BUILD"
# the shim directory comes first on the include path so that its nsr_dev.h / nsr_rt.h shadow the HIP ones
cp "$ROOT/nice_slam_amd/csrc/nsr_api.cpp" "$ROOT/nice_slam_amd/csrc/nsr_kernels.h" "$ROOT/nice_slam_amd/csrc/nsr_bwd2.h" "$ROOT/nice_slam_amd/csrc/nsr_fwd2.h" "$ROOT/nice_slam_amd/csrc/nsr_layout.h" "$BUILD/"
sed -i 's#"../../include/nsr.h"#"nsr.h"#' "$BUILD/nsr_kernels.h"
"$CXX" -O1 -std=c++17 -ffp-contract=off -fPIC -shared -pthread $NSR_EMU_DEFS \
    -I"$HERE" -I"$ROOT/include" \
    "$BUILD/nsr_api.cpp" "$HERE/emu_runtime.cpp" -o "$OUT"
echo "built $OUT"
