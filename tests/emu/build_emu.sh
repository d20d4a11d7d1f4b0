#!/bin/sh
# TEST INFRASTRUCTURE: builds tests/emu/libnsr_emu.so -- the kernel sources of nice_slam_amd/csrc compiled
# for the host against the fiber shim in this directory (same C ABI, host pointers).
set -e
HERE="$(cd "$(dirname "$0")" && pwd)"
ROOT="$(cd "$HERE/../.." && pwd)"
CXX="${NSR_EMU_CXX:-/opt/rocm/lib/llvm/bin/clang++}"
[ -x "$CXX" ] || CXX=clang++
mkdir -p "$HERE/_build"
# the shim directory comes first on the include path so that its nsr_dev.h / nsr_rt.h shadow the HIP ones
cp "$ROOT/nice_slam_amd/csrc/nsr_api.cpp" "$ROOT/nice_slam_amd/csrc/nsr_kernels.h" "$ROOT/nice_slam_amd/csrc/nsr_bwd2.h" "$ROOT/nice_slam_amd/csrc/nsr_fwd2.h" "$ROOT/nice_slam_amd/csrc/nsr_layout.h" "$HERE/_build/"
sed -i 's#"../../include/nsr.h"#"nsr.h"#' "$HERE/_build/nsr_kernels.h"
"$CXX" -O1 -std=c++17 -ffp-contract=off -fPIC -shared -pthread $NSR_EMU_DEFS \
    -I"$HERE" -I"$ROOT/include" \
    "$HERE/_build/nsr_api.cpp" "$HERE/emu_runtime.cpp" -o "$HERE/libnsr_emu.so"
echo "built $HERE/libnsr_emu.so"
