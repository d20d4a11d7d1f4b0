"""Build nice_slam_amd/libnsr.so for gfx950 with hipcc (in-tree; the .so travels with the repo snapshot)."""
from __future__ import annotations

import os
import shutil
import subprocess

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
OUT = os.path.join(HERE, "libnsr.so")
SOURCES = ("nsr_api.cpp",)
HEADERS = ("nsr_kernels.h", "nsr_bwd.h", "nsr_layout.h", "nsr_dev.h", "nsr_rt.h", os.path.join("..", "..", "include", "nsr.h"))
# -ffp-contract=off: every fused multiply-add in the kernels is written explicitly (fmaf / MFMA) so that
#   the CPU emulation used by the unit tests and the GPU agree operation by operation.
# -munsafe-fp-atomics: grid-gradient scatter uses hardware global_atomic_add_f32 (coarse-grained memory).
FLAGS = ("--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off", "-munsafe-fp-atomics",
         "-fPIC", "-shared", "-x", "hip")


def hipcc_path() -> str:
    for cand in (os.environ.get("HIPCC"), shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if cand and os.path.exists(cand):
            return cand
    raise RuntimeError("hipcc not found: cannot build libnsr.so")


def is_fresh() -> bool:
    if not os.path.exists(OUT):
        return False
    t = os.path.getmtime(OUT)
    deps = [os.path.join(CSRC, f) for f in SOURCES + HEADERS]
    return all(os.path.getmtime(d) <= t for d in deps)


def build_lib(force: bool = False, verbose: bool = False) -> str:
    if not force and is_fresh():
        return OUT
    cmd = [hipcc_path(), *FLAGS, *[os.path.join(CSRC, s) for s in SOURCES], "-o", OUT]
    if verbose:
        print(" ".join(cmd))
    res = subprocess.run(cmd, capture_output=True, text=True)
    if res.returncode != 0:
        raise RuntimeError("hipcc failed:\n" + res.stdout + res.stderr)
    return OUT


if __name__ == "__main__":
    print(build_lib(force=True, verbose=True))
