"""Build nice_slam_amd/libnsr.so for gfx950 with hipcc (in-tree; the .so travels with the repo snapshot)."""
from __future__ import annotations

import os
import shutil
import subprocess

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
OUT = os.path.join(HERE, "libnsr.so")
SOURCES = ("nsr_api.cpp",)
HEADERS = ("nsr_kernels.h", "nsr_bwd2.h", "nsr_fwd2.h", "nsr_layout.h", "nsr_dev.h", "nsr_rt.h", os.path.join("..", "..", "include", "nsr.h"))
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


STAMP = os.path.join(HERE, "libnsr.srchash")


def source_hash() -> str:
    """sha256 over the flags and every source / header the library is compiled from"""
    import hashlib
    h = hashlib.sha256(" ".join(FLAGS).encode())
    for f in SOURCES + HEADERS:
        with open(os.path.join(CSRC, f), "rb") as fh:
            h.update(f.encode() + b"\0" + fh.read())
    return h.hexdigest()


def is_fresh() -> bool:
    """The library is reused only if it was built from exactly these sources (a content hash written beside it by build_lib:
    modification times do not say so -- an edit during a compile leaves a NEWER library of OLDER sources)."""
    if not os.path.exists(OUT) or not os.path.exists(STAMP):
        return False
    try:
        with open(STAMP) as f:
            return f.read().strip() == source_hash()
    except OSError:
        return False


def build_lib(force: bool = False, verbose: bool = False) -> str:
    if not force and is_fresh():
        return OUT
    stamp = source_hash()                     # of what the compiler is about to read
    cmd = [hipcc_path(), *FLAGS, "-Rpass-analysis=kernel-resource-usage", *[os.path.join(CSRC, s) for s in SOURCES], "-o", OUT]
    if verbose:
        print(" ".join(cmd))
    res = subprocess.run(cmd, capture_output=True, text=True)
    if res.returncode != 0:
        raise RuntimeError("hipcc failed:\n" + res.stdout + res.stderr)
    _write_resources(res.stderr)
    with open(STAMP, "w") as f:
        f.write(stamp + "\n")
    return OUT


RESOURCES = os.path.join(HERE, "libnsr.resources.json")


def _write_resources(remarks: str):
    """Per-kernel register / scratch usage as the compiler reports it (the backward kernels run one wave per SIMD on the
    whole 512-entry register file; a change that pushes them into heavy scratch use shows up here, and in tests/test_capi.py)."""
    import json
    import re
    out, name = {}, None
    for line in remarks.splitlines():
        m = re.search(r"Function Name: (\S+)", line)
        if m:
            name = m.group(1)
            out[name] = {}
            continue
        for key, pat in (("vgprs", r" VGPRs: (\d+)"), ("agprs", r"AGPRs: (\d+)"), ("scratch_bytes_per_lane", r"ScratchSize \[bytes/lane\]: (\d+)"),
                         ("sgprs", r"TotalSGPRs: (\d+)"), ("occupancy_waves_per_simd", r"Occupancy \[waves/SIMD\]: (\d+)")):
            m = re.search(pat, line)
            if m and name:
                out[name][key] = int(m.group(1))
    with open(RESOURCES, "w") as f:
        json.dump(out, f, indent=1, sort_keys=True)


if __name__ == "__main__":
    print(build_lib(force=True, verbose=True))
