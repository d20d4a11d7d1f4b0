"""CPU: the C-ABI library builds for gfx950, loads, and exports every symbol include/nsr.h declares.
No compute call is made (there is no GPU here)."""
import os
import re
import shutil

import pytest

from conftest import ROOT


def _declared_symbols():
    txt = open(os.path.join(ROOT, "include", "nsr.h")).read()
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    return sorted(set(re.findall(r"\b(nsr_[a-z_]+)\s*\(", txt)))


@pytest.fixture(scope="module")
def lib():
    if shutil.which("hipcc") is None and not os.path.exists("/opt/rocm/bin/hipcc"):
        pytest.skip("hipcc not available")
    from nice_slam_amd import build, _capi
    build.build_lib()
    return _capi.Lib(_capi.LIB_PATH)


def test_header_symbols_are_bound_and_exported(lib):
    from nice_slam_amd import _capi
    declared = _declared_symbols()
    assert len(declared) >= 10
    assert sorted(n for n, _, _ in _capi.SYMBOLS) == declared      # the ctypes table covers the whole header
    for name in declared:
        assert hasattr(lib.cdll, name), name


def test_sizes_and_error_reporting(lib):
    from nice_slam_amd.layout import param_count
    assert lib.nsr_version() == 8
    # [passes][13 + 10 slots][points padded to 16][16] + d raw [.][4] + fp32 positions [.][4] + positions [.][4] doubles
    assert lib.nsr_acts_floats(0, 1000, 32) == 23 * 32000 * 16 + 32000 * 16
    assert lib.nsr_acts_floats(3, 1000, 48) == 3 * 23 * 48000 * 16 + 48000 * 16
    assert lib.nsr_acts_floats(1, 7, 47) == 23 * 336 * 16 + 336 * 16 and lib.nsr_acts_floats(4, 7, 48) == -1
    assert [lib.nsr_param_count(i) for i in range(4)] == [param_count(s) for s in ("coarse", "middle", "fine", "color")] \
        == [6337, 15800, 20920, 15899]
    # [aux table | forward operand stream | transposed stream of the split backward]
    assert [lib.nsr_packed_count(i) for i in range(4)] == [836 + 6144 + 6144, 836 + 15360 + 15360, 836 + 20480 + 15360, 836 + 15360 + 15360]
    assert lib.nsr_param_count(7) == -1
    # split backward: per pass `cap / passes` partial images (the largest decoder blob each) + as many d _B partials of 288 floats
    assert lib.nsr_bwd_workspace_floats(3, 1000, 48, 7) == 3 * (7 // 3) * (20920 + 288)
    assert lib.nsr_bwd_workspace_floats(0, 1000, 32, 5) == 1 * 5 * (6337 + 288)
    # argument validation happens before any device work
    assert lib.nsr_pack_params(9, None, None, None) != 0
    assert b"slot" in lib.nsr_last_error()
    assert lib.nsr_render_fwd(None, None) != 0 and b"null" in lib.nsr_last_error()


def test_ctypes_struct_layout_matches_header():
    """sizeof of the mirrored structs, computed independently from the header with the C compiler."""
    import ctypes
    import subprocess
    import tempfile
    from nice_slam_amd import _capi
    src = '#include <stdio.h>\n#include "nsr.h"\nint main(){printf("%zu %zu %zu %zu\\n", sizeof(nsr_grid), sizeof(nsr_decoder), ' \
          'sizeof(nsr_render_args), sizeof(nsr_bwd_args));return 0;}\n'
    with tempfile.TemporaryDirectory() as d:
        open(os.path.join(d, "t.c"), "w").write(src)
        subprocess.run(["gcc", "-I", os.path.join(ROOT, "include"), os.path.join(d, "t.c"), "-o", os.path.join(d, "t")], check=True)
        out = subprocess.run([os.path.join(d, "t")], capture_output=True, text=True, check=True).stdout.split()
    assert [int(v) for v in out] == [ctypes.sizeof(_capi.NsrGrid), ctypes.sizeof(_capi.NsrDecoder),
                                     ctypes.sizeof(_capi.NsrRenderArgs), ctypes.sizeof(_capi.NsrBwdArgs)]


def test_product_fails_loudly_without_gpu():
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    import nice_slam_amd as nsa
    from nice_slam_amd._capi import NsrError
    from scene_util import make_scene, build_product
    sc = make_scene(seed=1, n_rays=4, small=True)
    renderer, dec, grids = build_product(sc, "cpu")
    with pytest.raises(NsrError):
        renderer.render_batch_ray(grids, dec, sc["rays_d"], sc["rays_o"], "cpu", "middle", gt_depth=sc["gt_depth"])
    with pytest.raises(NsrError):
        nsa.get_samples(0, 8, 0, 8, 4, 48, 64, 60., 60., 31.5, 23.5, sc["c2w"], sc["depth_img"], sc["color_img"], "cpu")


def test_ctypes_struct_fields_follow_the_header():
    """Field names and order of the mirrored structs, parsed from include/nsr.h."""
    from nice_slam_amd import _capi
    txt = re.sub(r"/\*.*?\*/", "", open(os.path.join(ROOT, "include", "nsr.h")).read(), flags=re.S)
    for cname, ctype in (("nsr_grid", _capi.NsrGrid), ("nsr_decoder", _capi.NsrDecoder),
                         ("nsr_render_args", _capi.NsrRenderArgs), ("nsr_bwd_args", _capi.NsrBwdArgs),
                         ("nsr_adam_span", _capi.NsrAdamSpan), ("nsr_adam_grid", _capi.NsrAdamGrid)):
        body = re.search(r"typedef struct %s \{(.*?)\} %s;" % (cname, cname), txt, re.S).group(1)
        names = []
        for stmt in body.split(";"):
            for decl in stmt.split(","):
                m = re.search(r"(\w+)\s*(?:\[[^\]]*\])?\s*$", decl.strip())
                if m:
                    names.append(m.group(1))
        assert names == [f[0] for f in ctype._fields_], cname


def test_kernel_register_budget():
    """The build records what the compiler made of every kernel.  The split backward (csrc/nsr_bwd2.h) replaced a kernel that
    ran one wave per SIMD on the whole register file: its dX kernels must keep a forward-like budget (>= 3 waves/SIMD) and
    the dW kernels stay small (10 waves per block fit, no scratch).  A change that tips one of them over is a slowdown no
    numerical test notices -- it shows up here."""
    import json
    import os
    from nice_slam_amd import build
    if not os.path.exists(build.RESOURCES):
        pytest.skip("libnsr.resources.json not present (library built by an older build.py)")
    res = json.load(open(build.RESOURCES))
    bwd = {k: v for k, v in res.items() if "render_bwd_kernel" in k}
    dx = {k: v for k, v in res.items() if "render_bwd_dx_kernel" in k}
    dw = {k: v for k, v in res.items() if "render_bwd_dw_kernel" in k}
    fwd = {k: v for k, v in res.items() if "render_fwd_kernel" in k or "eval_points_kernel" in k}
    assert len(bwd) == 0 and len(dx) == 8 and len(dw) == 4 and len(fwd) == 8      # fwd: 4 stages + eval_points: 4 (the re-run backward kernel is gone)
    for k, v in dx.items():
        assert v["occupancy_waves_per_simd"] >= 3 and v["scratch_bytes_per_lane"] <= 192, (k, v)
    for k, v in dw.items():
        assert v["occupancy_waves_per_simd"] >= 3 and v["scratch_bytes_per_lane"] == 0, (k, v)
    for k, v in fwd.items():
        assert v["occupancy_waves_per_simd"] >= 3 and v["scratch_bytes_per_lane"] <= 96, (k, v)
    fwp = {k: v for k, v in res.items() if "render_fwd_pass_kernel" in k}          # the three-launch forward (csrc/nsr_fwd2.h)
    assert len(fwp) == 4
    for k, v in fwp.items():
        assert v["occupancy_waves_per_simd"] >= 3 and v["scratch_bytes_per_lane"] <= 128, (k, v)
