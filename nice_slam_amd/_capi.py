"""ctypes binding of the C ABI declared in include/nsr.h.

The product loads ``nice_slam_amd/libnsr.so`` (built by ``nice_slam_amd.build`` /
``__graft_entry__.build()`` with hipcc for gfx950) and fails loudly when it is missing: there is
no CPU or PyTorch fallback anywhere in this package.
"""
from __future__ import annotations

import ctypes as C
import os

import torch  # noqa: F401  -- must be loaded first: libnsr.so binds to the HIP runtime (libamdhip64.so.7) torch already mapped

MAX_SAMPLES = 64
ABI_VERSION = 8
STAGE_ID = {"coarse": 0, "middle": 1, "fine": 2, "color": 3}
SLOT_NAMES = ("coarse", "middle", "fine", "color")

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("NSR_LIB_PATH", os.path.join(_HERE, "libnsr.so"))   # override: A/B builds of the same HIP library


class NsrGrid(C.Structure):
    _fields_ = [("feat", C.c_void_p), ("dfeat", C.c_void_p),
                ("Z", C.c_int32), ("Y", C.c_int32), ("X", C.c_int32), ("pad_", C.c_int32),
                ("lo", C.c_double * 3), ("hi", C.c_double * 3)]


class NsrDecoder(C.Structure):
    _fields_ = [("params", C.c_void_p), ("packed", C.c_void_p), ("dparams", C.c_void_p)]


class NsrRenderArgs(C.Structure):
    _fields_ = [("stage", C.c_int32), ("n_samples", C.c_int32), ("n_surface", C.c_int32), ("pad_", C.c_int32),
                ("n_rays", C.c_int64),
                ("rays_o", C.c_void_p), ("rays_d", C.c_void_p), ("gt_depth", C.c_void_p), ("gt_max", C.c_void_p),
                ("bound_lo", C.c_double * 3), ("bound_hi", C.c_double * 3),
                ("t_uniform", C.c_float * MAX_SAMPLES), ("t_surface", C.c_double * MAX_SAMPLES),
                ("grid", NsrGrid * 4), ("dec", NsrDecoder * 4),
                ("depth", C.c_void_p), ("var", C.c_void_p), ("rgb", C.c_void_p), ("raw", C.c_void_p),
                ("zvals", C.c_void_p),
                ("gt_color", C.c_void_p), ("keep", C.c_void_p), ("loss", C.c_void_p), ("dl_depth", C.c_void_p), ("dl_rgb", C.c_void_p),
                ("w_color", C.c_float), ("acts_masks_only", C.c_int32), ("acts", C.c_void_p),
                ("ev_pass_start", C.c_void_p), ("ev_pass_stop", C.c_void_p),
                ("skip_masked", C.c_int32), ("pad2_", C.c_int32),
                ("grad_voxel_mask", C.c_void_p * 4)]


class NsrBwdArgs(C.Structure):
    _fields_ = [("d_depth", C.c_void_p), ("d_var", C.c_void_p), ("d_rgb", C.c_void_p), ("depth", C.c_void_p),
                ("d_rays_o", C.c_void_p), ("d_rays_d", C.c_void_p),
                ("workspace", C.c_void_p), ("workspace_floats", C.c_int64),
                ("max_blocks", C.c_int32), ("overwrite_dparams", C.c_int32),
                ("ev_start", C.c_void_p), ("ev_stop", C.c_void_p), ("grad_scale", C.c_void_p),
                ("loss_grads_from_forward", C.c_int32), ("pad_", C.c_int32),
                ("ev_dx_done", C.c_void_p), ("ev_dw_done", C.c_void_p)]


class NsrFrame(C.Structure):
    _fields_ = [("depth", C.c_void_p), ("color", C.c_void_p), ("c2w", C.c_void_p), ("c2w_stride", C.c_int32), ("pad_", C.c_int32)]


class NsrRows(C.Structure):
    _fields_ = [("grid", C.c_void_p), ("rows", C.c_void_p), ("n_rows", C.c_int64)]


class NsrSpan(C.Structure):
    _fields_ = [("ptr", C.c_void_p), ("n", C.c_int64)]


class NsrAdamSpan(C.Structure):
    _fields_ = [("p", C.c_void_p), ("g", C.c_void_p), ("m", C.c_void_p), ("v", C.c_void_p), ("n", C.c_int64), ("step", C.c_void_p),
                ("lr", C.c_float), ("pad_", C.c_int32)]


class NsrAdamGrid(C.Structure):
    _fields_ = [("p", C.c_void_p), ("g", C.c_void_p), ("m", C.c_void_p), ("v", C.c_void_p), ("voxel_mask", C.c_void_p),
                ("n_voxels", C.c_int64), ("step", C.c_void_p), ("lr", C.c_float), ("pad_", C.c_int32)]


# every symbol include/nsr.h declares: (name, restype, argtypes)
SYMBOLS = (
    ("nsr_version", C.c_int, []),
    ("nsr_last_error", C.c_char_p, []),
    ("nsr_param_count", C.c_int64, [C.c_int]),
    ("nsr_packed_count", C.c_int64, [C.c_int]),
    ("nsr_acts_floats", C.c_int64, [C.c_int, C.c_int64, C.c_int]),
    ("nsr_bwd_workspace_floats", C.c_int64, [C.c_int, C.c_int64, C.c_int, C.c_int]),
    ("nsr_pack_params", C.c_int, [C.c_int, C.c_void_p, C.c_void_p, C.c_void_p]),
    ("nsr_render_fwd", C.c_int, [C.POINTER(NsrRenderArgs), C.c_void_p]),
    ("nsr_render_bwd", C.c_int, [C.POINTER(NsrRenderArgs), C.POINTER(NsrBwdArgs), C.c_void_p]),
    ("nsr_eval_points_fwd", C.c_int, [C.POINTER(NsrRenderArgs), C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p]),
    ("nsr_masked_adam", C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64,
                                  C.c_float, C.c_float, C.c_float, C.c_float, C.c_float, C.c_void_p]),
    ("nsr_get_samples", C.c_int, [C.c_void_p, C.c_int64, C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_int32,
                                  C.c_float, C.c_float, C.c_float, C.c_float, C.c_void_p, C.c_int32,
                                  C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    ("nsr_get_samples_window", C.c_int, [C.c_void_p, C.c_int32, C.c_int64, C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_int32,
                                         C.c_float, C.c_float, C.c_float, C.c_float, C.POINTER(NsrFrame),
                                         C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p,
                                         C.POINTER(C.c_double), C.POINTER(C.c_double), C.c_void_p, C.c_void_p, C.c_void_p]),
    ("nsr_get_samples_window_draw", C.c_int, [C.c_void_p, C.c_void_p, C.c_int32, C.c_int64, C.c_int32, C.c_int32, C.c_int32, C.c_int32,
                                              C.c_int32, C.c_float, C.c_float, C.c_float, C.c_float, C.POINTER(NsrFrame),
                                              C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p,
                                              C.POINTER(C.c_double), C.POINTER(C.c_double), C.c_void_p, C.c_void_p, C.c_void_p]),
    ("nsr_get_samples_window_fused", C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int32, C.c_int64, C.c_int32, C.c_int32, C.c_int32, C.c_int32,
                                               C.c_int32, C.c_float, C.c_float, C.c_float, C.c_float, C.POINTER(NsrFrame),
                                               C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p,
                                               C.POINTER(C.c_double), C.POINTER(C.c_double), C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p]),
    ("nsr_get_samples_window_sharded", C.c_int, [C.c_void_p, C.c_void_p, C.POINTER(C.c_uint64), C.c_int32, C.c_int32, C.c_int64, C.c_int32, C.c_int32,
                                                 C.c_int32, C.c_int32, C.c_int32, C.c_float, C.c_float, C.c_float, C.c_float, C.POINTER(NsrFrame),
                                                 C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p,
                                                 C.POINTER(C.c_double), C.POINTER(C.c_double), C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p]),
    ("nsr_pose_grad", C.c_int, [C.c_void_p, C.c_int32, C.c_int64, C.c_int32, C.c_int32, C.c_int32, C.c_int32,
                                C.c_float, C.c_float, C.c_float, C.c_float, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int32, C.c_void_p]),
    ("nsr_masked_adam_multi", C.c_int, [C.POINTER(NsrAdamGrid), C.c_int32, C.c_double, C.c_double, C.c_double, C.c_int32,
                                        C.c_void_p, C.c_void_p]),
    ("nsr_flat_adam", C.c_int, [C.POINTER(NsrAdamSpan), C.c_int32, C.c_double, C.c_double, C.c_double, C.c_int32, C.c_void_p, C.c_void_p]),
    ("nsr_pack_rows", C.c_int, [C.POINTER(NsrRows), C.c_int32, C.POINTER(NsrSpan), C.c_int32, C.c_void_p, C.c_int32, C.c_void_p]),
    ("nsr_tracking_loss", C.c_int, [C.c_int64, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p,
                                    C.c_int32, C.c_int32, C.c_float, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    ("nsr_camera_from_tensor", C.c_int, [C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    ("nsr_aabb_keep", C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.POINTER(C.c_double), C.POINTER(C.c_double),
                                C.c_void_p, C.c_void_p, C.c_void_p]),
    ("nsr_frustum_workspace_floats", C.c_int64, [C.c_int64]),
    ("nsr_frustum_mask", C.c_int, [C.POINTER(C.c_float), C.POINTER(C.c_float), C.c_double, C.c_double, C.c_double, C.c_double,
                                   C.c_int32, C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p,
                                   C.c_int32, C.c_int32, C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p]),
)


class NsrError(RuntimeError):
    pass


import threading

restore_device = threading.local()      # .idx: the device common._stream() switched away from for the launch in flight


class on_device:
    """Device guard around an operation that makes SEVERAL library calls (a forward with its packing and window kernels, a
    backward): the tensors' device is current from entry to exit -- for every launch, the per-device LDS-attribute cache and the
    event records -- and the caller's device is restored on exit, exceptions included.  Inside the guard ``common._stream`` finds
    the device already current and arms no restore of its own."""

    def __init__(self, device):
        self.idx = torch.device(device).index if device is not None else None
        self.prev = None

    def __enter__(self):
        if self.idx is not None and torch.cuda.is_available():
            cur = torch.cuda.current_device()
            if cur != self.idx:
                self.prev = cur
                torch.cuda.set_device(self.idx)
        return self

    def __exit__(self, *exc):
        if self.prev is not None:
            torch.cuda.set_device(self.prev)
            self.prev = None
        return False


class Lib:
    """A loaded libnsr with typed entry points; ``check`` turns error codes into exceptions."""

    def __init__(self, path: str):
        if not os.path.exists(path):
            raise NsrError(
                f"{path} not found: the HIP extension has not been built. Run "
                f"`python -c 'import __graft_entry__ as g; g.build()'` (needs hipcc); there is no fallback path.")
        self.path = path
        self.cdll = C.CDLL(path)
        for name, res, args in SYMBOLS:
            fn = getattr(self.cdll, name)      # AttributeError here = ABI drift
            fn.restype = res
            fn.argtypes = args
            setattr(self, name, fn)
        if self.nsr_version() != ABI_VERSION:
            raise NsrError(f"{path}: ABI version {self.nsr_version()} != {ABI_VERSION}")

    def check(self, rc: int, what: str = "nsr"):
        prev = getattr(restore_device, "idx", None)
        if prev is not None:                 # the launch is enqueued: back to the caller's current device (common._stream)
            restore_device.idx = None
            torch.cuda.set_device(prev)
        if rc != 0:
            msg = self.nsr_last_error()
            raise NsrError(f"{what} failed: {msg.decode() if msg else rc}")


_lib = None


def get_lib() -> Lib:
    """The product library (lazy: Renderer objects are pickled into spawned processes before first use,
    reference src/NICE_SLAM.py:91,296-301)."""
    global _lib
    if _lib is None:
        _lib = Lib(LIB_PATH)
    return _lib
