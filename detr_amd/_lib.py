"""ctypes binding of libdetr_match.so (declared in include/detr_match.h).

There is no fallback: if the shared object is missing, importing this module
raises, and so does every matcher call.
"""
import ctypes
import os

# torch first: its bundled libamdhip64.so.7 must be the HIP runtime of the process.  Loading
# libdetr_match.so before torch would bind the same SONAME to /opt/rocm's copy, and torch's
# own HSA runtime then fails to find the device.
import torch  # noqa: F401

_PKG = os.path.dirname(os.path.abspath(__file__))
# DETR_AMD_LIB: an instrumented build of the same sources (profiling tools only)
LIB_PATH = os.environ.get("DETR_AMD_LIB") or os.path.join(_PKG, "libdetr_match.so")

ABI_VERSION = 2
MAX_SETS_PER_LAUNCH = 16
NOALLOC_MAX_COST_FLOATS = 8 << 20
CLS_LOSS_PARTIALS = 4096

OK, E_INVALID_ARG, E_LAUNCH, E_UNSUPPORTED = 0, -1, -2, -3
NEED_BUFFERS = 1     # dm_match_targets_f32 without pack buffers: nothing launched
ST_LABEL_RANGE, ST_DEGENERATE_BOX, ST_INVALID_COST, ST_INFEASIBLE, ST_BAD_SIZES = 1, 2, 4, 8, 16


class DmSet(ctypes.Structure):
    _fields_ = [("logits", ctypes.c_void_p), ("boxes", ctypes.c_void_p),
                ("cost_off", ctypes.c_int64), ("out_off", ctypes.c_int64),
                ("Q", ctypes.c_int32), ("q_stride", ctypes.c_int32),
                ("logits_stride", ctypes.c_int32), ("boxes_stride", ctypes.c_int32)]


class DmIndexSet(ctypes.Structure):
    _fields_ = [("pred_boxes", ctypes.c_void_p), ("grad_boxes", ctypes.c_void_p), ("target_cls", ctypes.c_void_p),
                ("out_off", ctypes.c_int64), ("Q", ctypes.c_int32), ("boxes_stride", ctypes.c_int32)]


class DmParams(ctypes.Structure):
    _fields_ = [("cls_mode", ctypes.c_int32), ("w_class", ctypes.c_float), ("w_bbox", ctypes.c_float),
                ("w_giou", ctypes.c_float), ("focal_alpha", ctypes.c_float), ("focal_gamma", ctypes.c_float)]


#: every symbol include/detr_match.h declares
EXPORTS = ["dm_pack_targets", "dm_abi_version", "dm_strerror", "dm_last_hip_error", "dm_device_count",
           "dm_cost_f32", "dm_lap_f32", "dm_match_f32", "dm_lap_flat_f32", "dm_match_flat_f32", "dm_match_fused_f32", "dm_match_targets_f32", "dm_match_noalloc_f32", "dm_noalloc_scratch",
           "dm_profile_enable", "dm_profile_read", "dm_match_plan",
           "dm_debug_set_knob",
           "dm_pair_loss_f32", "dm_pair_loss_backward_f32", "dm_pair_list_loss_f32", "dm_pair_list_backward_f32", "dm_target_classes_i64",
           "dm_focal_loss_f32", "dm_cross_entropy_f32",
           "dm_atss_f32", "dm_frcnn_f32", "dm_frcnn_dense_i64"]


def _load():
    if not os.path.exists(LIB_PATH):
        raise ImportError(
            f"{LIB_PATH} is missing: build the HIP extension with `python build_ext.py` "
            "(there is no CPU fallback for the matching path)")
    lib = ctypes.CDLL(LIB_PATH)
    vp, i32, i64p = ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p
    lib.dm_abi_version.restype = ctypes.c_int
    lib.dm_abi_version.argtypes = []
    lib.dm_strerror.restype = ctypes.c_char_p
    lib.dm_strerror.argtypes = [ctypes.c_int]
    lib.dm_last_hip_error.restype = ctypes.c_char_p
    lib.dm_last_hip_error.argtypes = []
    lib.dm_device_count.restype = ctypes.c_int
    lib.dm_device_count.argtypes = []
    # (the set descriptors are passed as a ctypes array of DmSet or as bytes holding the same records)
    setp, prmp = ctypes.c_void_p, ctypes.c_void_p
    lib.dm_pack_targets.restype = ctypes.c_int
    lib.dm_pack_targets.argtypes = [ctypes.c_char_p, ctypes.c_char_p, i32, vp, vp, vp, vp]
    lib.dm_cost_f32.restype = ctypes.c_int
    lib.dm_cost_f32.argtypes = [setp, i32, i32, i32, vp, vp, vp, i32, prmp, vp, vp, vp]
    lib.dm_lap_f32.restype = ctypes.c_int
    lib.dm_lap_f32.argtypes = [setp, i32, i32, vp, vp, vp, i32, i64p, i64p, vp, vp]
    lib.dm_match_f32.restype = ctypes.c_int
    lib.dm_match_f32.argtypes = [setp, i32, i32, i32, vp, vp, vp, vp, i32, prmp, vp, i64p, i64p, vp, vp]
    lib.dm_lap_flat_f32.restype = ctypes.c_int
    lib.dm_lap_flat_f32.argtypes = [setp, i32, i32, vp, vp, vp, i32, i64p, i64p, i64p, vp, vp]
    lib.dm_match_flat_f32.restype = ctypes.c_int
    lib.dm_match_flat_f32.argtypes = [setp, i32, i32, i32, vp, vp, vp, vp, i32, prmp, vp, i64p, i64p, i64p, vp, vp]
    lib.dm_match_fused_f32.restype = ctypes.c_int
    lib.dm_match_fused_f32.argtypes = lib.dm_match_flat_f32.argtypes
    lib.dm_match_targets_f32.restype = ctypes.c_int
    lib.dm_match_targets_f32.argtypes = [setp, i32, i32, i32, ctypes.c_char_p, ctypes.c_char_p, i32, prmp, vp, vp, vp, vp,
                                         i64p, i64p, i64p, vp, vp, ctypes.POINTER(ctypes.c_int)]
    lib.dm_match_noalloc_f32.restype = ctypes.c_int
    lib.dm_match_noalloc_f32.argtypes = [setp, i32, i32, i32, vp, vp, vp, vp, i32, i32, prmp, i64p, i64p, vp, vp]
    lib.dm_noalloc_scratch.restype = ctypes.c_int
    lib.dm_noalloc_scratch.argtypes = [ctypes.POINTER(ctypes.c_void_p), ctypes.POINTER(ctypes.c_uint64)]
    isetp = ctypes.POINTER(DmIndexSet)
    lib.dm_pair_loss_f32.restype = ctypes.c_int
    lib.dm_pair_loss_f32.argtypes = [isetp, i32, i32, i32, vp, vp, i64p, i64p, vp, vp, vp]
    lib.dm_pair_loss_backward_f32.restype = ctypes.c_int
    lib.dm_pair_loss_backward_f32.argtypes = [isetp, i32, i32, i32, vp, i64p, i64p, vp, vp, vp]
    lib.dm_pair_list_loss_f32.restype = ctypes.c_int
    lib.dm_pair_list_loss_f32.argtypes = [vp, i32, i32, i32, vp, i32, i64p, i64p, i64p, ctypes.c_int64, vp, vp, vp, vp, vp]
    lib.dm_pair_list_backward_f32.restype = ctypes.c_int
    lib.dm_pair_list_backward_f32.argtypes = [vp, vp, i64p, i64p, ctypes.c_int64, i32, i32, vp, vp]
    lib.dm_target_classes_i64.restype = ctypes.c_int
    lib.dm_target_classes_i64.argtypes = [isetp, i32, i32, i32, vp, vp, i64p, i64p, ctypes.c_int64, vp]
    lib.dm_focal_loss_f32.restype = ctypes.c_int
    lib.dm_focal_loss_f32.argtypes = [vp, vp, ctypes.c_int64, i32, ctypes.c_float, ctypes.c_float, ctypes.c_float,
                                      ctypes.c_int64, i32, vp, vp, vp, vp, vp]
    lib.dm_cross_entropy_f32.restype = ctypes.c_int
    lib.dm_cross_entropy_f32.argtypes = [vp, vp, vp, ctypes.c_int64, i32, ctypes.c_int64, vp, vp, vp, vp, vp]
    lib.dm_atss_f32.restype = ctypes.c_int
    lib.dm_atss_f32.argtypes = [isetp, i32, i32, i32, i32, vp, vp, vp, vp, vp]
    lib.dm_frcnn_f32.restype = ctypes.c_int
    lib.dm_frcnn_f32.argtypes = [isetp, i32, i32, i32, i32, ctypes.c_float, ctypes.c_float, vp, vp, vp, vp, vp, vp]
    lib.dm_frcnn_dense_i64.restype = ctypes.c_int
    lib.dm_frcnn_dense_i64.argtypes = [isetp, i32, i32, i32, ctypes.c_float, ctypes.c_float, vp, vp, vp, vp,
                                       ctypes.c_int64, vp, vp]
    lib.dm_match_plan.restype = ctypes.c_int
    lib.dm_match_plan.argtypes = [setp, i32, i32, i32, i32, ctypes.POINTER(ctypes.c_int)]
    lib.dm_debug_set_knob.restype = ctypes.c_int
    lib.dm_debug_set_knob.argtypes = [ctypes.c_int, ctypes.c_int]
    lib.dm_profile_enable.restype = ctypes.c_int
    lib.dm_profile_enable.argtypes = [ctypes.c_int]
    lib.dm_profile_read.restype = ctypes.c_int
    lib.dm_profile_read.argtypes = [ctypes.POINTER(ctypes.c_float), ctypes.POINTER(ctypes.c_float)]
    if lib.dm_abi_version() != ABI_VERSION:
        raise ImportError(f"{LIB_PATH}: ABI version {lib.dm_abi_version()} != {ABI_VERSION}; rebuild it")
    return lib


lib = _load()


#: launch-shape knobs of dm_debug_set_knob (testing / tuning aid; 0 = automatic)
KNOBS = {"lap_kernel": 0, "lap_max_waves": 1, "cost_qpt": 2, "cost_tsplit": 3, "tile_waves": 4, "lap_lds_kb": 5, "lap_wave_budget": 6, "no_fuse": 7, "no_direct": 8, "lap_plain_shape": 9}


def set_knob(name: str, value: int):
    check(lib.dm_debug_set_knob(KNOBS[name], int(value)), f"dm_debug_set_knob({name}, {value})")


def check(rc, what):
    if rc == OK:
        return
    msg = lib.dm_strerror(rc).decode()
    if rc == E_LAUNCH:
        msg += ": " + lib.dm_last_hip_error().decode()
    if rc == E_UNSUPPORTED:
        raise NotImplementedError(
            f"{what}: {msg} (B <= 65535 images, S <= 4096 sets, "
            "Q and T <= 4096 per problem; see include/detr_match.h)")
    raise RuntimeError(f"{what}: {msg}")
