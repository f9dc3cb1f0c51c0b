"""CPU checks of the C-ABI library: it loads without a GPU and exports every symbol that
include/detr_match.h declares.  No compute entry point is called here."""
import ctypes
import os
import re

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HEADER = os.path.join(ROOT, "include", "detr_match.h")


def declared_functions():
    text = open(HEADER).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(dm_[a-z0-9_]+)\s*\(", text)))


def test_header_declares_the_entry_points():
    names = declared_functions()
    for required in ("dm_cost_f32", "dm_lap_f32", "dm_match_f32", "dm_abi_version", "dm_strerror"):
        assert required in names


def test_library_loads_and_exports_every_declared_symbol():
    import detr_amd

    lib = ctypes.CDLL(detr_amd.LIB_PATH)
    for name in declared_functions():
        assert hasattr(lib, name), name
    assert sorted(detr_amd._lib.EXPORTS) == declared_functions()


def test_abi_version_and_strerror():
    import detr_amd

    assert detr_amd.lib.dm_abi_version() == 2
    assert detr_amd.lib.dm_strerror(0) == b"ok"
    assert b"invalid" in detr_amd.lib.dm_strerror(-1)
    assert detr_amd.lib.dm_device_count() >= 0


def test_struct_layouts_match_the_header():
    from detr_amd._lib import DmParams, DmSet

    assert ctypes.sizeof(DmSet) == 48      # 2 pointers, 2 int64, 4 int32
    assert DmSet.cost_off.offset == 16 and DmSet.Q.offset == 32 and DmSet.q_stride.offset == 36
    assert DmSet.logits_stride.offset == 40 and DmSet.boxes_stride.offset == 44
    assert ctypes.sizeof(DmParams) == 24


def test_argument_validation_without_gpu():
    """Argument errors are reported before any HIP call is made."""
    import detr_amd
    from detr_amd._lib import DmParams, DmSet

    lib = detr_amd.lib
    sets = (DmSet * 1)()
    prm = DmParams(0, 1.0, 1.0, 1.0, 0.25, 2.0)
    # negative sizes / missing status pointer
    assert lib.dm_cost_f32(sets, -1, 1, 3, None, None, None, 0, ctypes.byref(prm), None, None, None) == -1
    assert lib.dm_lap_f32(sets, 1, 1, None, None, None, 4, None, None, None, None) == -1
    # nothing to do: succeeds without touching the device
    status = (ctypes.c_int32 * 1)(0)
    assert lib.dm_cost_f32(sets, 0, 0, 3, None, None, None, 0, ctypes.byref(prm), None, status, None) == 0
    assert lib.dm_lap_f32(sets, 1, 4, None, None, None, 0, None, None, status, None) == 0
    # the variants with the packed-target index output share the checks
    assert lib.dm_lap_flat_f32(sets, 1, 1, None, None, None, 4, None, None, None, None, None) == -1
    assert lib.dm_lap_flat_f32(sets, 1, 4, None, None, None, 0, None, None, None, status, None) == 0
    assert lib.dm_match_flat_f32(sets, -1, 1, 3, None, None, None, None, 0, ctypes.byref(prm), None, None, None, None,
                                 status, None) == -1
    assert lib.dm_match_flat_f32(sets, 0, 0, 3, None, None, None, None, 0, ctypes.byref(prm), None, None, None, None,
                                 status, None) == 0


def test_noalloc_entry_point_checks_its_limit_without_gpu():
    """A request beyond the library's fixed cost array is refused before any HIP call."""
    from detr_amd import _lib
    from detr_amd._lib import DmParams, DmSet, lib

    arr = (DmSet * 1)()
    arr[0].cost_off, arr[0].out_off, arr[0].Q, arr[0].q_stride = 0, 0, 4096, 4096
    prm = DmParams(1, 2.0, 5.0, 2.0, 0.25, 2.0)
    M = _lib.NOALLOC_MAX_COST_FLOATS // 4096 + 1
    assert lib.dm_match_noalloc_f32(arr, 1, 1, 2, None, None, None, None, 1, M, ctypes.byref(prm), None, None,
                                    None, None) == _lib.E_UNSUPPORTED
    arr[0].cost_off = -4
    assert lib.dm_match_noalloc_f32(arr, 1, 1, 2, None, None, None, None, 1, 8, ctypes.byref(prm), None, None,
                                    None, None) == _lib.E_INVALID_ARG


def test_matcher_refuses_cpu_tensors():
    import pytest
    import torch

    import detr_amd

    m = detr_amd.HungarianMatcher("ce", 1.0, 1.0, 1.0)
    out = {"pred_logits": torch.zeros(1, 4, 3), "pred_boxes": torch.full((1, 4, 4), 0.5)}
    tg = [{"labels": torch.zeros(2, dtype=torch.int64), "boxes": torch.full((2, 4), 0.5)}]
    with pytest.raises(RuntimeError):
        m(out, tg)
    with pytest.raises(AssertionError):
        detr_amd.HungarianMatcher("ce", 0, 0, 0)


def test_index_set_layout_and_consumer_argument_validation():
    import detr_amd
    from detr_amd._lib import DmIndexSet

    assert ctypes.sizeof(DmIndexSet) == 40      # 3 pointers, 1 int64, 2 int32
    assert DmIndexSet.out_off.offset == 24 and DmIndexSet.Q.offset == 32
    lib = detr_amd.lib
    sets = (DmIndexSet * 1)()
    assert lib.dm_pair_loss_f32(sets, -1, 1, 0, None, None, None, None, None, None, None) == -1
    assert lib.dm_pair_loss_f32(sets, 1, 1, 0, None, None, None, None, None, None, None) == -1     # Q = 0
    assert lib.dm_target_classes_i64(sets, 0, 1, 0, None, None, None, None, 0, None) == 0          # nothing to do
    assert lib.dm_atss_f32(sets, 0, 1, 0, 65, None, None, None, None, None) == -3                  # k beyond 64
    assert lib.dm_frcnn_f32(sets, 0, 0, 0, 1, 0.7, 0.3, None, None, None, None, None, None) == 0
    assert lib.dm_frcnn_dense_i64(sets, 1, 1, 0, 0.7, 0.3, None, None, None, None, 91, None, None) == -1   # Q = 0
    assert lib.dm_frcnn_dense_i64(sets, 0, 1, 0, 0.7, 0.3, None, None, None, None, 91, None, None) == 0    # nothing to do


def test_knobs_reject_values_they_do_not_know():
    import detr_amd

    lib = detr_amd.lib
    assert lib.dm_debug_set_knob(99, 1) == -1
    assert lib.dm_debug_set_knob(detr_amd._lib.KNOBS["tile_waves"], 5) == -1
    assert lib.dm_debug_set_knob(detr_amd._lib.KNOBS["cost_tsplit"], 3) == -1
    for name in detr_amd._lib.KNOBS:
        detr_amd._lib.set_knob(name, 0)


def test_match_plan_reports_which_sets_need_the_cost_kernel():
    """dm_match_plan launches nothing: callable without a GPU."""
    import ctypes

    import detr_amd
    from detr_amd._lib import DmSet, lib

    def plan(queries, B, C, max_t):
        arr = (DmSet * len(queries))()
        for s, q in enumerate(queries):
            arr[s].Q, arr[s].q_stride = q, (q + 31) & ~31
        n = ctypes.c_int(-1)
        assert lib.dm_match_plan(arr, len(queries), B, C, max_t, ctypes.byref(n)) == 0
        return n.value

    assert plan([600] * 6, 64, 2, 26) == 0            # one kernel: every workgroup computes its own cost block
    assert plan([600] * 6, 64, 91, 26) == 6           # many classes: cost kernel first
    assert plan([600] * 12, 1024, 2, 28) == 12        # thousands of problems: the one-wavefront kernel reads cost blocks
    assert plan([300] * 6 + [20], 8, 2, 26) == 1      # T >= Q possible in the last set: the list-order kernel needs its blocks
    assert lib.dm_match_plan(None, 1, 1, 2, 1, None) == -1


def test_classification_loss_argument_validation_without_gpu():
    """The limits of dm_focal_loss_f32 / dm_cross_entropy_f32 are checked before any HIP call."""
    import detr_amd
    from detr_amd import _lib

    lib = detr_amd.lib
    ok_ptr = 4096                     # never dereferenced: every call below is refused first
    assert lib.dm_focal_loss_f32(ok_ptr, ok_ptr, -1, 2, 0.25, 2.0, 1.0, 0, 0, ok_ptr, None, ok_ptr, ok_ptr, None) == _lib.E_INVALID_ARG
    assert lib.dm_focal_loss_f32(ok_ptr, ok_ptr, 4, 0, 0.25, 2.0, 1.0, 0, 0, ok_ptr, None, ok_ptr, ok_ptr, None) == _lib.E_INVALID_ARG
    assert lib.dm_focal_loss_f32(ok_ptr, ok_ptr, 4, 2, 0.25, 2.0, 1.0, 0, 0, None, None, ok_ptr, ok_ptr, None) == _lib.E_INVALID_ARG
    assert lib.dm_focal_loss_f32(None, ok_ptr, 4, 2, 0.25, 2.0, 1.0, 0, 0, ok_ptr, None, ok_ptr, ok_ptr, None) == _lib.E_INVALID_ARG
    # logits off a 16-byte boundary
    assert lib.dm_focal_loss_f32(ok_ptr + 4, ok_ptr, 4, 2, 0.25, 2.0, 1.0, 0, 0, ok_ptr, None, ok_ptr, ok_ptr, None) == _lib.E_INVALID_ARG
    # rows * C must stay below 2^31
    assert lib.dm_focal_loss_f32(ok_ptr, ok_ptr, 1 << 30, 2, 0.25, 2.0, 1.0, 0, 0, ok_ptr, None, ok_ptr, ok_ptr, None) == _lib.E_UNSUPPORTED
    assert lib.dm_cross_entropy_f32(ok_ptr, ok_ptr, None, 1 << 30, 2, -100, ok_ptr, None, ok_ptr, ok_ptr, None) == _lib.E_UNSUPPORTED
    assert lib.dm_cross_entropy_f32(ok_ptr, None, None, 8, 3, -100, ok_ptr, None, ok_ptr, ok_ptr, None) == _lib.E_INVALID_ARG
    assert lib.dm_cross_entropy_f32(ok_ptr, ok_ptr, None, 8, 3, -100, ok_ptr, None, None, ok_ptr, None) == _lib.E_INVALID_ARG
    assert _lib.CLS_LOSS_PARTIALS == 4096
