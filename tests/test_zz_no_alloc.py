"""dm_match_noalloc_f32 (the entry point that allocates nothing: what the other entry points keep in a
caller-allocated workspace lives in a fixed device array of the library) against the two-kernel entry point
dm_match_flat_f32 with a workspace from the caller, through the C ABI.

Every shape here is one whose cost blocks really travel through that array (many classes, the two-kernel
path forced, an image with more targets than queries): dm_match_plan must say so, the array is filled with NaN
before the first call, and after a call its regions must hold, bit for bit, the cost blocks the workspace path
produced.  The second call of a pair is smaller and runs over what the first left behind."""
import ctypes

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _scratch():
    """The library's array on the current GPU as a float32 tensor (no copy)."""
    from detr_amd._lib import lib

    ptr, n = ctypes.c_void_p(), ctypes.c_uint64()
    assert lib.dm_noalloc_scratch(ctypes.byref(ptr), ctypes.byref(n)) == 0
    assert ptr.value and ptr.value % 256 == 0

    class _Array:
        __cuda_array_interface__ = {"shape": (int(n.value),), "typestr": "<f4", "data": (int(ptr.value), False),
                                    "version": 2}

    view = torch.as_tensor(_Array(), device="cuda")
    assert view.data_ptr() == ptr.value and view.numel() == n.value
    return view


def _run_both(wl, sets, targets, first_off=0, poison=False):
    """One call of each entry point on the same inputs.  Returns (M, [(offset, stride, Q) per set])."""
    import detr_amd
    from detr_amd import _lib
    from detr_amd._lib import DmSet, lib

    packed = detr_amd.PackedTargets(targets, torch.device("cuda", torch.cuda.current_device()))
    S, B, M = len(sets), packed.batch, packed.total
    arr = (DmSet * S)()
    off, regions = first_off, []
    for s, o in enumerate(sets):
        q = o["pred_logits"].shape[1]
        stride = (q + 31) & ~31
        arr[s].logits, arr[s].boxes = o["pred_logits"].data_ptr(), o["pred_boxes"].data_ptr()
        arr[s].cost_off, arr[s].out_off, arr[s].Q, arr[s].q_stride = off, s * M, q, stride
        regions.append((off, stride, q))
        off += M * stride
    assert off <= _lib.NOALLOC_MAX_COST_FLOATS
    # the call must be one that sends every set's cost blocks through memory: otherwise the array is a
    # workspace nobody touches and this test checks nothing
    cost_sets = ctypes.c_int(-1)
    assert lib.dm_match_plan(arr, S, B, wl.n_logits, packed.max_t, ctypes.byref(cost_sets)) == 0
    assert cost_sets.value == S, (cost_sets.value, S)
    prm = detr_amd.HungarianMatcher(**wl.matcher_kwargs)._params()
    stream = torch.cuda.current_stream().cuda_stream
    n = 2 * S * M + 1
    C = wl.n_logits
    tgt = (packed.labels.data_ptr(), packed.boxes.data_ptr(), packed.offsets_ptr, packed.order_ptr, packed.max_t)

    # the existing entry point: cost kernel, caller's workspace, assignment kernel
    cost = torch.full((off,), float("nan"), dtype=torch.float32, device="cuda")
    ref = torch.full((n,), -7, dtype=torch.int64, device="cuda")
    ref[-1] = 0
    rc = lib.dm_match_flat_f32(arr, S, B, C, *tgt, ctypes.byref(prm), cost.data_ptr(),
                               ref.data_ptr(), ref.data_ptr() + 8 * S * M, None, ref.data_ptr() + 16 * S * M, stream)
    assert rc == 0, rc

    # the entry point under test: every buffer it is handed exists before the call
    scratch = _scratch()
    if poison:
        scratch.fill_(float("nan"))
    got = torch.full((n,), -7, dtype=torch.int64, device="cuda")
    got[-1] = 0
    torch.cuda.synchronize()
    before = torch.cuda.memory_stats()["allocation.all.allocated"]
    free_before = torch.cuda.mem_get_info()[0]          # the driver's own count: sees any hipMalloc of the library
    rc = lib.dm_match_noalloc_f32(arr, S, B, C, *tgt, M, ctypes.byref(prm),
                                  got.data_ptr(), got.data_ptr() + 8 * S * M, got.data_ptr() + 16 * S * M, stream)
    assert rc == 0, (rc, lib.dm_last_hip_error())
    torch.cuda.synchronize()
    assert torch.cuda.mem_get_info()[0] >= free_before
    assert torch.cuda.memory_stats()["allocation.all.allocated"] == before
    ref_h, got_h = ref.cpu().numpy(), got.cpu().numpy()
    assert int(ref_h[-1]) == 0 and int(got_h[-1]) == 0           # status words
    np.testing.assert_array_equal(got_h, ref_h)                   # src | tgt | status, bit-exact
    pairs = sum(min(t, int(o["pred_logits"].shape[1])) for o in sets for t in packed.sizes)
    assert (got_h[:-1] >= 0).sum() == 2 * pairs                   # every pair written, nothing else touched
    # the array holds this call's cost blocks: the kernels wrote THROUGH the looked-up address
    for lo, stride, q in regions:
        mine = scratch[lo:lo + M * stride].view(M, stride)[:, :q]
        theirs = cost[lo:lo + M * stride].view(M, stride)[:, :q]
        assert not torch.isnan(mine).any()
        assert torch.equal(mine.view(torch.int32), theirs.view(torch.int32))
    return M, regions, off


def _custom(queries, counts, n_logits, cls_loss, seed):
    """A workload record + inputs with the given queries per set and targets per image."""
    from detr_amd import synth

    wl = synth.Workload("custom", len(counts), list(queries), n_logits, n_logits, cls_loss, "poisson16",
                        dict(cls_loss=cls_loss, cost_class=2.0, cost_bbox=5.0, cost_giou=2.0), seed)
    gen = torch.Generator(device="cpu").manual_seed(seed)
    targets = synth.make_targets(counts, n_logits, gen, "cuda")
    sets = [synth.make_output_set(len(counts), q, n_logits, gen, "cuda") for q in queries]
    return wl, sets, targets


def test_many_classes_big_then_small_over_the_same_array():
    """C = 91: cost kernel -> array -> assignment kernel; ~3 M floats, then ~0.5 M over their remains."""
    from detr_amd import synth

    wl, sets, targets = synth.make_workload("cfg4", device="cuda", batch=24)
    big, regions, end_big = _run_both(wl, sets, targets, poison=True)
    stale = _scratch()[:end_big].clone()
    wl, sets, targets = synth.make_workload("cfg4", device="cuda", batch=6, rank=1)
    small, _, end_small = _run_both(wl, sets[:4], targets)
    assert 0 < small < big and end_small < end_big
    # the smaller call wrote nothing beyond its own regions
    assert torch.equal(_scratch()[end_small:end_big].view(torch.int32), stale[end_small:].view(torch.int32))


def test_two_kernel_path_forced_at_two_logits():
    """cfg2 with the one-kernel step switched off: its ~3.7 M floats of cost blocks go through the array."""
    from detr_amd import _lib, synth

    _lib.set_knob("no_fuse", 1)
    try:
        wl, sets, targets = synth.make_workload("cfg2", device="cuda")
        big, _, _ = _run_both(wl, sets, targets, poison=True)
        wl, sets, targets = synth.make_workload("cfg3", device="cuda", batch=8)
        small, _, _ = _run_both(wl, sets[:3], targets)
        assert small < big
    finally:
        _lib.set_knob("no_fuse", 0)


def test_images_with_more_targets_than_queries():
    """T >= Q: the list-order assignment kernel reads its blocks from the array (no knob involved)."""
    wl, sets, targets = _custom([12, 20], [16, 3, 12, 0, 30, 20], 2, "focal", 77)
    _run_both(wl, sets, targets, poison=True)
    wl, sets, targets = _custom([5], [9, 5, 1], 3, "ce", 78)
    _run_both(wl, sets, targets)


def test_a_region_that_ends_exactly_at_the_limit_and_one_beyond():
    from detr_amd import _lib
    from detr_amd._lib import DmParams, DmSet, lib

    wl, sets, targets = _custom([900], [40, 7, 0, 64], 91, "focal", 79)
    M, stride = 111, 928
    first = _lib.NOALLOC_MAX_COST_FLOATS - M * stride
    got_m, regions, end = _run_both(wl, sets, targets, first_off=first, poison=True)
    assert got_m == M and end == _lib.NOALLOC_MAX_COST_FLOATS
    # one float further: rejected, nothing launched (no pointer of the call is looked at)
    arr = (DmSet * 1)()
    arr[0].cost_off, arr[0].out_off, arr[0].Q, arr[0].q_stride = first + 1, 0, 900, stride
    prm = DmParams(1, 2.0, 5.0, 2.0, 0.25, 2.0)
    rc = lib.dm_match_noalloc_f32(arr, 1, 4, 91, None, None, None, None, 64, M, ctypes.byref(prm), None, None, None,
                                  None)
    assert rc == _lib.E_UNSUPPORTED


def test_noalloc_entry_point_rejects_sizes_beyond_its_array():
    from detr_amd import _lib
    from detr_amd._lib import DmParams, DmSet, lib

    arr = (DmSet * 1)()
    arr[0].cost_off, arr[0].out_off, arr[0].Q, arr[0].q_stride = 0, 0, 4096, 4096
    M = _lib.NOALLOC_MAX_COST_FLOATS // 4096 + 1
    prm = DmParams(1, 2.0, 5.0, 2.0, 0.25, 2.0)
    rc = lib.dm_match_noalloc_f32(arr, 1, 1, 2, None, None, None, None, 1, M, ctypes.byref(prm), None, None, None,
                                  None)
    assert rc == _lib.E_UNSUPPORTED
    with pytest.raises(NotImplementedError, match="limit"):
        _lib.check(rc, "dm_match_noalloc_f32")
