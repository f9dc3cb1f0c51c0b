"""dm_match_noalloc_f32 (the entry point that allocates nothing: its cost blocks live in a fixed
device array of the library) against dm_match_f32 (caller-allocated cost workspace), through the
C ABI: bit-identical indices and status on two calls in a row, the second with a smaller input
(stale contents of the shared array must not matter)."""
import ctypes

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _run_both(name, batch, n_sets):
    import detr_amd
    from detr_amd import _lib, synth
    from detr_amd._lib import DmSet, lib

    wl, sets, targets = synth.make_workload(name, device="cuda", batch=batch)
    sets = sets[:n_sets]
    packed = detr_amd.PackedTargets(targets, torch.device("cuda", torch.cuda.current_device()))
    S, B, M = len(sets), packed.batch, packed.total
    arr = (DmSet * S)()
    off = 0
    for s, o in enumerate(sets):
        q = o["pred_logits"].shape[1]
        stride = (q + 31) & ~31
        arr[s].logits, arr[s].boxes = o["pred_logits"].data_ptr(), o["pred_boxes"].data_ptr()
        arr[s].cost_off, arr[s].out_off, arr[s].Q, arr[s].q_stride = off, s * M, q, stride
        off += M * stride
    assert off <= _lib.NOALLOC_MAX_COST_FLOATS
    prm = detr_amd.HungarianMatcher(**wl.matcher_kwargs)._params()
    stream = torch.cuda.current_stream().cuda_stream
    n = 2 * S * M + 1

    # the existing entry point: cost workspace from the caller
    cost = torch.empty(off, dtype=torch.float32, device="cuda")
    ref = torch.full((n,), -7, dtype=torch.int64, device="cuda")
    ref[-1] = 0
    rc = lib.dm_match_f32(arr, S, B, wl.n_logits, packed.labels.data_ptr(), packed.boxes.data_ptr(),
                          packed.offsets_ptr, packed.order_ptr, packed.max_t, ctypes.byref(prm), cost.data_ptr(),
                          ref.data_ptr(), ref.data_ptr() + 8 * S * M, ref.data_ptr() + 16 * S * M, stream)
    assert rc == 0, rc

    # the entry point under test: every buffer it is handed exists before the call
    got = torch.full((n,), -7, dtype=torch.int64, device="cuda")
    got[-1] = 0
    torch.cuda.synchronize()
    before = torch.cuda.memory_stats()["allocation.all.allocated"]
    rc = lib.dm_match_noalloc_f32(arr, S, B, wl.n_logits, packed.labels.data_ptr(), packed.boxes.data_ptr(),
                                  packed.offsets_ptr, packed.order_ptr, packed.max_t, M, ctypes.byref(prm),
                                  got.data_ptr(), got.data_ptr() + 8 * S * M, got.data_ptr() + 16 * S * M, stream)
    assert rc == 0, (rc, lib.dm_last_hip_error())
    torch.cuda.synchronize()
    assert torch.cuda.memory_stats()["allocation.all.allocated"] == before
    ref_h, got_h = ref.cpu().numpy(), got.cpu().numpy()
    assert int(ref_h[-1]) == 0 and int(got_h[-1]) == 0           # status words
    np.testing.assert_array_equal(got_h, ref_h)                   # src | tgt | status, bit-exact
    pairs = sum(min(t, int(o["pred_logits"].shape[1])) for o in sets for t in packed.sizes)
    assert (got_h[:-1] >= 0).sum() == 2 * pairs                   # every pair written, nothing else touched
    return M


def test_noalloc_entry_point_matches_workspace_entry_point_twice():
    big = _run_both("cfg2", 64, 6)       # 6 sets x 64 images, ~15 MB of cost blocks
    small = _run_both("cfg3", 8, 3)      # then a smaller input over the same array
    assert small < big


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
