"""CPU tests of the host-side logic: result splitting, status mapping, synthetic workloads,
and the data-parallel helpers (world_size 2 over gloo)."""
import os
import socket
import sys
import time

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

import detr_amd
from detr_amd import synth
from detr_amd.dist import all_reduce_num_boxes, matched_box_counts, shard_range, start_num_boxes_all_reduce
from detr_amd.matcher import HungarianMatcher, _raise_for_status


def test_split_regular_and_ragged():
    sizes, queries = [3, 0, 2], [5, 4]
    S, M = 2, 5
    flat = torch.arange(2 * S * M + 1, dtype=torch.int64)
    off = np.array([0, 3, 3, 5])
    res = HungarianMatcher._split(flat, S, M, sizes, queries, off)
    assert [len(r) for r in res] == [3, 3]
    assert res[0][0][0].tolist() == [0, 1, 2] and res[0][0][1].tolist() == [10, 11, 12]
    assert res[0][1][0].numel() == 0
    assert res[1][2][0].tolist() == [8, 9] and res[1][2][1].tolist() == [18, 19]
    # more targets than queries: only min(Q, T) pairs per image
    res = HungarianMatcher._split(flat, S, M, sizes, [2, 1], off)
    assert res[0][0][0].tolist() == [0, 1] and res[1][0][0].tolist() == [5]
    assert res[1][2][1].tolist() == [18]
    # no targets at all
    res = HungarianMatcher._split(torch.zeros(1, dtype=torch.int64), 2, 0, [0, 0], [5, 4], np.array([0, 0, 0]))
    assert all(a.numel() == 0 and b.numel() == 0 for r in res for a, b in r)


def test_status_precedence_matches_reference_call_order():
    from detr_amd import _lib

    with pytest.raises(IndexError):
        _raise_for_status(_lib.ST_LABEL_RANGE | _lib.ST_DEGENERATE_BOX | _lib.ST_INVALID_COST)
    with pytest.raises(AssertionError):
        _raise_for_status(_lib.ST_DEGENERATE_BOX | _lib.ST_INVALID_COST)
    with pytest.raises(ValueError):
        _raise_for_status(_lib.ST_INVALID_COST)
    with pytest.raises(ValueError):
        _raise_for_status(_lib.ST_INFEASIBLE)
    _raise_for_status(0)


def test_synthetic_workloads_are_deterministic_and_valid():
    for name in ("cfg1", "cfg2", "cfg3", "cfg3_hw"):
        wl, sets, targets = synth.make_workload(name, batch=4)
        wl2, sets2, targets2 = synth.make_workload(name, batch=4)
        assert len(sets) == len(wl.set_queries)
        for a, b in zip(sets, sets2):
            assert torch.equal(a["pred_logits"], b["pred_logits"]) and torch.equal(a["pred_boxes"], b["pred_boxes"])
        for s, q in zip(sets, wl.set_queries):
            assert s["pred_logits"].shape == (4, q, wl.n_logits) and s["pred_boxes"].shape == (4, q, 4)
            bx = s["pred_boxes"]
            assert (bx[..., 2:] > 0).all() and (bx >= 0).all()
        for t in targets:
            assert t["labels"].dtype == torch.int64 and t["boxes"].shape == (len(t["labels"]), 4)
            assert (t["labels"] >= 0).all() and (t["labels"] < wl.n_label_classes).all()
    a = synth.make_workload("cfg2", batch=4, rank=0)[1][0]["pred_logits"]
    b = synth.make_workload("cfg2", batch=4, rank=1)[1][0]["pred_logits"]
    assert not torch.equal(a, b)
    counts = synth.target_counts("heavy_tail", 2000, np.random.default_rng(0))
    assert counts.min() >= 0 and counts.max() <= 100 and counts.mean() > 10


def test_matched_box_counts_and_shards():
    assert matched_box_counts([3, 0, 9], [5, 2]) == [3 + 0 + 5, 2 + 0 + 2]
    covered = []
    for r in range(4):
        lo, hi = shard_range(10, r, 4)
        covered += list(range(lo, hi))
    assert covered == list(range(10))
    assert shard_range(8, 1, 2) == (4, 8)
    # no process group: local counts, clamped
    out = all_reduce_num_boxes([0, 7], "cpu")
    assert out.tolist() == [1.0, 7.0]


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _dp_worker(rank, world, port, results):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        # every rank owns a shard of the images; the matched-box counts are summed in ONE collective
        _, _, targets = synth.make_workload("cfg2", batch=8)
        lo, hi = shard_range(8, rank, world)
        sizes = [len(t["labels"]) for t in targets[lo:hi]]
        counts = matched_box_counts(sizes, [600] * 6 + [10])
        pending = start_num_boxes_all_reduce(counts, "cpu")      # asynchronous: the matcher would run here
        out = pending.result()
        assert pending.result() is out                            # waiting twice is harmless
        results[rank] = out.tolist()
    finally:
        dist.destroy_process_group()


def test_num_boxes_all_reduce_world_size_2():
    world, port = 2, _free_port()
    with mp.Manager() as mgr:
        results = mgr.dict()
        mp.spawn(_dp_worker, args=(world, port, results), nprocs=world, join=True)
        got = [results[r] for r in range(world)]
    _, _, targets = synth.make_workload("cfg2", batch=8)
    sizes = [len(t["labels"]) for t in targets]
    expect = [max(c / world, 1.0) for c in matched_box_counts(sizes, [600] * 6 + [10])]
    assert got[0] == got[1]
    np.testing.assert_allclose(got[0], expect, rtol=1e-6)


def test_balanced_shards_even_out_images_and_targets():
    from detr_amd import synth
    from detr_amd.dist import balanced_shards

    rng = np.random.default_rng(4)
    sizes = synth.target_counts("heavy_tail", 256, rng).tolist()
    for world in (1, 2, 4, 8):
        shards = balanced_shards(sizes, world)
        assert sorted(i for s in shards for i in s) == list(range(256))
        counts = [len(s) for s in shards]
        assert max(counts) - min(counts) <= 1
        loads = [sum(sizes[i] for i in s) for s in shards]
        assert max(loads) - min(loads) <= max(sizes)          # within one heaviest image of each other
        contiguous = [sum(sizes[r * 256 // world:(r + 1) * 256 // world]) for r in range(world)]
        assert max(loads) - min(loads) <= max(contiguous) - min(contiguous) or world == 1
    assert balanced_shards([], 4) == [[], [], [], []]


def test_strong_scaling_shards_cover_the_workload():
    """What `bench.py --scaling strong` gives every rank: its share of the workload's target counts."""
    from detr_amd.dist import balanced_shards

    wl = synth.WORKLOADS["cfg4"]
    counts = synth.target_counts(wl.t_law, wl.batch, np.random.default_rng(wl.seed))
    for world in (2, 4, 8):
        shards = balanced_shards(counts.tolist(), world)
        seen = []
        for rank, mine in enumerate(shards):
            _, sets, targets = synth.make_workload("cfg3", rank=rank, counts=counts[mine][:3])
            assert [len(t["labels"]) for t in targets] == counts[mine][:3].tolist()
            assert sets[0]["pred_logits"].shape[0] == 3
            seen += mine
        assert sorted(seen) == list(range(wl.batch))


def test_bench_without_launcher_reports_a_failed_rank():
    """`python bench.py --gpus 2` starts its own ranks; without a GPU they fail and the parent says so."""
    import subprocess
    import sys

    if torch.cuda.is_available():
        pytest.skip("the ranks would run for real on a GPU box (covered by the gpu tests)")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    run = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0"],
                         capture_output=True, text=True, timeout=120)
    assert run.returncode != 0 and "exited with status" in run.stderr
    assert not [l for l in run.stdout.splitlines() if l.startswith("{")]


def _python_split(flat, S, M, sizes, queries, offsets):
    import detr_amd.matcher as mm

    helper, mm._index_views = mm._index_views, None
    try:
        return HungarianMatcher._split(flat, S, M, sizes, queries, offsets)
    finally:
        mm._index_views = helper


def test_index_view_helper_equals_python_split_and_never_touches_live_tensors():
    """detr_amd._index_views builds the per-image (src, tgt) views in bulk and recycles the tensor
    objects of its previous call -- only those nobody else references any more."""
    import weakref

    from detr_amd.matcher import _index_views

    assert _index_views is not None, "build_ext.py builds detr_amd/_index_views.so"
    rng = np.random.default_rng(5)
    for queries in ([600] * 6, [300, 7, 1050], [4]):
        sizes = rng.poisson(9, 17).tolist() + [0]
        S, M = len(queries), sum(sizes)
        off = np.concatenate([[0], np.cumsum(sizes)]).tolist()
        flat = torch.arange(2 * S * M + 1, dtype=torch.int64)
        got = HungarianMatcher._split(flat, S, M, sizes, queries, off)
        want = _python_split(flat, S, M, sizes, queries, off)
        assert isinstance(got, list) and len(got) == S
        for s in range(S):
            assert isinstance(got[s], list) and len(got[s]) == len(sizes)
            for pair, (w0, w1) in zip(got[s], want[s]):
                assert isinstance(pair, tuple) and len(pair) == 2
                assert pair[0].dtype == torch.int64 and torch.equal(pair[0], w0) and torch.equal(pair[1], w1)
        # a view into a buffer with a storage offset (the matcher's [status | src | tgt] buffer)
        shifted = torch.arange(2 * S * M + 2, dtype=torch.int64)[1:]
        got = HungarianMatcher._split(shifted, S, M, sizes, queries, off)
        want = _python_split(shifted, S, M, sizes, queries, off)
        assert all(torch.equal(a0, b0) and torch.equal(a1, b1)
                   for s in range(S) for (a0, a1), (b0, b1) in zip(got[s], want[s]))

    sizes, queries = [3, 2], [5, 5]
    S, M, off = 2, 5, [0, 3, 5]
    first = torch.arange(21, dtype=torch.int64)
    second = torch.arange(100, 121, dtype=torch.int64)
    r1 = HungarianMatcher._split(first, S, M, sizes, queries, off)
    held = r1[0][0][0]
    r2 = HungarianMatcher._split(second, S, M, sizes, queries, off)     # r1 is alive: nothing of it may change
    assert held.tolist() == [0, 1, 2] and r1[1][1][1].tolist() == [18, 19]
    assert r2[0][0][0].tolist() == [100, 101, 102] and r2[0][0][0] is not held
    del r1
    r3 = HungarianMatcher._split(second, S, M, sizes, queries, off)     # `held` is still referenced
    assert held.tolist() == [0, 1, 2] and r3[0][0][0] is not held
    ids = {id(t) for r in r3 for pair in r for t in pair}
    ref = weakref.ref(r3[1][0][0])
    del r3
    # two generations of objects are used in turn: the second call from here meets r3's objects again
    r4 = HungarianMatcher._split(first, S, M, sizes, queries, off)
    r5 = HungarianMatcher._split(first, S, M, sizes, queries, off)      # now they are recycled ...
    recycled = {id(t) for r in r5 for pair in r for t in pair}
    assert len(ids & recycled) >= 6
    assert ref() is None                                                  # ... except the weakly referenced one
    assert r5[1][1][1].tolist() == [18, 19] and r4[1][1][1].tolist() == [18, 19] and r2[1][1][1].tolist() == [118, 119]
    # the usual loop keeps the previous result while the next is made: objects come back every other call
    seen = []
    result = None
    for k in range(6):
        result = HungarianMatcher._split(first if k % 2 else second, S, M, sizes, queries, off)
        seen.append({id(t) for r in result for pair in r for t in pair})
        assert result[0][0][0].tolist() == ([0, 1, 2] if k % 2 else [100, 101, 102])
    assert len(seen[5] & seen[3]) >= 6


def test_tensor_scans_of_the_host_helper():
    """detr_amd._index_views.scan_targets / same_targets / scan_outputs read tensor metadata in bulk; anything
    unusual is reported as None (the Python code then takes its general path)."""
    from detr_amd.matcher import _index_views as iv

    targets = [{"labels": torch.tensor([1, 0, 1]), "boxes": torch.rand(3, 4)},
               {"labels": torch.zeros(0, dtype=torch.int64), "boxes": torch.zeros(0, 4)},
               {"labels": torch.tensor([0] * 5), "boxes": torch.rand(5, 4)}]
    sizes, total, max_t, pointers, tensors, versions, offsets, meta = iv.scan_targets(targets, -1)   # -1: host tensors
    assert sizes == [3, 0, 5] and total == 8 and max_t == 5 and offsets == [0, 3, 3, 8]
    ptr = np.frombuffer(pointers, dtype=np.uint64)
    assert ptr[0] == targets[0]["labels"].data_ptr() and ptr[5] == targets[2]["boxes"].data_ptr()
    assert tensors[0] is targets[0]["labels"] and tensors[5] is targets[2]["boxes"]
    assert np.frombuffer(meta, dtype=np.int32).tolist() == [0, 3, 3, 8, 2, 0, 1]      # offsets | largest first
    assert iv.same_targets(targets, tensors, versions)
    assert iv.same_targets([dict(t) for t in targets], tensors, versions)            # same tensors in new dicts
    targets[2]["boxes"][0, 0] = 0.5                                                    # in-place change
    assert not iv.same_targets(targets, tensors, versions)
    assert not iv.same_targets(targets[:2], tensors, versions)
    other = [dict(t) for t in targets]
    other[1] = {"labels": targets[1]["labels"].clone(), "boxes": targets[1]["boxes"]}
    assert not iv.same_targets(other, tensors, versions)
    # unusual inputs -> None
    assert iv.scan_targets([{"labels": torch.zeros(2, dtype=torch.int32), "boxes": torch.rand(2, 4)}], -1) is None
    assert iv.scan_targets([{"labels": torch.zeros(2, dtype=torch.int64), "boxes": torch.rand(3, 4)}], -1) is None
    assert iv.scan_targets([{"labels": torch.zeros(2, dtype=torch.int64), "boxes": torch.rand(4, 2).t()}], -1) is None
    assert iv.scan_targets([{"labels": torch.zeros(2, dtype=torch.int64)}], -1) is None
    assert iv.scan_targets(targets, 0) is None                                         # not on cuda:0

    outs = [{"pred_logits": torch.rand(3, 10, 2), "pred_boxes": torch.rand(3, 10, 4)},
            {"pred_logits": torch.rand(3, 70, 2), "pred_boxes": torch.rand(3, 70, 4)}]
    assert iv.scan_outputs(outs, 3, 8) is None                                         # host tensors need the test flag
    index, C, queries, recs, cost_floats = iv.scan_outputs(outs, 3, 8, True)
    assert C == 2 and queries == [10, 70] and cost_floats == 8 * 32 + 8 * 96
    from detr_amd._lib import DmSet
    sets = (DmSet * 2).from_buffer_copy(recs)
    assert sets[0].logits == outs[0]["pred_logits"].data_ptr() and sets[1].boxes == outs[1]["pred_boxes"].data_ptr()
    assert (sets[0].Q, sets[0].q_stride, sets[0].cost_off, sets[0].out_off) == (10, 32, 0, 0)
    assert (sets[1].Q, sets[1].q_stride, sets[1].cost_off, sets[1].out_off) == (70, 96, 256, 8)
    assert iv.scan_outputs(outs, 4, 8, True) is None                                   # batch size mismatch
    assert iv.scan_outputs([{"pred_logits": torch.rand(3, 10, 2).double(), "pred_boxes": torch.rand(3, 10, 4)}], 3, 8, True) is None
    assert iv.scan_outputs([outs[0], {"pred_logits": torch.rand(3, 10, 3), "pred_boxes": torch.rand(3, 10, 4)}], 3, 8, True) is None
    assert iv.scan_outputs([{"pred_logits": torch.rand(3, 10, 2), "pred_boxes": torch.rand(3, 4, 10).transpose(1, 2)}], 3, 8, True) is None


def test_launch_targets_refuses_what_it_cannot_launch():
    """detr_amd._index_views.launch_targets on the CPU: inconsistent arguments raise, host tensors are
    'not this path' (None) -- in neither case is the entry point called (its address here is a dummy)."""
    from detr_amd.matcher import _index_views as iv

    targets = [{"labels": torch.tensor([1, 0, 1]), "boxes": torch.rand(3, 4)},
               {"labels": torch.tensor([0]), "boxes": torch.rand(1, 4)}]
    sizes, total, max_t, pointers, tensors, versions, offsets, meta = iv.scan_targets(targets, -1)
    outs = [{"pred_logits": torch.rand(2, 10, 2), "pred_boxes": torch.rand(2, 10, 4)}]
    dummy = 0xDEAD0
    assert iv.launch_targets(outs, 2, total, max_t, pointers, meta, dummy, dummy, dummy, 1 << 20, 0, 0) is None
    with pytest.raises(ValueError):      # pointer table of another batch size
        iv.launch_targets(outs, 3, total, max_t, pointers, meta, dummy, dummy, dummy, 1 << 20, 0, 0)
    with pytest.raises(ValueError):      # no entry point
        iv.launch_targets(outs, 2, total, max_t, pointers, meta, 0, dummy, dummy, 1 << 20, 0, 0)


def test_index_pairs_recycles_containers_only_when_nobody_holds_them():
    """The helper hands out the lists and tuples of the call before the last again when the caller has dropped
    all of them and the shape is unchanged; a tensor, tuple or list that is still held is never touched."""
    iv = pytest.importorskip("detr_amd._index_views")
    S, B, M = 3, 4, 6
    sizes, offs = [2, 0, 3, 1], [0, 2, 2, 5]

    def make(val):
        flat = torch.arange(2 * S * M + 1, dtype=torch.int64) + val
        return flat, iv.index_pairs(flat, M, sizes, [10] * S, offs)

    def expect(val, s, b, side):
        return [val + side * S * M + s * M + offs[b] + k for k in range(sizes[b])]

    def check(result, val):
        for s in range(S):
            for b in range(B):
                assert result[s][b][0].tolist() == expect(val, s, b, 0)
                assert result[s][b][1].tolist() == expect(val, s, b, 1)

    iv.clear_pool()
    try:
        f1, r1 = make(0)
        f2, r2 = make(100)
        outer, inner = id(r1), [id(x) for x in r1]
        del r1
        f3, r3 = make(200)                       # the containers of the first call come back, around re-pointed tensors
        assert id(r3) == outer and [id(x) for x in r3] == inner
        check(r3, 200)
        check(r2, 100)
        keep = r2[1][2][0]                       # a tensor kept from a dropped result is left alone
        del r2
        f4, r4 = make(300)
        assert keep.tolist() == expect(100, 1, 2, 0) and r4[1][2][0] is not keep
        check(r4, 300)
        held = r3[0]                             # a list of images kept: nothing of that result is reused
        del r3
        f5, r5 = make(400)
        assert held[0][0].tolist() == expect(200, 0, 0, 0) and held[2][1].tolist() == expect(200, 0, 2, 1)
        check(r5, 400)
        r4[0].append(None)                       # a result the caller changed before dropping it
        del r4
        f6, r6 = make(500)
        assert all(len(x) == B for x in r6)
        check(r6, 500)
        other = iv.index_pairs(torch.arange(100, dtype=torch.int64), 3, [1, 2], [5, 5], [0, 1])      # another shape
        assert len(other) == 2 and other[1][1][1].tolist() == [10, 11]
    finally:
        iv.clear_pool()


def test_index_view_helper_is_pinned_to_its_torch_version():
    """The helper reaches into tensor internals; it says which torch it was compiled against and the matcher
    module only keeps it under that very version."""
    import detr_amd.matcher as mm

    assert mm._index_views is not None
    assert mm._index_views.built_for_torch == torch.__version__


def _interleaved_user(name, steps, seed, errors, barrier=None):
    """One caller of the shared helper: its own buffers, holds its latest result while the next is made (as a
    training loop does) and checks every value of it after every call -- another caller's calls in between
    must never show in them."""
    rng = np.random.default_rng(seed)
    held = None
    try:
        for k in range(steps):
            S, B = int(rng.integers(1, 5)), int(rng.integers(1, 9))
            if k % 3:                       # mostly a steady shape, so that recycling really happens
                S, B = 3, 6
            sizes = rng.integers(0, 5, B).tolist()
            queries = rng.choice([2, 3, 50], S).tolist()
            M = sum(sizes)
            off = np.concatenate([[0], np.cumsum(sizes)]).astype(int).tolist()
            flat = torch.arange(2 * S * M + 1, dtype=torch.int64) + 1000 * (k + 1) + seed
            got = HungarianMatcher._split(flat, S, M, sizes, queries, off)
            if barrier is not None and k % 7 == 0:
                time.sleep(0)               # let the other thread in
            for result, base, s_, m_, sz, qs, of in ((got, flat, S, M, sizes, queries, off),) + ((held,) if held else ()):
                for s in range(s_):
                    for b in range(len(sz)):
                        n = min(sz[b], qs[s])
                        lo = s * m_ + of[b]
                        want0 = base[lo:lo + n]
                        want1 = base[s_ * m_ + lo:s_ * m_ + lo + n]
                        a0, a1 = result[s][b]
                        if not (torch.equal(a0, want0) and torch.equal(a1, want1)):
                            errors.append((name, k, s, b))
            held = (got, flat, S, M, sizes, queries, off)
    except Exception as exc:      # noqa: BLE001  (reported to the main thread)
        errors.append((name, repr(exc)))


def test_index_view_helper_shared_by_alternating_callers():
    """Two matchers (Co-DETR: the Hungarian matcher and the auxiliary heads' matcher in one step) use the
    helper's process-global pools in turn, each holding its own previous result: no held tensor ever changes."""
    from detr_amd.matcher import _index_views as iv

    iv.clear_pool()
    errors = []
    users = [dict(rng=np.random.default_rng(s), held=None) for s in (1, 2, 3)]
    for k in range(200):
        for u, user in enumerate(users):
            S, B = (3, 6) if k % 4 else (int(user["rng"].integers(1, 5)), int(user["rng"].integers(1, 9)))
            sizes = user["rng"].integers(0, 5, B).tolist()
            queries = user["rng"].choice([2, 3, 50], S).tolist()
            M = sum(sizes)
            off = np.concatenate([[0], np.cumsum(sizes)]).astype(int).tolist()
            flat = torch.arange(2 * S * M + 1, dtype=torch.int64) + 100000 * u + 100 * k
            got = HungarianMatcher._split(flat, S, M, sizes, queries, off)
            want = _python_split(flat, S, M, sizes, queries, off)
            snapshot = [[(a.clone(), b.clone()) for a, b in per] for per in want]
            for other in users:            # everything anybody still holds is as it was handed out
                if other["held"] is None:
                    continue
                res, snap = other["held"]
                for s in range(len(snap)):
                    for (a0, a1), (w0, w1) in zip(res[s], snap[s]):
                        if not (torch.equal(a0, w0) and torch.equal(a1, w1)):
                            errors.append((k, u))
            for s in range(S):
                for (a0, a1), (w0, w1) in zip(got[s], snapshot[s]):
                    if not (torch.equal(a0, w0) and torch.equal(a1, w1)):
                        errors.append((k, u, "fresh"))
            user["held"] = (got, snapshot)
    iv.clear_pool()
    assert not errors, errors[:5]


def test_index_view_helper_from_two_threads():
    """Two Python threads, each with its own buffers, interleave on the shared pools (every helper call runs
    under the interpreter lock from start to end): each thread's results stay what they were."""
    import threading

    from detr_amd.matcher import _index_views as iv

    iv.clear_pool()
    errors = []
    old = sys.getswitchinterval()
    sys.setswitchinterval(1e-5)           # switch threads as often as the interpreter allows
    try:
        threads = [threading.Thread(target=_interleaved_user, args=("t%d" % i, 400, 10 + i, errors, True))
                   for i in range(2)]
        for t in threads:
            t.start()
        for t in threads:
            t.join(120)
            assert not t.is_alive()
    finally:
        sys.setswitchinterval(old)
        iv.clear_pool()
    assert not errors, errors[:5]


def test_input_dtypes_follow_the_reference():
    """float64 predictions / targets are accepted (rounded to float32), as the reference accepts them; box dtypes
    that differ raise RuntimeError and half precision NotImplementedError, as tests/golden/dtype records for the
    reference (matcher.py:59-90 on such inputs)."""
    conv = HungarianMatcher._float32_inputs
    mk = lambda ld, bd: {"pred_logits": torch.zeros(2, 3, 2, dtype=ld), "pred_boxes": torch.zeros(2, 3, 4, dtype=bd)}
    tg = lambda bd: [{"labels": torch.zeros(1, dtype=torch.int64), "boxes": torch.zeros(1, 4, dtype=bd)} for _ in range(2)]
    outs, t32 = [mk(torch.float32, torch.float32)], tg(torch.float32)
    same_o, same_t = conv(outs, t32)
    assert same_o is outs and same_t is t32                                    # the usual case is not touched
    o, t = conv([mk(torch.float64, torch.float64)], tg(torch.float64))
    assert o[0]["pred_logits"].dtype == o[0]["pred_boxes"].dtype == t[0]["boxes"].dtype == torch.float32
    assert t[0]["labels"].dtype == torch.int64
    o, t = conv([mk(torch.float64, torch.float32)], t32)                       # float64 logits beside float32 boxes
    assert o[0]["pred_logits"].dtype == torch.float32
    with pytest.raises(RuntimeError):
        conv([mk(torch.float64, torch.float64)], t32)
    for half in (torch.float16, torch.bfloat16):
        with pytest.raises(NotImplementedError):
            conv([mk(half, half)], tg(half))
    with pytest.raises(RuntimeError):
        conv([mk(torch.float64, torch.float64), mk(torch.float32, torch.float32)], tg(torch.float64))
