"""The one-kernel path (dm_match_fused_f32: every assignment workgroup computes its own cost block)
against the two-kernel path (cost kernel, then assignment kernel): identical indices, bit for bit,
and the same exceptions."""
import numpy as np
import pytest
import torch

import golden_io

pytestmark = pytest.mark.gpu

OK_CASES = [n for n in golden_io.case_names() if golden_io.load(n).error is None]


def _both(matcher, sets, targets, **knobs):
    """match_many through the fused entry (default) and with fusion switched off."""
    import detr_amd

    for k, v in knobs.items():
        detr_amd._lib.set_knob(k, v)
    try:
        fused = matcher.match_many(sets, targets)
        detr_amd._lib.set_knob("no_fuse", 1)
        try:
            plain = matcher.match_many(sets, targets)
        finally:
            detr_amd._lib.set_knob("no_fuse", 0)
    finally:
        for k in knobs:
            detr_amd._lib.set_knob(k, 0)
    return fused, plain


def _assert_identical(fused, plain):
    assert len(fused) == len(plain)
    for fs, ps in zip(fused, plain):
        assert len(fs) == len(ps)
        for (a0, a1), (b0, b1) in zip(fs, ps):
            assert torch.equal(a0, b0) and torch.equal(a1, b1)


@pytest.mark.parametrize("name", OK_CASES)
def test_fused_equals_two_kernel_path_on_goldens(name):
    import detr_amd

    g = golden_io.load(name)
    outputs, targets = g.torch_inputs("cuda")
    matcher = detr_amd.HungarianMatcher(**g.matcher_kwargs())
    for knobs in ({}, {"lap_kernel": 2}, {"lap_kernel": 3}, {"lap_lds_kb": 12}):
        fused, plain = _both(matcher, [outputs], targets, **knobs)
        _assert_identical(fused, plain)
    # and the fused result is the reference's (or an equal-cost optimum)
    for b in range(g.B):
        gs, gt = fused[0][b][0].numpy(), fused[0][b][1].numpy()
        rs, rt = g.indices[b]
        if not (np.array_equal(gs, rs) and np.array_equal(gt, rt)):
            a = golden_io.assignment_cost(g.cost[b], gs, gt)
            r = golden_io.assignment_cost(g.cost[b], rs, rt)
            assert abs(a - r) <= 1e-5 * max(1.0, abs(r))


@pytest.mark.parametrize("name,batch", [("cfg1", None), ("cfg2", None), ("cfg3", None), ("cfg3_hw", None),
                                        ("cfg3_hw2", 16), ("cfg5", None)])
def test_fused_equals_two_kernel_path_on_workloads(name, batch):
    import detr_amd
    from detr_amd import synth

    wl, sets, targets = synth.make_workload(name, device="cuda", batch=batch)
    matcher = detr_amd.HungarianMatcher(**wl.matcher_kwargs)
    fused, plain = _both(matcher, sets, targets)
    _assert_identical(fused, plain)
    if name in ("cfg1", "cfg2"):
        # rows of the cost blocks that do not fit in LDS go through the workspace: same answer
        spill, _ = _both(matcher, sets, targets, lap_lds_kb=10)
        _assert_identical(spill, plain)
        wave, _ = _both(matcher, sets, targets, lap_kernel=3)
        _assert_identical(wave, plain)


@pytest.mark.parametrize("name,batch,queries", [("cfg2", 16, None), ("cfg3_hw", 8, None), ("cfg4", 8, None),
                                                ("cfg2", 8, 832), ("cfg2", 8, 1024),
                                                # enough problems for the automatic shape to leave one wavefront per slice
                                                ("cfg2", 160, None), ("cfg3", 320, None), ("cfg4", 160, None)])
def test_wavefront_counts_do_not_change_the_pairs(name, batch, queries):
    """The workgroup kernel with every wavefront count per problem: wavefronts whose second (third) column
    slice lies past the last query run the solver compiled for fewer columns, and sets narrower than the
    launch's shape leave whole wavefronts without a real column.  Same pairs as the one-wavefront-per-slice
    shape, in the one-kernel and the two-kernel path."""
    import dataclasses

    import detr_amd
    from detr_amd import synth

    if queries is not None:
        base = synth.WORKLOADS[name]
        synth.WORKLOADS[name] = dataclasses.replace(base, set_queries=[queries] * 3)
    try:
        wl, sets, targets = synth.make_workload(name, device="cuda", batch=batch)
    finally:
        if queries is not None:
            synth.WORKLOADS[name] = base
    matcher = detr_amd.HungarianMatcher(**wl.matcher_kwargs)
    want, _ = _both(matcher, sets, targets, lap_plain_shape=1)
    automatic, plain = _both(matcher, sets, targets)
    _assert_identical(automatic, want)
    _assert_identical(plain, want)
    for waves in ((2, 3, 5, 7, 8, 11, 12, 13, 16) if batch <= 16 else ()):
        fused, plain = _both(matcher, sets, targets, lap_plain_shape=waves)
        _assert_identical(fused, want)
        _assert_identical(plain, want)


def test_fused_path_reports_the_reference_errors():
    """Degenerate prediction boxes of an image WITHOUT targets, a label out of range, a NaN logit:
    the fused kernels raise what the two-kernel path raises."""
    import detr_amd
    from detr_amd import synth

    wl, sets, targets = synth.make_workload("cfg2", device="cuda", batch=4)
    matcher = detr_amd.HungarianMatcher(**wl.matcher_kwargs)
    empty = [dict(t) for t in targets]
    empty[2] = {"labels": targets[2]["labels"][:0], "boxes": targets[2]["boxes"][:0]}
    bad = {k: v.clone() for k, v in sets[0].items()}
    bad["pred_boxes"][2, 17, 2] = -0.5            # negative width in the image that has no targets
    with pytest.raises(AssertionError):
        matcher.match_many([bad], empty)
    matcher.match_many([sets[0]], empty)          # the kept status word was cleared
    lab = [dict(t) for t in targets]
    lab[1] = {"labels": targets[1]["labels"].clone(), "boxes": targets[1]["boxes"]}
    lab[1]["labels"][0] = 2
    with pytest.raises(IndexError):
        matcher.match_many([sets[0]], lab)
    nan = {k: v.clone() for k, v in sets[0].items()}
    nan["pred_logits"][0, 5, 1] = float("nan")
    with pytest.raises(ValueError):
        matcher.match_many([nan], targets)
    ok = matcher.match_many([sets[0]], targets)
    assert len(ok[0]) == 4


@pytest.mark.parametrize("name,batch", [("cfg1", None), ("cfg2", None), ("cfg3_hw", 16), ("cfg5", 128), ("cfg5", 160)])
def test_targets_read_in_place_equal_packed_targets(name, batch):
    """A step that runs as one fused kernel reads every image's labels and boxes through the pointers it is handed
    (up to 128 images); with more images, or with the knob, the targets are packed first.  Same indices."""
    import detr_amd
    from detr_amd import synth

    wl, sets, targets = synth.make_workload(name, device="cuda", batch=batch)
    matcher = detr_amd.HungarianMatcher(**wl.matcher_kwargs)
    direct = matcher.match_many(sets, targets)
    was_direct = matcher._packed._unpacked is not None          # the packed arrays were never filled
    assert was_direct == (len(targets) <= 128)
    detr_amd._lib.set_knob("no_direct", 1)
    try:
        packed_first = matcher.match_many(sets, [dict(t) for t in targets[:-1]] + [
            {"labels": targets[-1]["labels"].clone(), "boxes": targets[-1]["boxes"]}])      # (defeats the reuse)
        assert matcher._packed._unpacked is None
    finally:
        detr_amd._lib.set_knob("no_direct", 0)
    _assert_identical(direct, packed_first)
    # the packed arrays appear when somebody asks for them
    p = matcher._packed_for(targets, torch.device("cuda", torch.cuda.current_device()))
    ref = detr_amd.PackedTargets(targets, p.device)
    assert torch.equal(p.labels, ref.labels) and torch.equal(p.boxes, ref.boxes) and torch.equal(p.meta, ref.meta)
    flats = matcher.match_flat(sets, targets)
    for s, f in enumerate(flats):
        assert torch.equal(f.src_idx.cpu(), torch.cat([i for i, _ in direct[s]]))


@pytest.mark.parametrize("name,batch", [("cfg2", 16), ("cfg4", 8), ("cfg5", 8)])
def test_indices_stored_to_host_memory_equal_copied_indices(name, batch):
    """The CPU tuples are views of a pinned host tensor that the kernels either write themselves or that a copy
    fills from a device buffer: same tuples, same exceptions, and a raised status does not leak into the next call."""
    import detr_amd
    from detr_amd import synth

    wl, sets, targets = synth.make_workload(name, device="cuda", batch=batch)
    matcher = detr_amd.HungarianMatcher(**wl.matcher_kwargs)
    assert matcher.indices_to_host
    direct = matcher.match_many(sets, targets)
    assert not direct[0][0][0].is_cuda and direct[0][0][0].is_pinned()
    matcher.indices_to_host = False
    copied = matcher.match_many(sets, targets)
    _assert_identical(direct, copied)
    C = sets[0]["pred_logits"].shape[-1]
    bad = [dict(t) for t in targets]
    bad[0] = {"labels": torch.full_like(targets[0]["labels"], C), "boxes": targets[0]["boxes"]}
    for to_host in (True, False, True):
        matcher.indices_to_host = to_host
        with pytest.raises(IndexError):
            matcher.match_many(sets, bad)
        _assert_identical(direct, matcher.match_many(sets, targets))


def test_one_call_launch_equals_the_general_path():
    """The usual step goes through ONE call of the host helper (output scan, pinned result tensor, entry point);
    everything else through the Python path.  Same tuples; the helper asks for a larger workspace instead of
    launching into a small one, and a step that has to pack its targets does so inside the same call once it has room for them."""
    import detr_amd
    import detr_amd.matcher as mm
    from detr_amd import synth

    wl, sets, targets = synth.make_workload("cfg2", device="cuda", batch=16)
    fast = detr_amd.HungarianMatcher(**wl.matcher_kwargs)
    calls = []
    real = mm._index_views.launch_targets

    class Spy:
        def __getattr__(self, name):
            return getattr(mm._index_views_real, name)

        @staticmethod
        def launch_targets(*a):
            r = real(*a)
            calls.append(r if type(r) is int else (None if r is None else r[2]))
            return r

    mm._index_views_real, mm._index_views = mm._index_views, Spy()
    try:
        a = fast.match_many(sets, targets)
        assert len(calls) == 2 and calls[0] > 0 and calls[1] == 0        # workspace request, then the launch
        b = fast.match_many(sets[:2], targets)                           # smaller step, workspace large enough
        assert calls[2:] == [0]
        # 160 images: more than the kernel-argument table holds -> the call reports that it has to pack, gets room
        # for the packed targets and is made again
        wl5, sets5, targets5 = synth.make_workload("cfg5", device="cuda", batch=160)
        big = detr_amd.HungarianMatcher(**wl5.matcher_kwargs)
        del calls[:]
        c = big.match_many(sets5[:2], targets5)
        assert calls[-2:] == [detr_amd._lib.NEED_BUFFERS, 0] and big._packed._unpacked is None
        n = len(calls)
        c2 = big.match_many(sets5[:2], targets5[:-1] + [dict(targets5[-1], labels=targets5[-1]["labels"].clone())])
        assert calls[n:] == [detr_amd._lib.NEED_BUFFERS, 0]                  # new targets: packed again, in the same call
        _assert_identical(c, c2)
    finally:
        mm._index_views = mm._index_views_real
        del mm._index_views_real
    slow = detr_amd.HungarianMatcher(**wl.matcher_kwargs)
    helper, mm._index_views = mm._index_views, None
    try:
        _assert_identical(a, slow.match_many(sets, targets))
        _assert_identical(b, slow.match_many(sets[:2], targets))
        _assert_identical(c, detr_amd.HungarianMatcher(**wl5.matcher_kwargs).match_many(sets5[:2], targets5))
    finally:
        mm._index_views = helper


def test_output_sets_with_different_class_counts():
    """Heads with different class counts in one call (a 3-logit and a 7-logit head here): one pass per class
    count, results in the caller's order, equal to calling the matcher once per set as the reference does;
    the status of a device-resident call is the OR over the passes."""
    import detr_amd
    from detr_amd import synth

    wl, sets, targets = synth.make_workload("cfg1", device="cuda", batch=5)        # 3 logits, labels < 2
    g = torch.Generator(device="cpu").manual_seed(5)
    wide = synth.make_output_set(5, 130, 7, g, "cuda")
    mixed = [sets[0], wide, sets[1]]
    matcher = detr_amd.HungarianMatcher(**wl.matcher_kwargs)
    together = matcher.match_many(mixed, targets)
    one_by_one = [matcher(o, targets) for o in mixed]
    _assert_identical(together, one_by_one)
    res, costs = matcher.match_many(mixed, targets, return_cost=True)
    _assert_identical(res, one_by_one)
    assert [tuple(c.shape) for c in costs] == [(sum(len(t["labels"]) for t in targets), o["pred_logits"].shape[1]) for o in mixed]
    flats = matcher.match_flat(mixed, targets)
    matcher.check_status()
    for f, ref in zip(flats, one_by_one):
        assert torch.equal(f.src_idx.cpu(), torch.cat([i for i, _ in ref]))
        assert torch.equal(f.tgt_idx.cpu(), torch.cat([j for _, j in ref]))
    # a label that only the 3-logit heads reject
    bad = [dict(t) for t in targets]
    bad[2] = {"labels": torch.full_like(targets[2]["labels"], 5), "boxes": targets[2]["boxes"]}
    assert len(bad[2]["labels"]) > 0
    matcher.match_many([wide], bad)                      # fine for the 7-logit head alone
    with pytest.raises(IndexError):
        matcher.match_many(mixed, bad)
    matcher.match_many(mixed, bad, device_indices=True)
    with pytest.raises(IndexError):
        matcher.check_status()
    matcher.match_flat([wide, sets[0]], bad)
    with pytest.raises(IndexError):
        matcher.check_status()


def test_while_waiting_runs_once_per_call_on_every_path():
    """`match_many(..., while_waiting=fn)`: fn runs exactly once, after the kernels have been enqueued, whichever way the
    call takes -- the one-call launch, the device buffer + copy, the operators, several class counts -- and not at all
    for a call that returns nothing to wait for; the indices are unaffected."""
    import detr_amd
    from detr_amd import synth

    wl, sets, targets = synth.make_workload("cfg1", device="cuda", batch=4)
    g = torch.Generator(device="cpu").manual_seed(9)
    other = synth.make_output_set(4, 90, 5, g, "cuda")
    matcher = detr_amd.HungarianMatcher(**wl.matcher_kwargs)
    plain = matcher.match_many(sets[:2], targets)
    calls = []

    def hook():
        calls.append(torch.cuda.current_stream().query())        # (any host work; the kernel may still be running)

    assert matcher.match_many(sets[:2], targets, while_waiting=hook) is not None and len(calls) == 1
    matcher.indices_to_host = False
    _assert_identical(plain, matcher.match_many(sets[:2], targets, while_waiting=hook))
    assert len(calls) == 2
    matcher.indices_to_host, matcher.through_ops = True, True
    _assert_identical(plain, matcher.match_many(sets[:2], targets, while_waiting=hook))
    assert len(calls) == 3
    matcher.through_ops = False
    mixed = matcher.match_many([sets[0], other, sets[1]], targets, while_waiting=hook)
    assert len(calls) == 4
    _assert_identical([mixed[0], mixed[2]], plain)
    res, _ = matcher.match_many(sets[:2], targets, return_cost=True, while_waiting=hook)
    assert len(calls) == 5
    _assert_identical(res, plain)
    assert matcher.match_many([], targets, while_waiting=hook) == [] and len(calls) == 5
