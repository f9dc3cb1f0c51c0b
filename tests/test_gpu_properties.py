"""Randomised GPU parity: arbitrary small shapes (T = 0, T > Q, C on both sides of the
register/LDS-tile split, odd and even C), both class costs, random weights -- each checked against
scipy on the GPU's own cost (bit-exact) and against the CPU oracle (tie-tolerant)."""
import numpy as np
import pytest
import torch
from hypothesis import HealthCheck, given, settings
from hypothesis import strategies as st
from scipy.optimize import linear_sum_assignment

import golden_io

pytestmark = pytest.mark.gpu


def _boxes(rng, n):
    return np.concatenate([rng.random((n, 2)), rng.random((n, 2)) * 0.3 + 0.01], axis=1).astype(np.float32)


@st.composite
def problems(draw):
    B = draw(st.integers(1, 5))
    Q = draw(st.sampled_from([1, 2, 5, 17, 63, 64, 65, 100, 130, 257]))
    C = draw(st.sampled_from([1, 2, 3, 4, 5, 6, 8, 11, 12, 20, 32, 33, 64, 80, 91, 92, 128]))
    sizes = [draw(st.sampled_from([0, 0, 1, 2, 3, 7, 16, 40, 70])) for _ in range(B)]
    cls = draw(st.sampled_from(["ce", "focal"]))
    w = (draw(st.sampled_from([0.0, 1.0, 2.0])), draw(st.sampled_from([0.0, 1.0, 5.0])),
         draw(st.sampled_from([0.0, 2.0])))
    if w == (0.0, 0.0, 0.0):
        w = (1.0, 5.0, 2.0)
    seed = draw(st.integers(0, 2 ** 31 - 1))
    return B, Q, C, sizes, cls, w, seed


@settings(max_examples=150, deadline=None, suppress_health_check=list(HealthCheck), derandomize=True)
@given(problems())
def test_random_problems_match_oracle_and_scipy(p):
    import detr_amd
    from oracle import oracle

    B, Q, C, sizes, cls, w, seed = p
    rng = np.random.default_rng(seed)
    logits = (rng.standard_normal((B, Q, C)) * 2).astype(np.float32)
    pboxes = _boxes(rng, B * Q).reshape(B, Q, 4)
    labels = [rng.integers(0, C, t).astype(np.int64) for t in sizes]
    tboxes = [_boxes(rng, t) for t in sizes]
    kw = dict(cls_loss=cls, cost_class=w[0], cost_bbox=w[1], cost_giou=w[2], focal_alpha=0.25, focal_gamma=2.0)

    outputs = {"pred_logits": torch.from_numpy(logits).cuda(), "pred_boxes": torch.from_numpy(pboxes).cuda()}
    targets = [{"labels": torch.from_numpy(l).cuda(), "boxes": torch.from_numpy(b).cuda()}
               for l, b in zip(labels, tboxes)]
    matcher = detr_amd.HungarianMatcher(**kw)
    got = matcher(outputs, targets)
    blocks = matcher.cost_blocks(outputs, targets)
    ref, ref_cost = oracle.match(logits, pboxes, labels, tboxes, return_cost=True, **kw)
    assert len(got) == B
    for b in range(B):
        n = min(Q, sizes[b])
        gs, gt = got[b][0].numpy(), got[b][1].numpy()
        assert gs.shape == (n,) and gt.shape == (n,) and gs.dtype == np.int64
        if n == 0:
            continue
        blk = blocks[b].cpu().numpy()
        golden_io.assert_cost_close(blk, ref_cost[b], logits[b], labels[b], cls, w[0], 0.25, 2.0)
        rs, cs = linear_sum_assignment(blk)
        np.testing.assert_array_equal(gs, rs)
        np.testing.assert_array_equal(gt, cs)
        if not (np.array_equal(gs, ref[b][0]) and np.array_equal(gt, ref[b][1])):
            a = golden_io.assignment_cost(ref_cost[b], gs, gt)
            r = golden_io.assignment_cost(ref_cost[b], ref[b][0], ref[b][1])
            assert abs(a - r) <= 1e-5 * max(1.0, abs(r))


def test_non_contiguous_and_offset_inputs():
    """Views with strides / storage offsets are accepted and give the same answer."""
    import detr_amd

    g = golden_io.load("focal_c2_q50")
    outputs, targets = g.torch_inputs("cuda")
    matcher = detr_amd.HungarianMatcher(**g.matcher_kwargs())
    base = matcher(outputs, targets)
    big_logits = torch.zeros(g.B, g.Q, g.C + 3, device="cuda")
    big_logits[..., 1:1 + g.C] = outputs["pred_logits"]
    big_boxes = torch.zeros(g.B, g.Q + 1, 5, device="cuda")
    big_boxes[:, 1:, 1:] = outputs["pred_boxes"]
    views = {"pred_logits": big_logits[..., 1:1 + g.C], "pred_boxes": big_boxes[:, 1:, 1:]}
    assert not views["pred_logits"].is_contiguous()
    got = matcher(views, targets)
    for (a0, a1), (b0, b1) in zip(base, got):
        assert torch.equal(a0, b0) and torch.equal(a1, b1)


def test_match_many_equals_per_set_calls():
    import detr_amd
    from detr_amd import synth

    wl, sets, targets = synth.make_workload("cfg3_hw", device="cuda", batch=6)
    matcher = detr_amd.HungarianMatcher(**wl.matcher_kwargs)
    packed = detr_amd.PackedTargets(targets, torch.device("cuda", 0))
    many = matcher.match_many(sets, packed)
    for s, out in enumerate(sets):
        single = matcher(out, targets)
        for (a0, a1), (b0, b1) in zip(single, many[s]):
            assert torch.equal(a0, b0) and torch.equal(a1, b1)


def test_dtype_and_shape_errors():
    import detr_amd

    g = golden_io.load("ce_c3_q50")
    outputs, targets = g.torch_inputs("cuda")
    matcher = detr_amd.HungarianMatcher(**g.matcher_kwargs())
    with pytest.raises(RuntimeError):      # box dtypes that differ (the reference's torch.cdist raises)
        matcher({"pred_logits": outputs["pred_logits"], "pred_boxes": outputs["pred_boxes"].double()}, targets)
    # float64 logits beside float32 boxes are fine in the reference (the sum is promoted), and here
    same = matcher({"pred_logits": outputs["pred_logits"].double(), "pred_boxes": outputs["pred_boxes"]}, targets)
    for (a0, a1), (b0, b1) in zip(same, matcher(outputs, targets)):
        assert torch.equal(a0, b0) and torch.equal(a1, b1)
    with pytest.raises(ValueError):
        matcher(outputs, targets[:-1])


def test_matcher_survives_deepcopy_and_pickle_after_use():
    """The criterion that owns the matcher is an nn.Module that users copy and checkpoint."""
    import copy
    import pickle

    import detr_amd

    g = golden_io.load("ce_c3_q50")
    outputs, targets = g.torch_inputs("cuda")
    matcher = detr_amd.HungarianMatcher(**g.matcher_kwargs())
    base = matcher(outputs, targets)
    for clone in (copy.deepcopy(matcher), pickle.loads(pickle.dumps(matcher))):
        got = clone(outputs, targets)
        for (a0, a1), (b0, b1) in zip(base, got):
            assert torch.equal(a0, b0) and torch.equal(a1, b1)


@pytest.mark.parametrize("cls_loss,C,gamma", [("focal", 2, 2.0), ("focal", 7, 1.5), ("ce", 3, 2.0), ("ce", 12, 2.0)])
def test_extreme_logits_match_the_oracle(cls_loss, C, gamma):
    """Saturated sigmoids / softmaxes (logits up to +-90) stay finite and agree with the CPU restatement."""
    import detr_amd
    from oracle import oracle

    rng = np.random.default_rng(17)
    B, Q, sizes = 3, 70, [5, 9, 2]
    logits = rng.choice(np.array([-90.0, -30.0, -8.0, 0.0, 8.0, 30.0, 90.0], np.float32), size=(B, Q, C))
    logits += rng.standard_normal((B, Q, C)).astype(np.float32) * 0.01
    boxes = np.stack([_boxes(rng, Q) for _ in range(B)])
    labels = [rng.integers(0, C - (cls_loss == "ce"), t) for t in sizes]
    tboxes = [_boxes(rng, t) for t in sizes]
    kw = dict(cls_loss=cls_loss, cost_class=2.0, cost_bbox=5.0, cost_giou=2.0, focal_alpha=0.25, focal_gamma=gamma)
    ref, ref_cost = oracle.match(logits, boxes, labels, tboxes, return_cost=True, **kw)
    outputs = {"pred_logits": torch.from_numpy(logits).cuda(), "pred_boxes": torch.from_numpy(boxes).cuda()}
    targets = [{"labels": torch.from_numpy(np.asarray(l, np.int64)).cuda(), "boxes": torch.from_numpy(b).cuda()}
               for l, b in zip(labels, tboxes)]
    blocks = detr_amd.HungarianMatcher(**kw).cost_blocks(outputs, targets)
    for b in range(B):
        got = blocks[b].cpu().numpy()
        assert np.isfinite(got).all()
        golden_io.assert_cost_close(got, ref_cost[b], logits[b], labels[b], cls_loss, 2.0, 0.25, gamma)


def test_more_output_sets_than_one_launch_takes():
    """20 output sets (a launch carries at most 16 by value), mixed Q: same result as set-by-set calls;
    the fused consumers agree with the torch expressions on every one of them."""
    import detr_amd
    from detr_amd import losses, synth

    gen = torch.Generator().manual_seed(5)
    _, _, targets = synth.make_workload("cfg1", device="cuda", batch=3)
    sets = []
    for s in range(20):
        q = [40, 75, 130][s % 3]
        sets.append(synth.make_output_set(3, q, 3, gen, device="cuda"))
    matcher = detr_amd.HungarianMatcher("ce", 2.0, 5.0, 2.0)
    many = matcher.match_many(sets, targets)
    flats = matcher.match_flat(sets, targets)
    maps = losses.target_class_maps(flats, 2)
    assert len(many) == 20
    for s, out in enumerate(sets):
        single = matcher(out, targets)
        for (a0, a1), (b0, b1) in zip(single, many[s]):
            assert torch.equal(a0, b0) and torch.equal(a1, b1)
        assert torch.equal(flats[s].src_idx.cpu(), torch.cat([i for i, _ in single]))
        assert torch.equal(maps[s], flats[s].target_classes(2))


def _tail_views(sets, extra_rows):
    """Every output set as the tail of a wider tensor, the way the denoising models hand the matched queries over:
    pred_logits.split([n_dn, Q], dim=1)[1] (/root/reference/models/dino_detr/dino_func.py:164-182)."""
    views = []
    for k, o in enumerate(sets):
        n_dn = extra_rows * (k + 1)
        out = {}
        for key, t in o.items():
            B, Q, K = t.shape
            wide = torch.full((B, n_dn + Q, K), float("nan"), dtype=t.dtype, device=t.device)
            wide[:, n_dn:] = t
            out[key] = wide.split([n_dn, Q], dim=1)[1]
            assert not out[key].is_contiguous() or B == 1
        views.append(out)
    return views


def _same_result(a, b):
    for pa, pb in zip(a, b):
        for (a0, a1), (b0, b1) in zip(pa, pb):
            assert torch.equal(a0.cpu(), b0.cpu()) and torch.equal(a1.cpu(), b1.cpu())


@pytest.mark.parametrize("name,batch,extra", [("cfg3_hw", 8, 10), ("cfg1", 5, 7), ("cfg2", 130, 4), ("cfg4", 6, 9),
                                              ("cfg4_ce", 5, 3), ("wide_ce", 3, 5), ("cfg5", 300, 2)])
def test_tails_of_wider_tensors_are_read_in_place(name, batch, extra):
    """The matched queries of DN-DETR / DINO / Co-DETR are views with the images further apart than Q rows: the
    kernels read them where they are (no copy), in every path -- one kernel, two kernels, the logit-tile and
    the wide-class cost kernels, the wavefront-per-problem kernel -- with the results of the contiguous tensors."""
    import detr_amd
    import detr_amd.matcher as mm
    from detr_amd import synth

    wl, sets, targets = synth.make_workload(name, device="cuda", batch=batch)
    views = _tail_views(sets, extra)
    matcher = detr_amd.HungarianMatcher(**wl.matcher_kwargs)
    # the host helper takes them as they are and records the distance between images
    packed = detr_amd.PackedTargets(targets, torch.device("cuda", torch.cuda.current_device()))
    scan = mm._index_views.scan_outputs(views, packed.batch, packed.total)
    assert scan is not None
    from detr_amd._lib import DmSet
    recs = (DmSet * len(views)).from_buffer_copy(scan[3])
    for r, v in zip(recs, views):
        assert r.logits == v["pred_logits"].data_ptr() and r.logits_stride == (v["pred_logits"].stride(0) if batch > 1 else 0)
        assert r.boxes == v["pred_boxes"].data_ptr() and r.boxes_stride == (v["pred_boxes"].stride(0) if batch > 1 else 0)
    base, base_cost = matcher.match_many(sets, targets, return_cost=True)
    got, got_cost = matcher.match_many(views, targets, return_cost=True)
    _same_result(base, got)
    for a, b in zip(base_cost, got_cost):
        assert torch.equal(a, b)
    _same_result(base, matcher.match_many(views, targets))                          # CPU tuples, the usual call
    _same_result(base, matcher.match_many(views, targets, device_indices=True))
    _same_result([base[1]], [matcher(views[1], targets)])
    for a, b in zip(matcher.match_flat(sets, targets), matcher.match_flat(views, targets)):
        assert torch.equal(a.src_idx, b.src_idx) and torch.equal(a.tgt_flat, b.tgt_flat)
    matcher.through_ops = True                                                      # the registered operators
    _same_result(base, matcher.match_many(views, targets))
    matcher.through_ops = False
    detr_amd._lib.set_knob("no_fuse", 1)
    try:
        _same_result(base, matcher.match_many(views, targets))
    finally:
        detr_amd._lib.set_knob("no_fuse", 0)


def test_strided_views_with_more_targets_than_queries_and_errors():
    import detr_amd
    from detr_amd import synth

    g = golden_io.load("focal_t_gt_q")
    outputs, targets = g.torch_inputs("cuda")
    matcher = detr_amd.HungarianMatcher(**g.matcher_kwargs())
    view = _tail_views([outputs], 3)[0]
    _same_result([matcher(outputs, targets)], [matcher(view, targets)])
    # a NaN in the rows the view does NOT cover is nobody's business; one inside it is
    bad = {k: v.clone() for k, v in outputs.items()}
    bad["pred_boxes"][1, 2, 0] = float("nan")
    with pytest.raises(AssertionError):
        matcher(_tail_views([bad], 3)[0], targets)
    # rows that are not dense inside an image are copied (every other query of a wider tensor)
    wide = {k: torch.repeat_interleave(v, 2, dim=1) for k, v in outputs.items()}
    sparse = {k: v[:, ::2] for k, v in wide.items()}
    assert not sparse["pred_boxes"].is_contiguous()
    _same_result([matcher(outputs, targets)], [matcher(sparse, targets)])
