"""The registered operators torch.ops.detr_amd.{match, cost, lap}: same results as the matcher's direct
calls, operator checks (schema, fake tensors, AOT dispatch) and tracing with torch.compile."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _workload(name="cfg3_hw", batch=6):
    import detr_amd
    from detr_amd import synth

    wl, sets, targets = synth.make_workload(name, device="cuda", batch=batch)
    matcher = detr_amd.HungarianMatcher(**wl.matcher_kwargs)
    packed = detr_amd.PackedTargets(targets, torch.device("cuda", torch.cuda.current_device()))
    return wl, sets, targets, matcher, packed


def _op_args(matcher, sets, packed):
    return ([s["pred_logits"] for s in sets], [s["pred_boxes"] for s in sets], packed.labels, packed.boxes,
            packed.meta, packed.total, packed.max_t, 0 if matcher.cls_loss == "ce" else 1, float(matcher.cost_class),
            float(matcher.cost_bbox), float(matcher.cost_giou), float(matcher.focal_alpha), float(matcher.focal_gamma))


def test_operators_equal_the_direct_calls():
    wl, sets, targets, matcher, packed = _workload()
    S, M = len(sets), packed.total
    direct, costs = matcher.match_many(sets, targets, return_cost=True)
    out = torch.ops.detr_amd.match(*_op_args(matcher, sets, packed)).cpu()
    assert out.shape == (2 * S * M + 1,) and int(out[-1]) == 0
    for s in range(S):
        for b, (src, tgt) in enumerate(direct[s]):
            lo = s * M + packed.offsets_list[b]
            assert torch.equal(out[lo:lo + len(src)], src) and torch.equal(out[S * M + lo:S * M + lo + len(tgt)], tgt)
    # cost, then lap on that cost: the two-kernel path as operators
    cost = torch.ops.detr_amd.cost(*_op_args(matcher, sets, packed))
    off = 0
    for s, c in enumerate(costs):
        q, stride = c.shape[1], (c.shape[1] + 31) // 32 * 32
        assert torch.equal(cost[off:off + M * stride].view(M, stride)[:, :q], c)
        off += M * stride
    again = torch.ops.detr_amd.lap(cost, [int(s["pred_logits"].shape[1]) for s in sets], packed.meta, M, packed.max_t)
    assert torch.equal(again.cpu(), out)
    flat = torch.ops.detr_amd.match(*_op_args(matcher, sets, packed), True).cpu()
    assert flat.shape == (3 * S * M + 1,) and torch.equal(flat[:2 * S * M + 1], out)


def test_matcher_through_the_operators():
    wl, sets, targets, matcher, packed = _workload("cfg2", 8)
    direct = matcher.match_many(sets, targets)
    matcher.through_ops = True
    via_ops = matcher.match_many(sets, targets)
    via_ops_cost, costs = matcher.match_many(sets, targets, return_cost=True)
    flats = matcher.match_flat(sets, targets)
    matcher.through_ops = False
    flats_direct = matcher.match_flat(sets, targets)
    for s in range(len(sets)):
        for (a0, a1), (b0, b1), (c0, c1) in zip(direct[s], via_ops[s], via_ops_cost[s]):
            assert torch.equal(a0, b0) and torch.equal(a1, b1) and torch.equal(a0, c0) and torch.equal(a1, c1)
        assert torch.equal(flats[s].src_idx, flats_direct[s].src_idx)
        assert torch.equal(flats[s].tgt_flat, flats_direct[s].tgt_flat)
        assert costs[s].shape == (packed.total, sets[s]["pred_logits"].shape[1])
    # the reference's exceptions also arrive through the operators
    bad = {k: v.clone() for k, v in sets[0].items()}
    bad["pred_boxes"][1, 3, 3] = float("nan")
    matcher.through_ops = True
    with pytest.raises(AssertionError):
        matcher(bad, targets)


def test_opcheck():
    wl, sets, targets, matcher, packed = _workload("cfg3_hw", 3)
    args = _op_args(matcher, sets, packed)
    torch.library.opcheck(torch.ops.detr_amd.match.default, args)
    torch.library.opcheck(torch.ops.detr_amd.match.default, args, {"flat": True})
    torch.library.opcheck(torch.ops.detr_amd.cost.default, args)
    cost = torch.ops.detr_amd.cost(*args)
    torch.library.opcheck(torch.ops.detr_amd.lap.default,
                          (cost, [int(s["pred_logits"].shape[1]) for s in sets], packed.meta, packed.total, packed.max_t))


def test_traced_by_torch_compile():
    """A criterion-shaped function -- match, then gather the matched predictions and targets -- traced as
    ONE graph (the operator is a node in it), and the whole matcher under torch.compile with graph breaks."""
    import detr_amd

    wl, sets, targets, matcher, packed = _workload("cfg2", 4)
    M = packed.total
    args = _op_args(matcher, sets[:1], packed)

    def box_l1(logits, boxes, labels, tboxes, meta):
        idx = torch.ops.detr_amd.match([logits], [boxes], labels, tboxes, meta, *args[5:], True)
        src, flat = idx[:M], idx[2 * M + 1:]
        batch = torch.bucketize(torch.arange(M, device=boxes.device), meta[1:(meta.shape[0] + 1) // 2].long(), right=True)
        return (boxes[batch, src] - tboxes[flat]).abs().sum()

    eager = box_l1(sets[0]["pred_logits"], sets[0]["pred_boxes"], packed.labels, packed.boxes, packed.meta)
    compiled = torch.compile(box_l1, backend="aot_eager", fullgraph=True)
    traced = compiled(sets[0]["pred_logits"], sets[0]["pred_boxes"], packed.labels, packed.boxes, packed.meta)
    assert torch.equal(eager, traced)
    # the reference's own expression on the matcher's tuples
    ref = sum((sets[0]["pred_boxes"][b][i.cuda()] - targets[b]["boxes"][j.cuda()]).abs().sum()
              for b, (i, j) in enumerate(matcher(sets[0], targets)))
    torch.testing.assert_close(eager, ref, rtol=1e-5, atol=1e-5)

    def whole(outputs, tg):
        return matcher(outputs, tg)

    import warnings

    with warnings.catch_warnings(record=True) as caught:
        warnings.simplefilter("always")
        got = torch.compile(whole, backend="aot_eager", fullgraph=False)(sets[0], targets)
    want = matcher(sets[0], targets)
    for (a0, a1), (b0, b1) in zip(got, want):
        assert torch.equal(a0, b0) and torch.equal(a1, b1)
    # the host helper's C functions are not called while tracing: nothing for Dynamo to complain about
    complaints = [str(w.message) for w in caught
                  if "ynamo" in str(w.message) or "_index_views" in str(w.message) or "torch._dynamo" in (w.filename or "")]
    assert not complaints, complaints


def test_max_targets_smaller_than_the_offsets_say():
    """The operators trust two host integers about the device-side offsets; max_targets too small must end in
    the status word (bit 16), not in out-of-bounds LDS accesses -- in every assignment kernel."""
    import detr_amd
    from detr_amd import _lib

    wl, sets, targets, matcher, packed = _workload("cfg2", 8)
    S, M = len(sets), packed.total
    args = list(_op_args(matcher, sets, packed))
    assert packed.max_t >= 8
    args[6] = packed.max_t // 2
    cost = torch.ops.detr_amd.cost(*_op_args(matcher, sets, packed))
    queries = [int(s["pred_logits"].shape[1]) for s in sets]
    for mode in (0, 1, 2, 3):
        _lib.set_knob("lap_kernel", mode)
        try:
            out = torch.ops.detr_amd.match(*args)
            assert int(out[2 * S * M]) & _lib.ST_BAD_SIZES, mode
            out = torch.ops.detr_amd.lap(cost, queries, packed.meta, M, packed.max_t // 2)
            assert int(out[2 * S * M]) & _lib.ST_BAD_SIZES, mode
            ok = torch.ops.detr_amd.lap(cost, queries, packed.meta, M, packed.max_t)
            assert int(ok[2 * S * M]) == 0
        finally:
            _lib.set_knob("lap_kernel", 0)
    with pytest.raises(RuntimeError, match="max_targets"):
        detr_amd.matcher._raise_for_status(_lib.ST_BAD_SIZES)
