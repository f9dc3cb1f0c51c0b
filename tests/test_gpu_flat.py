"""Flat device-resident matches and one-upload target packing, checked against what the
reference's criterion derives from the per-image CPU tuples (models/detr/loss.py:71-75,
97-133): the gather indices, the matched target boxes / labels and the dense class map."""
import numpy as np
import pytest
import torch

import golden_io

pytestmark = pytest.mark.gpu


def _criterion_view(indices, targets, batch, queries, no_object):
    """What the reference's losses build from a list of (src, tgt) tuples, computed on the CPU."""
    batch_idx = torch.cat([torch.full_like(src, b) for b, (src, _) in enumerate(indices)])
    src_idx = torch.cat([src for src, _ in indices])
    boxes = torch.cat([t["boxes"].cpu()[j] for t, (_, j) in zip(targets, indices)])
    labels = torch.cat([t["labels"].cpu()[j] for t, (_, j) in zip(targets, indices)])
    dense = torch.full((batch, queries), no_object, dtype=torch.int64)
    dense[batch_idx, src_idx] = labels
    return batch_idx, src_idx, boxes, labels, dense


@pytest.mark.parametrize("name,batch", [("cfg1", 8), ("cfg2", 16), ("cfg3_hw", 6), ("cfg5", 4)])
def test_match_flat_equals_criterion_view_of_tuples(name, batch):
    import detr_amd
    from detr_amd import synth

    wl, sets, targets = synth.make_workload(name, device="cuda", batch=batch)
    matcher = detr_amd.HungarianMatcher(**wl.matcher_kwargs)
    tuples = matcher.match_many(sets, targets)
    flats = matcher.match_flat(sets, targets)
    matcher.check_status()
    assert len(flats) == len(sets)
    for s, (flat, out) in enumerate(zip(flats, sets)):
        B, Q, C = out["pred_logits"].shape
        bi, si, boxes, labels, dense = _criterion_view(tuples[s], targets, B, Q, C - 1)
        fb, fs = flat.src_permutation_idx()
        assert fb.is_cuda and fs.is_cuda and fb.dtype == torch.int64
        assert flat.num_matched == bi.shape[0]
        assert torch.equal(fb.cpu(), bi) and torch.equal(fs.cpu(), si)
        assert torch.equal(flat.tgt_idx.cpu(), torch.cat([j for _, j in tuples[s]]))
        assert torch.equal(flat.target_boxes().cpu(), boxes)
        assert torch.equal(flat.target_labels().cpu(), labels)
        assert torch.equal(flat.target_classes(C - 1).cpu(), dense)
        # the gathers the losses perform give the same tensors either way
        assert torch.equal(out["pred_boxes"][fb, fs].cpu(), out["pred_boxes"].cpu()[bi, si])


def test_match_flat_more_targets_than_queries():
    """Images with T > Q match only Q of their targets: the flat view drops the unused slots."""
    import detr_amd

    g = golden_io.load("ce_t_gt_q")
    outputs, targets = g.torch_inputs("cuda")
    matcher = detr_amd.HungarianMatcher(**g.matcher_kwargs())
    tuples = matcher(outputs, targets)
    flat = matcher.match_flat([outputs], targets)[0]
    bi, si, boxes, labels, dense = _criterion_view(tuples, targets, g.B, g.Q, g.C - 1)
    assert torch.equal(flat.batch_idx.cpu(), bi) and torch.equal(flat.src_idx.cpu(), si)
    assert torch.equal(flat.target_boxes().cpu(), boxes)
    assert torch.equal(flat.target_labels().cpu(), labels)
    assert torch.equal(flat.target_classes(g.C - 1).cpu(), dense)


def test_match_flat_no_targets_and_status():
    import detr_amd
    from detr_amd import synth

    wl, sets, targets = synth.make_workload("cfg1", device="cuda", batch=3)
    empty = [{"labels": t["labels"][:0], "boxes": t["boxes"][:0]} for t in targets]
    matcher = detr_amd.HungarianMatcher(**wl.matcher_kwargs)
    flat = matcher.match_flat(sets[:1], empty)[0]
    assert flat.num_matched == 0 and flat.target_boxes().shape == (0, 4)
    Q, C = sets[0]["pred_logits"].shape[1:]
    assert bool((flat.target_classes(C - 1) == C - 1).all())
    # an out-of-range label is reported by check_status(), not by the launch
    bad = [{"labels": t["labels"].clone(), "boxes": t["boxes"]} for t in targets]
    bad[1]["labels"][0] = C + 5
    matcher.match_flat(sets[:1], bad)
    with pytest.raises(IndexError):
        matcher.check_status()


@pytest.mark.parametrize("name,batch", [("cfg2", 16), ("cfg3_hw", 5)])
def test_packed_targets_from_host(name, batch):
    """One pinned upload of CPU targets gives the same packed tensors and the same matches as
    packing targets that were moved to the device one tensor at a time."""
    import detr_amd
    from detr_amd import synth

    wl, sets, targets = synth.make_workload(name, device="cuda", batch=batch)
    cpu_targets = [{k: v.cpu() for k, v in t.items()} for t in targets]
    dev = torch.device("cuda", 0)
    a = detr_amd.PackedTargets(targets, dev)
    b = detr_amd.PackedTargets.from_host(cpu_targets, dev)
    assert a.sizes == b.sizes and a.total == b.total and a.max_t == b.max_t
    assert torch.equal(a.labels, b.labels) and torch.equal(a.boxes, b.boxes) and torch.equal(a.meta, b.meta)
    np.testing.assert_array_equal(a.offsets_host, b.offsets_host)
    assert b.boxes.data_ptr() % 16 == 0 and b.meta.data_ptr() % 4 == 0
    matcher = detr_amd.HungarianMatcher(**wl.matcher_kwargs)
    ra, rb = matcher.match_many(sets, a), matcher.match_many(sets, b)
    for sa, sb in zip(ra, rb):
        for (a0, a1), (b0, b1) in zip(sa, sb):
            assert torch.equal(a0, b0) and torch.equal(a1, b1)


def test_packed_targets_from_host_empty_batch_member():
    import detr_amd

    dev = torch.device("cuda", 0)
    tg = [{"labels": torch.zeros(0, dtype=torch.int64), "boxes": torch.zeros(0, 4)},
          {"labels": torch.tensor([1, 0]), "boxes": torch.tensor([[0.5, 0.5, 0.2, 0.2], [0.3, 0.3, 0.1, 0.4]])}]
    p = detr_amd.PackedTargets.from_host(tg, dev)
    assert p.sizes == [0, 2] and p.offsets_host.tolist() == [0, 0, 2]
    none = detr_amd.PackedTargets.from_host([tg[0], tg[0]], dev)
    assert none.total == 0 and none.max_t == 0


def test_pack_kernel_equals_the_torch_packing():
    """The usual inputs are packed by dm_pack_targets (one launch, pointers as kernel arguments); unusual ones by
    torch.cat and an upload.  Same labels, boxes, offsets and order either way -- also with empty images and with
    more images than one launch takes."""
    import detr_amd
    import detr_amd.matcher as mm
    from detr_amd import synth

    device = torch.device("cuda", torch.cuda.current_device())
    for name, batch in (("cfg2", 7), ("cfg4", 256), ("cfg1", 1)):
        _, _, targets = synth.make_workload(name, device="cuda", batch=batch)
        if batch == 7:
            targets[3] = {"labels": targets[3]["labels"][:0], "boxes": targets[3]["boxes"][:0]}
        fast = detr_amd.PackedTargets(targets, device)
        assert fast._scanned is not None
        helper, mm._index_views = mm._index_views, None
        try:
            slow = detr_amd.PackedTargets(targets, device)
        finally:
            mm._index_views = helper
        assert slow._scanned is None and fast.sizes == slow.sizes and fast.total == slow.total
        assert fast.max_t == slow.max_t and fast.offsets_list == slow.offsets_list
        assert torch.equal(fast.labels[:fast.total], slow.labels[:slow.total])
        assert torch.equal(fast.boxes[:fast.total], slow.boxes[:slow.total])
        assert torch.equal(fast.meta, slow.meta)
        assert fast.same_targets(targets) and slow.same_targets(targets)
        assert not fast.same_targets(targets[:-1] + [dict(labels=targets[-1]["labels"].clone(), boxes=targets[-1]["boxes"])])
    # batches beyond 96 images hand their table over in a ring of pinned slots: more packs in a row than the ring has
    # slots, the batch growing in between (the ring is then re-made)
    for k, batch in enumerate([130, 130, 300, 97, 300, 1024, 130, 1024, 200, 130, 130, 256]):
        _, _, targets = synth.make_workload("cfg5", device="cuda", batch=batch, rank=k)
        fast = detr_amd.PackedTargets(targets, device)
        assert fast._scanned is not None
        sizes = [len(t["labels"]) for t in targets]
        assert torch.equal(fast.labels[:fast.total], torch.cat([t["labels"] for t in targets]))
        assert torch.equal(fast.boxes[:fast.total], torch.cat([t["boxes"] for t in targets]))
        meta = fast.meta.tolist()
        assert meta[:batch + 1] == [sum(sizes[:b]) for b in range(batch + 1)]
        assert meta[batch + 1:] == sorted(range(batch), key=lambda b: -sizes[b])
    # all images empty; int32 labels and float64 boxes take the general path (and its exceptions)
    empty = [{"labels": torch.zeros(0, dtype=torch.int64, device="cuda"), "boxes": torch.zeros(0, 4, device="cuda")}] * 3
    p = detr_amd.PackedTargets(empty, device)
    assert p.total == 0 and p.meta.tolist() == [0, 0, 0, 0, 0, 1, 2]
    odd = [{"labels": torch.tensor([1, 0], dtype=torch.int32, device="cuda"), "boxes": torch.rand(2, 4, device="cuda")}]
    p = detr_amd.PackedTargets(odd, device)
    assert p._scanned is None and p.labels.dtype == torch.int64 and p.labels.tolist() == [1, 0]
    with pytest.raises(RuntimeError):
        detr_amd.PackedTargets([{"labels": torch.tensor([1], device="cuda"), "boxes": torch.rand(1, 4, device="cuda").double()}], device)


@pytest.mark.parametrize("name", [n for n in golden_io.case_names() if golden_io.load(n).error is None])
def test_match_flat_against_the_reference_golden_indices(name):
    """The flat view checked against the REFERENCE's own indices (golden vectors), not against the product's
    tuples: what the criterion builds from the golden (src, tgt) lists must be what match_flat hands over
    (identical pairs, or -- where the optimum is not unique -- pairs of equal total cost on the golden cost)."""
    import detr_amd

    g = golden_io.load(name)
    outputs, targets = g.torch_inputs("cuda")
    matcher = detr_amd.HungarianMatcher(**g.matcher_kwargs())
    flat = matcher.match_flat([outputs], targets)[0]
    matcher.check_status()
    golden = [(torch.from_numpy(np.asarray(i)), torch.from_numpy(np.asarray(j))) for i, j in g.indices]
    bi, si, boxes, labels, dense = _criterion_view(golden, targets, g.B, g.Q, g.C - 1)
    fb, fs, ft = flat.batch_idx.cpu(), flat.src_idx.cpu(), flat.tgt_idx.cpu()
    assert fb.shape == bi.shape and torch.equal(fb, bi)
    if torch.equal(fs, si):
        assert torch.equal(flat.target_boxes().cpu(), boxes) and torch.equal(flat.target_labels().cpu(), labels)
        assert torch.equal(flat.target_classes(g.C - 1).cpu(), dense)
    else:
        for b in range(g.B):
            mine = fb == b
            a = golden_io.assignment_cost(g.cost[b], fs[mine].numpy(), ft[mine].numpy())
            r = golden_io.assignment_cost(g.cost[b], np.asarray(g.indices[b][0]), np.asarray(g.indices[b][1]))
            assert abs(a - r) <= 1e-5 * max(1.0, abs(r))
