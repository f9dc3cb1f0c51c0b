"""GPU auxiliary assigners (detr_amd.aux_assign) against the golden vectors produced by the
reference's module and against the numpy restatement on workload-sized random inputs.
Bar: identical index lists (ATSS: a target may differ only where the reference's own distances
have a near-tie, see aux_golden.assert_atss_equal; Faster-RCNN: identical sets per target)."""
import numpy as np
import pytest
import torch

import aux_golden

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("name", aux_golden.AUX_CASES)
def test_atss_golden(name):
    from detr_amd.aux_assign import ATSS

    c = aux_golden.AuxCase(name)
    outputs, targets = c.torch_inputs("cuda")
    got = ATSS(c.k)(outputs, targets)
    assert len(got) == c.B
    for b, (src, tgt) in enumerate(got):
        assert src.device.type == "cpu" and src.dtype == torch.int64 and tgt.dtype == torch.int64
        aux_golden.assert_atss_equal(c, (src.numpy(), tgt.numpy()), b)


@pytest.mark.parametrize("name", aux_golden.AUX_CASES)
def test_faster_rcnn_golden(name):
    from detr_amd.aux_assign import FasterRCNN

    c = aux_golden.AuxCase(name)
    outputs, targets = c.torch_inputs("cuda")
    pos, neg = FasterRCNN()(outputs, targets)
    for b in range(c.B):
        assert pos[b][0].dtype == torch.int64 and neg[b].dtype == torch.int64 and neg[b].device.type == "cpu"
        aux_golden.assert_rcnn_equal(c, (pos[b][0].numpy(), pos[b][1].numpy()), neg[b].numpy(), b)


def _workload(seed, B, Q, S):
    from detr_amd import synth

    rng = np.random.default_rng(seed)
    gen = torch.Generator().manual_seed(seed)
    sizes = synth.target_counts("heavy_tail", B, rng)
    targets = [{"boxes": synth.random_boxes(int(t), gen), "labels": torch.zeros(int(t), dtype=torch.int64)}
               for t in sizes]
    sets = []
    for _ in range(S):
        pred = synth.random_boxes(B * Q, gen).view(B, Q, 4).clone()
        for b, t in enumerate(targets):       # noisy copies of the targets so that positives exist
            n = len(t["boxes"])
            if n:
                reps = min(3, Q // n)
                noisy = t["boxes"].repeat(reps, 1) * (1 + 0.1 * (torch.rand(reps * n, 4, generator=gen) - 0.5))
                pred[b, :reps * n] = noisy.clamp(0.005, 0.995)
        sets.append({"pred_boxes": pred})
    return sets, targets


def test_all_sets_of_a_step_against_the_restatement():
    from detr_amd.aux_assign import ATSS, FasterRCNN
    from oracle import aux_assign

    sets, targets = _workload(5, B=12, Q=900, S=3)
    dev_sets = [{"pred_boxes": s["pred_boxes"].cuda()} for s in sets]
    dev_targets = [{k: v.cuda() for k, v in t.items()} for t in targets]
    tb = [t["boxes"].numpy() for t in targets]
    atss = ATSS(9).assign_many(dev_sets, dev_targets)
    rcnn = FasterRCNN(0.6, 0.2).assign_many(dev_sets, dev_targets)
    assert len(atss) == 3 and len(rcnn) == 3
    for s, out in enumerate(sets):
        ref = aux_assign.atss(out["pred_boxes"].numpy(), tb, 9)
        ref_pos, ref_neg = aux_assign.faster_rcnn(out["pred_boxes"].numpy(), tb, 0.6, 0.2)
        for b in range(12):
            np.testing.assert_array_equal(atss[s][b][0].numpy(), ref[b][0])
            np.testing.assert_array_equal(atss[s][b][1].numpy(), ref[b][1])
            np.testing.assert_array_equal(rcnn[s][0][b][0].numpy(), ref_pos[b][0])
            np.testing.assert_array_equal(rcnn[s][0][b][1].numpy(), ref_pos[b][1])
            np.testing.assert_array_equal(rcnn[s][1][b].numpy(), ref_neg[b])
    # single-set calls agree with the batched call; CPU-resident targets are accepted like the reference's
    one = ATSS(9)(dev_sets[1], targets)
    for (a0, a1), (b0, b1) in zip(one, atss[1]):
        assert torch.equal(a0, b0) and torch.equal(a1, b1)


def test_dense_class_targets_against_the_scatter_of_the_lists():
    """The device-resident class map equals the reference's scatter (loss.py:153-159) of the index lists."""
    from detr_amd.aux_assign import FasterRCNN
    from oracle import aux_assign

    sets, targets = _workload(11, B=9, Q=700, S=3)
    sets.append({"pred_boxes": sets[0]["pred_boxes"][:, :130].clone()})        # a shorter set in the same launch
    rng = np.random.default_rng(3)
    for t in targets:
        t["labels"] = torch.from_numpy(rng.integers(0, 91, len(t["boxes"])))
    dev_sets = [{"pred_boxes": s["pred_boxes"].cuda()} for s in sets]
    dev_targets = [{k: v.cuda() for k, v in t.items()} for t in targets]
    tb = [t["boxes"].numpy() for t in targets]
    labels = [t["labels"].numpy() for t in targets]
    head = FasterRCNN(0.5, 0.3)
    maps, num_pos = head.dense_class_targets(dev_sets, dev_targets, 91)
    assert num_pos.dtype == torch.int32 and num_pos.is_cuda and len(maps) == 4
    seen = set()
    for s, out in enumerate(sets):
        Q = out["pred_boxes"].shape[1]
        pos, neg = aux_assign.faster_rcnn(out["pred_boxes"].numpy(), tb, 0.5, 0.3)
        want, pairs = aux_assign.faster_rcnn_class_targets(pos, neg, labels, Q, 91)
        assert maps[s].shape == (9, Q) and maps[s].dtype == torch.int64 and maps[s].is_cuda
        np.testing.assert_array_equal(maps[s].cpu().numpy(), want)
        assert int(num_pos[s]) == pairs
        seen |= set(np.unique(want).tolist())
    assert -1 in seen and 91 in seen and len(seen) > 5          # ignored, negative and positive entries all occur
    # and the scatter of the product's own lists gives the same map
    lists = head.assign_many(dev_sets, dev_targets)
    for s, (pos, neg) in enumerate(lists):
        want, pairs = aux_assign.faster_rcnn_class_targets([(a.numpy(), b.numpy()) for a, b in pos],
                                                           [n.numpy() for n in neg], labels, maps[s].shape[1], 91)
        np.testing.assert_array_equal(maps[s].cpu().numpy(), want)
    # images without targets: everything negative, no pairs
    none = [{"boxes": t["boxes"][:0], "labels": t["labels"][:0]} for t in dev_targets]
    maps, num_pos = head.dense_class_targets(dev_sets[:2], none, 7)
    assert all(bool((m == 7).all()) for m in maps) and num_pos.tolist() == [0, 0]


@pytest.mark.parametrize("name", aux_golden.AUX_CASES)
def test_dense_class_targets_golden(name):
    """Scatter of the golden lists (produced by the reference's module) against the device map."""
    from detr_amd.aux_assign import FasterRCNN
    from oracle import aux_assign

    c = aux_golden.AuxCase(name)
    outputs, targets = c.torch_inputs("cuda")
    labels = [np.arange(len(t["boxes"]), dtype=np.int64) * 3 + 1 for t in targets]
    targets = [dict(t, labels=torch.from_numpy(l).cuda()) for t, l in zip(targets, labels)]
    maps, num_pos = FasterRCNN().dense_class_targets([outputs], targets, 1000)
    # the golden pairs are target-major; their order within one target does not matter to the scatter
    # (the predictions of one target are distinct)
    want, pairs = aux_assign.faster_rcnn_class_targets(c.pos, c.neg, labels, c.Q, 1000)
    np.testing.assert_array_equal(maps[0].cpu().numpy(), want)
    assert int(num_pos[0]) == pairs


def test_errors_and_edges():
    from detr_amd.aux_assign import ATSS, FasterRCNN

    c = aux_golden.AuxCase("q20_k20")
    outputs, targets = c.torch_inputs("cuda")
    with pytest.raises(RuntimeError):
        ATSS(21)(outputs, targets)                       # torch.topk: k out of range
    with pytest.raises(RuntimeError):
        ATSS(3)({"pred_boxes": outputs["pred_boxes"].cpu()}, targets)
    with pytest.raises(NotImplementedError):
        big = {"pred_boxes": torch.rand(2, 100, 4, device="cuda")}
        ATSS(65)(big, targets)
    none = [{"boxes": t["boxes"][:0], "labels": t["labels"][:0]} for t in targets]
    assert all(len(i) == 0 for i, _ in ATSS(3)(outputs, none))
    pos, neg = FasterRCNN()(outputs, none)
    assert all(len(i) == 0 for i, _ in pos) and all(torch.equal(n, torch.arange(c.Q)) for n in neg)
