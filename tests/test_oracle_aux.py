"""The numpy restatement of the Co-DETR auxiliary assigners (oracle/aux_assign.py) against the
golden vectors produced by the reference's own module (tests/golden/aux/)."""
import numpy as np
import pytest

import aux_golden
from oracle import aux_assign


@pytest.mark.parametrize("name", aux_golden.AUX_CASES)
def test_atss(name):
    c = aux_golden.AuxCase(name)
    got = aux_assign.atss(c.pred_boxes, c.tboxes, c.k)
    assert len(got) == c.B
    for b in range(c.B):
        assert got[b][0].dtype == np.int64
        aux_golden.assert_atss_equal(c, got[b], b)


@pytest.mark.parametrize("name", aux_golden.AUX_CASES)
def test_faster_rcnn(name):
    c = aux_golden.AuxCase(name)
    pos, neg = aux_assign.faster_rcnn(c.pred_boxes, c.tboxes)
    for b in range(c.B):
        aux_golden.assert_rcnn_equal(c, pos[b], neg[b], b)


def test_atss_k_larger_than_queries_raises():
    c = aux_golden.AuxCase("q20_k20")
    with pytest.raises(RuntimeError):
        aux_assign.atss(c.pred_boxes, c.tboxes, 21)


@pytest.mark.parametrize("name", aux_golden.AUX_CASES)
def test_class_targets_scatter_matches_torch_indexing(name):
    """faster_rcnn_class_targets against the torch statements of the reference
    (co_detr/loss.py:149-159) executed on the golden index lists (CPU tensors)."""
    import torch

    c = aux_golden.AuxCase(name)
    n_cls = 50
    labels = [np.arange(t, dtype=np.int64) % 7 + 2 for t in c.sizes]
    pos = [(torch.from_numpy(s), torch.from_numpy(t)) for s, t in c.pos]
    neg = [torch.from_numpy(n) for n in c.neg]
    pidx = (torch.cat([torch.full_like(s, i) for i, (s, _) in enumerate(pos)]), torch.cat([s for s, _ in pos]))
    nidx = (torch.cat([torch.full_like(s, i) for i, s in enumerate(neg)]), torch.cat(neg))
    want = torch.full((c.B, c.Q), -1)
    want[pidx] = torch.cat([torch.from_numpy(l)[i] for l, (_, i) in zip(labels, pos)])
    want[nidx] = torch.cat([torch.full(i.shape, n_cls) for i in neg])
    got, pairs = aux_assign.faster_rcnn_class_targets(c.pos, c.neg, labels, c.Q, n_cls)
    np.testing.assert_array_equal(got, want.numpy())
    assert pairs == len(pidx[0])
