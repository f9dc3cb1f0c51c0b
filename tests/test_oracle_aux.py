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
