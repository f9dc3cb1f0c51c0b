"""The numpy restatement of the box loss (oracle/pair_loss.py) on index tuples that do not come from the matcher --
denoising groups, one prediction paired with several targets -- against what the reference's calc_box_loss + autograd
produced for them (tests/golden/pairs, gen_golden_pairs.py)."""
import numpy as np
import pytest

import golden_io
from oracle import pair_loss


@pytest.mark.parametrize("name", golden_io.pair_case_names())
def test_box_losses_and_gradients_of_pair_lists(name):
    c = golden_io.PairListCase(name)
    l1, giou, grad_l1, grad_giou = pair_loss.box_losses(c.pred_boxes, c.target_boxes, c.indices)
    assert abs(l1 - c.l1) <= 1e-5 * max(1.0, abs(c.l1)) and abs(giou - c.giou) <= 1e-5 * max(1.0, abs(c.giou))
    np.testing.assert_allclose(grad_l1, c.grad_l1, rtol=1e-5, atol=1e-7)
    np.testing.assert_allclose(grad_giou, c.grad_giou, rtol=1e-5, atol=1e-6)


def test_the_cases_hold_what_they_are_for():
    dup = golden_io.PairListCase("pairs_dup")
    src0 = dup.indices[0][0]
    assert len(set(src0.tolist())) < len(src0)                      # a prediction in several pairs
    assert any(len(s) == 0 for s, _ in dup.indices)                 # an image without pairs
    cdn = golden_io.PairListCase("pairs_cdn")
    assert any(len(b) == 0 for b in cdn.target_boxes) and sum(len(s) for s, _ in cdn.indices) == 45
