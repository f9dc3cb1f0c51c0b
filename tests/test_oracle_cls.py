"""CPU checks of the classification-loss restatement (oracle/cls_loss.py) against the golden vectors the reference's
own FocalLoss / CrossEntropy / calc_cls_loss_with_ignore produced (tests/golden/cls, gen_golden_cls.py)."""
import numpy as np
import pytest

import golden_io
from oracle import cls_loss as oracle_cls

# loss: relative; gradient: relative to the entry plus a floor of 1e-6 of the largest entry (float32 operation order
# of numpy vs torch; saturated logits lose digits in 1 - sigmoid(x))
LOSS_RTOL, GRAD_RTOL = 2e-6, 2e-5


def oracle_of(case):
    if case.kind == "focal":
        return oracle_cls.focal_loss(case.logits, case.target_cls, case.num_boxes, case.alpha, case.gamma,
                                     ignore_index=case.ignore_index)
    return oracle_cls.cross_entropy(case.logits, case.target_cls, case.weight)


@pytest.mark.parametrize("name", golden_io.cls_case_names())
def test_oracle_matches_the_references_loss_and_gradient(name):
    case = golden_io.ClsLossCase(name)
    loss, grad = oracle_of(case)
    assert abs(float(loss) - case.loss) <= LOSS_RTOL * abs(case.loss) + 1e-7, (float(loss), case.loss)
    floor = 1e-6 * float(np.abs(case.grad).max())
    np.testing.assert_allclose(grad, case.grad, rtol=GRAD_RTOL, atol=floor)


def test_golden_set_covers_what_the_kernels_branch_on():
    cases = [golden_io.ClsLossCase(n) for n in golden_io.cls_case_names()]
    gammas = {c.gamma for c in cases if c.kind == "focal"}
    assert {0.0, 1.5, 2.0, 3.0} <= gammas
    assert any(c.alpha is not None and c.alpha < 0 for c in cases)
    assert any(c.ignore_index == -1 for c in cases)
    assert any(c.kind == "ce" and (c.target_cls == -100).any() for c in cases)
    assert any(c.kind == "ce" and c.logits.shape[-1] > 64 for c in cases)
    assert any(c.kind == "focal" and c.logits.shape[-1] == 1 for c in cases)
    assert any(c.logits.size % 4 for c in cases)              # a tail that is not a whole 16-byte group


def test_oracle_refuses_class_values_out_of_range():
    x = np.zeros((1, 3, 2), np.float32)
    with pytest.raises(RuntimeError):
        oracle_cls.focal_loss(x, np.array([[0, 3, 2]]), 1)
    with pytest.raises(RuntimeError):
        oracle_cls.cross_entropy(x, np.array([[0, 2, 1]]))
