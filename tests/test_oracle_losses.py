"""The numpy restatement of the criterion's index consumers (oracle/pair_loss.py) against the
golden vectors produced by the reference's own methods (tests/golden/loss/)."""
import os

import numpy as np
import pytest

import golden_io
from oracle import pair_loss

LOSS_DIR = os.path.join(golden_io.GOLDEN_DIR, "loss")
LOSS_CASES = sorted(os.path.splitext(f)[0] for f in os.listdir(LOSS_DIR) if f.endswith(".npz"))


def load_loss(name):
    return np.load(os.path.join(LOSS_DIR, name + ".npz"), allow_pickle=False)


@pytest.mark.parametrize("name", LOSS_CASES)
def test_box_losses_and_gradients(name):
    g, z = golden_io.load(name), load_loss(name)
    l1, giou, grad_l1, grad_giou = pair_loss.box_losses(g.pboxes, g.tboxes, g.indices)
    # 1e-5 relative on the two scalars (fp32 summation order differs), gradients elementwise
    assert abs(l1 - z["l1"]) <= 1e-5 * max(1.0, abs(z["l1"]))
    assert abs(giou - z["giou"]) <= 1e-5 * max(1.0, abs(z["giou"]))
    np.testing.assert_allclose(grad_l1, z["grad_l1"], rtol=1e-5, atol=1e-7)
    np.testing.assert_allclose(grad_giou, z["grad_giou"], rtol=1e-5, atol=1e-6)


@pytest.mark.parametrize("name", LOSS_CASES)
def test_target_class_map(name):
    g, z = golden_io.load(name), load_loss(name)
    dense = pair_loss.target_classes((g.B, g.Q), g.labels, g.indices, int(z["n_cls"]))
    np.testing.assert_array_equal(dense, z["target_cls"])
    bi, _ = pair_loss.src_permutation_idx(g.indices)
    assert max(len(bi), 1) == int(z["num_boxes"])
