"""Pins the CPU oracle (oracle/) against the golden vectors and against scipy.

The goldens were produced by running the reference's own matcher.py/box_ops.py
(tests/golden/gen_golden.py); scipy is the third-party solver the reference
calls at matcher.py:94.
"""
import numpy as np
import pytest
from scipy.optimize import linear_sum_assignment

import golden_io
from oracle import oracle

OK_CASES = [n for n in golden_io.case_names() if golden_io.load(n).error is None]
ERR_CASES = [n for n in golden_io.case_names() if golden_io.load(n).error is not None]


@pytest.mark.parametrize("name", OK_CASES)
def test_cost_block_matches_reference(name):
    g = golden_io.load(name)
    for b in range(g.B):
        if g.sizes[b] == 0:
            continue
        got = oracle.cost_block(g.logits[b], g.pboxes[b], g.labels[b], g.tboxes[b], **g.matcher_kwargs())
        # north-star tolerance: 1e-5 relative in fp32 (transcendentals differ by an ulp between libms)
        golden_io.assert_cost_close(got, g.cost[b], g.logits[b], g.labels[b], g.cls_loss, g.weights[0],
                                    g.alpha, g.gamma)


@pytest.mark.parametrize("name", OK_CASES)
def test_lsap_on_reference_cost_is_bit_exact(name):
    g = golden_io.load(name)
    for b in range(g.B):
        src, tgt = oracle.lsap(g.cost[b]) if g.sizes[b] else (np.zeros(0, np.int64), np.zeros(0, np.int64))
        np.testing.assert_array_equal(src, g.indices[b][0])
        np.testing.assert_array_equal(tgt, g.indices[b][1])


@pytest.mark.parametrize("name", OK_CASES)
def test_full_match_matches_reference(name):
    g = golden_io.load(name)
    got = oracle.match(g.logits, g.pboxes, g.labels, g.tboxes, **g.matcher_kwargs())
    for b in range(g.B):
        rs, rt = g.indices[b]
        if np.array_equal(got[b][0], rs) and np.array_equal(got[b][1], rt):
            continue
        # an ulp of difference in the cost may pick another optimum of equal value
        assert len(got[b][0]) == len(rs)
        a = golden_io.assignment_cost(g.cost[b], got[b][0], got[b][1])
        r = golden_io.assignment_cost(g.cost[b], rs, rt)
        assert abs(a - r) <= 1e-5 * max(1.0, abs(r))


@pytest.mark.parametrize("name", ERR_CASES)
def test_error_cases_raise_like_reference(name):
    g = golden_io.load(name)
    with pytest.raises(g.error):
        oracle.match(g.logits, g.pboxes, g.labels, g.tboxes, **g.matcher_kwargs())


def test_scipy_docstring_known_answer():
    cost = np.array([[4, 1, 3], [2, 0, 5], [3, 2, 2]], dtype=np.float32)
    r, c = oracle.lsap(cost)
    assert r.tolist() == [0, 1, 2] and c.tolist() == [1, 0, 2]
    assert cost[r, c].sum() == 5


def _random_matrices(rng):
    for _ in range(60):
        nr, nc = int(rng.integers(1, 40)), int(rng.integers(1, 40))
        yield rng.standard_normal((nr, nc)).astype(np.float32)
    for _ in range(60):   # three-level integer costs: heavy ties
        nr, nc = int(rng.integers(1, 30)), int(rng.integers(1, 30))
        yield rng.integers(0, 3, (nr, nc)).astype(np.float32)
    for _ in range(20):   # one-decimal costs
        nr, nc = int(rng.integers(1, 30)), int(rng.integers(1, 60))
        yield np.round(rng.random((nr, nc)), 1).astype(np.float32)
    for shape in [(1, 1), (1, 9), (9, 1), (7, 7), (12, 5), (5, 12)]:
        yield np.zeros(shape, dtype=np.float32)
    m = rng.random((6, 9)).astype(np.float32)
    m[2, :] = np.inf
    m[2, 4] = 1.0
    yield m


def test_lsap_bit_exact_vs_scipy_including_ties():
    rng = np.random.default_rng(1234)
    for m in _random_matrices(rng):
        r, c = oracle.lsap(m)
        rs, cs = linear_sum_assignment(m)
        np.testing.assert_array_equal(r, rs)
        np.testing.assert_array_equal(c, cs)


def test_lsap_errors():
    m = np.ones((3, 4), dtype=np.float32)
    m[1, 2] = np.nan
    with pytest.raises(ValueError):
        oracle.lsap(m)
    m[1, 2] = -np.inf
    with pytest.raises(ValueError):
        oracle.lsap(m)
    m = np.full((3, 4), np.inf, dtype=np.float32)
    with pytest.raises(ValueError):
        oracle.lsap(m)
    with pytest.raises(ValueError):
        linear_sum_assignment(m)


def test_torch_pipeline_restatement_matches_reference():
    import torch

    from oracle import torch_pipeline

    for name in ["ce_c3_q50", "focal_c2_q50", "focal_c4_gamma15", "focal_t_gt_q", "dup_gt"]:
        g = golden_io.load(name)
        outputs, targets = g.torch_inputs()
        got = torch_pipeline.match(outputs["pred_logits"], outputs["pred_boxes"], targets, **g.matcher_kwargs())
        for b in range(g.B):
            assert torch.equal(got[b][0], torch.from_numpy(g.indices[b][0]))
            assert torch.equal(got[b][1], torch.from_numpy(g.indices[b][1]))
