"""GPU parity tests: the HIP path (through the C ABI) against the golden vectors, the CPU
oracle and scipy.

Bars: cost entries within 1e-5 relative in fp32 (for the focal
class term widened by its sensitivity to a 2-ulp change of sigmoid(x), see
golden_io.assert_cost_close); assignments
bit-exact wherever the optimum is unique; total matched cost within 1e-5 relative where
ties exist.  The assignment kernel on its own is held to a stricter bar: bit-exact against
scipy on the SAME cost matrix, ties included.
"""
import numpy as np
import pytest
import torch
from scipy.optimize import linear_sum_assignment

import golden_io

pytestmark = pytest.mark.gpu

OK_CASES = [n for n in golden_io.case_names() if golden_io.load(n).error is None]
ERR_CASES = [n for n in golden_io.case_names() if golden_io.load(n).error is not None]


def _matcher(**kw):
    import detr_amd

    return detr_amd.HungarianMatcher(**kw)


def _assert_same_or_equal_cost(cost_qt, got, ref):
    gs, gt = got[0].numpy(), got[1].numpy()
    rs, rt = np.asarray(ref[0]), np.asarray(ref[1])
    assert gs.dtype == np.int64 and gt.dtype == np.int64
    assert len(gs) == len(rs) == len(gt)
    if np.array_equal(gs, rs) and np.array_equal(gt, rt):
        return True
    # not the same pairs: must be another optimum of equal value on the reference's cost
    assert len(set(gs.tolist())) == len(gs) and len(set(gt.tolist())) == len(gt)
    a = golden_io.assignment_cost(cost_qt, gs, gt)
    r = golden_io.assignment_cost(cost_qt, rs, rt)
    assert abs(a - r) <= 1e-5 * max(1.0, abs(r)), (a, r)
    return False


PARITY_REPORT = {}


@pytest.fixture(scope="module", autouse=True)
def _write_parity_report():
    """Per golden case: share of cost entries inside the plain 1e-5 bar and the worst relative error, written
    where the GPU run's artefacts go (copied to profiles/ by hand when they are to be kept)."""
    yield
    if PARITY_REPORT:
        import json
        import os

        out = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "gpurun_out")
        os.makedirs(out, exist_ok=True)
        with open(os.path.join(out, "parity_report.json"), "w") as f:
            json.dump(PARITY_REPORT, f, indent=1, sort_keys=True)


@pytest.mark.parametrize("name", OK_CASES)
def test_golden_costs(name):
    g = golden_io.load(name)
    outputs, targets = g.torch_inputs("cuda")
    blocks = _matcher(**g.matcher_kwargs()).cost_blocks(outputs, targets)
    stats = {}
    for b in range(g.B):
        got = blocks[b].cpu().numpy()
        assert got.shape == (g.Q, g.sizes[b])
        golden_io.assert_cost_close(got, g.cost[b], g.logits[b], g.labels[b], g.cls_loss,
                                    g.weights[0], g.alpha, g.gamma, stats)
    if stats.get("entries"):
        stats["share_inside_plain_bar"] = stats["inside_plain_bar"] / stats["entries"]
        PARITY_REPORT[name] = stats
        # every entry whose logit is not saturated meets the plain 1e-5 bar (assert_cost_close widens the bar
        # for saturated ones only); over all entries the share stays above 99.8 % (lowest observed: 99.90 %,
        # focal_c2_q1050_first_stage) except in the case that was built to be saturated
        assert stats["inside_plain_bar"] >= stats["entries"] - stats["saturated_entries"]
        assert stats["share_inside_plain_bar"] >= (0.9 if name == "focal_c2_saturated" else 0.998)


@pytest.mark.parametrize("name", OK_CASES)
def test_golden_indices(name):
    g = golden_io.load(name)
    outputs, targets = g.torch_inputs("cuda")
    got = _matcher(**g.matcher_kwargs())(outputs, targets)
    assert len(got) == g.B
    for b in range(g.B):
        assert got[b][0].device.type == "cpu" and got[b][0].dtype == torch.int64
        _assert_same_or_equal_cost(g.cost[b], got[b], g.indices[b])


@pytest.mark.parametrize("name", ERR_CASES)
def test_golden_errors(name):
    g = golden_io.load(name)
    outputs, targets = g.torch_inputs("cuda")
    with pytest.raises(g.error):
        _matcher(**g.matcher_kwargs())(outputs, targets)


def test_cpu_tensors_are_rejected():
    g = golden_io.load("ce_c3_q50")
    outputs, targets = g.torch_inputs("cpu")
    with pytest.raises(RuntimeError):
        _matcher(**g.matcher_kwargs())(outputs, targets)


def _lsap_matrices(rng):
    mats = []
    for _ in range(40):     # T < Q: the register kernel
        q, t = int(rng.integers(2, 700)), int(rng.integers(1, 60))
        mats.append(rng.standard_normal((q, t)).astype(np.float32))
    for _ in range(40):     # heavy ties
        q, t = int(rng.integers(2, 300)), int(rng.integers(1, 40))
        mats.append(rng.integers(0, 3, (q, t)).astype(np.float32))
    for _ in range(20):     # one-decimal costs
        q, t = int(rng.integers(2, 200)), int(rng.integers(1, 50))
        mats.append(np.round(rng.random((q, t)), 1).astype(np.float32))
    for _ in range(20):     # T >= Q: scipy does not transpose
        q = int(rng.integers(1, 40))
        t = q + int(rng.integers(0, 40))
        mats.append(rng.integers(0, 4, (q, t)).astype(np.float32))
    for shape in [(1, 1), (9, 1), (1, 9), (7, 7), (64, 5), (65, 64), (128, 127), (129, 3), (1088, 17), (1089, 5),
                  (1500, 30), (320, 100), (321, 100), (640, 2), (641, 2), (2048, 40), (2049, 7), (3072, 25),
                  (4096, 12)]:
        mats.append(np.zeros(shape, dtype=np.float32))
        mats.append(rng.standard_normal(shape).astype(np.float32))
    mats.append(rng.integers(0, 3, (2500, 20)).astype(np.float32))     # heavy ties beyond 1088 columns
    mats.append(np.round(rng.random((4000, 15)), 1).astype(np.float32))
    m = rng.random((60, 9)).astype(np.float32)
    m[:, 2] = np.inf
    m[4, 2] = 0.25
    mats.append(m)
    return mats


@pytest.mark.parametrize("force_generic", [0, 1, 2, 3])
def test_lsap_bit_exact_vs_scipy(force_generic):
    import detr_amd

    rng = np.random.default_rng(99)
    mats = _lsap_matrices(rng)
    detr_amd._lib.set_knob("lap_kernel", force_generic)
    try:
        got = detr_amd.batched_linear_sum_assignment([torch.from_numpy(m).cuda() for m in mats])
    finally:
        detr_amd._lib.set_knob("lap_kernel", 0)
    for m, (r, c) in zip(mats, got):
        rs, cs = linear_sum_assignment(m)
        np.testing.assert_array_equal(r.numpy(), rs, err_msg=str(m.shape))
        np.testing.assert_array_equal(c.numpy(), cs, err_msg=str(m.shape))


def test_lsap_errors():
    import detr_amd

    for force in (0, 1, 2, 3):
        detr_amd._lib.set_knob("lap_kernel", force)
        try:
            for bad in (np.nan, -np.inf):
                m = np.ones((30, 4), dtype=np.float32)
                m[17, 2] = bad
                with pytest.raises(ValueError):
                    detr_amd.batched_linear_sum_assignment([torch.from_numpy(m).cuda()])
            m = np.full((30, 4), np.inf, dtype=np.float32)
            with pytest.raises(ValueError):
                detr_amd.batched_linear_sum_assignment([torch.from_numpy(m).cuda()])
        finally:
            detr_amd._lib.set_knob("lap_kernel", 0)


def _check_workload(name, batch=None, lap_sample=None):
    """match_many on a full workload: GPU assignment bit-exact vs scipy on the GPU's own cost,
    cost vs the CPU oracle within tolerance, end-to-end vs the CPU oracle."""
    import detr_amd
    from detr_amd import synth
    from oracle import oracle

    wl, sets, targets = synth.make_workload(name, device="cpu", batch=batch)
    dev_sets = [{k: v.cuda() for k, v in s.items()} for s in sets]
    dev_targets = [{k: v.cuda() for k, v in t.items()} for t in targets]
    matcher = detr_amd.HungarianMatcher(**wl.matcher_kwargs)
    result, costs = matcher.match_many(dev_sets, dev_targets, return_cost=True)
    sizes = [len(t["labels"]) for t in targets]
    off = np.concatenate([[0], np.cumsum(sizes)])
    rng = np.random.default_rng(0)
    exact = total = 0
    for s, out in enumerate(sets):
        B, Q = out["pred_logits"].shape[:2]
        cost_s = costs[s].cpu().numpy()          # [M, Q]
        images = range(B) if lap_sample is None else rng.choice(B, size=min(lap_sample, B), replace=False)
        ref = None
        if lap_sample is None:
            ref, ref_cost = oracle.match(out["pred_logits"].numpy(), out["pred_boxes"].numpy(),
                                         [t["labels"].numpy() for t in targets],
                                         [t["boxes"].numpy() for t in targets],
                                         return_cost=True, **wl.matcher_kwargs)
        for b in images:
            gs, gt = result[s][b][0].numpy(), result[s][b][1].numpy()
            n = min(Q, sizes[b])
            assert len(gs) == n and len(gt) == n
            if n == 0:
                continue
            blk = cost_s[off[b]:off[b + 1]].T    # [Q, T_b], the argument scipy would get
            rs, cs = linear_sum_assignment(blk)
            np.testing.assert_array_equal(gs, rs)
            np.testing.assert_array_equal(gt, cs)
            if ref is not None:
                kw = wl.matcher_kwargs
                golden_io.assert_cost_close(blk, ref_cost[b], out["pred_logits"][b].numpy(),
                                            targets[b]["labels"].numpy(), kw["cls_loss"], kw["cost_class"],
                                            kw["focal_alpha"], kw["focal_gamma"])
                exact += _assert_same_or_equal_cost(ref_cost[b], result[s][b], ref[b])
                total += 1
    return exact, total


def test_workload_cfg1_full():
    exact, total = _check_workload("cfg1")
    assert exact >= 0.99 * total


def test_workload_cfg2_full():
    exact, total = _check_workload("cfg2")
    assert exact >= 0.99 * total


def test_workload_cfg3_full():
    """cfg3 as stated: B=64, six decoder sets and the two-stage encoder set, all at Q=300."""
    exact, total = _check_workload("cfg3")
    assert exact >= 0.99 * total


def test_workload_cfg3_first_stage_hw():
    exact, total = _check_workload("cfg3_hw", batch=16)
    assert exact >= 0.99 * total


def test_workload_cfg3_wide_first_stage():
    """First stage over 2100 encoder tokens: beyond the one-wavefront kernel's 1088 columns."""
    exact, total = _check_workload("cfg3_hw2", batch=8)
    assert exact >= 0.99 * total


@pytest.mark.parametrize("name", ["wide_focal", "wide_ce", "wide_focal_g15"])
def test_class_counts_beyond_the_lds_tile(name):
    """More classes than a 64-row logit tile in LDS can hold (C > ~600, e.g. LVIS's 1203): the cost kernel that
    reads the logits from memory; same bars as every other shape."""
    exact, total = _check_workload(name)
    assert exact >= 0.99 * total


def test_workload_cfg4_reduced_batch():
    exact, total = _check_workload("cfg4", batch=24)
    assert exact >= 0.99 * total


def test_workload_cfg4_full_vs_oracle():
    """cfg4 at the size it is quoted at (B=256, Q=900, C=91, heavy-tail T, 1536 problems): every problem against
    scipy on the GPU's own cost, every cost block and every index pair against the C oracle (matcher.py:59-95)."""
    exact, total = _check_workload("cfg4")
    assert total == 6 * 256 - _empty_problems("cfg4") and exact >= 0.99 * total


def test_workload_cfg5_full_vs_oracle():
    """cfg5 on one GPU (B=1024, 12 sets of Q=600, 12 288 problems), problem by problem like cfg4 above."""
    exact, total = _check_workload("cfg5")
    assert total == 12 * 1024 - _empty_problems("cfg5") and exact >= 0.99 * total


def _empty_problems(name):
    from detr_amd import synth

    wl, sets, targets = synth.make_workload(name, device="cpu")
    return len(sets) * sum(1 for t in targets if len(t["labels"]) == 0)


def test_workload_cfg4_full_size_properties():
    """Full cfg4 (B=256, Q=900, C=91, heavy-tail T): permutation validity everywhere,
    scipy equality on a sample of problems."""
    import detr_amd
    from detr_amd import synth

    wl, sets, targets = synth.make_workload("cfg4", device="cuda")
    matcher = detr_amd.HungarianMatcher(**wl.matcher_kwargs)
    result = matcher.match_many(sets, targets)
    sizes = [len(t["labels"]) for t in targets]
    for s in range(len(sets)):
        Q = sets[s]["pred_logits"].shape[1]
        for b, (src, tgt) in enumerate(result[s]):
            n = min(Q, sizes[b])
            src, tgt = src.numpy(), tgt.numpy()
            assert len(src) == n
            assert np.all(np.diff(src) > 0) and (n == 0 or (src[0] >= 0 and src[-1] < Q))
            assert sorted(tgt.tolist()) == list(range(n)) if n == sizes[b] else len(set(tgt.tolist())) == n
    # idempotence: a second call gives the same answer
    again = matcher.match_many(sets, targets)
    for s in range(len(sets)):
        for (a0, a1), (b0, b1) in zip(result[s], again[s]):
            assert torch.equal(a0, b0) and torch.equal(a1, b1)
    _check_workload("cfg4", lap_sample=24)


def test_workload_cfg5_full_size_on_one_gpu():
    """All 12 288 problems of cfg5 (B=1024, 12 sets) in one call: counts and index validity on every
    problem, scipy equality on the GPU's own cost for a sample of images of every set."""
    import detr_amd
    from detr_amd import synth

    wl, sets, targets = synth.make_workload("cfg5", device="cuda")
    result = detr_amd.HungarianMatcher(**wl.matcher_kwargs).match_many(sets, targets)
    sizes = [len(t["labels"]) for t in targets]
    assert len(result) == 12 and all(len(r) == 1024 for r in result)
    for s in range(12):
        for b, (src, tgt) in enumerate(result[s]):
            src, tgt = src.numpy(), tgt.numpy()
            assert len(src) == sizes[b] and np.all(np.diff(src) > 0)
            assert sizes[b] == 0 or (src[0] >= 0 and src[-1] < 600)
            assert sorted(tgt.tolist()) == list(range(sizes[b]))
    del result, sets, targets
    torch.cuda.empty_cache()
    _check_workload("cfg5", lap_sample=6)


@pytest.mark.parametrize("name", ["focal_c91_q100", "ce_c92_q100", "focal_c4_gamma15", "focal_c2_q300"])
def test_cost_kernel_shape_knobs_do_not_change_the_cost(name):
    """Wavefronts per workgroup (C > 4 kernel), queries per thread and target split (C <= 4 kernel)
    are launch shapes only: every one of them yields bit-identical cost blocks."""
    import detr_amd

    g = golden_io.load(name)
    outputs, targets = g.torch_inputs("cuda")
    matcher = _matcher(**g.matcher_kwargs())
    base = [blk.clone() for blk in matcher.cost_blocks(outputs, targets)]
    try:
        for knob, value in (("tile_waves", 4), ("tile_waves", 8), ("cost_qpt", 2), ("cost_qpt", 1),
                            ("cost_tsplit", 1), ("cost_tsplit", 2), ("cost_tsplit", 4)):
            detr_amd._lib.set_knob(knob, value)
            for a, b in zip(base, matcher.cost_blocks(outputs, targets)):
                assert torch.equal(a, b), (knob, value)
            detr_amd._lib.set_knob(knob, 0)
    finally:
        for knob in ("tile_waves", "cost_qpt", "cost_tsplit"):
            detr_amd._lib.set_knob(knob, 0)


def test_many_problems_wavefront_kernel_without_resident_rows():
    """More than 512 problems: the one-wavefront kernel keeps only its state in LDS and reads the cost
    rows from memory.  Same pairs as the workgroup kernel on every problem, scipy on a sample."""
    import detr_amd
    from detr_amd import synth

    wl, sets, targets = synth.make_workload("cfg5", device="cuda", batch=64)      # 12 sets x 64 images = 768 problems
    matcher = detr_amd.HungarianMatcher(**wl.matcher_kwargs)
    base, costs = matcher.match_many(sets, targets, return_cost=True)
    detr_amd._lib.set_knob("lap_kernel", 3)
    try:
        wave = matcher.match_many(sets, targets)
    finally:
        detr_amd._lib.set_knob("lap_kernel", 0)
    for s in range(len(sets)):
        for (a0, a1), (b0, b1) in zip(base[s], wave[s]):
            assert torch.equal(a0, b0) and torch.equal(a1, b1)
    sizes = [len(t["labels"]) for t in targets]
    off = np.concatenate([[0], np.cumsum(sizes)])
    rng = np.random.default_rng(1)
    for s in rng.choice(len(sets), 3, replace=False):
        cost_s = costs[s].cpu().numpy()
        for b in rng.choice(64, 8, replace=False):
            if sizes[b] == 0:
                continue
            rs, cs = linear_sum_assignment(cost_s[off[b]:off[b + 1]].T)
            np.testing.assert_array_equal(wave[s][b][0].numpy(), rs)
            np.testing.assert_array_equal(wave[s][b][1].numpy(), cs)


def test_image_order_invariance():
    import detr_amd
    from detr_amd import synth

    wl, sets, targets = synth.make_workload("cfg2", device="cuda", batch=16)
    matcher = detr_amd.HungarianMatcher(**wl.matcher_kwargs)
    base = matcher(sets[0], targets)
    perm = torch.randperm(16, generator=torch.Generator().manual_seed(3)).tolist()
    shuffled = {k: v[perm].contiguous() for k, v in sets[0].items()}
    got = matcher(shuffled, [targets[i] for i in perm])
    for k, i in enumerate(perm):
        assert torch.equal(got[k][0], base[i][0]) and torch.equal(got[k][1], base[i][1])


def test_device_indices_mode_matches_cpu_mode():
    import detr_amd
    from detr_amd import synth

    wl, sets, targets = synth.make_workload("cfg3", device="cuda", batch=8)
    matcher = detr_amd.HungarianMatcher(**wl.matcher_kwargs)
    cpu = matcher.match_many(sets, targets)
    dev = matcher.match_many(sets, targets, device_indices=True)
    matcher.check_status()
    for s in range(len(sets)):
        for (a0, a1), (b0, b1) in zip(cpu[s], dev[s]):
            assert b0.device.type == "cuda"
            assert torch.equal(a0, b0.cpu()) and torch.equal(a1, b1.cpu())


def test_tensors_on_a_device_that_is_not_current():
    """Predictions and targets on cuda:1 while cuda:0 is the current device: the kernels, the copy and the
    completion event all go to the tensors' device (needs two GPUs)."""
    import detr_amd
    from detr_amd import synth

    if torch.cuda.device_count() < 2:
        pytest.skip("needs two GPUs")
    wl, sets, targets = synth.make_workload("cfg2", device="cpu", batch=8)
    matcher = detr_amd.HungarianMatcher(**wl.matcher_kwargs)
    here = matcher.match_many([{k: v.cuda(0) for k, v in s.items()} for s in sets],
                              [{k: v.cuda(0) for k, v in t.items()} for t in targets])
    torch.cuda.set_device(0)
    there = matcher.match_many([{k: v.cuda(1) for k, v in s.items()} for s in sets],
                               [{k: v.cuda(1) for k, v in t.items()} for t in targets])
    for s in range(len(sets)):
        for (a0, a1), (b0, b1) in zip(here[s], there[s]):
            assert torch.equal(a0, b0) and torch.equal(a1, b1)


def test_device_and_host_index_tuples_alternate():
    """The bulk view helper recycles tensor objects only for storage of the kind they were made for: calls that
    return device tuples and calls that return CPU tuples can alternate."""
    import detr_amd
    from detr_amd import synth

    wl, sets, targets = synth.make_workload("cfg2", device="cuda", batch=4)
    matcher = detr_amd.HungarianMatcher(**wl.matcher_kwargs)
    want = [[(a.clone(), b.clone()) for a, b in per_set] for per_set in matcher.match_many(sets, targets)]
    for k in range(6):
        got = matcher.match_many(sets, targets, device_indices=bool(k % 2))
        for s in range(len(sets)):
            for (a0, a1), (b0, b1) in zip(want[s], got[s]):
                assert b0.device.type == ("cuda" if k % 2 else "cpu")
                assert torch.equal(a0, b0.cpu()) and torch.equal(a1, b1.cpu())
        del got


@pytest.mark.parametrize("name", golden_io.dtype_case_names())
def test_inputs_that_are_not_float32(name):
    """float64 inputs (the reference computes in the dtype it is handed, matcher.py:59-90): the kernels round them
    to float32 and compute there.  Against the reference's own float64 cost blocks and indices: cost within the
    float32 bar, indices identical or of equal total cost; the reference's exceptions for mixed box dtypes
    (RuntimeError) and half precision (NotImplementedError)."""
    g = golden_io.DtypeCase(name)
    matcher = _matcher(**g.matcher_kwargs())
    outputs = {"pred_logits": g.logits.cuda(), "pred_boxes": g.pboxes.cuda()}
    targets = [{k: v.cuda() for k, v in t.items()} for t in g.targets]
    if g.error is not None:
        with pytest.raises(g.error):
            matcher(outputs, targets)
        with pytest.raises(g.error):
            matcher.match_flat([outputs], targets)
        return
    got = matcher(outputs, targets)
    blocks = matcher.cost_blocks({k: v.float() for k, v in outputs.items()},
                                 [dict(t, boxes=t["boxes"].float()) for t in targets])
    for b, ((src, tgt), (rs, rt)) in enumerate(zip(got, g.indices)):
        assert src.dtype == torch.int64 and src.device.type == "cpu"
        ref_cost = g.cost[b]
        golden_io.assert_cost_close(blocks[b].cpu().numpy(), ref_cost, g.logits[b].numpy(), g.targets[b]["labels"].numpy(),
                                    g.cls_loss, g.weights[0], 0.25, 2.0)
        _assert_same_or_equal_cost(ref_cost, (src, tgt), (rs, rt))
    flat = matcher.match_flat([outputs], targets)[0]
    assert torch.equal(flat.src_idx.cpu(), torch.cat([s for s, _ in got]))
