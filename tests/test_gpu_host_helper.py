"""The host helper's shared pools under the use a Co-DETR step makes of them
(/root/reference/models/co_detr/loss.py:199-241): the Hungarian matcher, a second matcher instance and the
auxiliary heads called in turn while the caller holds results of each; two threads with a matcher each; and a
short soak of the path that stores the indices straight into pinned host memory, over varying shapes."""
import threading

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _clone(result):
    return [[(a.clone(), b.clone()) for a, b in per] for per in result]


def _equal(result, snapshot):
    return all(torch.equal(a0, b0) and torch.equal(a1, b1)
               for per, snap in zip(result, snapshot) for (a0, a1), (b0, b1) in zip(per, snap))


def _without_helper(fn):
    import detr_amd.matcher as mm

    helper, mm._index_views = mm._index_views, None
    try:
        return fn()
    finally:
        mm._index_views = helper


def test_two_matchers_and_the_auxiliary_heads_in_turn():
    import detr_amd
    from detr_amd import synth
    from detr_amd.aux_assign import ATSS, FasterRCNN

    wl, sets, targets = synth.make_workload("cfg3_hw", device="cuda", batch=12)
    decoder, first_stage = sets[:6], sets[6:]
    m1 = detr_amd.HungarianMatcher(**wl.matcher_kwargs)
    m2 = detr_amd.HungarianMatcher(**dict(wl.matcher_kwargs, cost_class=1.0))
    atss, rcnn = ATSS(9), FasterRCNN()
    # what each call must return, computed without the helper (the Python split, fresh tensors)
    want1 = _clone(_without_helper(lambda: m1.match_many(decoder, targets)))
    want2 = _clone(_without_helper(lambda: m2.match_many(first_stage, targets)))
    want_single = _clone([_without_helper(lambda: m1(decoder[2], targets))])
    held = []
    for step in range(12):
        fresh = [{k: v[:] for k, v in t.items()} for t in targets]
        r1 = m1.match_many(decoder, fresh)
        a = atss(decoder[step % 6], fresh)
        r2 = m2.match_many(first_stage, fresh)
        pos, neg = rcnn(decoder[(step + 1) % 6], fresh)
        r3 = m1(decoder[2], fresh)
        held.append((r1, want1))
        held.append((r2, want2))
        held.append(([r3], want_single))
        assert len(a) == len(targets) and len(pos) == len(targets) and len(neg) == len(targets)
        for result, want in held:          # everything handed out so far, this step's and earlier ones'
            assert _equal(result, want)
        if step % 3 == 2:
            del held[:4]                    # the caller drops some old results, keeps others


def test_two_threads_with_a_matcher_each():
    import detr_amd
    from detr_amd import synth

    errors = []

    def worker(name, workload, batch, steps):
        try:
            stream = torch.cuda.Stream()
            with torch.cuda.stream(stream):
                wl, sets, targets = synth.make_workload(workload, device="cuda", batch=batch)
                matcher = detr_amd.HungarianMatcher(**wl.matcher_kwargs)
                want = _clone(matcher.match_many(sets, targets, device_indices=True))
                want = [[(a.cpu(), b.cpu()) for a, b in per] for per in want]
                previous = None
                for k in range(steps):
                    fresh = [{k_: v[:] for k_, v in t.items()} for t in targets]
                    got = matcher.match_many(sets, fresh)
                    if not _equal(got, want) or (previous is not None and not _equal(previous, want)):
                        errors.append((name, k))
                    previous = got
        except Exception as exc:      # noqa: BLE001
            errors.append((name, repr(exc)))

    threads = [threading.Thread(target=worker, args=("a", "cfg2", 16, 150)),
               threading.Thread(target=worker, args=("b", "cfg3", 9, 150))]
    for t in threads:
        t.start()
    for t in threads:
        t.join(100)
        assert not t.is_alive()
    assert not errors, errors[:5]


def test_soak_of_the_indices_stored_to_host_memory():
    """A few hundred steps of varying shape: the path that lets the kernels store into the pinned result tensor
    (views recycled by the helper) against the device-buffer + copy path -- identical every time."""
    import detr_amd
    from detr_amd import synth

    rng = np.random.default_rng(11)
    wl = synth.WORKLOADS["cfg2"]
    direct = detr_amd.HungarianMatcher(**wl.matcher_kwargs)
    copied = detr_amd.HungarianMatcher(**wl.matcher_kwargs)
    copied.indices_to_host = False
    steps = 0
    kept = []
    for shape in range(40):
        B = int(rng.choice([1, 3, 16, 64, 96, 130]))
        S = int(rng.integers(1, 7))
        Q = int(rng.choice([100, 300, 600, 900]))
        counts = rng.poisson(12, B)
        counts[rng.random(B) < 0.1] = 0
        gen = torch.Generator().manual_seed(int(rng.integers(1 << 30)))
        targets = synth.make_targets(counts, 2, gen, "cuda")
        sets = [synth.make_output_set(B, Q, 2, gen, "cuda") for _ in range(S)]
        for rep in range(8):           # the same shapes a few times: buffers and tensor objects get recycled
            a = direct.match_many(sets, [dict(t) for t in targets])
            b = copied.match_many(sets, targets)
            steps += 1
            assert _equal(a, b), (shape, rep, B, S, Q)
            if rep == 3:
                kept.append((a, _clone(b)))      # some results outlive many later calls
    for result, snapshot in kept:
        assert _equal(result, snapshot)
    assert steps == 320


def test_exception_between_launch_and_completion_waits_for_the_kernels():
    """An exception raised by the caller's `while_waiting` hook leaves match_many after the kernels have been
    enqueued: the pinned result tensor they store into must not be released before they finish, and the
    matcher must work as before on the next call."""
    import detr_amd
    from detr_amd import synth

    wl, sets, targets = synth.make_workload("cfg2", device="cuda", batch=32)
    matcher = detr_amd.HungarianMatcher(**wl.matcher_kwargs)
    want = _clone(matcher.match_many(sets, targets))

    def boom():
        raise KeyError("from the hook")

    for _ in range(5):
        with pytest.raises(KeyError):
            matcher.match_many(sets, [dict(t) for t in targets], while_waiting=boom)
        # pinned allocations right after the failed call: what would be handed the released block
        scribble = [torch.full((2 * len(sets) * sum(len(t["labels"]) for t in targets) + 1,), -3, dtype=torch.int64,
                               pin_memory=True) for _ in range(4)]
        assert _equal(matcher.match_many(sets, [dict(t) for t in targets]), want)
        assert all(int(x.min()) == -3 and int(x.max()) == -3 for x in scribble)


def test_num_boxes_steps_run_ahead_of_a_stalled_side_stream():
    """Thirty steps issued while the side stream is held up, none consumed in between: every step's counts
    arrive as they were issued (a staging buffer is only rewritten after its upload has run)."""
    from detr_amd import dist as dd

    device = torch.device("cuda", torch.cuda.current_device())
    with torch.cuda.stream(dd._side_stream(device)):
        torch.cuda._sleep(200_000_000)          # ~0.1 s of nothing in front of the uploads
    pending = [dd.start_num_boxes_all_reduce([k, 2 * k + 1, 0], device) for k in range(30)]
    for k, p in enumerate(pending):
        assert p.result().tolist() == [float(max(k, 1)), float(2 * k + 1), 1.0]
    torch.cuda.synchronize()
