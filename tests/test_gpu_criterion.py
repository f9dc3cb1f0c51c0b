"""detr_amd.criterion.attach: an unmodified criterion loop (one matcher call per output set, as in
/root/reference/models/detr/loss.py:160-183 and dino_detr/loss.py:162-203) gets all its sets matched in one
batched call made by a forward pre-hook; the loop's own calls find their results waiting."""
import types

import pytest
import torch


class LoopCriterion(torch.nn.Module):
    """The calling pattern of the reference's criteria, nothing else: walks the decoder-layer sets, skips the
    denoising sets (they bring their indices), then the first-stage sets; keeps what the matcher returned."""

    def __init__(self, matcher, aux_flag=True):
        super().__init__()
        self.matcher = matcher
        self.aux_flag = aux_flag
        self.seen = []

    def forward(self, outputs, targets):
        self.seen = []
        for key in ("model", "cdn"):
            group = outputs.get(key, [])
            if not self.aux_flag:
                group = group[-1:]
            for output in group:
                if key == "cdn":
                    continue
                self.seen.append(self.matcher(output, targets))
        for output in outputs.get("first_stage", []):
            self.seen.append(self.matcher(output, targets))
        return len(self.seen)


def test_the_sets_a_loop_will_match_cpu():
    from detr_amd.criterion import matcher_like, sets_the_loop_will_match

    mk = lambda: {"pred_logits": torch.zeros(1, 2, 2), "pred_boxes": torch.zeros(1, 2, 4)}
    outputs = {"model": [mk(), mk(), mk()], "cdn": [mk()], "first_stage": [mk()], "cdn_indices": [], "aux": {"atss": [mk()]}}
    found = sets_the_loop_will_match(outputs)
    assert [id(o) for o in found] == [id(o) for o in outputs["model"] + outputs["first_stage"]]
    last = sets_the_loop_will_match(outputs, last_only=True)
    assert [id(o) for o in last] == [id(outputs["model"][-1]), id(outputs["first_stage"][-1])]
    assert sets_the_loop_will_match({"model": []}) == []
    ref = types.SimpleNamespace(cls_loss="focal", cost_class=2.0, cost_bbox=5.0, cost_giou=2.0, focal_alpha=0.3,
                                focal_gamma=1.5)
    m = matcher_like(ref)
    assert (m.cls_loss, m.cost_class, m.cost_bbox, m.cost_giou, m.focal_alpha, m.focal_gamma) == \
        ("focal", 2.0, 5.0, 2.0, 0.3, 1.5)
    assert matcher_like(m) is m


def _same(a, b):
    assert len(a) == len(b)
    for (a0, a1), (b0, b1) in zip(a, b):
        assert torch.equal(a0, b0) and torch.equal(a1, b1)


@pytest.mark.gpu
@pytest.mark.parametrize("aux_flag", [True, False])
def test_attached_criterion_matches_once_per_step(aux_flag):
    import detr_amd
    from detr_amd import synth
    from detr_amd.criterion import attach, detach

    wl, sets, targets = synth.make_workload("cfg3_hw", device="cuda", batch=8)
    outputs = {"model": sets[:6], "cdn": [dict(sets[0])], "first_stage": sets[6:], "cdn_indices": None}
    plain = detr_amd.HungarianMatcher(**wl.matcher_kwargs)
    want = [plain(o, targets) for o in (sets[:6] if aux_flag else sets[5:6]) + sets[6:]]
    want = [[(a.clone(), b.clone()) for a, b in per] for per in want]

    ref_like = types.SimpleNamespace(**wl.matcher_kwargs)       # what the reference's matcher object exposes
    crit = LoopCriterion(ref_like, aux_flag=aux_flag)
    matcher = attach(crit)
    assert crit.matcher is matcher and attach(crit) is matcher
    calls = []
    inner = matcher.match_many
    matcher.match_many = lambda *a, **k: (calls.append(len(a[0])), inner(*a, **k))[1]
    for step in range(3):
        fresh = [{k: v[:] for k, v in t.items()} for t in targets]
        n = crit(outputs, fresh)
        assert n == len(want)
        for got, ref in zip(crit.seen, want):
            _same(got, ref)
    assert calls == [len(want)] * 3          # one batched call per step, none from the loop
    assert matcher._prefetched is None       # dropped after the criterion returned

    # a set the hook could not foresee, and targets it was not given, are matched on the spot
    calls.clear()
    matcher.prefetch(sets[:2], targets)
    _same(matcher(sets[0], targets), want[0] if aux_flag else plain(sets[0], targets))
    assert calls == [2]
    other = dict(sets[0])
    matcher(other, targets)
    assert calls == [2, 1]
    matcher(sets[1], [{k: v[:] for k, v in t.items()} for t in targets])
    assert calls == [2, 1, 1]
    # predictions changed in place after the prefetch: not served from it
    matcher.prefetch(sets[:2], targets)
    sets[1]["pred_boxes"].mul_(1.0)
    matcher(sets[1], targets)
    assert calls == [2, 1, 1, 2, 1]
    detach(crit)
    matcher.clear_prefetched()
    calls.clear()
    crit(outputs, targets)
    assert calls == [1] * len(want)


@pytest.mark.gpu
def test_attached_criterion_raises_what_the_loop_would():
    import detr_amd
    from detr_amd import synth
    from detr_amd.criterion import attach

    wl, sets, targets = synth.make_workload("cfg2", device="cuda", batch=4)
    sets = sets[:3]
    sets[2]["pred_boxes"][1, 5, 2] = float("nan")
    crit = LoopCriterion(detr_amd.HungarianMatcher(**wl.matcher_kwargs))
    attach(crit)
    with pytest.raises(AssertionError):
        crit({"model": sets}, targets)
    assert crit.matcher._prefetched is None


@pytest.mark.gpu
def test_predictions_that_require_grad():
    """In training the predictions are model outputs with a gradient history (and, for the denoising models,
    non-contiguous views of them): matched as they are, nothing recorded on the autograd tape."""
    import detr_amd
    from detr_amd import synth
    from detr_amd.criterion import attach

    wl, sets, targets = synth.make_workload("cfg3", device="cuda", batch=6)
    want = detr_amd.HungarianMatcher(**wl.matcher_kwargs).match_many(sets, targets)
    want = [[(a.clone(), b.clone()) for a, b in per] for per in want]
    live = []
    for o in sets:
        wide_l = torch.cat([torch.zeros_like(o["pred_logits"][:, :5]), o["pred_logits"]], dim=1).requires_grad_(True)
        wide_b = torch.cat([torch.zeros_like(o["pred_boxes"][:, :5]), o["pred_boxes"]], dim=1).requires_grad_(True)
        live.append({"pred_logits": (wide_l * 1.0).split([5, 300], dim=1)[1], "pred_boxes": (wide_b * 1.0).split([5, 300], dim=1)[1]})
        assert live[-1]["pred_boxes"].requires_grad and not live[-1]["pred_boxes"].is_contiguous()
    crit = LoopCriterion(types.SimpleNamespace(**wl.matcher_kwargs))
    attach(crit)
    assert crit({"model": live[:6], "first_stage": live[6:]}, targets) == 7
    for got, ref in zip(crit.seen, want):
        _same(got, ref)
        assert not got[0][0].requires_grad


@pytest.mark.gpu
def test_attach_swaps_auxiliary_assigners_that_then_run_on_the_gpu():
    """A Co-DETR-shaped criterion: its `aux_heads` dict holds objects with the reference's parameters; after attach
    the same keys hold the GPU assigners, which give what the standalone ones give."""
    import detr_amd
    from detr_amd import aux_assign, synth
    from detr_amd.criterion import attach

    wl, sets, targets = synth.make_workload("cfg3", device="cuda", batch=5)
    crit = LoopCriterion(types.SimpleNamespace(**wl.matcher_kwargs))

    class ATSSLike:          # what the reference's ATSS object exposes
        k = 9

    crit.aux_heads = {"atss": ATSSLike(), "faster_rcnn": types.SimpleNamespace(pos_threshold=0.7, neg_threshold=0.3)}
    attach(crit)
    assert isinstance(crit.aux_heads["atss"], aux_assign.ATSS) and isinstance(crit.aux_heads["faster_rcnn"], aux_assign.FasterRCNN)
    got = crit.aux_heads["atss"](sets[0], targets)
    want = aux_assign.ATSS(9)(sets[0], targets)
    assert len(got) == 5 and all(torch.equal(a0, b0) and torch.equal(a1, b1) for (a0, a1), (b0, b1) in zip(got, want))
    pos, neg = crit.aux_heads["faster_rcnn"](sets[1], targets)
    assert len(pos) == 5 and len(neg) == 5


@pytest.mark.gpu
def test_attach_with_device_resident_indices():
    """attach(criterion, device_indices=True): the loop gets device tensors (same values), the expressions the
    reference's losses apply to them work without an upload, and bad inputs raise when the criterion returns."""
    import detr_amd
    from detr_amd import synth
    from detr_amd.criterion import attach

    wl, sets, targets = synth.make_workload("cfg3_hw", device="cuda", batch=6)
    want = detr_amd.HungarianMatcher(**wl.matcher_kwargs).match_many(sets, targets)
    want = [[(a.clone(), b.clone()) for a, b in per] for per in want]
    crit = LoopCriterion(types.SimpleNamespace(**wl.matcher_kwargs))
    attach(crit, device_indices=True)
    assert crit({"model": sets[:6], "first_stage": sets[6:]}, targets) == 7
    for s, (got, ref) in enumerate(zip(crit.seen, want)):
        for (g0, g1), (r0, r1) in zip(got, ref):
            assert g0.device.type == "cuda" and torch.equal(g0.cpu(), r0) and torch.equal(g1.cpu(), r1)
        # detr/loss.py:71-75, 102-103: permutation index and gathers, all on the device
        batch_idx = torch.cat([torch.full_like(src, i) for i, (src, _) in enumerate(got)])
        src_idx = torch.cat([src for src, _ in got])
        picked = sets[s]["pred_boxes"][batch_idx, src_idx]
        tgt_boxes = torch.cat([t["boxes"][j] for t, (_, j) in zip(targets, got)])
        assert picked.shape == tgt_boxes.shape == (sum(len(t["labels"]) for t in targets), 4)
    bad = [{k: v.clone() for k, v in o.items()} for o in sets[:2]]
    bad[1]["pred_boxes"][2, 7, 3] = float("nan")
    with pytest.raises(AssertionError):
        crit({"model": bad}, targets)
    assert crit({"model": sets[:2]}, targets) == 2          # and the next step is fine again


class BoxLossCriterion(LoopCriterion):
    """LoopCriterion plus the box loss as the reference's criteria compute it (permutation index, gathers, the
    [N, N] GIoU matrix and its diagonal: /root/reference/models/detr/loss.py:97-114), restated with stock torch ops."""

    def _get_src_permutation_idx(self, indices):
        return (torch.cat([torch.full_like(src, i) for i, (src, _) in enumerate(indices)]),
                torch.cat([src for src, _ in indices]))

    def calc_box_loss(self, outputs, targets, indices):
        batch_idx, src_idx = self._get_src_permutation_idx(indices)
        n = max(len(batch_idx), 1)
        a = outputs["pred_boxes"][batch_idx, src_idx]
        g = torch.cat([t["boxes"][j] for t, (_, j) in zip(targets, indices)])
        l1 = (a - g).abs().sum() / n

        def xyxy(b):
            return torch.cat([b[:, :2] - 0.5 * b[:, 2:], b[:, :2] + 0.5 * b[:, 2:]], -1)

        a, g = xyxy(a), xyxy(g)
        area_a, area_g = (a[:, 2] - a[:, 0]) * (a[:, 3] - a[:, 1]), (g[:, 2] - g[:, 0]) * (g[:, 3] - g[:, 1])
        wh = (torch.min(a[:, None, 2:], g[:, 2:]) - torch.max(a[:, None, :2], g[:, :2])).clamp(min=0)
        inter = wh[..., 0] * wh[..., 1]
        union = area_a[:, None] + area_g - inter
        ewh = (torch.max(a[:, None, 2:], g[:, 2:]) - torch.min(a[:, None, :2], g[:, :2])).clamp(min=0)
        earea = ewh[..., 0] * ewh[..., 1]
        giou = inter / union - (earea - union) / earea
        return l1, (1 - torch.diag(giou)).sum() / n

    n_cls = 2

    def calc_cls_loss(self, outputs, targets, indices, check_acc=False):
        """The dense class map and a classification loss on it, the way the reference builds them (loss.py:116-127)."""
        idx = self._get_src_permutation_idx(indices)
        logits = outputs["pred_logits"]
        target_cls = torch.full(logits.shape[:2], self.n_cls, device=logits.device)
        target_cls_o = torch.cat([t["labels"][i] for t, (_, i) in zip(targets, indices)])
        target_cls[idx] = target_cls_o
        self.class_maps.append(target_cls)
        onehot = torch.nn.functional.one_hot(target_cls, self.n_cls + 1)[..., :-1].float()
        loss = torch.nn.functional.binary_cross_entropy_with_logits(logits, onehot, reduction="sum") / max(len(idx[0]), 1)
        if check_acc:
            self.accuracy = (logits[idx].argmax(-1) == target_cls_o).float().mean()
        return loss

    def forward(self, outputs, targets):
        self.seen, self.box_losses, self.class_maps, total = [], [], [], 0.0
        for key in ("model", "cdn"):
            for output in outputs.get(key, []):
                indices = outputs["cdn_indices"] if key == "cdn" else self.matcher(output, targets)
                tg = outputs["cdn_targets"] if key == "cdn" else targets
                self.seen.append(indices)
                self.box_losses.append(self.calc_box_loss(output, tg, indices))
                total = total + self.calc_cls_loss(output, tg, indices, key == "model")
        return total + sum(5.0 * l1 + 2.0 * g for l1, g in self.box_losses)


@pytest.mark.gpu
def test_attach_with_the_fused_consumers():
    """attach(criterion, fused_consumers=True): calc_box_loss is answered by the fused pair-loss kernel for the sets
    this matcher served (same values and gradients as the criterion's own method), while indices from elsewhere --
    the denoising group's -- still go through the criterion's own method."""
    import detr_amd
    from detr_amd import losses, synth
    from detr_amd.criterion import attach, detach

    wl, sets, targets = synth.make_workload("cfg3", device="cuda", batch=5)
    sets = sets[:3]
    for o in sets:
        o["pred_boxes"].requires_grad_(True)
        o["pred_logits"].requires_grad_(True)
    cdn = {"pred_logits": sets[0]["pred_logits"].detach().clone(), "pred_boxes": sets[0]["pred_boxes"].detach().clone().requires_grad_(True)}
    cdn_indices = [(torch.arange(min(3, len(t["labels"])), device="cuda"), torch.arange(min(3, len(t["labels"])), device="cuda"))
                   for t in targets]
    outputs = {"model": sets, "cdn": [cdn], "cdn_indices": cdn_indices, "cdn_targets": targets}
    plain = BoxLossCriterion(detr_amd.HungarianMatcher(**wl.matcher_kwargs))
    want = plain(outputs, targets)
    want_losses = [(a.item(), b.item()) for a, b in plain.box_losses]
    want_maps, want_acc = [m.clone() for m in plain.class_maps], plain.accuracy.item()
    want_grads = torch.autograd.grad(want, [o["pred_boxes"] for o in sets] + [cdn["pred_boxes"]] + [o["pred_logits"] for o in sets])

    crit = BoxLossCriterion(types.SimpleNamespace(**wl.matcher_kwargs))
    attach(crit, fused_consumers=True)
    calls = []
    inner = losses.matched_box_losses
    losses.matched_box_losses = lambda *a, **k: (calls.append(1), inner(*a, **k))[1]
    try:
        got = crit(outputs, targets)
    finally:
        losses.matched_box_losses = inner
    assert len(calls) == 3                                   # the three matched sets; the denoising set went by its list of pairs
    for (a, b), (wa, wb) in zip(crit.box_losses, want_losses):
        assert abs(a.item() - wa) <= 1e-5 * max(1.0, abs(wa)) and abs(b.item() - wb) <= 1e-5 * max(1.0, abs(wb))
    # the class maps come out of the criterion's own calc_cls_loss, handed the batch as one pseudo-image
    assert len(crit.class_maps) == 4 and all(torch.equal(a, b) for a, b in zip(crit.class_maps, want_maps))
    assert abs(crit.accuracy.item() - want_acc) < 1e-6
    torch.testing.assert_close(got, want, rtol=1e-5, atol=1e-5)
    grads = torch.autograd.grad(got, [o["pred_boxes"] for o in sets] + [cdn["pred_boxes"]] + [o["pred_logits"] for o in sets])
    for g, w in zip(grads, want_grads):
        torch.testing.assert_close(g, w, rtol=1e-5, atol=1e-6)
    assert crit.seen[0][0][0].device.type == "cuda"
    # the permutation index of a served result comes from the flat buffer; equal to the loop's own expression
    own = BoxLossCriterion._get_src_permutation_idx.__get__(crit)
    matcher = crit.matcher
    assert "_get_src_permutation_idx" in crit.__dict__
    matcher.prefetch(sets, targets, device_indices=True, flat=True)
    served = matcher(sets[1], targets)
    fast, slow = crit._get_src_permutation_idx(served), own(served)
    assert torch.equal(fast[0], slow[0]) and torch.equal(fast[1], slow[1])
    foreign = crit._get_src_permutation_idx(cdn_indices)
    assert torch.equal(foreign[1], torch.cat([s for s, _ in cdn_indices]))
    matcher.clear_prefetched()
    detach(crit)
    assert not {"calc_box_loss", "_get_src_permutation_idx", "calc_cls_loss"} & set(crit.__dict__)
    torch.testing.assert_close(crit(outputs, targets), want, rtol=1e-6, atol=1e-6)


class FocalLoss(torch.nn.Module):
    """The reference's classification criterion of that name, restated with stock torch ops
    (/root/reference/models/detr/loss.py:198-213, 217-242): input [B, n_cls, Q]."""

    def __init__(self, alpha, gamma, n_cls):
        super().__init__()
        self.alpha, self.gamma, self.n_cls = alpha, gamma, n_cls

    def forward(self, inputs, targets, num_boxes):
        F = torch.nn.functional
        inputs = inputs.transpose(1, 2)
        onehot = F.one_hot(targets, self.n_cls + 1)[:, :, :-1].float()
        prob = inputs.sigmoid()
        ce = F.binary_cross_entropy_with_logits(inputs, onehot, reduction="none")
        p_t = prob * onehot + (1 - prob) * (1 - onehot)
        loss = (self.alpha * onehot + (1 - self.alpha) * (1 - onehot)) * ce * ((1 - p_t) ** self.gamma)
        return loss.mean(1).sum() / num_boxes * inputs.shape[1]


class CrossEntropy(torch.nn.Module):
    """/root/reference/models/detr/loss.py:185-195."""

    def __init__(self, weight):
        super().__init__()
        self.criterion = torch.nn.CrossEntropyLoss(weight=weight)

    def forward(self, inputs, targets, num_boxes):
        return self.criterion(inputs, targets)


class ClsCriterion(BoxLossCriterion):
    """BoxLossCriterion with the classification loss behind a `cls_criterion` module, called as the reference's
    calc_cls_loss calls it (loss.py:127)."""

    def __init__(self, matcher, cls_criterion, n_cls):
        super().__init__(matcher)
        self.cls_criterion, self.n_cls = cls_criterion, n_cls

    def calc_cls_loss(self, outputs, targets, indices, check_acc=False):
        idx = self._get_src_permutation_idx(indices)
        logits = outputs["pred_logits"]
        target_cls = torch.full(logits.shape[:2], self.n_cls, device=logits.device)
        target_cls[idx] = torch.cat([t["labels"][i] for t, (_, i) in zip(targets, indices)])
        self.class_maps.append(target_cls)
        return self.cls_criterion(logits.transpose(1, 2), target_cls, max(len(idx[0]), 1))


@pytest.mark.gpu
@pytest.mark.parametrize("kind", ["focal", "ce"])
def test_fused_consumers_front_the_classification_criterion(kind):
    """attach(criterion, fused_consumers=True) puts the one-pass loss-and-gradient kernel in front of the criterion's
    FocalLoss / CrossEntropy module: same loss, same gradients w.r.t. logits and boxes, for the matched sets and the
    denoising group; the module itself (and its state_dict) stays."""
    import detr_amd
    from detr_amd import losses, synth
    from detr_amd.criterion import attach, detach

    wl, sets, targets = synth.make_workload("cfg1" if kind == "ce" else "cfg3", device="cuda", batch=5)
    sets = sets[:3]
    n_cls = wl.n_label_classes
    for o in sets:
        o["pred_boxes"].requires_grad_(True)
        o["pred_logits"].requires_grad_(True)
    cdn = {"pred_logits": sets[0]["pred_logits"].detach().clone().requires_grad_(True),
           "pred_boxes": sets[0]["pred_boxes"].detach().clone().requires_grad_(True)}
    cdn_indices = [(torch.arange(min(3, len(t["labels"])), device="cuda"), torch.arange(min(3, len(t["labels"])), device="cuda"))
                   for t in targets]
    outputs = {"model": sets, "cdn": [cdn], "cdn_indices": cdn_indices, "cdn_targets": targets}
    leaves = [o["pred_logits"] for o in sets] + [cdn["pred_logits"]] + [o["pred_boxes"] for o in sets]

    def make():
        if kind == "focal":
            return FocalLoss(0.25, 2.0, n_cls)
        w = torch.ones(n_cls + 1)
        w[-1] = 0.1
        return CrossEntropy(w).cuda()

    plain = ClsCriterion(detr_amd.HungarianMatcher(**wl.matcher_kwargs), make(), n_cls)
    want = plain(outputs, targets)
    want_grads = torch.autograd.grad(want, leaves)

    crit = ClsCriterion(types.SimpleNamespace(**wl.matcher_kwargs), make(), n_cls)
    keys = sorted(crit.state_dict())
    attach(crit, fused_consumers=True)
    assert "forward" in crit.cls_criterion.__dict__ and sorted(crit.state_dict()) == keys
    calls = []
    name = "focal_loss" if kind == "focal" else "cross_entropy"
    inner = getattr(losses, name)
    setattr(losses, name, lambda *a, **k: (calls.append(1), inner(*a, **k))[1])
    try:
        got = crit(outputs, targets)
    finally:
        setattr(losses, name, inner)
    assert len(calls) == 4                       # three matched sets and the denoising set
    torch.testing.assert_close(got, want, rtol=1e-5, atol=1e-5)
    for g, w in zip(torch.autograd.grad(got, leaves), want_grads):
        torch.testing.assert_close(g, w, rtol=5e-5, atol=2e-6 * float(w.abs().max()))
    # a class value the reference's one_hot / cross_entropy refuses: raised when the criterion returns
    bad = [dict(t) for t in targets]
    k = next(i for i, t in enumerate(bad) if len(t["labels"]))
    bad[k]["labels"] = bad[k]["labels"].clone()
    outputs_bad = dict(outputs, cdn_targets=bad)
    bad[k]["labels"][0] = n_cls + 3
    with pytest.raises(RuntimeError):
        crit(outputs_bad, targets)
    torch.testing.assert_close(crit(outputs, targets), want, rtol=1e-5, atol=1e-5)     # the next step is fine again
    detach(crit)
    assert "forward" not in crit.cls_criterion.__dict__
    torch.testing.assert_close(crit(outputs, targets), want, rtol=1e-6, atol=1e-6)


class IgnoreCriterion(torch.nn.Module):
    """The part of Co-DETR's criterion that serves its Faster-RCNN style assigner, restated with stock torch ops
    (/root/reference/models/co_detr/loss.py:143-167, 234-237)."""

    def __init__(self, head, n_cls):
        super().__init__()
        self.matcher = types.SimpleNamespace(cls_loss="focal", cost_class=2.0, cost_bbox=5.0, cost_giou=2.0,
                                             focal_alpha=0.25, focal_gamma=2.0)
        self.aux_heads, self.n_cls = {"faster_rcnn": head}, n_cls

    def calc_cls_loss_with_ignore(self, outputs, targets, pos_indices, neg_indices):
        F = torch.nn.functional
        logits = outputs["pred_logits"]
        pidx = (torch.cat([torch.full_like(s, i) for i, (s, _) in enumerate(pos_indices)]), torch.cat([s for s, _ in pos_indices]))
        nidx = (torch.cat([torch.full_like(s, i) for i, s in enumerate(neg_indices)]), torch.cat(list(neg_indices)))
        num_boxes = max(len(pidx[0]), 1)
        target_cls = torch.full(logits.shape[:2], -1, device=logits.device)
        target_cls[pidx] = torch.cat([t["labels"][i] for t, (_, i) in zip(targets, pos_indices)])
        target_cls[nidx] = torch.cat([torch.full(i.shape, self.n_cls) for i in neg_indices]).to(logits.device)
        fg = target_cls != -1
        src, tgt = logits[fg], target_cls[fg]
        onehot = F.one_hot(tgt, self.n_cls + 1)[..., :-1].float()
        p = src.sigmoid()
        ce = F.binary_cross_entropy_with_logits(src, onehot, reduction="none")
        p_t = p * onehot + (1 - p) * (1 - onehot)
        loss = (0.25 * onehot + 0.75 * (1 - onehot)) * ce * (1 - p_t) ** 2.0
        return loss[None].mean(1).sum() / num_boxes * src.shape[0]

    def forward(self, outputs, targets):
        total = 0.0
        for output in outputs["aux"]["faster_rcnn"]:
            pos, neg = self.aux_heads["faster_rcnn"](output, targets)
            total = total + self.calc_cls_loss_with_ignore(output, targets, pos, neg)
        return total


@pytest.mark.gpu
def test_fused_consumers_front_the_loss_with_ignored_predictions():
    from detr_amd import aux_assign, synth
    from detr_amd.criterion import attach, detach

    wl, sets, targets = synth.make_workload("cfg3", device="cuda", batch=6)
    sets = sets[:2]
    for o in sets:
        # predictions close to the targets, so that the assigner finds positives above its threshold
        boxes = o["pred_boxes"].clone()
        for b, t in enumerate(targets):
            k = len(t["labels"])
            boxes[b, :k] = t["boxes"] + 0.002
        o["pred_boxes"] = boxes
        o["pred_logits"].requires_grad_(True)
    outputs = {"aux": {"faster_rcnn": sets}}
    crit = IgnoreCriterion(aux_assign.FasterRCNN(), wl.n_label_classes)
    want = crit(outputs, targets)
    want_grads = torch.autograd.grad(want, [o["pred_logits"] for o in sets])
    attach(crit, fused_consumers=True)
    assert "calc_cls_loss_with_ignore" in crit.__dict__
    got = crit(outputs, targets)
    torch.testing.assert_close(got, want, rtol=1e-5, atol=1e-6)
    for g, w in zip(torch.autograd.grad(got, [o["pred_logits"] for o in sets]), want_grads):
        torch.testing.assert_close(g, w, rtol=5e-5, atol=2e-6 * float(w.abs().max()))
    detach(crit)
    assert "calc_cls_loss_with_ignore" not in crit.__dict__


@pytest.mark.gpu
@pytest.mark.parametrize("kind", ["focal", "ce"])
def test_set_losses_and_their_total_equal_the_loop_of_a_reference_style_criterion(kind):
    """detr_amd.losses.set_losses / set_loss_total: the body of Criterion.forward's loop (loss.py:164-176) and the group's
    weighted total (:135-158) -- the same per-set values, the same total and the same gradients w.r.t. logits and boxes as
    a criterion in the reference's style computes with stock torch ops."""
    import detr_amd
    from detr_amd import losses, synth

    wl, sets, targets = synth.make_workload("cfg1" if kind == "ce" else "cfg3", device="cuda", batch=5)
    sets = sets[:3]
    n_cls = wl.n_label_classes
    for o in sets:
        o["pred_boxes"].requires_grad_(True)
        o["pred_logits"].requires_grad_(True)
    leaves = [o["pred_logits"] for o in sets] + [o["pred_boxes"] for o in sets]
    weight = torch.ones(n_cls + 1)
    weight[-1] = 0.1
    weight = weight.cuda()
    cls_criterion = FocalLoss(0.25, 2.0, n_cls) if kind == "focal" else CrossEntropy(weight).cuda()
    matcher = detr_amd.HungarianMatcher(**wl.matcher_kwargs)
    plain = ClsCriterion(matcher, cls_criterion, n_cls)
    want_sets = []
    for o in sets:                                     # the loop of Criterion.forward
        plain.class_maps = []
        indices = matcher(o, targets)
        l1, giou = plain.calc_box_loss(o, targets, indices)
        want_sets.append((l1, giou, plain.calc_cls_loss(o, targets, indices)))
    # calc_total_loss: the mean of each kind over the layers, weighted and summed
    want = sum(w * sum(v[k] for v in want_sets) / len(want_sets) for k, w in enumerate((5.0, 2.0, 2.0)))
    want_grads = torch.autograd.grad(want, leaves)

    got_sets = losses.set_losses(matcher, sets, targets, n_cls, kind, class_weight=weight)
    for g3, w3 in zip(got_sets, want_sets):
        for g, w in zip(g3, w3):
            torch.testing.assert_close(g, w, rtol=1e-5, atol=1e-6)
    got = losses.set_loss_total(matcher, sets, targets, n_cls, kind, 5.0, 2.0, 2.0, class_weight=weight)
    torch.testing.assert_close(got, want, rtol=1e-5, atol=1e-5)
    for g, w in zip(torch.autograd.grad(got, leaves), want_grads):
        torch.testing.assert_close(g, w, rtol=5e-5, atol=2e-6 * float(w.abs().max()))
    total_sum = losses.set_loss_total(matcher, sets, targets, n_cls, kind, 5.0, 2.0, 2.0, class_weight=weight, reduction="sum")
    torch.testing.assert_close(total_sum, want * len(sets), rtol=1e-5, atol=1e-5)
    matcher.check_status()
    losses.check_cls_status()
    # logits that need no gradient: only the boxes get one
    frozen = [dict(o, pred_logits=o["pred_logits"].detach()) for o in sets]
    part = losses.set_loss_total(matcher, frozen, targets, n_cls, kind, 5.0, 2.0, 2.0, class_weight=weight)
    for g, w in zip(torch.autograd.grad(part, [o["pred_boxes"] for o in sets]), want_grads[3:]):
        torch.testing.assert_close(g, w, rtol=5e-5, atol=2e-6 * float(w.abs().max()))
    with pytest.raises(ValueError):
        losses.set_loss_total(matcher, sets, targets, n_cls, kind, reduction="max")
