"""detr_amd.criterion.attach on the REFERENCE's own Criterion classes (all six model families), in the build
container only: the modules are loaded from /root/reference by path (the GPU box has no reference: skipped there).

What is checked is the plumbing around the kernels, which needs no GPU: that `attach` finds the matcher and the
`aux_flag` of each real criterion, that its forward pre-hook foresees exactly the output sets each criterion's loop
hands to the matcher (`outputs['model']`, `outputs['first_stage']`, never the denoising groups), that the loop's calls
are served from the one batched call, and that the loss comes out the same as with the reference's own matcher.
The kernels are stood in for by the CPU oracle (tests may use it): `match_many` of the attached matcher is replaced by
a function that returns the oracle's indices, so the test runs on CPU tensors.
"""
import importlib.util
import os
import sys
import types

import numpy as np
import pytest
import torch

REF = "/root/reference"
pytestmark = pytest.mark.skipif(not os.path.isdir(REF), reason="needs the reference's source tree (build container only)")

FAMILIES = ["detr", "conditional_detr", "dab_detr", "dn_detr", "dino_detr", "co_detr"]


def _load(name, path, package_path=None):
    spec = importlib.util.spec_from_file_location(name, path, submodule_search_locations=package_path)
    mod = importlib.util.module_from_spec(spec)
    sys.modules[name] = mod
    spec.loader.exec_module(mod)
    return mod


@pytest.fixture(scope="module")
def reference():
    """The reference's matcher and the six loss modules, loaded by path behind the stand-ins the container needs:
    torchvision is absent (its box_area / box_iou are one-line definitions), and utils/misc.py imports PIL and
    torchvision at module level although the criteria only take the logging helper AverageMeter from it."""
    saved = dict(sys.modules)
    tv, ops, bx = (types.ModuleType(n) for n in ("torchvision", "torchvision.ops", "torchvision.ops.boxes"))
    bx.box_area = lambda b: (b[:, 2] - b[:, 0]) * (b[:, 3] - b[:, 1])

    def box_iou(a, b):
        lt, rb = torch.max(a[:, None, :2], b[:, :2]), torch.min(a[:, None, 2:], b[:, 2:])
        inter = (rb - lt).clamp(min=0).prod(-1)
        return inter / (bx.box_area(a)[:, None] + bx.box_area(b) - inter)

    ops.boxes, ops.box_iou, tv.ops = bx, box_iou, ops
    sys.modules.update({"torchvision": tv, "torchvision.ops": ops, "torchvision.ops.boxes": bx})
    for pkg in ("utils", "models"):
        m = types.ModuleType(pkg)
        m.__path__ = []
        sys.modules[pkg] = m
    _load("utils.box_ops", os.path.join(REF, "utils/box_ops.py"))
    misc = types.ModuleType("utils.misc")

    class AverageMeter:
        def update(self, val, n=1):
            pass

    misc.AverageMeter = AverageMeter
    sys.modules["utils.misc"] = misc
    matcher_mod = _load("models.matcher", os.path.join(REF, "models/matcher.py"))
    losses = {}
    for fam in FAMILIES:
        pkg = types.ModuleType("models." + fam)
        pkg.__path__ = [os.path.join(REF, "models", fam)]
        sys.modules["models." + fam] = pkg
        losses[fam] = _load(f"models.{fam}.loss", os.path.join(REF, "models", fam, "loss.py"))
    yield matcher_mod, losses
    for k in list(sys.modules):
        if k not in saved:
            del sys.modules[k]
    sys.modules.update(saved)


def _boxes(n, gen):
    return torch.cat([torch.rand(n, 2, generator=gen), torch.rand(n, 2, generator=gen) * 0.3 + 0.01], -1)


def _set(B, Q, C, gen, n_dn=0):
    """One output set; with n_dn the matched queries are the tail view the denoising models hand over."""
    logits = torch.randn(B, n_dn + Q, C, generator=gen) * 2.0
    boxes = _boxes(B * (n_dn + Q), gen).view(B, n_dn + Q, 4)
    if n_dn:
        (dl, ml), (db, mb) = logits.split([n_dn, Q], dim=1), boxes.split([n_dn, Q], dim=1)
        return {"pred_logits": ml, "pred_boxes": mb}, {"pred_logits": dl, "pred_boxes": db}
    return {"pred_logits": logits, "pred_boxes": boxes}, None


def _inputs(fam, cls_loss, gen):
    B, Q, L, n_cls, n_dn = 3, 40, 3, 2, 6
    C = n_cls + 1 if cls_loss == "ce" else n_cls
    sizes = [5, 0, 9]
    targets = [{"labels": torch.randint(0, n_cls, (t,), generator=gen), "boxes": _boxes(t, gen)} for t in sizes]
    denoising = fam in ("dn_detr", "dino_detr", "co_detr")
    pairs = [_set(B, Q, C, gen, n_dn if denoising else 0) for _ in range(L)]
    outputs = {"model": [p[0] for p in pairs]}
    if denoising:
        key = "noised" if fam == "dn_detr" else "cdn"
        outputs[key] = [p[1] for p in pairs]
        # the denoising groups bring their own targets and indices: query k of an image is matched to target k % T
        dn_targets = [t for t in targets]
        idx = [(torch.arange(min(n_dn, len(t["labels"]))), torch.arange(min(n_dn, len(t["labels"])))) for t in targets]
        outputs[key + "_indices"], outputs[key + "_targets"] = idx, dn_targets
    if fam in ("dino_detr", "co_detr"):
        outputs["first_stage"] = [_set(B, 70, C, gen)[0]]
    return outputs, targets, n_cls


class _KnowsItsTargets:
    device = torch.device("cpu")

    def __init__(self, targets):
        self.targets = targets

    def same_targets(self, targets):
        return targets is self.targets


def _oracle_match_many(matcher, calls):
    """Stands where the kernels are: the CPU oracle's indices for every set of the call, and the bookkeeping
    `prefetch` relies on (what match_many has made of the targets)."""
    from oracle import oracle

    def match_many(outputs_list, targets, *args, **kwargs):
        calls.append(len(outputs_list))
        kw = dict(cls_loss=matcher.cls_loss, cost_class=matcher.cost_class, cost_bbox=matcher.cost_bbox,
                  cost_giou=matcher.cost_giou, focal_alpha=matcher.focal_alpha, focal_gamma=matcher.focal_gamma)
        result = []
        for o in outputs_list:
            ref = oracle.match(np.ascontiguousarray(o["pred_logits"].detach().numpy()),
                               np.ascontiguousarray(o["pred_boxes"].detach().numpy()),
                               [t["labels"].numpy() for t in targets], [t["boxes"].numpy() for t in targets], **kw)
            result.append([(torch.from_numpy(np.asarray(i, dtype=np.int64)), torch.from_numpy(np.asarray(j, dtype=np.int64)))
                           for i, j in ref])
        matcher.__dict__["_packed"] = _KnowsItsTargets(targets)
        return result

    return match_many


@pytest.mark.parametrize("aux_flag", [True, False])
@pytest.mark.parametrize("cls_loss", ["ce", "focal"])
@pytest.mark.parametrize("fam", FAMILIES)
def test_attach_on_the_references_own_criteria(reference, fam, cls_loss, aux_flag):
    import detr_amd
    from detr_amd.criterion import attach

    matcher_mod, losses = reference
    gen = torch.Generator().manual_seed(FAMILIES.index(fam) * 10 + (cls_loss == "ce") + 2 * aux_flag)
    outputs, targets, n_cls = _inputs(fam, cls_loss, gen)
    ref_matcher = matcher_mod.HungarianMatcher(cls_loss=cls_loss, cost_class=2.0, cost_bbox=5.0, cost_giou=2.0)
    criterion = losses[fam].Criterion(matcher=ref_matcher, n_cls=n_cls, cls_loss=cls_loss, aux_flag=aux_flag)
    want = criterion(outputs, targets)

    matcher = attach(criterion)
    assert isinstance(matcher, detr_amd.HungarianMatcher) and criterion.matcher is matcher
    assert (matcher.cls_loss, matcher.cost_class, matcher.cost_bbox, matcher.cost_giou) == (cls_loss, 2.0, 5.0, 2.0)
    calls = []
    matcher.match_many = _oracle_match_many(matcher, calls)
    got = criterion(outputs, targets)
    # ONE batched call with exactly the sets the loop asks for; every call of the loop was served from it
    n_model = len(outputs["model"]) if aux_flag else 1
    n_first = len(outputs.get("first_stage", []))
    assert calls == [n_model + n_first], (fam, calls)
    assert matcher._prefetched is None
    torch.testing.assert_close(got, want, rtol=1e-5, atol=1e-6)
    # and again with new targets objects (a new step)
    fresh = [dict(t) for t in targets]
    torch.testing.assert_close(criterion(outputs, fresh), want, rtol=1e-5, atol=1e-6)
    assert calls == [n_model + n_first] * 2


def test_attach_swaps_the_auxiliary_assigners_of_co_detr(reference):
    """Co-DETR keeps its ATSS / Faster-RCNN assigners in a plain dict on the criterion (co_detr/loss.py:64): attach
    replaces them by the GPU ones with the same parameters."""
    from detr_amd import aux_assign
    from detr_amd.criterion import attach

    matcher_mod, losses = reference
    aux_mod = sys.modules["models.co_detr.auxiliary_head"]
    args = types.SimpleNamespace(aux_heads=["atss", "faster_rcnn"], atss_k=7)
    heads = aux_mod.get_aux_heads(args)
    heads["faster_rcnn"].pos_threshold = 0.6
    criterion = losses["co_detr"].Criterion(matcher=matcher_mod.HungarianMatcher("focal", 2.0, 5.0, 2.0), n_cls=2,
                                            cls_loss="focal", aux_heads=heads)
    attach(criterion)
    assert isinstance(criterion.aux_heads["atss"], aux_assign.ATSS) and criterion.aux_heads["atss"].k == 7
    rcnn = criterion.aux_heads["faster_rcnn"]
    assert isinstance(rcnn, aux_assign.FasterRCNN) and (rcnn.pos_threshold, rcnn.neg_threshold) == (0.6, 0.3)
    keep = losses["co_detr"].Criterion(matcher=matcher_mod.HungarianMatcher("focal", 2.0, 5.0, 2.0), n_cls=2,
                                       cls_loss="focal", aux_heads=aux_mod.get_aux_heads(args))
    attach(keep, aux_heads=False)
    assert type(keep.aux_heads["atss"]).__module__ == "models.co_detr.auxiliary_head"


@pytest.mark.parametrize("fam", FAMILIES)
def test_attach_with_fused_consumers_wraps_the_real_calc_box_loss(reference, fam):
    """fused_consumers=True puts a wrapper in front of the criterion's own calc_box_loss (the same signature in all six
    families).  Here no fused result exists (the stand-in for the kernels yields index tuples only), so every call
    must fall through to the criterion's own method and the loss must be unchanged."""
    from detr_amd.criterion import attach, detach

    matcher_mod, losses = reference
    gen = torch.Generator().manual_seed(100 + FAMILIES.index(fam))
    outputs, targets, n_cls = _inputs(fam, "focal", gen)
    criterion = losses[fam].Criterion(matcher=matcher_mod.HungarianMatcher("focal", 2.0, 5.0, 2.0), n_cls=n_cls,
                                      cls_loss="focal")
    want = criterion(outputs, targets)
    matcher = attach(criterion, fused_consumers=True)
    assert "calc_box_loss" in criterion.__dict__ and "_get_src_permutation_idx" in criterion.__dict__
    calls = []
    as_many = _oracle_match_many(matcher, calls)
    matcher.match_flat = lambda outs, tg, with_tuples=False: ([], as_many(outs, tg))
    matcher.match_many = as_many
    matcher.check_status = lambda: None          # (nothing ran on a device)
    torch.testing.assert_close(criterion(outputs, targets), want, rtol=1e-5, atol=1e-6)
    assert calls == [len(outputs["model"]) + len(outputs.get("first_stage", []))]
    detach(criterion)
    assert "calc_box_loss" not in criterion.__dict__ and "_get_src_permutation_idx" not in criterion.__dict__


def _cpu_flat_stand_in(matcher, calls):
    """match_flat(with_tuples=True) as the kernels would answer it, from the CPU oracle: FlatMatches whose tensors live on
    the CPU (batch / source / flat target index per matched pair, the packed labels and boxes) beside the tuples."""
    from detr_amd.matcher import FlatMatches

    as_many = _oracle_match_many(matcher, calls)

    def match_flat(outputs_list, targets, with_tuples=False):
        tuples = as_many(outputs_list, targets)
        sizes = [len(t["labels"]) for t in targets]
        off = np.concatenate([[0], np.cumsum(sizes)]).astype(np.int64)
        packed = types.SimpleNamespace(labels=torch.cat([t["labels"] for t in targets]),
                                       boxes=torch.cat([t["boxes"].reshape(-1, 4) for t in targets]),
                                       total=int(off[-1]), batch=len(targets), max_t=max(sizes))
        flats = []
        for o, per_image in zip(outputs_list, tuples):
            b_idx = torch.cat([torch.full_like(i, b) for b, (i, _) in enumerate(per_image)])
            src = torch.cat([i for i, _ in per_image])
            tgt = torch.cat([j for _, j in per_image])
            flat_t = torch.cat([j + int(off[b]) for b, (_, j) in enumerate(per_image)])
            flats.append(FlatMatches(b_idx, src, tgt, flat_t, packed, int(o["pred_logits"].shape[1])))
        return (flats, tuples) if with_tuples else flats

    return match_flat


def _torch_pair_losses(outputs_list, flats):
    """What detr_amd.losses.matched_box_losses computes, with stock torch ops on the flat indices (pairwise, O(pairs))."""
    result = []
    for o, f in zip(outputs_list, flats):
        a, g = o["pred_boxes"][f.batch_idx, f.src_idx], f.packed.boxes[f.tgt_flat]
        n = float(max(len(f.src_idx), 1))
        l1 = (a - g).abs().sum() / n
        a = torch.cat([a[:, :2] - 0.5 * a[:, 2:], a[:, :2] + 0.5 * a[:, 2:]], -1)
        g = torch.cat([g[:, :2] - 0.5 * g[:, 2:], g[:, :2] + 0.5 * g[:, 2:]], -1)
        area_a, area_g = (a[:, 2] - a[:, 0]) * (a[:, 3] - a[:, 1]), (g[:, 2] - g[:, 0]) * (g[:, 3] - g[:, 1])
        wh = (torch.min(a[:, 2:], g[:, 2:]) - torch.max(a[:, :2], g[:, :2])).clamp(min=0)
        inter = wh[:, 0] * wh[:, 1]
        union = area_a + area_g - inter
        ewh = (torch.max(a[:, 2:], g[:, 2:]) - torch.min(a[:, :2], g[:, :2])).clamp(min=0)
        earea = ewh[:, 0] * ewh[:, 1]
        result.append((l1, (1 - (inter / union - (earea - union) / earea)).sum() / n))
    return result


@pytest.mark.parametrize("aux_flag", [True, False])
@pytest.mark.parametrize("cls_loss", ["ce", "focal"])
@pytest.mark.parametrize("fam", FAMILIES)
def test_fused_consumers_inside_the_references_own_criteria(reference, fam, cls_loss, aux_flag, monkeypatch):
    """fused_consumers=True with flat results present: the real criteria's calc_box_loss is answered from the flat
    indices, their own calc_cls_loss runs on the one-pseudo-image stand-in and their own _get_src_permutation_idx is
    answered for it -- and the total loss equals the unmodified criterion's.  (The fused kernel is stood in for by
    its stock-torch equivalent, the matching by the CPU oracle.)"""
    from detr_amd import losses as dm_losses
    from detr_amd.criterion import attach

    matcher_mod, losses = reference
    gen = torch.Generator().manual_seed(200 + FAMILIES.index(fam) * 4 + (cls_loss == "ce") + 2 * aux_flag)
    outputs, targets, n_cls = _inputs(fam, cls_loss, gen)
    criterion = losses[fam].Criterion(matcher=matcher_mod.HungarianMatcher(cls_loss, 2.0, 5.0, 2.0), n_cls=n_cls,
                                      cls_loss=cls_loss, aux_flag=aux_flag)
    want = criterion(outputs, targets)
    matcher = attach(criterion, fused_consumers=True)
    calls, fused_calls = [], []
    matcher.match_flat = _cpu_flat_stand_in(matcher, calls)
    matcher.check_status = lambda: None
    monkeypatch.setattr(dm_losses, "matched_box_losses",
                        lambda outs, flats: (fused_calls.append(len(outs)), _torch_pair_losses(outs, flats))[1])
    got = criterion(outputs, targets)
    served = (len(outputs["model"]) if aux_flag else 1) + len(outputs.get("first_stage", []))
    assert calls == [served] and fused_calls == [1] * served          # every served set took the fused path, no other
    torch.testing.assert_close(got, want, rtol=1e-5, atol=1e-6)

    # the classification criterion: its forward is fronted by the one-pass loss (stood in for by stock torch ops on the
    # class-last view it is handed); every family's layout -- [B, C, Q], co_detr [B, Q, C] -- must arrive class-last
    assert "forward" in criterion.cls_criterion.__dict__
    cls_calls = []

    def focal_stand_in(x, cls, num_boxes, alpha, gamma, ignore_index=None):
        assert tuple(x.shape) == tuple(cls.shape) + (n_cls,)
        cls_calls.append("focal")
        onehot = torch.nn.functional.one_hot(cls, n_cls + 1)[..., :-1].float()
        p = x.sigmoid()
        ce = torch.nn.functional.binary_cross_entropy_with_logits(x, onehot, reduction="none")
        p_t = p * onehot + (1 - p) * (1 - onehot)
        return ((alpha * onehot + (1 - alpha) * (1 - onehot)) * ce * (1 - p_t) ** gamma).sum() / num_boxes

    def ce_stand_in(x, cls, weight=None, ignore_index=-100):
        assert tuple(x.shape) == tuple(cls.shape) + (n_cls + 1,)
        cls_calls.append("ce")
        return torch.nn.functional.cross_entropy(x.reshape(-1, x.shape[-1]), cls.reshape(-1), weight=weight,
                                                 ignore_index=ignore_index)

    monkeypatch.setattr(dm_losses, "_takes", lambda x, t: True)
    monkeypatch.setattr(dm_losses, "focal_loss", focal_stand_in)
    monkeypatch.setattr(dm_losses, "cross_entropy", ce_stand_in)
    monkeypatch.setattr(dm_losses, "check_cls_status", lambda device=None: None)
    calls.clear()
    torch.testing.assert_close(criterion(outputs, targets), want, rtol=1e-5, atol=1e-6)
    # every classification loss of the step went through it: the served sets and the denoising groups
    n_dn = sum(len(outputs.get(k, [])) for k in ("cdn", "noised")) if aux_flag else sum(1 for k in ("cdn", "noised") if k in outputs)
    assert cls_calls == ["ce" if cls_loss == "ce" else "focal"] * len(cls_calls) and len(cls_calls) >= served
    assert len(cls_calls) == served + n_dn, (fam, len(cls_calls), served, n_dn)
    from detr_amd.criterion import detach

    detach(criterion)
    assert "forward" not in criterion.cls_criterion.__dict__
    state = criterion.state_dict()
    assert all(".fallback" not in k for k in state)


@pytest.mark.parametrize("aux_flag", [True, False])
def test_fused_consumers_front_co_detrs_loss_with_ignore(reference, aux_flag, monkeypatch):
    """Co-DETR with its auxiliary heads (the reference's own ATSS / Faster-RCNN assigners, left in place here because the
    tensors are on the CPU): `calc_cls_loss_with_ignore` (co_detr/loss.py:143-167) is answered by the focal kernel on the
    dense map with the ignored rows marked -- stood in for by stock torch ops that leave those rows out -- and the total
    loss equals the unmodified criterion's."""
    from detr_amd import losses as dm_losses
    from detr_amd.criterion import attach, detach

    matcher_mod, losses = reference
    aux_mod = sys.modules["models.co_detr.auxiliary_head"]
    gen = torch.Generator().manual_seed(900 + aux_flag)
    outputs, targets, n_cls = _inputs("co_detr", "focal", gen)
    # predictions close to the targets, so that the IoU thresholds of the Faster-RCNN assigner find positives
    aux_sets = []
    for _ in range(2):
        o, _ = _set(3, 40, n_cls, gen)
        for b, t in enumerate(targets):
            k = len(t["labels"])
            o["pred_boxes"][b, :k] = t["boxes"] + 0.002
            o["pred_boxes"][b, k:2 * k] = t["boxes"] * torch.tensor([1.0, 1.0, 1.3, 1.3])
        aux_sets.append(o)
    outputs["aux"] = {"faster_rcnn": aux_sets, "atss": [_set(3, 40, n_cls, gen)[0]]}
    heads = aux_mod.get_aux_heads(types.SimpleNamespace(aux_heads=["atss", "faster_rcnn"], atss_k=5))
    criterion = losses["co_detr"].Criterion(matcher=matcher_mod.HungarianMatcher("focal", 2.0, 5.0, 2.0), n_cls=n_cls,
                                            cls_loss="focal", aux_flag=aux_flag, aux_heads=heads)
    want = criterion(outputs, targets)
    pos, neg = heads["faster_rcnn"](aux_sets[-1], targets)
    assert sum(len(s) for s, _ in pos) > 0 and sum(len(s) for s in neg) > 0         # the case is not vacuous

    matcher = attach(criterion, fused_consumers=True, aux_heads=False)
    assert "calc_cls_loss_with_ignore" in criterion.__dict__
    matcher.match_flat = _cpu_flat_stand_in(matcher, [])
    matcher.match_many = _oracle_match_many(matcher, [])          # (what the loop's calls reach once detached)
    matcher.check_status = lambda: None
    monkeypatch.setattr(dm_losses, "matched_box_losses", _torch_pair_losses)
    ignore_calls = []

    def focal_stand_in(x, cls, num_boxes, alpha, gamma, ignore_index=None):
        onehot = torch.nn.functional.one_hot(cls.clamp(min=0), n_cls + 1)[..., :-1].float()
        p = x.sigmoid()
        ce = torch.nn.functional.binary_cross_entropy_with_logits(x, onehot, reduction="none")
        p_t = p * onehot + (1 - p) * (1 - onehot)
        loss = (alpha * onehot + (1 - alpha) * (1 - onehot)) * ce * (1 - p_t) ** gamma
        if ignore_index is not None:
            ignore_calls.append(int((cls == ignore_index).sum()))
            loss = loss * (cls != ignore_index)[..., None]
        return loss.sum() / num_boxes

    monkeypatch.setattr(dm_losses, "_takes", lambda x, t: True)
    monkeypatch.setattr(dm_losses, "focal_loss", focal_stand_in)
    monkeypatch.setattr(dm_losses, "check_cls_status", lambda device=None: None)
    got = criterion(outputs, targets)
    torch.testing.assert_close(got, want, rtol=1e-5, atol=1e-6)
    assert len(ignore_calls) == (2 if aux_flag else 1) and all(n > 0 for n in ignore_calls)
    detach(criterion)
    assert "calc_cls_loss_with_ignore" not in criterion.__dict__
    torch.testing.assert_close(criterion(outputs, targets), want, rtol=1e-6, atol=1e-6)


class _PackedOnCpu:
    """Stands where PackedTargets is when the tensors are on the CPU: the packed boxes / labels and the offsets."""

    def __init__(self, targets, device):
        self.sizes = [int(t["boxes"].shape[0]) for t in targets]
        self.offsets_list = [0] + np.cumsum(self.sizes).tolist()
        self.total, self.batch, self.device = self.offsets_list[-1], len(targets), device
        self.boxes = torch.cat([t["boxes"].reshape(-1, 4) for t in targets])
        self.labels = torch.cat([t["labels"] for t in targets])


@pytest.mark.parametrize("cls_loss", ["ce", "focal"])
@pytest.mark.parametrize("fam", ["dn_detr", "dino_detr", "co_detr"])
def test_fused_consumers_take_the_denoising_groups_by_their_list_of_pairs(reference, fam, cls_loss, monkeypatch):
    """The denoising groups bring their own indices and targets (dino_detr/loss.py:170-172): with fused_consumers=True
    their box loss goes through the pair-list kernel (stood in for by stock torch ops on the three flat index tensors)
    and their class loss through the criterion's own body on the pseudo-image -- total loss equal to the unmodified
    criterion's, inside the reference's three real criteria that have such groups."""
    import detr_amd.matcher as dm_matcher
    from detr_amd import losses as dm_losses
    from detr_amd.criterion import attach

    matcher_mod, losses = reference
    gen = torch.Generator().manual_seed(300 + FAMILIES.index(fam) * 2 + (cls_loss == "ce"))
    outputs, targets, n_cls = _inputs(fam, cls_loss, gen)
    criterion = losses[fam].Criterion(matcher=matcher_mod.HungarianMatcher(cls_loss, 2.0, 5.0, 2.0), n_cls=n_cls,
                                      cls_loss=cls_loss)
    want = criterion(outputs, targets)
    matcher = attach(criterion, fused_consumers=True)
    matcher.match_flat = _cpu_flat_stand_in(matcher, [])
    matcher.check_status = lambda: None
    list_calls = []

    def pair_list_stand_in(pred, tgt_boxes, b, q, t):
        list_calls.append(int(b.shape[0]))
        one = types.SimpleNamespace(batch_idx=b, src_idx=q, tgt_flat=t, packed=types.SimpleNamespace(boxes=tgt_boxes))
        l1, giou = _torch_pair_losses([{"pred_boxes": pred}], [one])[0]
        return torch.stack([l1, giou]) * float(max(len(b), 1))

    monkeypatch.setattr(dm_losses, "matched_box_losses", _torch_pair_losses)
    monkeypatch.setattr(dm_losses, "pair_list_box_loss_sums", pair_list_stand_in)
    monkeypatch.setattr(dm_losses, "_takes_boxes", lambda p: True)
    monkeypatch.setattr(dm_losses, "check_cls_status", lambda device=None: None)
    monkeypatch.setattr(dm_matcher, "PackedTargets", _PackedOnCpu)
    got = criterion(outputs, targets)
    torch.testing.assert_close(got, want, rtol=1e-5, atol=1e-6)
    key = "noised" if fam == "dn_detr" else "cdn"
    n_pairs = sum(len(s) for s, _ in outputs[key + "_indices"])
    assert list_calls == [n_pairs] * len(outputs[key])            # every denoising set, and only those
