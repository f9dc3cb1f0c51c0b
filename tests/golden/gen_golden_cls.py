#!/usr/bin/env python3
"""Golden vectors for the classification loss on the dense class map, made by RUNNING THE REFERENCE:

  * `Criterion.calc_cls_loss` of models/detr/loss.py (:116-133, with its `FocalLoss` :198-213 / `CrossEntropy`
    :185-195 and `sigmoid_focal_loss` :217-242) on the inputs and indices of matcher golden cases in the parent
    directory: `cls/<case>.npz` holds the loss and its gradient w.r.t. pred_logits (torch autograd); the logits are
    the matcher golden's, the class map is the one in `loss/<case>.npz`.
  * the same two modules called the way calc_cls_loss calls them (`logits.transpose(1, 2), target_cls, num_boxes`) on
    seeded synthetic inputs with other parameters (gamma 0 / 1.5 / 3, no alpha weighting, saturated logits, one class,
    class weights with ignored rows, 92 classes): `cls/<case>.npz` then also holds logits, target_cls and parameters.
  * `Criterion.calc_cls_loss_with_ignore` of models/co_detr/loss.py (:143-167) on synthetic positives / negatives:
    the class map with -1 = ignored is the one the method builds (captured), the loss and gradient are its own.

Runs only in the build container (needs /root/reference).  Loading: as gen_golden_losses.py (box_area stand-in, a
no-op AverageMeter); co_detr's loss module additionally imports its auxiliary_head module, which is loaded by path
with a one-line `box_iou` stand-in for the absent torchvision.

Usage:  python tests/golden/gen_golden_cls.py
"""
import importlib.util
import os
import sys
import types

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
import gen_golden  # noqa: E402
import gen_golden_losses  # noqa: E402
import golden_io  # noqa: E402

FROM_MATCHER_CASES = ["focal_c2_q50", "ce_c3_q600", "focal_c91_q100", "focal_t_gt_q", "ce_t_gt_q", "all_empty"]


def load_co_detr_loss():
    bx = sys.modules["torchvision.ops.boxes"]

    def box_iou(a, b):
        lt, rb = torch.max(a[:, None, :2], b[:, :2]), torch.min(a[:, None, 2:], b[:, 2:])
        inter = (rb - lt).clamp(min=0).prod(-1)
        return inter / (bx.box_area(a)[:, None] + bx.box_area(b) - inter)

    sys.modules["torchvision.ops"].box_iou = box_iou
    pkg = types.ModuleType("models.co_detr")
    pkg.__path__ = [os.path.join(gen_golden.REF, "models", "co_detr")]
    sys.modules["models.co_detr"] = pkg
    spec = importlib.util.spec_from_file_location("models.co_detr.loss",
                                                  os.path.join(gen_golden.REF, "models/co_detr/loss.py"))
    mod = importlib.util.module_from_spec(spec)
    sys.modules["models.co_detr.loss"] = mod
    spec.loader.exec_module(mod)
    return mod


def save(name, **arrays):
    os.makedirs(os.path.join(HERE, "cls"), exist_ok=True)
    np.savez_compressed(os.path.join(HERE, "cls", name + ".npz"), **arrays)
    print(f"{name:28s} loss={float(arrays['loss']):.7f}")


def class_map(B, Q, n_cls, gen, matched=0.1):
    cls = torch.full((B, Q), n_cls, dtype=torch.int64)
    pick = torch.rand(B, Q, generator=gen) < matched
    cls[pick] = torch.randint(0, n_cls, (int(pick.sum()),), generator=gen)
    return cls, max(int(pick.sum()), 1)


def main():
    ref_matcher = gen_golden.load_reference()
    loss_mod = gen_golden_losses.load_criterion_module(ref_matcher)

    for name in FROM_MATCHER_CASES:
        g = golden_io.load(name)
        outputs, targets = g.torch_inputs("cpu")
        indices = [(torch.from_numpy(np.ascontiguousarray(i)), torch.from_numpy(np.ascontiguousarray(j)))
                   for i, j in g.indices]
        n_cls = g.C - 1 if g.cls_loss == "ce" else g.C
        crit = loss_mod.Criterion(matcher=None, n_cls=n_cls, cls_loss=g.cls_loss)
        logits = outputs["pred_logits"].clone().requires_grad_(True)
        loss = crit.calc_cls_loss({"pred_logits": logits, "pred_boxes": outputs["pred_boxes"]}, targets, indices)
        (grad,) = torch.autograd.grad(loss, logits)
        extra = {}
        if g.cls_loss == "ce":
            extra["weight"] = crit.cls_criterion.criterion.weight.numpy()
        else:
            extra.update(alpha=np.float32(crit.cls_criterion.alpha), gamma=np.float32(crit.cls_criterion.gamma))
        save(name, kind=np.array("ce" if g.cls_loss == "ce" else "focal"), loss=np.float32(loss.item()),
             grad=grad.numpy(), n_cls=np.int64(n_cls), from_matcher_case=np.array(name), **extra)

    def focal_case(name, seed, B, Q, n_cls, alpha, gamma, spread=2.0):
        gen = torch.Generator().manual_seed(seed)
        logits = (torch.randn(B, Q, n_cls, generator=gen) * spread).requires_grad_(True)
        cls, num_boxes = class_map(B, Q, n_cls, gen)
        loss = loss_mod.FocalLoss(alpha=alpha, gamma=gamma, n_cls=n_cls)(logits.transpose(1, 2), cls, num_boxes)
        (grad,) = torch.autograd.grad(loss, logits)
        save(name, kind=np.array("focal"), logits=logits.detach().numpy(), target_cls=cls.numpy(),
             num_boxes=np.int64(num_boxes), n_cls=np.int64(n_cls), alpha=np.float32(alpha), gamma=np.float32(gamma),
             loss=np.float32(loss.item()), grad=grad.numpy())

    focal_case("syn_focal_gamma1p5", 501, 2, 37, 5, 0.4, 1.5)
    focal_case("syn_focal_gamma0", 502, 2, 37, 5, 0.25, 0.0)
    focal_case("syn_focal_gamma3", 503, 3, 21, 4, 0.25, 3.0)
    focal_case("syn_focal_no_alpha", 504, 2, 37, 5, -1.0, 2.0)
    focal_case("syn_focal_saturated", 505, 2, 50, 3, 0.25, 2.0, spread=15.0)
    focal_case("syn_focal_c1", 506, 2, 33, 1, 0.25, 2.0)
    focal_case("syn_focal_c91", 507, 2, 45, 91, 0.25, 2.0)

    def ce_case(name, seed, B, Q, n_cls, weight, ignored=0.0):
        gen = torch.Generator().manual_seed(seed)
        C = n_cls + 1
        logits = (torch.randn(B, Q, C, generator=gen) * 2.0).requires_grad_(True)
        cls, num_boxes = class_map(B, Q, n_cls, gen, matched=0.3)
        if ignored:
            cls[torch.rand(B, Q, generator=gen) < ignored] = -100
        loss = loss_mod.CrossEntropy(weight=weight)(logits.transpose(1, 2), cls, num_boxes)
        (grad,) = torch.autograd.grad(loss, logits)
        save(name, kind=np.array("ce"), logits=logits.detach().numpy(), target_cls=cls.numpy(), n_cls=np.int64(n_cls),
             weight=weight.numpy(), loss=np.float32(loss.item()), grad=grad.numpy())

    gen = torch.Generator().manual_seed(600)
    ce_case("syn_ce_weighted_c7", 601, 2, 41, 6, torch.rand(7, generator=gen) + 0.1, ignored=0.15)
    w92 = torch.ones(92)
    w92[-1] = 0.1
    ce_case("syn_ce_c92", 602, 2, 50, 91, w92)
    w2 = torch.ones(2)
    w2[-1] = 0.1
    ce_case("syn_ce_c2", 603, 3, 29, 1, w2)

    # Co-DETR: positives / negatives of an auxiliary assigner, every other prediction ignored
    co = load_co_detr_loss()
    gen = torch.Generator().manual_seed(700)
    B, Q, n_cls = 2, 30, 3
    logits = (torch.randn(B, Q, n_cls, generator=gen) * 2.0).requires_grad_(True)
    targets = [{"labels": torch.randint(0, n_cls, (4,), generator=gen)} for _ in range(B)]
    pos = [(torch.tensor([3, 7, 7, 12]), torch.tensor([0, 1, 2, 3])), (torch.tensor([1, 20]), torch.tensor([2, 0]))]
    neg = [torch.tensor([0, 1, 2, 15, 16, 29]), torch.tensor([5, 6, 7, 8])]
    crit = co.Criterion(matcher=None, n_cls=n_cls, cls_loss="focal", aux_heads={})
    captured = {}
    real_one_hot = co.F.one_hot

    def recording_one_hot(t, n):          # the class vector of the foreground rows, as the method hands it on (:165)
        captured["tgt"] = t.clone()
        return real_one_hot(t, n)

    co.F.one_hot = recording_one_hot
    try:
        loss = crit.calc_cls_loss_with_ignore({"pred_logits": logits}, targets, pos, neg)
    finally:
        co.F.one_hot = real_one_hot
    (grad,) = torch.autograd.grad(loss, logits)
    # the dense map the method builds (:153-159): -1, then the positives' labels, then n_cls on the negatives
    cls = torch.full((B, Q), -1, dtype=torch.int64)
    for b, (src, tgt) in enumerate(pos):
        cls[b, src] = targets[b]["labels"][tgt]
    for b, src in enumerate(neg):
        cls[b, src] = n_cls
    assert torch.equal(cls[cls != -1], captured["tgt"])
    num_boxes = max(sum(len(s) for s, _ in pos), 1)
    save("syn_focal_with_ignore", kind=np.array("focal"), logits=logits.detach().numpy(), target_cls=cls.numpy(),
         num_boxes=np.int64(num_boxes), n_cls=np.int64(n_cls), alpha=np.float32(0.25), gamma=np.float32(2.0),
         ignore_index=np.int64(-1), loss=np.float32(loss.item()), grad=grad.numpy())


if __name__ == "__main__":
    main()
