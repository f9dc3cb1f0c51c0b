#!/usr/bin/env python3
"""What a criterion in the reference's style gets out of each way of wiring the matcher in: forward + backward of one
step (all decoder-layer sets) with the loss code left as the reference has it -- a loop over the output sets, one
matcher call each, permutation index, per-image gathers, the [N, N] GIoU matrix and its diagonal, the dense class
map, a sigmoid focal classification loss (/root/reference/models/detr/loss.py:71-75, 97-127, 160-183; restated here
with stock torch ops).

    python tools/criterion_bench.py [workload]

  import swapped         from detr_amd import HungarianMatcher, nothing else
  attach                 + detr_amd.criterion.attach(criterion): one batched matcher call per step
  attach, device indices + device_indices=True: the loss code's gathers need no index upload
  attach, fused consumers + fused_consumers=True: calc_box_loss and _get_src_permutation_idx answered from the flat buffer,
                         calc_cls_loss on one pseudo-image, the classification criterion fronted by the one-pass kernel
                         (the line before it: the same with the criterion's own FocalLoss / CrossEntropy left in place)
  forward on set_losses  the loop of Criterion.forward replaced by detr_amd.losses.set_losses (an edited criterion)
  forward on set_loss_total  ... by detr_amd.losses.set_loss_total: the group's weighted total as one scalar
"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import torch.nn.functional as F

import detr_amd
from detr_amd import synth
from detr_amd.criterion import attach, detach, unfuse_cls_criterion


def xyxy(b):
    return torch.cat([b[:, :2] - 0.5 * b[:, 2:], b[:, :2] + 0.5 * b[:, 2:]], -1)


class FocalLoss(torch.nn.Module):              # /root/reference/models/detr/loss.py:198-213, 217-242
    def __init__(self, alpha, gamma, n_cls):
        super().__init__()
        self.alpha, self.gamma, self.n_cls = alpha, gamma, n_cls

    def forward(self, inputs, targets, num_boxes):
        inputs = inputs.transpose(1, 2)
        onehot = F.one_hot(targets, self.n_cls + 1)[:, :, :-1].float()
        prob = inputs.sigmoid()
        ce = F.binary_cross_entropy_with_logits(inputs, onehot, reduction="none")
        p_t = prob * onehot + (1 - prob) * (1 - onehot)
        loss = ce * ((1 - p_t) ** self.gamma)
        loss = (self.alpha * onehot + (1 - self.alpha) * (1 - onehot)) * loss
        return loss.mean(1).sum() / num_boxes * inputs.shape[1]


class CrossEntropy(torch.nn.Module):           # /root/reference/models/detr/loss.py:185-195
    def __init__(self, weight):
        super().__init__()
        self.criterion = torch.nn.CrossEntropyLoss(weight=weight)

    def forward(self, inputs, targets, num_boxes):
        return self.criterion(inputs, targets)


class Criterion(torch.nn.Module):
    def __init__(self, matcher, n_cls, cls_loss):
        super().__init__()
        self.matcher, self.n_cls, self.cls_loss, self.aux_flag = matcher, n_cls, cls_loss, True
        if cls_loss == "ce":
            weight = torch.ones(n_cls + 1)
            weight[-1] = 0.1
            self.cls_criterion = CrossEntropy(weight)
        else:
            self.cls_criterion = FocalLoss(0.25, 2.0, n_cls)

    def _get_src_permutation_idx(self, indices):
        return (torch.cat([torch.full_like(src, i) for i, (src, _) in enumerate(indices)]),
                torch.cat([src for src, _ in indices]))

    def calc_box_loss(self, outputs, targets, indices):
        idx = self._get_src_permutation_idx(indices)
        n = max(len(idx[0]), 1)
        a = outputs["pred_boxes"][idx]
        g = torch.cat([t["boxes"][j] for t, (_, j) in zip(targets, indices)])
        l1 = F.l1_loss(a, g, reduction="none").sum() / n
        a, g = xyxy(a), xyxy(g)
        area_a, area_g = (a[:, 2] - a[:, 0]) * (a[:, 3] - a[:, 1]), (g[:, 2] - g[:, 0]) * (g[:, 3] - g[:, 1])
        wh = (torch.min(a[:, None, 2:], g[:, 2:]) - torch.max(a[:, None, :2], g[:, :2])).clamp(min=0)
        inter = wh[..., 0] * wh[..., 1]
        union = area_a[:, None] + area_g - inter
        ewh = (torch.max(a[:, None, 2:], g[:, 2:]) - torch.min(a[:, None, :2], g[:, :2])).clamp(min=0)
        earea = ewh[..., 0] * ewh[..., 1]
        giou = inter / union - (earea - union) / earea
        return l1, (1 - torch.diag(giou)).sum() / n

    def calc_cls_loss(self, outputs, targets, indices):
        idx = self._get_src_permutation_idx(indices)
        n = max(len(idx[0]), 1)
        logits = outputs["pred_logits"]
        target_cls = torch.full(logits.shape[:2], self.n_cls, device=logits.device)
        target_cls[idx] = torch.cat([t["labels"][j] for t, (_, j) in zip(targets, indices)])
        return self.cls_criterion(logits.transpose(1, 2), target_cls, n)

    def calc_cls_loss_with_ignore(self, outputs, targets, pos_indices, neg_indices):      # co_detr/loss.py:143-167
        logits = outputs["pred_logits"]
        pidx = self._get_src_permutation_idx(pos_indices)
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
        for output in outputs["model"]:
            indices = self.matcher(output, targets)
            l1, giou = self.calc_box_loss(output, targets, indices)
            total = total + 5.0 * l1 + 2.0 * giou + 2.0 * self.calc_cls_loss(output, targets, indices)
        total = total / len(outputs["model"])
        if "aux" in outputs:            # Co-DETR's collaborative heads (co_detr/loss.py:222-241)
            for key, head in self.aux_heads.items():
                part = 0.0
                for output in outputs["aux"][key]:
                    if key == "atss":
                        indices = head(output, targets)
                        l1, giou = self.calc_box_loss(output, targets, indices)
                        cls = self.calc_cls_loss(output, targets, indices)
                    else:
                        pos, neg = head(output, targets)
                        l1, giou = self.calc_box_loss(output, targets, pos)
                        cls = self.calc_cls_loss_with_ignore(output, targets, pos, neg)
                    part = part + 5.0 * l1 + 2.0 * giou + 2.0 * cls
                total = total + part / len(outputs["aux"][key])
        if "cdn" in outputs:            # the denoising groups bring their indices and targets (dino_detr/loss.py:170-172)
            dn = 0.0
            for output in outputs["cdn"]:
                l1, giou = self.calc_box_loss(output, outputs["cdn_targets"], outputs["cdn_indices"])
                dn = dn + 5.0 * l1 + 2.0 * giou + 2.0 * self.calc_cls_loss(output, outputs["cdn_targets"], outputs["cdn_indices"])
            total = total + dn / len(outputs["cdn"])
        return total


class RewrittenCriterion(torch.nn.Module):
    """The same total with the loop replaced by detr_amd.losses.set_losses (what a maintainer who edits
    Criterion.forward gets)."""

    def __init__(self, matcher, n_cls, cls_loss):
        super().__init__()
        self.matcher, self.n_cls, self.cls_loss, self.one_scalar = matcher, n_cls, cls_loss, False
        weight = torch.ones(n_cls + 1)
        weight[-1] = 0.1
        self.register_buffer("weight", weight)

    def forward(self, outputs, targets):
        from detr_amd import losses

        if self.one_scalar:
            return losses.set_loss_total(self.matcher, outputs["model"], targets, self.n_cls, self.cls_loss, 5.0, 2.0, 2.0,
                                         class_weight=self.weight)
        per_set = losses.set_losses(self.matcher, outputs["model"], targets, self.n_cls, self.cls_loss,
                                    class_weight=self.weight)
        total = 0.0
        for l1, giou, cls in per_set:
            total = total + 5.0 * l1 + 2.0 * giou + 2.0 * cls
        return total / len(per_set)


args = [a for a in sys.argv[1:] if not a.startswith("--")]
with_aux = "--aux" in sys.argv          # + Co-DETR's auxiliary heads (ATSS, Faster-RCNN style), focal workloads only
with_dn = "--dn" in sys.argv            # + denoising groups laid out as /root/reference/models/dino_detr/dino_func.py:57-129 does
name = args[0] if args else "cfg2"
wl, sets, targets = synth.make_workload(name, device="cuda")
for o in sets:
    o["pred_boxes"].requires_grad_(True)
    o["pred_logits"].requires_grad_(True)
outputs = {"model": sets}
pool = [[{k: v[:] for k, v in t.items()} for t in targets] for _ in range(4)]
if with_dn:
    num_group, sizes = 5, [len(t["labels"]) for t in targets]
    max_obj = max(sizes)
    gen = torch.Generator().manual_seed(11)
    dn_targets, dn_indices = [], []
    for t, n in zip(targets, sizes):
        boxes, labels = t["boxes"], t["labels"]
        if 0 < n < max_obj:
            extra = torch.randint(0, n, (max_obj - n,), generator=gen).cuda()
            boxes, labels = torch.cat([boxes, boxes[extra]]), torch.cat([labels, labels[extra]])
        if n:
            boxes, labels = boxes.repeat(num_group, 1), labels.repeat(num_group)
        dn_targets.append({"boxes": boxes.contiguous(), "labels": labels.contiguous()})
        num_pos = max_obj if n else 0
        idx = torch.arange(num_pos).unsqueeze(0).repeat(num_group, 1)
        dn_indices.append(((idx + torch.arange(num_group).unsqueeze(1) * num_pos * 2).view(-1),
                           (idx + torch.arange(num_group).unsqueeze(1) * num_pos).view(-1)))
    q_dn = num_group * max_obj * 2
    outputs["cdn"] = [{"pred_logits": (torch.randn(wl.batch, q_dn, wl.n_logits, generator=gen) * 2).cuda().requires_grad_(True),
                       "pred_boxes": torch.cat([torch.rand(wl.batch, q_dn, 2, generator=gen),
                                                torch.rand(wl.batch, q_dn, 2, generator=gen) * 0.3 + 0.01], -1).cuda().requires_grad_(True)}
                      for _ in sets]
    outputs["cdn_targets"], outputs["cdn_indices"] = dn_targets, dn_indices
    print(f"+ denoising groups: {len(sets)} sets of {q_dn} queries, {sum(len(s) for s, _ in dn_indices)} pairs each")


def timeit(criterion, n=30):
    for k in range(5):
        criterion(outputs, pool[k & 3]).backward()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for k in range(n):
        loss = criterion(outputs, pool[k & 3])
        loss.backward()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3, float(loss.detach())


if with_aux:
    gen = torch.Generator().manual_seed(12)
    aux_sets = {}
    for key in ("atss", "faster_rcnn"):
        aux_sets[key] = []
        for o in sets:
            boxes = o["pred_boxes"].detach().clone()
            for b, t in enumerate(targets):          # some predictions close to the targets, as a trained head has
                k = len(t["labels"])
                boxes[b, :k] = (t["boxes"] + 0.003).clamp(0.01, 0.99)
            aux_sets[key].append({"pred_logits": (torch.randn(o["pred_logits"].shape, generator=gen) * 2).cuda().requires_grad_(True),
                                  "pred_boxes": boxes.requires_grad_(True)})
    outputs["aux"] = aux_sets
    print(f"+ auxiliary heads: ATSS(k=9) and Faster-RCNN on {len(sets)} sets each")
crit = Criterion(detr_amd.HungarianMatcher(**wl.matcher_kwargs), wl.n_label_classes, wl.cls_loss).cuda()
if with_aux:
    from detr_amd import aux_assign

    crit.aux_heads = {"atss": aux_assign.ATSS(9), "faster_rcnn": aux_assign.FasterRCNN()}
print(f"{name}: criterion forward + backward per step (ms), loss value")
print("  import swapped           %8.2f   %.6f" % timeit(crit))
attach(crit)
print("  attach                   %8.2f   %.6f" % timeit(crit))
detach(crit)
attach(crit, device_indices=True)
print("  attach, device indices   %8.2f   %.6f" % timeit(crit))
detach(crit)
attach(crit, fused_consumers=True)
unfuse_cls_criterion(crit.cls_criterion)
print("  fused, own class loss    %8.2f   %.6f" % timeit(crit))
detach(crit)
attach(crit, fused_consumers=True)
print("  attach, fused consumers  %8.2f   %.6f" % timeit(crit))
detach(crit)
if with_dn or with_aux:
    sys.exit(0)
new = RewrittenCriterion(detr_amd.HungarianMatcher(**wl.matcher_kwargs), wl.n_label_classes, wl.cls_loss).cuda()
print("  forward on set_losses    %8.2f   %.6f" % timeit(new))
new.one_scalar = True
print("  forward on set_loss_total %7.2f   %.6f" % timeit(new))
