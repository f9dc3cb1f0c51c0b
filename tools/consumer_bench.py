#!/usr/bin/env python3
"""Time of the criterion's first steps after matching, for all output sets of one step:
(a) the reference's way: CPU index tuples -> per set permutation index, gathers, [N, N] GIoU
    + diag, dense class map with stock torch ops (loss.py:71-75, 97-125), forward + backward;
(a') the same stock torch ops handed DEVICE index tuples (match_many(device_indices=True), what
     detr_amd.criterion.attach(criterion, device_indices=True) serves the loop): no index upload per gather;
(b) detr_amd.losses on the FlatMatches of match_flat.
Matching itself is outside both timings."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

import detr_amd
from detr_amd import losses, synth


def xyxy(b):
    cx, cy, w, h = b.unbind(-1)
    return torch.stack([cx - 0.5 * w, cy - 0.5 * h, cx + 0.5 * w, cy + 0.5 * h], -1)


def giou_matrix(a, g):
    area_a = (a[:, 2] - a[:, 0]) * (a[:, 3] - a[:, 1])
    area_g = (g[:, 2] - g[:, 0]) * (g[:, 3] - g[:, 1])
    wh = (torch.min(a[:, None, 2:], g[:, 2:]) - torch.max(a[:, None, :2], g[:, :2])).clamp(min=0)
    inter = wh[..., 0] * wh[..., 1]
    union = area_a[:, None] + area_g - inter
    ewh = (torch.max(a[:, None, 2:], g[:, 2:]) - torch.min(a[:, None, :2], g[:, :2])).clamp(min=0)
    earea = ewh[..., 0] * ewh[..., 1]
    return inter / union - (earea - union) / earea


def reference_way(sets, targets, tuples, no_object):
    total = 0.0
    for o, indices in zip(sets, tuples):
        bi = torch.cat([torch.full_like(s, b) for b, (s, _) in enumerate(indices)])
        si = torch.cat([s for s, _ in indices])
        n = max(len(bi), 1)
        sel = o["pred_boxes"][bi, si]
        tgt = torch.cat([t["boxes"][j] for t, (_, j) in zip(targets, indices)])
        l1 = (sel - tgt).abs().sum() / n
        giou = (1 - torch.diag(giou_matrix(xyxy(sel), xyxy(tgt)))).sum() / n
        cls = torch.full(o["pred_logits"].shape[:2], no_object, device=sel.device)
        cls[bi, si] = torch.cat([t["labels"][j] for t, (_, j) in zip(targets, indices)])
        total = total + 5.0 * l1 + 2.0 * giou
    total.backward()


def mixed_way(sets, targets, tuples, flats, no_object):
    """What attach(criterion, device_indices=True, fused_consumers=True) leaves of the reference's work: the class map by
    stock torch ops on device tuples, the box losses by one fused call per output set."""
    total = 0.0
    for o, indices, f in zip(sets, tuples, flats):
        bi = torch.cat([torch.full_like(s, b) for b, (s, _) in enumerate(indices)])
        si = torch.cat([s for s, _ in indices])
        cls = torch.full(o["pred_logits"].shape[:2], no_object, device=bi.device)
        cls[bi, si] = torch.cat([t["labels"][j] for t, (_, j) in zip(targets, indices)])
        l1, giou = losses.matched_box_losses([o], [f])[0]
        total = total + 5.0 * l1 + 2.0 * giou
    total.backward()


def fused_way(sets, flats, no_object):
    got = losses.matched_box_losses(sets, flats)
    losses.target_class_maps(flats, no_object)
    total = sum(5.0 * l1 + 2.0 * g for l1, g in got)
    total.backward()


def fused_total(sets, flats, no_object):
    losses.target_class_maps(flats, no_object)
    losses.matched_box_loss_total(sets, flats, 5.0, 2.0, reduction="sum").backward()


def timeit(fn, n=50):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e6


name = sys.argv[1] if len(sys.argv) > 1 else "cfg2"
wl, sets, targets = synth.make_workload(name, device="cuda")
for o in sets:
    o["pred_boxes"].requires_grad_(True)
m = detr_amd.HungarianMatcher(**wl.matcher_kwargs)
tuples = m.match_many(sets, targets)
dev_tuples = m.match_many(sets, targets, device_indices=True)
flats = m.match_flat(sets, targets)
no_object = wl.n_logits - 1
print(f"{name}: reference-style torch ops {timeit(lambda: reference_way(sets, targets, tuples, no_object)):9.1f} us/step")
print(f"{name}: the same on device index tuples {timeit(lambda: reference_way(sets, targets, dev_tuples, no_object)):9.1f} us/step")
print(f"{name}: device tuples, class map by torch ops, box losses fused per set {timeit(lambda: mixed_way(sets, targets, dev_tuples, flats, no_object)):9.1f} us/step")
print(f"{name}: fused consumers           {timeit(lambda: fused_way(sets, flats, no_object)):9.1f} us/step")
print(f"{name}: fused, weighted total     {timeit(lambda: fused_total(sets, flats, no_object)):9.1f} us/step")
