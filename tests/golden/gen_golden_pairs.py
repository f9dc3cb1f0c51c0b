#!/usr/bin/env python3
"""Golden vectors for the box loss on index tuples that do not come from the matcher, made by RUNNING THE REFERENCE's
`Criterion.calc_box_loss` (models/detr/loss.py:97-114; the same method in all six families) + torch autograd on

  * `pairs_cdn`: indices and targets laid out like the denoising groups of DINO (models/dino_detr/dino_func.py:117-129):
    per image `num_group` copies of its (padded) targets, query k of group g matched to target slot g * max_obj + k,
    an image without objects contributing no pair;
  * `pairs_dup`: a prediction paired with several targets and a target paired with several predictions (what an
    IoU-threshold assigner yields, models/co_detr/auxiliary_head.py:69-96).

Each `pairs/<case>.npz` holds pred_boxes [B, Q, 4], the per-image target boxes (flattened, with sizes), the per-image
index pairs (flattened, with sizes), l1, giou and their gradients w.r.t. pred_boxes.

Runs only in the build container (needs /root/reference); loading as gen_golden_losses.py.
Usage:  python tests/golden/gen_golden_pairs.py
"""
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
import gen_golden  # noqa: E402
import gen_golden_losses  # noqa: E402


def save(name, loss_mod, pred, target_boxes, indices):
    os.makedirs(os.path.join(HERE, "pairs"), exist_ok=True)
    crit = loss_mod.Criterion(matcher=None, n_cls=2, cls_loss="focal")
    boxes = pred.clone().requires_grad_(True)
    targets = [{"boxes": b} for b in target_boxes]
    l1, giou = crit.calc_box_loss({"pred_boxes": boxes}, targets, indices)
    grad_l1 = torch.autograd.grad(l1, boxes, retain_graph=True)[0]
    grad_giou = torch.autograd.grad(giou, boxes)[0]
    np.savez_compressed(
        os.path.join(HERE, "pairs", name + ".npz"), pred_boxes=pred.numpy(),
        tboxes=torch.cat(target_boxes).numpy(), tsizes=np.array([len(b) for b in target_boxes], np.int64),
        src=torch.cat([s for s, _ in indices]).numpy(), tgt=torch.cat([t for _, t in indices]).numpy(),
        isizes=np.array([len(s) for s, _ in indices], np.int64),
        l1=np.float32(l1.item()), giou=np.float32(giou.item()), grad_l1=grad_l1.numpy(), grad_giou=grad_giou.numpy())
    print(f"{name:12s} pairs={sum(len(s) for s, _ in indices)} l1={l1.item():.6f} giou={giou.item():.6f}")


def main():
    loss_mod = gen_golden_losses.load_criterion_module(gen_golden.load_reference())
    gen = torch.Generator().manual_seed(4242)
    # denoising groups: B = 4 images with 3, 0, 5, 2 objects, padded to max_obj = 5, 3 groups, positives + negatives
    num_group, max_obj, sizes = 3, 5, [3, 0, 5, 2]
    Q = num_group * max_obj * 2
    pred = gen_golden.boxes(len(sizes) * Q, gen).view(len(sizes), Q, 4)
    target_boxes, indices = [], []
    for n in sizes:
        if n:
            own = gen_golden.boxes(n, gen)
            padded = torch.cat([own, own[torch.randint(0, n, (max_obj - n,), generator=gen)]]) if n < max_obj else own
            target_boxes.append(padded.repeat(num_group, 1))
        else:
            target_boxes.append(torch.zeros(0, 4))
        num_pos = max_obj if n else 0
        idx = torch.arange(num_pos).unsqueeze(0).repeat(num_group, 1)
        indices.append(((idx + torch.arange(num_group).unsqueeze(1) * num_pos * 2).view(-1),
                        (idx + torch.arange(num_group).unsqueeze(1) * num_pos).view(-1)))
    save("pairs_cdn", loss_mod, pred, target_boxes, indices)
    # one prediction with several targets, one target with several predictions, an image without pairs
    pred = gen_golden.boxes(3 * 12, gen).view(3, 12, 4)
    target_boxes = [gen_golden.boxes(4, gen), gen_golden.boxes(2, gen), gen_golden.boxes(3, gen)]
    indices = [(torch.tensor([5, 5, 5, 0, 11]), torch.tensor([0, 1, 3, 3, 3])),
               (torch.zeros(0, dtype=torch.int64), torch.zeros(0, dtype=torch.int64)),
               (torch.tensor([2, 7, 7]), torch.tensor([1, 1, 0]))]
    save("pairs_dup", loss_mod, pred, target_boxes, indices)


if __name__ == "__main__":
    main()
