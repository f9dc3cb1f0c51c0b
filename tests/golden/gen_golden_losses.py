#!/usr/bin/env python3
"""Golden vectors for the consumers of the matching indices, made by RUNNING THE REFERENCE's
criterion methods (models/detr/loss.py: `_get_src_permutation_idx` :71-75, `calc_box_loss`
:97-114, the target scatter of `calc_cls_loss` :116-127) on the inputs and indices of the
matcher golden cases in the parent directory.

Runs only in the build container (needs /root/reference).  Loading: as gen_golden.py does for
the matcher (box_area stand-in for the absent torchvision), plus
  * `models.matcher` is the reference's matcher module loaded by path (importing the `models`
    package itself would pull in the torchvision backbones);
  * `utils.misc` is a stand-in holding only a no-op `AverageMeter`: the reference's module
    imports PIL and torchvision at module level, and the criterion only uses `AverageMeter` for
    logging (`set_summary`), never in the values recorded here.
Each `loss/<case>.npz` holds: l1, giou (the two returned losses), grad_l1, grad_giou (their
gradients w.r.t. pred_boxes from torch autograd) and target_cls (the dense [B, Q] class map
handed to the classification criterion).

Usage:  python tests/golden/gen_golden_losses.py
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
import golden_io  # noqa: E402

CASES = ["focal_c2_q50", "ce_c3_q600", "focal_c91_q100", "focal_t_gt_q", "ce_t_gt_q", "all_empty", "dup_gt",
         "identical_preds", "masked_first_stage"]


def load_criterion_module(ref_matcher):
    models = types.ModuleType("models")
    models.__path__ = []
    models.matcher = ref_matcher
    misc = types.ModuleType("utils.misc")

    class AverageMeter:          # logging helper of the reference, not part of any recorded value
        def update(self, val, n=1):
            pass

    misc.AverageMeter = AverageMeter
    sys.modules.update({"models": models, "models.matcher": ref_matcher, "utils.misc": misc})
    spec = importlib.util.spec_from_file_location("ref_detr_loss", os.path.join(gen_golden.REF, "models/detr/loss.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def main():
    ref_matcher = gen_golden.load_reference()
    loss_mod = load_criterion_module(ref_matcher)
    os.makedirs(os.path.join(HERE, "loss"), exist_ok=True)
    for name in CASES:
        g = golden_io.load(name)
        outputs, targets = g.torch_inputs("cpu")
        indices = [(torch.from_numpy(np.ascontiguousarray(i)), torch.from_numpy(np.ascontiguousarray(j)))
                   for i, j in g.indices]
        n_cls = g.C - 1 if g.cls_loss == "ce" else g.C
        crit = loss_mod.Criterion(matcher=None, n_cls=n_cls, cls_loss=g.cls_loss)
        boxes = outputs["pred_boxes"].clone().requires_grad_(True)
        l1, giou = crit.calc_box_loss({"pred_boxes": boxes, "pred_logits": outputs["pred_logits"]}, targets, indices)
        zero = torch.zeros_like(boxes)
        grad_l1 = torch.autograd.grad(l1, boxes, retain_graph=True, allow_unused=True)[0]
        grad_giou = torch.autograd.grad(giou, boxes, allow_unused=True)[0]
        captured = {}

        class Recorder(torch.nn.Module):       # stands where the classification criterion is called (:127)
            def forward(self, logits_t, target_cls, num_boxes):
                captured["target_cls"] = target_cls.clone()
                captured["num_boxes"] = num_boxes
                return logits_t.sum() * 0.0

        crit.cls_criterion = Recorder()
        crit.calc_cls_loss(outputs, targets, indices)
        np.savez_compressed(
            os.path.join(HERE, "loss", name + ".npz"),
            l1=np.float32(l1.item()), giou=np.float32(giou.item()),
            grad_l1=(zero if grad_l1 is None else grad_l1).numpy(),
            grad_giou=(zero if grad_giou is None else grad_giou).numpy(),
            target_cls=captured["target_cls"].numpy().astype(np.int64),
            num_boxes=np.int64(captured["num_boxes"]), n_cls=np.int64(n_cls))
        print(f"{name:28s} l1={l1.item():.6f} giou={giou.item():.6f} num_boxes={captured['num_boxes']}")


if __name__ == "__main__":
    main()
