#!/usr/bin/env python3
"""Golden vectors for the Co-DETR auxiliary assigners, made by RUNNING THE REFERENCE's
models/co_detr/auxiliary_head.py (ATSS :17-60, FasterRCNN :63-96) on the CPU.

Runs only in the build container (needs /root/reference).  The module is loaded by path;
torchvision is not installed, so the two symbols it takes from it are one-line stand-ins with
torchvision's documented definitions: `box_area` (as in gen_golden.py) and
`torchvision.ops.box_iou(a, b) = inter / (area_a[:, None] + area_b - inter)`.

Each `aux/<case>.npz` holds the inputs (pred_boxes [B, Q, 4], target boxes packed + sizes), the
ATSS pairs for the case's k and the Faster-RCNN positives / negatives, concatenated over the
images with per-image counts.  The ATSS centre distances of the reference (torch.cdist) are
stored too: where two candidates are closer than its rounding noise the choice between them is
not a property of the algorithm.

Usage:  python tests/golden/gen_golden_aux.py
"""
import importlib.util
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
import gen_golden  # noqa: E402


def load_aux_module():
    gen_golden.load_reference()          # registers the torchvision stand-in modules and utils.box_ops
    ops = sys.modules["torchvision.ops"]

    def box_iou(a, b):
        area_a = (a[:, 2] - a[:, 0]) * (a[:, 3] - a[:, 1])
        area_b = (b[:, 2] - b[:, 0]) * (b[:, 3] - b[:, 1])
        lt = torch.max(a[:, None, :2], b[:, :2])
        rb = torch.min(a[:, None, 2:], b[:, 2:])
        wh = (rb - lt).clamp(min=0)
        inter = wh[:, :, 0] * wh[:, :, 1]
        return inter / (area_a[:, None] + area_b - inter)

    ops.box_iou = box_iou
    spec = importlib.util.spec_from_file_location(
        "ref_aux_head", os.path.join(gen_golden.REF, "models/co_detr/auxiliary_head.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def make_case(seed, B, Q, sizes, jitter):
    """Predictions that partly sit on the targets (jittered copies), so both assigners find positives."""
    gen = torch.Generator().manual_seed(seed)
    targets = [{"boxes": gen_golden.boxes(t, gen), "labels": torch.zeros(t, dtype=torch.int64)} for t in sizes]
    pred = gen_golden.boxes(B * Q, gen).view(B, Q, 4)
    for b, t in enumerate(targets):
        n = len(t["boxes"])
        for r in range(min(4, Q // max(n, 1))):        # a few noisy copies of every target
            if n:
                noise = (torch.rand(n, 4, generator=gen) - 0.5) * jitter
                pred[b, r * n:(r + 1) * n] = (t["boxes"] + noise * t["boxes"][:, 2:].repeat(1, 2)).clamp(0.005, 0.995)
    return pred, targets


CASES = {
    "q300_k9": dict(seed=71, B=4, Q=300, sizes=[16, 0, 37, 5], jitter=0.25, k=9),
    "q900_k9": dict(seed=72, B=3, Q=900, sizes=[60, 3, 100], jitter=0.15, k=9),
    "q100_k4": dict(seed=73, B=3, Q=100, sizes=[7, 1, 12], jitter=0.4, k=4),
    "q20_k20": dict(seed=74, B=2, Q=20, sizes=[3, 2], jitter=0.3, k=20),
    "all_empty": dict(seed=75, B=2, Q=50, sizes=[0, 0], jitter=0.3, k=5),
}


def main():
    aux = load_aux_module()
    os.makedirs(os.path.join(HERE, "aux"), exist_ok=True)
    for name, c in CASES.items():
        pred, targets = make_case(c["seed"], c["B"], c["Q"], c["sizes"], c["jitter"])
        outputs = {"pred_boxes": pred}
        a = aux.ATSS(c["k"])(outputs, targets)
        pos, neg = aux.FasterRCNN()(outputs, targets)
        dist = [torch.cdist(t["boxes"][:, :2], p[:, :2], p=2.0).numpy().reshape(-1) for t, p in zip(targets, pred)]
        cat = lambda xs: np.concatenate([np.asarray(x, dtype=np.int64).reshape(-1) for x in xs] + [np.zeros(0, np.int64)])
        np.savez_compressed(
            os.path.join(HERE, "aux", name + ".npz"),
            pred_boxes=pred.numpy(), sizes=np.asarray(c["sizes"], np.int64), k=np.int64(c["k"]),
            tboxes=np.concatenate([t["boxes"].numpy().reshape(-1, 4) for t in targets]).astype(np.float32),
            atss_src=cat([i for i, _ in a]), atss_tgt=cat([j for _, j in a]),
            atss_counts=np.asarray([len(i) for i, _ in a], np.int64),
            pos_src=cat([i for i, _ in pos]), pos_tgt=cat([j for _, j in pos]),
            pos_counts=np.asarray([len(i) for i, _ in pos], np.int64),
            neg=cat(neg), neg_counts=np.asarray([len(n) for n in neg], np.int64),
            ref_dist=np.concatenate(dist + [np.zeros(0, np.float32)]).astype(np.float32))
        print(f"{name:12s} atss pairs {sum(len(i) for i, _ in a):4d}  rcnn pos {sum(len(i) for i, _ in pos):4d}  "
              f"neg {sum(len(n) for n in neg):5d}")


if __name__ == "__main__":
    main()
