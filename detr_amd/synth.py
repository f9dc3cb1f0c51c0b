"""Seeded synthetic matcher inputs for the five benchmark/parity workloads.

Shapes follow what the reference's criteria hand to the matcher
(/root/reference/models/matcher.py:43-50): per output set ``pred_logits
[B, Q, C]`` and ``pred_boxes [B, Q, 4]`` (cxcywh in [0, 1]), per image
``labels int64[T]`` and ``boxes f32[T, 4]``.  Everything is drawn from a CPU
``torch.Generator`` so the GPU path and the CPU checker see identical bits.
"""
from dataclasses import dataclass, field
from typing import Dict, List

import numpy as np
import torch


@dataclass
class Workload:
    name: str
    batch: int
    set_queries: List[int]          # Q of every Hungarian-matched output set of one step
    n_logits: int                   # C, the logits' last dimension
    n_label_classes: int            # labels are drawn from [0, n_label_classes)
    cls_loss: str                   # 'ce' or 'focal'
    t_law: str                      # 'poisson16' or 'heavy_tail'
    matcher_kwargs: Dict = field(default_factory=dict)
    seed: int = 0


_W = dict(cost_class=2.0, cost_bbox=5.0, cost_giou=2.0, focal_alpha=0.25, focal_gamma=2.0)

WORKLOADS = {
    # DETR, CE class cost; C = 2 classes + no-object (detr/detr.py:20-22)
    "cfg1": Workload("cfg1", 8, [600] * 6, 3, 2, "ce", "poisson16", dict(cls_loss="ce", **_W), 1000),
    # Conditional / DAB-DETR, focal class cost
    "cfg2": Workload("cfg2", 64, [600] * 6, 2, 2, "focal", "poisson16", dict(cls_loss="focal", **_W), 2000),
    # DINO: 6 decoder sets + 1 two-stage encoder set
    "cfg3": Workload("cfg3", 64, [300] * 6 + [300], 2, 2, "focal", "poisson16", dict(cls_loss="focal", **_W), 3000),
    # DINO with the first stage over all H*W encoder tokens (25 x 42)
    "cfg3_hw": Workload("cfg3_hw", 64, [300] * 6 + [1050], 2, 2, "focal", "poisson16",
                        dict(cls_loss="focal", **_W), 3500),
    # the same with a larger image: 50 x 42 encoder tokens, beyond the 1088 columns of the one-wavefront kernel
    "cfg3_hw2": Workload("cfg3_hw2", 64, [300] * 6 + [2100], 2, 2, "focal", "poisson16",
                         dict(cls_loss="focal", **_W), 3600),
    # COCO-like stress
    "cfg4": Workload("cfg4", 256, [900] * 6, 91, 91, "focal", "heavy_tail", dict(cls_loss="focal", **_W), 4000),
    # Co-DETR scale: 6 layers x 2 heads
    # cfg4 with the softmax class cost of the reference's 'ce' models: 91 classes + no-object = 92 logits
    "cfg4_ce": Workload("cfg4_ce", 256, [900] * 6, 92, 91, "ce", "heavy_tail", dict(cls_loss="ce", **_W), 4000),
    # an LVIS-size class head (1203 classes): beyond what a 64-row logit tile in LDS holds
    "wide_focal": Workload("wide_focal", 4, [300, 100], 1203, 1203, "focal", "poisson16", dict(cls_loss="focal", **_W), 6000),
    "wide_ce": Workload("wide_ce", 4, [300, 100], 1204, 1203, "ce", "poisson16", dict(cls_loss="ce", **_W), 6100),
    "wide_focal_g15": Workload("wide_focal_g15", 3, [130], 700, 700, "focal", "heavy_tail",
                               dict(cls_loss="focal", **dict(_W, focal_gamma=1.5)), 6200),
    "cfg5": Workload("cfg5", 1024, [600] * 12, 2, 2, "focal", "poisson16", dict(cls_loss="focal", **_W), 5000),
}


def target_counts(law: str, batch: int, rng: np.random.Generator) -> np.ndarray:
    """Number of ground-truth boxes per image."""
    if law == "poisson16":
        return np.clip(rng.poisson(16.0, batch), 0, 100).astype(np.int64)
    if law == "heavy_tail":
        body = np.clip(np.rint(rng.lognormal(1.8, 0.9, batch)), 0, 100)
        tail = rng.integers(60, 101, batch)
        return np.where(rng.random(batch) < 0.8, body, tail).astype(np.int64)
    raise ValueError(law)


def random_boxes(n: int, gen: torch.Generator) -> torch.Tensor:
    """[n, 4] cxcywh boxes: centre U(0,1), size U(0.01, 0.31) -- never degenerate."""
    centre = torch.rand(n, 2, generator=gen)
    size = torch.rand(n, 2, generator=gen) * 0.30 + 0.01
    return torch.cat([centre, size], dim=-1)


def make_targets(counts, n_label_classes: int, gen: torch.Generator, device="cpu"):
    out = []
    for t in counts:
        t = int(t)
        out.append({
            "labels": torch.randint(0, n_label_classes, (t,), generator=gen, dtype=torch.int64).to(device),
            "boxes": random_boxes(t, gen).to(device),
        })
    return out


def make_output_set(batch: int, queries: int, n_logits: int, gen: torch.Generator, device="cpu"):
    logits = torch.randn(batch, queries, n_logits, generator=gen) * 2.0
    boxes = random_boxes(batch * queries, gen).view(batch, queries, 4)
    return {"pred_logits": logits.to(device), "pred_boxes": boxes.to(device)}


def make_workload(name: str, device="cpu", batch: int = None, rank: int = 0, counts=None):
    """Returns (workload, list of output-set dicts, targets) for one training step.

    ``rank`` offsets the seed so data-parallel ranks hold different images.  ``counts`` (targets
    per image) overrides the workload's law and sets the batch size: a rank's share of a batch
    whose images were dealt out by target count.
    """
    wl = WORKLOADS[name]
    if counts is not None:
        counts = np.asarray(counts, dtype=np.int64)
        B = len(counts)
    else:
        B = wl.batch if batch is None else batch
        rng = np.random.default_rng(wl.seed + 7919 * rank)
        counts = target_counts(wl.t_law, B, rng)
    gen = torch.Generator(device="cpu").manual_seed(wl.seed + 7919 * rank)
    targets = make_targets(counts, wl.n_label_classes, gen, device)
    sets = []
    for layer, q in enumerate(wl.set_queries):
        g = torch.Generator(device="cpu").manual_seed(wl.seed + 7919 * rank + layer + 1)
        sets.append(make_output_set(B, q, wl.n_logits, g, device))
    return wl, sets, targets
