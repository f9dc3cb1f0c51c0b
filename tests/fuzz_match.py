#!/usr/bin/env python3
"""Differential fuzz of HungarianMatcher (every path the shapes select: one kernel with the targets read in place,
packed targets, two kernels, the list-order kernel for T >= Q, many classes) against the CPU oracle: random class
counts, query counts (also different ones for the output sets of a call), batch sizes, weights, focal parameters, empty images, images with more targets than queries,
python-style negative labels.  An image counts as a mismatch when its pairs differ AND their total cost on the
oracle's cost block differs by more than 1e-5 relative."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

import detr_amd
from detr_amd import synth
from oracle import oracle

seconds = float(sys.argv[1]) if len(sys.argv) > 1 else 60.0
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 0)
bad = same = other = calls = 0
t_end, t_say = time.time() + seconds, time.time() + 20
while time.time() < t_end:
    if time.time() > t_say:
        print(f"... {calls} calls, {same + other} images, {bad} mismatches", flush=True)
        t_say += 20
    cls_loss = str(rng.choice(["ce", "focal"]))
    C = int(rng.choice([1, 2, 3, 4, 5, 17, 91, 92, 700])) if cls_loss == "focal" else int(rng.choice([2, 3, 4, 7, 92, 650]))
    Q = int(rng.choice([3, 10, 64, 65, 100, 300, 600, 832, 900, 1024, 1100]))
    B = int(rng.choice([1, 2, 5, 16, 40, 100, 128, 160]))      # (128: the most the in-place target table holds)
    if C >= 91:
        B = min(B, 5)
    S = int(rng.integers(1, 4))
    # output sets of different widths in one call (decoder layers beside a first stage): one launch shaped for the widest
    widths = [Q if rng.random() < 0.7 else int(rng.choice([50, 300, 600, 1050])) for _ in range(S)]
    counts = rng.poisson(rng.choice([3, 12, 30]), B)
    counts[rng.random(B) < 0.15] = 0
    if rng.random() < 0.2:
        counts[rng.integers(0, B)] = min(widths) + int(rng.integers(0, 5))          # an image with T >= Q
    kw = dict(cls_loss=cls_loss, cost_class=float(rng.choice([1.0, 2.0, 0.5])), cost_bbox=float(rng.choice([5.0, 1.0])),
              cost_giou=float(rng.choice([2.0, 1.0])), focal_alpha=float(rng.choice([0.25, 0.4])),
              focal_gamma=float(rng.choice([2.0, 2.0, 1.5, 1.0])))
    gen = torch.Generator().manual_seed(int(rng.integers(1 << 30)))
    n_label = C if cls_loss == "focal" else C - 1
    targets = synth.make_targets(counts, max(n_label, 1), gen, "cuda")
    if rng.random() < 0.3:                       # python-style negative labels address the same classes
        for t in targets:
            if len(t["labels"]):
                t["labels"] = torch.where(torch.rand(len(t["labels"]), device="cuda") < 0.3, t["labels"] - C, t["labels"])
    sets = [synth.make_output_set(B, q, C, gen, "cuda") for q in widths]
    matcher = detr_amd.HungarianMatcher(**kw)
    got = matcher.match_many(sets, targets)
    calls += 1
    s = int(rng.integers(0, S))
    ref, ref_cost = oracle.match(sets[s]["pred_logits"].cpu().numpy(), sets[s]["pred_boxes"].cpu().numpy(),
                                 [t["labels"].cpu().numpy() % C for t in targets], [t["boxes"].cpu().numpy() for t in targets],
                                 return_cost=True, **kw)
    for b, ((g0, g1), (r0, r1)) in enumerate(zip(got[s], ref)):
        g0, g1 = g0.numpy(), g1.numpy()
        if np.array_equal(g0, r0) and np.array_equal(g1, r1):
            same += 1
            continue
        cost = ref_cost[b]                       # [Q, T_b]
        a = float(cost[g0, g1].astype(np.float64).sum()) if len(g0) else 0.0
        r = float(cost[r0, r1].astype(np.float64).sum()) if len(r0) else 0.0
        ok = len(g0) == len(r0) and len(set(g0.tolist())) == len(g0) and len(set(g1.tolist())) == len(g1) and \
            abs(a - r) <= 1e-5 * max(1.0, abs(r))
        other += 1 if ok else 0
        if not ok:
            bad += 1
            print("MISMATCH", kw, "C", C, "Q", widths, "set", s, "B", B, "image", b, "T", int(counts[b]), a, r, flush=True)
print(f"{calls} calls, {same} images with the oracle's pairs, {other} with another optimum of equal cost, {bad} mismatches")
