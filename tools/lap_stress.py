#!/usr/bin/env python3
"""One-off stress of the assignment kernels against scipy: many random matrices with heavy ties,
rectangular both ways, with +inf entries, in every kernel mode.  Prints the number of mismatches."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from scipy.optimize import linear_sum_assignment

import detr_amd

seed = int(sys.argv[1]) if len(sys.argv) > 1 else 0
rounds = int(sys.argv[2]) if len(sys.argv) > 2 else 20
rng = np.random.default_rng(seed)
bad = total = 0
for r in range(rounds):
    mats = []
    for _ in range(64):
        kind = rng.integers(0, 5)
        q = int(rng.choice([5, 17, 64, 65, 100, 300, 600, 900, 1100, 2100]))
        t = int(rng.integers(1, min(q + 3, 110)))
        if kind == 0:
            m = rng.random((q, t), dtype=np.float32)
        elif kind == 1:
            m = rng.integers(0, 3, (q, t)).astype(np.float32)                 # heavy ties
        elif kind == 2:
            m = np.round(rng.random((q, t), dtype=np.float32) * 4, 1).astype(np.float32) - 2.0
        elif kind == 3:
            m = np.zeros((q, t), dtype=np.float32)
        else:
            m = rng.random((q, t), dtype=np.float32)
            m[rng.random((q, t)) < 0.2] = np.inf
            m[rng.integers(0, q, t), np.arange(t)] = 1.0                       # keep it feasible
        mats.append(m)
    for mode in (0, 1, 2, 3):
        detr_amd._lib.set_knob("lap_kernel", mode)
        try:
            got = detr_amd.batched_linear_sum_assignment([torch.from_numpy(m).cuda() for m in mats])
        finally:
            detr_amd._lib.set_knob("lap_kernel", 0)
        for m, (gr, gc) in zip(mats, got):
            rs, cs = linear_sum_assignment(m)
            total += 1
            if not (np.array_equal(gr.numpy(), rs) and np.array_equal(gc.numpy(), cs)):
                bad += 1
                print("MISMATCH mode", mode, m.shape)
    print(f"round {r}: {total} problems, {bad} mismatches", flush=True)
sys.exit(1 if bad else 0)
