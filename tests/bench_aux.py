#!/usr/bin/env python3
"""(Not collected by pytest; lives under tests/ because it times the CPU checker.)  Auxiliary assigners: all output sets of one step on the GPU (kernels + D2H + building the
index lists) next to the numpy restatement of the reference's per-image loops on the host."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import numpy as np
import torch

from detr_amd.aux_assign import ATSS, FasterRCNN
from oracle import aux_assign
from test_gpu_aux import _workload

B, Q, S = 64, 900, 6
sets, targets = _workload(9, B=B, Q=Q, S=S)
dev_sets = [{"pred_boxes": s["pred_boxes"].cuda()} for s in sets]
dev_targets = [{k: v.cuda() for k, v in t.items()} for t in targets]


def timeit(fn, n):
    fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3


atss, rcnn = ATSS(9), FasterRCNN()
print(f"B={B} Q={Q} S={S} sum T={sum(len(t['boxes']) for t in targets)}")
print(f"ATSS        GPU {timeit(lambda: atss.assign_many(dev_sets, dev_targets), 20):8.2f} ms/step")
print(f"Faster-RCNN GPU {timeit(lambda: rcnn.assign_many(dev_sets, dev_targets), 20):8.2f} ms/step")
print(f"Faster-RCNN GPU {timeit(lambda: rcnn.dense_class_targets(dev_sets, dev_targets, 91), 20):8.2f} ms/step "
      "(dense class maps + pair counts, device-resident)")
tb = [t["boxes"].numpy() for t in targets]
t0 = time.perf_counter()
aux_assign.atss(sets[0]["pred_boxes"].numpy(), tb, 9)
print(f"ATSS        numpy restatement {(time.perf_counter() - t0) * S * 1e3:8.1f} ms/step (one set timed, x{S})")
t0 = time.perf_counter()
aux_assign.faster_rcnn(sets[0]["pred_boxes"].numpy(), tb)
print(f"Faster-RCNN numpy restatement {(time.perf_counter() - t0) * S * 1e3:8.1f} ms/step (one set timed, x{S})")
