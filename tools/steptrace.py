#!/usr/bin/env python3
"""Host timeline of one match_many step (perf_counter stamps between its phases)."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

import detr_amd
from detr_amd import synth
from detr_amd.matcher import PackedTargets, _raise_for_status

wl, sets, targets = synth.make_workload(sys.argv[1] if len(sys.argv) > 1 else "cfg2", device="cuda")
m = detr_amd.HungarianMatcher(**wl.matcher_kwargs)
dev = torch.device("cuda", 0)
N = 300
EV = torch.cuda.Event() if os.environ.get("USE_EVENT") else None
acc = np.zeros(6)
S = len(sets)
for it in range(N + 20):
    t0 = time.perf_counter()
    packed = PackedTargets(targets, dev)
    if os.environ.get("SYNC_AFTER_PACK"):
        torch.cuda.current_stream(dev).synchronize()
    t1 = time.perf_counter()
    out, queries, cost, strides = m._launch(sets, packed, own_buffers=True)
    t2 = time.perf_counter()
    n = 2 * S * packed.total + 1
    host = torch.empty(n, dtype=torch.int64, pin_memory=True)
    host.copy_(out[:n], non_blocking=True)
    if EV is not None:
        EV.record()
    if os.environ.get("SYNC_BEFORE_SPLIT"):
        torch.cuda.current_stream(dev).synchronize()
    t3 = time.perf_counter()
    res = None          # (drop the previous step's tuples first, as `indices = matcher(...)` does not: see below)
    res = m._split(host[1:], S, packed.total, packed.sizes, queries, packed.offsets_list)
    t4 = time.perf_counter()
    if EV is not None:
        while not EV.query():
            pass
    else:
        torch.cuda.current_stream(dev).synchronize()
    t5 = time.perf_counter()
    _raise_for_status(int(host[0]) & 0xFFFFFFFF)
    t6 = time.perf_counter()
    if it >= 20:
        acc += np.diff([t0, t1, t2, t3, t4, t5, t6])
names = ["pack", "launch", "d2h enqueue", "split", "sync wait", "status"]
for n, v in zip(names, acc / N * 1e6):
    print(f"{n:12s} {v:8.1f} us")
print(f"{'total':12s} {acc.sum() / N * 1e6:8.1f} us")
# the same step through the public call, with new target dicts every step (as a training loop has)
import gc
gc.disable()
pool = [[{k: v[:] for k, v in t.items()} for t in targets] for _ in range(4)]
for label, fresh in (("match_many, new target tensors every step", True), ("match_many, same target tensors", False)):
    for _ in range(20):
        m.match_many(sets, targets)
    torch.cuda.synchronize()
    t = time.perf_counter()
    for k in range(N):
        m.match_many(sets, pool[k & 3] if fresh else targets)
    torch.cuda.synchronize()
    print(f"{label}: {(time.perf_counter() - t) / N * 1e6:.1f} us")
