#!/usr/bin/env python3
"""Host timeline of one match_many step (perf_counter stamps between its phases).

    python tools/steptrace.py [workload] [--per-layer]

--per-layer: the step as the reference's criterion loop drives it, one call per output set
(/root/reference/models/detr/loss.py:164-169): the phases are summed over the calls of a step."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

import detr_amd
from detr_amd import synth
from detr_amd.matcher import _raise_for_status

PER_LAYER = "--per-layer" in sys.argv
names_given = [a for a in sys.argv[1:] if not a.startswith("--")]
wl, sets, targets = synth.make_workload(names_given[0] if names_given else "cfg2", device="cuda")
m = detr_amd.HungarianMatcher(**wl.matcher_kwargs)
dev = torch.device("cuda", 0)
N = 300
acc = np.zeros(6)
S = len(sets)
pool = [[{k: v[:] for k, v in t.items()} for t in targets] for _ in range(4)]
done = torch.cuda.Event()
import ctypes
calls = [[o] for o in sets] if PER_LAYER else [sets]
for it in range(N + 20):
  tg = pool[it & 3]            # target tensors the matcher did not see in the step before
  for sets_of_call in calls:
    S = len(sets_of_call)
    t0 = time.perf_counter()
    packed = m._packed_for(tg, dev)            # checks + pointer table; the kernel reads the targets in place
    t1 = time.perf_counter()
    host, queries = m._launch_to_host(sets_of_call, packed, dev)     # output scan, pinned result tensor, the C entry point
    t2 = time.perf_counter()
    done.record()
    t3 = time.perf_counter()
    res = m._split(host[1:], S, packed.total, packed.sizes, queries, packed.offsets_list)
    t4 = time.perf_counter()
    if not done.query():
        done.synchronize()
    t5 = time.perf_counter()
    _raise_for_status(ctypes.c_int64.from_address(host.data_ptr()).value & 0xFFFFFFFF)
    t6 = time.perf_counter()
    if it >= 20:
        acc += np.diff([t0, t1, t2, t3, t4, t5, t6])
names = ["targets", "launch", "event", "tuples", "wait", "status"]
for n, v in zip(names, acc / N * 1e6):
    print(f"{n:14s} {v:8.1f} us")
print(f"{'total':14s} {acc.sum() / N * 1e6:8.1f} us" + (f"  ({len(calls)} calls per step)" if PER_LAYER else ""))
S = len(sets)
# the same step through the public call, with new target dicts every step (as a training loop has)
import gc
gc.disable()
for label, fresh in (("match_many, new target tensors every step", True), ("match_many, same target tensors", False)):
    for _ in range(20):
        m.match_many(sets, targets)
    torch.cuda.synchronize()
    t = time.perf_counter()
    for k in range(N):
        m.match_many(sets, pool[k & 3] if fresh else targets)
    torch.cuda.synchronize()
    print(f"{label}: {(time.perf_counter() - t) / N * 1e6:.1f} us")
for label, ahead in (("loop of per-set calls, new target tensors every step", False), ("the same loop behind prefetch()", True)):
    for k in range(20):
        [m(o, targets) for o in sets]
    torch.cuda.synchronize()
    t = time.perf_counter()
    for k in range(N):
        tg = pool[k & 3]
        if ahead:
            m.prefetch(sets, tg)
        for o in sets:
            held = m(o, tg)          # one set's indices at a time, as the criterion's loop keeps them
    torch.cuda.synchronize()
    print(f"{label}: {(time.perf_counter() - t) / N * 1e6:.1f} us")
