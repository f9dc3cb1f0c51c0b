#!/usr/bin/env python3
"""Where the host time of one match_many step goes (cProfile + coarse timers)."""
import cProfile
import os
import pstats
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

import detr_amd
from detr_amd import synth

name = sys.argv[1] if len(sys.argv) > 1 else "cfg2"
wl, sets, targets = synth.make_workload(name, device="cuda")
m = detr_amd.HungarianMatcher(**wl.matcher_kwargs)
for _ in range(20):
    m.match_many(sets, targets)
torch.cuda.synchronize()
t0 = time.perf_counter()
N = 200
for _ in range(N):
    m.match_many(sets, targets)
torch.cuda.synchronize()
print("us/step", (time.perf_counter() - t0) / N * 1e6)
packed = detr_amd.PackedTargets(targets, torch.device("cuda", 0))
t0 = time.perf_counter()
for _ in range(N):
    m.match_many(sets, packed)
torch.cuda.synchronize()
print("us/step with pre-packed targets", (time.perf_counter() - t0) / N * 1e6)
t0 = time.perf_counter()
for _ in range(N):
    m.match_many(sets, packed, device_indices=True)
torch.cuda.synchronize()
print("us/step pre-packed, device indices", (time.perf_counter() - t0) / N * 1e6)
pr = cProfile.Profile()
pr.enable()
for _ in range(N):
    m.match_many(sets, targets)
pr.disable()
pstats.Stats(pr).sort_stats("cumulative").print_stats(25)
t0 = time.perf_counter()
for _ in range(N):
    m.match_flat(sets, packed)
torch.cuda.synchronize()
print("us/step pre-packed, match_flat", (time.perf_counter() - t0) / N * 1e6)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(N):
    m.match_flat(sets, packed)
t1 = time.perf_counter()
torch.cuda.synchronize()
t2 = time.perf_counter()
print("match_flat host us/call", (t1 - t0) / N * 1e6, " drain us", (t2 - t1) * 1e6)
pr = cProfile.Profile()
pr.enable()
for _ in range(N):
    m.match_flat(sets, packed)
pr.disable()
torch.cuda.synchronize()
pstats.Stats(pr).sort_stats("tottime").print_stats(14)
