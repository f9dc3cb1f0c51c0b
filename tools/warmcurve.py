#!/usr/bin/env python3
"""What a device-wide synchronisation does to the steps behind it (why bench.py runs a short untimed region bracketed
like the timed one before it measures):

    python tools/warmcurve.py plain|rehearse3|fence_only|stamps

75 steps without any device-wide synchronisation, then: two 20-step regions between fences (plain), the same behind a
3-step region (rehearse3) or behind two more fences (fence_only), or per-step stamps of the 20 steps behind a fence
(stamps: steps 2-7 take 90-115 us instead of 73-76)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, detr_amd
from detr_amd import synth
wl, sets, targets = synth.make_workload("cfg2", device="cuda")
m = detr_amd.HungarianMatcher(**wl.matcher_kwargs)
pool = [[{k: v[:] for k, v in t.items()} for t in targets] for _ in range(4)]
def fence():
    torch.cuda.synchronize(); torch.cuda.synchronize()
def region(n):
    fence(); t0 = time.perf_counter()
    for k in range(n):
        held = m.match_many(sets, pool[k & 3])
    fence(); return (time.perf_counter() - t0) / n * 1e6
def loop(n):
    for k in range(n):
        held = m.match_many(sets, pool[k & 3])
mode = sys.argv[1]
loop(75)
if mode == "plain":
    print(mode, "%.1f" % region(20), "%.1f" % region(20))
elif mode == "rehearse3":
    region(3); print(mode, "%.1f" % region(20), "%.1f" % region(20))
elif mode == "fence_only":
    fence(); fence(); print(mode, "%.1f" % region(20), "%.1f" % region(20))
elif mode == "stamps":
    fence(); ts = []
    for k in range(20):
        t0 = time.perf_counter(); held = m.match_many(sets, pool[k & 3]); ts.append((time.perf_counter() - t0) * 1e6)
    t0 = time.perf_counter(); fence(); tf = (time.perf_counter() - t0) * 1e6
    print(mode, " ".join("%.0f" % t for t in ts), "| trailing fence %.0f" % tf)
