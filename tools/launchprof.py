#!/usr/bin/env python3
"""Micro-timings of the host pieces of HungarianMatcher._launch (cfg2 shapes)."""
import ctypes
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

import detr_amd
from detr_amd import synth
from detr_amd._lib import DmSet, lib
from detr_amd.matcher import HungarianMatcher, PackedTargets

dev = torch.device("cuda", 0)
wl, sets, targets = synth.make_workload(sys.argv[1] if len(sys.argv) > 1 else "cfg2", device=dev)
packed = PackedTargets(targets, dev)
m = HungarianMatcher(**wl.matcher_kwargs)
S, B, M = len(sets), packed.batch, packed.total


def timeit(name, fn, n=2000, sync_every=50):
    for _ in range(20):
        fn()
    torch.cuda.synchronize()
    t = 0.0
    for i in range(n):
        t0 = time.perf_counter()
        fn()
        t += time.perf_counter() - t0
        if i % sync_every == sync_every - 1:
            torch.cuda.synchronize()
    print(f"{name:28s} {1e6 * t / n:7.2f} us")


timeit("_prepare_sets", lambda: m._prepare_sets(sets, B))
device, C, keep, queries = m._prepare_sets(sets, B)


def fill():
    arr = (DmSet * S)()
    off = 0
    for s, ((logits, boxes), q) in enumerate(zip(keep, queries)):
        stride = (q + 31) & ~31
        arr[s].logits, arr[s].boxes = logits.data_ptr(), boxes.data_ptr()
        arr[s].cost_off, arr[s].out_off = off, s * M
        arr[s].Q, arr[s].q_stride = q, stride
        off += M * stride
    return arr, off


timeit("DmSet fill", fill)
arr, off = fill()
timeit("empty cost + zeros out", lambda: (torch.empty(off, dtype=torch.float32, device=dev),
                                          torch.zeros(2 * S * M + 1, dtype=torch.int64, device=dev)))
timeit("current_stream", lambda: torch.cuda.current_stream(dev).cuda_stream)


def ctx():
    with torch.cuda.device(dev):
        pass


timeit("with torch.cuda.device", ctx)
cost = torch.empty(off, dtype=torch.float32, device=dev)
out = torch.zeros(2 * S * M + 1, dtype=torch.int64, device=dev)
prm = m._params()
timeit("_params", m._params)
stream = torch.cuda.current_stream(dev).cuda_stream


def call():
    lib.dm_match_f32(arr, S, B, C, packed.labels.data_ptr(), packed.boxes.data_ptr(), packed.offsets_ptr,
                     packed.order_ptr, packed.max_t, ctypes.byref(prm), cost.data_ptr(), out.data_ptr(),
                     out.data_ptr() + 8 * S * M, out.data_ptr() + 16 * S * M, stream)


timeit("dm_match_f32 (ctypes)", call, sync_every=10)
lib.dm_profile_enable(1)
timeit("dm_match_f32 + events", call, sync_every=10)
lib.dm_profile_enable(0)
timeit("_launch (all of it)", lambda: m._launch(sets, packed), sync_every=10)
timeit("out[-1:]", lambda: out[-1:])
