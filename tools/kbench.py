#!/usr/bin/env python3
"""Kernel-level timing of the two kernels through the C ABI, back-to-back launches.

    python tools/kbench.py --workload cfg2 [--batch B] [--iters 200] [--which cost,lap]

Prints per-launch time from torch events over `iters` back-to-back launches, achieved
algorithmic GB/s for the cost kernel and problems/s for the assignment kernel.
"""
import argparse
import ctypes
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

import torch

import detr_amd
from detr_amd import synth
from detr_amd._lib import DmParams, DmSet, lib
from detr_amd.matcher import PackedTargets


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--workload", default="cfg2")
    ap.add_argument("--batch", type=int, default=None)
    ap.add_argument("--iters", type=int, default=200)
    ap.add_argument("--targets", type=int, default=None, help="the same number of targets for every image")
    ap.add_argument("--queries", type=int, default=None, help="this many queries in every output set instead of the workload's")
    ap.add_argument("--which", default="cost,lap,both,fused")
    ap.add_argument("--knob", action="append", default=[], metavar="NAME=VALUE",
                    help="launch-shape knob, e.g. lap_kernel=3, lap_max_waves=5, cost_tsplit=2, tile_waves=8")
    args = ap.parse_args()
    dev = torch.device("cuda", 0)
    for kv in args.knob:
        name, value = kv.split("=")
        detr_amd._lib.set_knob(name, int(value))
    if args.queries is not None:
        import dataclasses
        base = synth.WORKLOADS[args.workload]
        synth.WORKLOADS[args.workload] = dataclasses.replace(base, set_queries=[args.queries] * len(base.set_queries))
    counts = None
    if args.targets is not None:
        counts = [args.targets] * (args.batch or synth.WORKLOADS[args.workload].batch)
    wl, sets, targets = synth.make_workload(args.workload, device=dev, batch=args.batch, counts=counts)
    packed = PackedTargets(targets, dev)
    S, B, M = len(sets), packed.batch, packed.total
    C = wl.n_logits
    arr = (DmSet * S)()
    off = 0
    for s, o in enumerate(sets):
        q = o["pred_logits"].shape[1]
        st = (q + 31) & ~31
        arr[s].logits, arr[s].boxes = o["pred_logits"].data_ptr(), o["pred_boxes"].data_ptr()
        arr[s].cost_off, arr[s].out_off, arr[s].Q, arr[s].q_stride = off, s * M, q, st
        off += M * st
    cost = torch.empty(off, dtype=torch.float32, device=dev)
    out = torch.zeros(2 * S * M + 1, dtype=torch.int64, device=dev)
    m = detr_amd.HungarianMatcher(**wl.matcher_kwargs)
    prm = m._params()
    stream = torch.cuda.current_stream().cuda_stream
    status = out.data_ptr() + 16 * S * M

    def run_cost():
        rc = lib.dm_cost_f32(arr, S, B, C, packed.labels.data_ptr(), packed.boxes.data_ptr(), packed.offsets_ptr,
                             packed.max_t, ctypes.byref(prm), cost.data_ptr(), status, stream)
        assert rc == 0, rc

    def run_lap():
        rc = lib.dm_lap_f32(arr, S, B, cost.data_ptr(), packed.offsets_ptr, packed.order_ptr, packed.max_t,
                            out.data_ptr(), out.data_ptr() + 8 * S * M, status, stream)
        assert rc == 0, rc

    def timeit(fn, iters):
        for _ in range(10):
            fn()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(iters):
            fn()
        e1.record()
        torch.cuda.synchronize()
        return e0.elapsed_time(e1) / iters * 1e3  # us

    sizes = packed.sizes
    algo = sum(q * C * 4 + q * 16 + t * 24 + q * t * 4 for q in wl.set_queries for t in sizes)
    run_cost()
    torch.cuda.synchronize()
    if "cost" in args.which:
        us = timeit(run_cost, args.iters)
        print(f"cost: {us:8.2f} us/launch  {algo / us / 1e3:8.1f} GB/s algorithmic ({algo / 1e6:.1f} MB)  "
              f"frac of 8 TB/s = {algo / us / 1e3 / 8000:.3f}")
    if "lap" in args.which:
        us = timeit(run_lap, args.iters)
        print(f"lap : {us:8.2f} us/launch  {S * B / us:8.2f} problems/us  (P={S * B}, max T={packed.max_t})")
    if "match" in args.which:
        def run_match():
            rc = lib.dm_match_f32(arr, S, B, C, packed.labels.data_ptr(), packed.boxes.data_ptr(), packed.offsets_ptr,
                                  packed.order_ptr, packed.max_t, ctypes.byref(prm), cost.data_ptr(), out.data_ptr(),
                                  out.data_ptr() + 8 * S * M, status, stream)
            assert rc == 0, rc
        us = timeit(run_match, args.iters)
        print(f"match: {us:7.2f} us per dm_match_f32 (cost kernel in the targets' processing order)")
    if "fused" in args.which:
        def run_fused():
            rc = lib.dm_match_fused_f32(arr, S, B, C, packed.labels.data_ptr(), packed.boxes.data_ptr(),
                                        packed.offsets_ptr, packed.order_ptr, packed.max_t, ctypes.byref(prm),
                                        cost.data_ptr(), out.data_ptr(), out.data_ptr() + 8 * S * M, None, status, stream)
            assert rc == 0, rc
        us = timeit(run_fused, args.iters)
        print(f"fused: {us:7.2f} us per dm_match_fused_f32  {S * B / us:8.2f} problems/us")
    if "both" in args.which:
        def run_both():
            run_cost()
            run_lap()
        us = timeit(run_both, args.iters)
        print(f"both: {us:8.2f} us per cost+lap pair")
    print("status", int(out[-1].item()))


if __name__ == "__main__":
    main()
