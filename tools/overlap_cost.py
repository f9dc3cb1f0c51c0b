#!/usr/bin/env python3
"""What the step's collective costs when it is issued from match_many's `while_waiting` hook (behind the kernels)
and when it is issued before the call: a single-rank RCCL group, so the numbers are host and launch cost only."""
import gc
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import torch.distributed as dist

os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
os.environ.setdefault("MASTER_PORT", "29534")
torch.cuda.set_device(0)
dev = torch.device("cuda", 0)
dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
import detr_amd
from detr_amd import synth

wl, sets, targets = synth.make_workload("cfg2", device=dev)
pool = [[{k: v[:] for k, v in t.items()} for t in targets] for _ in range(4)]
m = detr_amd.HungarianMatcher(**wl.matcher_kwargs)
counts = [1018.0] * 6


from detr_amd.dist import start_num_boxes_all_reduce

pend = [None]


def collective_main_stream():
    staged = torch.tensor(counts, dtype=torch.float32, pin_memory=True)
    x = staged.to(dev, non_blocking=True)
    work = dist.all_reduce(x, op=dist.ReduceOp.SUM, async_op=True)      # (a one-rank group still goes through RCCL)
    work.wait()
    return x.clamp_(min=1.0)


def collective():
    pend[0] = start_num_boxes_all_reduce(counts, dev, _always_reduce=True)     # on its own stream


def run(kind, n=300):
    gc.disable()
    for k in range(n + 30):
        if k == 30:
            torch.cuda.synchronize()
            t = time.perf_counter()
        if kind == "none":
            m.match_many(sets, pool[k & 3])
        elif kind == "main stream, before":
            collective_main_stream()
            m.match_many(sets, pool[k & 3])
        elif kind == "main stream, hook":
            m.match_many(sets, pool[k & 3], while_waiting=collective_main_stream)
        elif kind == "side stream, before":
            collective()
            m.match_many(sets, pool[k & 3])
            pend[0].result()
        else:
            m.match_many(sets, pool[k & 3], while_waiting=collective)
            pend[0].result()
    torch.cuda.synchronize()
    gc.enable()
    return (time.perf_counter() - t) / n * 1e6


for kind in ("none", "main stream, before", "main stream, hook", "side stream, before", "side stream, hook") * 2:
    print(f"{kind:22s} {run(kind):7.1f} us/step")
dist.destroy_process_group()
