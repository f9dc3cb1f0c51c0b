#!/usr/bin/env python3
"""Host cost of the step's one collective (all_reduce_num_boxes) on a single-rank RCCL group."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import torch.distributed as dist

os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
os.environ.setdefault("MASTER_PORT", "29533")
torch.cuda.set_device(0)
dev = torch.device("cuda", 0)
dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
from detr_amd.dist import all_reduce_num_boxes

counts = [1018] * 6
for _ in range(20):
    all_reduce_num_boxes(counts, dev)
torch.cuda.synchronize()
N = 300
t0 = time.perf_counter()
for _ in range(N):
    all_reduce_num_boxes(counts, dev)
t1 = time.perf_counter()
torch.cuda.synchronize()
print(f"all_reduce_num_boxes host {1e6 * (t1 - t0) / N:.1f} us/call (world size 1: the division is skipped)")
x = torch.ones(6, device=dev)
t0 = time.perf_counter()
for _ in range(N):
    dist.all_reduce(x)
t1 = time.perf_counter()
torch.cuda.synchronize()
print(f"dist.all_reduce alone host {1e6 * (t1 - t0) / N:.1f} us/call")
dist.destroy_process_group()
