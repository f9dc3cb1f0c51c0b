#!/usr/bin/env python3
"""Device memory bandwidth reached by plain streaming kernels (copy, fill, sum) on this GPU:
the practical ceiling next to the nominal 8 TB/s."""
import torch

dev = torch.device("cuda", 0)
n = 1 << 28   # 1 GiB of float32
a = torch.empty(n, dtype=torch.float32, device=dev).normal_()
b = torch.empty_like(a)


def timeit(fn, iters=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e-3


gb = n * 4 / 1e9
t = timeit(lambda: b.copy_(a)); print(f"copy  : {2 * gb / t:8.1f} GB/s (read+write)")
t = timeit(lambda: b.zero_()); print(f"fill  : {gb / t:8.1f} GB/s (write)")
t = timeit(lambda: a.sum()); print(f"sum   : {gb / t:8.1f} GB/s (read)")
m = 1 << 25   # 128 MiB: fits the 256 MB infinity cache
t = timeit(lambda: b[:m].copy_(a[:m])); print(f"copy 128 MiB: {2 * m * 4 / 1e9 / t:8.1f} GB/s (read+write)")
