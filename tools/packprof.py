import sys, time, os
sys.path.insert(0, os.getcwd())
import numpy as np, torch
import detr_amd
from detr_amd import synth
wl, sets, targets = synth.make_workload("cfg2", device="cuda")
dev = torch.device("cuda", 0)
def bench(f, N=300):
    for _ in range(20): f()
    torch.cuda.synchronize(); t = time.perf_counter()
    for _ in range(N): f()
    torch.cuda.synchronize()
    return (time.perf_counter() - t) / N * 1e6
print("PackedTargets", bench(lambda: detr_amd.PackedTargets(targets, dev)))
print("sizes listcomp", bench(lambda: [int(t["boxes"].shape[0]) for t in targets]))
print("cat labels", bench(lambda: torch.cat([t["labels"] for t in targets])))
print("cat boxes", bench(lambda: torch.cat([t["boxes"] for t in targets])))
sizes = [int(t["boxes"].shape[0]) for t in targets]
def meta_pinned():
    meta = np.empty(2 * 64 + 1, dtype=np.int32); meta[0] = 0
    np.cumsum(sizes, out=meta[1:65]); meta[65:] = np.argsort(-np.asarray(sizes, dtype=np.int64), kind="stable")
    st = torch.empty(129, dtype=torch.int32, pin_memory=True); st.numpy()[:] = meta
    return st.to(dev, non_blocking=True)
print("meta pinned", bench(meta_pinned))
def meta_np_only():
    meta = np.empty(2 * 64 + 1, dtype=np.int32); meta[0] = 0
    np.cumsum(sizes, out=meta[1:65]); meta[65:] = np.argsort(-np.asarray(sizes, dtype=np.int64), kind="stable")
    return meta
print("meta numpy only", bench(meta_np_only))
def meta_pageable():
    return torch.from_numpy(meta_np_only()).to(dev)
print("meta pageable .to", bench(meta_pageable))
def meta_tensor():
    return torch.tensor(meta_np_only(), device=dev)
print("meta torch.tensor", bench(meta_tensor))
pin = torch.empty(129, dtype=torch.int32, pin_memory=True)
print("pinned alloc only", bench(lambda: torch.empty(129, dtype=torch.int32, pin_memory=True)))
def meta_reuse_pin():
    pin.numpy()[:] = meta_np_only(); return pin.to(dev, non_blocking=True)
print("meta reuse pinned", bench(meta_reuse_pin))
m = detr_amd.HungarianMatcher(**wl.matcher_kwargs)
print("_prepare_sets", bench(lambda: m._prepare_sets(sets, 64)))
print("zeros out", bench(lambda: torch.zeros(12217, dtype=torch.int64, device=dev)))
print("empty cost", bench(lambda: torch.empty(3664800, dtype=torch.float32, device=dev)))
out = torch.zeros(12217, dtype=torch.int64, device=dev)
def d2h():
    h = torch.empty(12217, dtype=torch.int64, pin_memory=True); h.copy_(out, non_blocking=True); torch.cuda.current_stream().synchronize(); return h
print("pinned alloc + d2h + sync", bench(d2h))
h = torch.empty(12217, dtype=torch.int64)
sz = sizes * 6
f = torch.ops.aten.unsafe_split_with_sizes.default
print("unsafe split x2", bench(lambda: (f(h[:6108], sz), f(h[6108:12216], sz))))
print("split x2", bench(lambda: (h[:6108].split(sz), h[6108:12216].split(sz))))
sz2 = sz * 2
print("unsafe split, one call for both sides", bench(lambda: f(h[:12216], sz2)))
print("Tensor.split, one call for both sides", bench(lambda: h[:12216].split(sz2)))
def one_call_pairs():
    parts = f(h[:12216], sz2)
    n = len(sz)
    pairs = list(zip(parts[:n], parts[n:]))
    return [pairs[s * 64:(s + 1) * 64] for s in range(6)]
def two_call_pairs():
    pairs = list(zip(f(h[:6108], sz), f(h[6108:12216], sz)))
    return [pairs[s * 64:(s + 1) * 64] for s in range(6)]
print("one call + pairs", bench(one_call_pairs))
print("two calls + pairs", bench(two_call_pairs))
import gc
gc.disable()
print("gc off: one call + pairs", bench(one_call_pairs))
print("gc off: two calls + pairs", bench(two_call_pairs))
