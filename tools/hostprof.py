#!/usr/bin/env python3
"""Cost of the host-side pieces of one match_many step, each timed alone (us per call, GPU box's host)."""
import ctypes
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

import detr_amd
from detr_amd import synth
from detr_amd.matcher import PackedTargets, _index_views
from detr_amd._lib import lib

wl, sets, targets = synth.make_workload(sys.argv[1] if len(sys.argv) > 1 else "cfg2", device="cuda")
m = detr_amd.HungarianMatcher(**wl.matcher_kwargs)
dev = torch.device("cuda", 0)
m.match_many(sets, targets)
torch.cuda.synchronize()
pool = [[{k: v[:] for k, v in t.items()} for t in targets] for _ in range(4)]
packed = PackedTargets(targets, dev, defer=True)
S, B, M = len(sets), packed.batch, packed.total
N = 2000


def bench(label, fn):
    for _ in range(50):
        fn()
    t = time.perf_counter()
    for _ in range(N):
        fn()
    print(f"{label:44s} {(time.perf_counter() - t) / N * 1e6:7.2f} us")


k = [0]
def nxt():
    k[0] += 1
    return pool[k[0] & 3]

bench("same_targets (miss)", lambda: packed.same_targets(nxt()))
bench("same_targets (hit)", lambda: packed.same_targets(targets))
bench("scan_targets", lambda: _index_views.scan_targets(nxt(), 0))
bench("PackedTargets(defer=True)", lambda: PackedTargets(nxt(), dev, defer=True))
bench("torch.empty(device uint8)", lambda: torch.empty(40000, dtype=torch.uint8, device=dev))
n = 2 * S * M + 1
bench("torch.empty(pinned int64)", lambda: torch.empty(n, dtype=torch.int64, pin_memory=True))
host = torch.empty(n, dtype=torch.int64, pin_memory=True)
def z():
    host[0] = 0
bench("host[0] = 0", z)
bench("scan_outputs", lambda: _index_views.scan_outputs(sets, B, M))
bench("current_stream().cuda_stream", lambda: torch.cuda.current_stream(dev).cuda_stream)
bench("_params_ref", m._params_ref)
ev = torch.cuda.Event()
st = torch.cuda.current_stream(dev)
bench("event.record", lambda: ev.record(st))
torch.cuda.synchronize()
bench("event.query", ev.query)
bench("int(host[0])", lambda: int(host[0]))
bench("dm_strerror (a trivial ctypes call)", lambda: lib.dm_strerror(0))
# the C call alone (ctypes marshalling + planning + hipLaunchKernel), prepared arguments, 100 launches in the queue
import detr_amd.matcher as mm
pk = m._packed_for(targets, dev)
scan = _index_views.scan_outputs(sets, B, M)
_, C, queries, setbytes, cost_off = scan
ws = m._workspace(dev, 0, max(cost_off, 1))
pointers, meta, _ = PackedTargets(pool[0], dev, defer=True)._unpacked
did = ctypes.c_int(0)
stream = mm._raw_stream(0)
sp = host.data_ptr()
args = (setbytes, S, B, C, pointers, meta, pk.max_t, m._params_ref(), None, None, None, ws["cost"].data_ptr(),
        sp + 8, sp + 8 + 8 * S * M, None, sp, stream, ctypes.byref(did))
for _ in range(5):
    lib.dm_match_targets_f32(*args)
torch.cuda.synchronize()
t = time.perf_counter()
for _ in range(100):
    lib.dm_match_targets_f32(*args)
dt = (time.perf_counter() - t) / 100 * 1e6
torch.cuda.synchronize()
print(f"{'dm_match_targets_f32 call (host side)':44s} {dt:7.2f} us")
arr = (detr_amd._lib.DmSet * S).from_buffer_copy(setbytes)
need = (ctypes.c_ubyte * 4096)()
bench("data_ptr()", host.data_ptr)


def launch():
    m._launch(sets, m._packed_for(nxt(), dev), own_buffers=True, host_out=host)
for _ in range(20):
    launch()
torch.cuda.synchronize()
t = time.perf_counter()
for _ in range(300):
    launch()
dt = (time.perf_counter() - t) / 300 * 1e6
torch.cuda.synchronize()
print(f"{'_packed_for + _launch, back to back (queue full)':44s} {dt:7.2f} us")
