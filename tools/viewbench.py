#!/usr/bin/env python3
"""Step time when the output sets are the views the denoising models hand over (the tail of every image's queries,
pred_logits.split([n_dn, Q], dim=1)[1]): read in place against copied to contiguous tensors first (what a matcher
that insists on contiguous inputs has to do every step).

    python tools/viewbench.py [workload] [n_dn]
"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

import detr_amd
from detr_amd import synth

name = sys.argv[1] if len(sys.argv) > 1 else "cfg3"
n_dn = int(sys.argv[2]) if len(sys.argv) > 2 else 200
wl, sets, targets = synth.make_workload(name, device="cuda")
views = []
for o in sets:
    out = {}
    for key, t in o.items():
        B, Q, K = t.shape
        wide = torch.zeros(B, n_dn + Q, K, device="cuda")
        wide[:, n_dn:] = t
        out[key] = wide.split([n_dn, Q], dim=1)[1]
    views.append(out)
m = detr_amd.HungarianMatcher(**wl.matcher_kwargs)
pool = [[{k: v[:] for k, v in t.items()} for t in targets] for _ in range(4)]


def run(label, make_sets, n=300):
    for k in range(30):
        m.match_many(make_sets(), pool[k & 3])
    torch.cuda.synchronize()
    t = time.perf_counter()
    for k in range(n):
        held = m.match_many(make_sets(), pool[k & 3])
    torch.cuda.synchronize()
    print(f"{name} {label:52s} {(time.perf_counter() - t) / n * 1e6:8.1f} us/step", flush=True)


for rep in range(2):
    run("contiguous tensors", lambda: sets)
    run(f"tail views (n_dn={n_dn}), read in place", lambda: views)
    run("tail views copied to contiguous tensors every step", lambda: [{k: v.contiguous() for k, v in o.items()} for o in views])
