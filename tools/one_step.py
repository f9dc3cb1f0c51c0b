import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, detr_amd
from detr_amd import synth
wl, sets, targets = synth.make_workload("cfg2", device="cuda")
m = detr_amd.HungarianMatcher(**wl.matcher_kwargs)
pool = [[{k: v[:] for k, v in t.items()} for t in targets] for _ in range(4)]
for k in range(30):
    r = m.match_many(sets, pool[k & 3])
torch.cuda.synchronize()
