#!/usr/bin/env python3
"""Diagnostic: builds the library with -DDM_LAP_STAMPS into a scratch .so and prints the
per-problem cycle breakdown of the assignment kernel (staging vs solve, cycles per step)."""
import ctypes, os, subprocess, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
so = os.path.join(ROOT, "tools", "libdm_stamps.so")      # (built beforehand where there is no GPU: python tools/lapstamps.py --build)
src = [os.path.join(ROOT, "detr_amd/csrc", f) for f in ("cost_kernel.hip", "lap_kernel.hip", "pack_kernel.hip", "abi.hip")]
if "--build" in sys.argv or not os.path.exists(so):
    subprocess.check_call(["hipcc", "--offload-arch=gfx950", "-O3", "-ffp-contract=off", "-fPIC", "-shared", "-std=c++17",
                           "-DDM_LAP_STAMPS", "-Wno-pass-failed", "-I", os.path.join(ROOT, "include"),
                           "-I", os.path.join(ROOT, "detr_amd/csrc"), "-o", so] + src)
    if "--build" in sys.argv:
        sys.exit(0)
import detr_amd
from detr_amd import synth
from detr_amd._lib import DmSet
from detr_amd.matcher import PackedTargets
lib = ctypes.CDLL(so)
lib.dm_debug_set_knob(0, int(os.environ.get("DM_LAPMODE", "0")))      # knob 0: which assignment kernel
name = ([a for a in sys.argv[1:] if not a.startswith("--")] or ["cfg2"])[0]
dev = torch.device("cuda", 0)
wl, sets, targets = synth.make_workload(name, device=dev)
packed = PackedTargets(targets, dev)
S, B, M = len(sets), packed.batch, packed.total
arr = (DmSet * S)()
off = 0
for s, o in enumerate(sets):
    q = o["pred_logits"].shape[1]; st = (q + 31) & ~31
    arr[s].logits, arr[s].boxes = o["pred_logits"].data_ptr(), o["pred_boxes"].data_ptr()
    arr[s].cost_off, arr[s].out_off, arr[s].Q, arr[s].q_stride = off, s * M, q, st
    off += M * st
cost = torch.empty(off, dtype=torch.float32, device=dev)
out = torch.zeros(2 * S * M, dtype=torch.int64, device=dev)
P = S * B
status = torch.zeros(16 + P * 24 + P * 36, dtype=torch.int32, device=dev)
prm = detr_amd.HungarianMatcher(**wl.matcher_kwargs)._params()
vp = ctypes.c_void_p
lib.dm_match_f32.argtypes = [ctypes.POINTER(DmSet), ctypes.c_int, ctypes.c_int, ctypes.c_int, vp, vp, vp, vp, ctypes.c_int,
                             ctypes.c_void_p, vp, vp, vp, vp, vp]
for _ in range(3):
    rc = lib.dm_match_f32(arr, S, B, wl.n_logits, packed.labels.data_ptr(), packed.boxes.data_ptr(), packed.offsets_ptr,
                          packed.order_ptr, packed.max_t, ctypes.byref(prm), cost.data_ptr(), out.data_ptr(),
                          out.data_ptr() + 8 * S * M, status.data_ptr(), None)
    assert rc == 0
torch.cuda.synchronize()
raw = status[16:].cpu().numpy().view(np.uint64)
rec = raw[:P * 12].reshape(P, 12).astype(np.float64)
ext = raw[P * 12:P * 30].reshape(P, 18).astype(np.float64)
ext = ext[rec[:, 4] > 0]
rec = rec[rec[:, 4] > 0]
clk = rec[:, 5] / (rec[:, 2] * 10e-9) / 1e9   # s_memrealtime ticks at 100 MHz
print("problems", len(rec), "clock GHz median %.2f" % np.median(clk))
print("staging cycles median %.0f max %.0f" % (np.median(rec[:, 0]), rec[:, 0].max()))
print("solve cycles median %.0f max %.0f" % (np.median(rec[:, 1]), rec[:, 1].max()))
print("steps per problem median %.0f max %.0f; rows median %.0f max %.0f" % (np.median(rec[:, 3]), rec[:, 3].max(), np.median(rec[:, 4]), rec[:, 4].max()))
print("solve cycles per step (incl. row-end work) median %.0f" % np.median(rec[:, 1] / rec[:, 3]))
A = np.stack([rec[:, 3], rec[:, 4], np.ones(len(rec))], 1)
coef, *_ = np.linalg.lstsq(A, rec[:, 1], rcond=None)
print("least squares: cycles = %.0f * steps + %.0f * rows + %.0f" % tuple(coef))
seg = rec[:, 6:11] / rec[:, 3:4]
print("per-step segment cycles (median): seg0 %.0f | seg1 %.0f | seg2 %.0f | seg3 %.0f | seg4 %.0f" % tuple(np.median(seg, 0)))
print("sum of segments per step %.0f ; row-end etc per row %.0f" % (np.median(seg.sum(1)), np.median((rec[:, 1] - rec[:, 6:11].sum(1)) / rec[:, 4])))
if ext[:, 1].max() > 0:      # the workgroup kernel: arrival spread of its wavefronts at the step's barrier
    steps = rec[:, 3:4]
    print("per step (median over problems): arrival spread of the wavefronts %.0f cycles, last arrival -> wave 0 released %.0f" %
          (np.median(rec[:, 11] / rec[:, 3]), np.median(ext[:, 0] / rec[:, 3])))
    W = int(ext[:, 1].max())
    late = np.median(ext[:, 2:2 + W] / steps, 0)
    print("mean lateness per wavefront behind the first arrival:", " ".join("%.0f" % x for x in late))
