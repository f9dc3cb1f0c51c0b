#!/usr/bin/env python3
"""Kernel durations of the classification-loss kernels from a rocprofv3 kernel trace of
`python3 tools/clsbench.py B Q C` (the launches of that run in order: 110 focal calls without a gradient, 110 with,
then the forward + backward loops).

    rocprofv3 --kernel-trace --stats --output-format csv -d OUT -- python3 tools/clsbench.py 256 900 91
    python tools/cls_trace_summary.py OUT/.../*_kernel_trace.csv 256 900 91
"""
import csv
import sys

path, B, Q, C = sys.argv[1], int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4])
rows = sorted(csv.DictReader(open(path)), key=lambda r: int(r["Start_Timestamp"]))
seq = [(r["Kernel_Name"], (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3) for r in rows]
focal = [d for n, d in seq if "focal_loss_kernel" in n]
ce = [d for n, d in seq if "cross_entropy_" in n]
fin = [d for n, d in seq if "cls_loss_finish_kernel" in n]
mb_in = (4 * B * Q * C + 8 * B * Q) / 1e6
mb_both = (8 * B * Q * C + 8 * B * Q) / 1e6


def st(v):
    v = sorted(v)
    return "n=%d min %.1f median %.1f mean %.1f max %.1f us" % (len(v), v[0], v[len(v) // 2], sum(v) / len(v), v[-1])


def median(v):
    return sorted(v)[len(v) // 2]


print(f"rocprofv3 --kernel-trace, python3 tools/clsbench.py {B} {Q} {C} (one output set: {B * Q * C} logits)")
print(f"focal_loss_kernel, launches 11-110 (loss only, {mb_in:.1f} MB read):      ", st(focal[10:110]))
print(f"focal_loss_kernel, launches 121-220 (loss + gradient, {mb_both:.1f} MB):  ", st(focal[120:220]))
print(f"cross_entropy_kernel (loss + gradient, {mb_both:.1f} MB):                 ", st(ce[10:]))
print("cls_loss_finish_kernel (one workgroup, the partial sums in index order):  ", st(fin))
for name, v in (("focal", focal[120:220]), ("cross entropy", ce[10:])):
    m = median(v)
    print(f"{name}, loss + gradient: {mb_both:.1f} MB / {m:.1f} us = {mb_both / m:.2f} TB/s = {mb_both / m / 8 * 100:.0f} % of 8 TB/s")
