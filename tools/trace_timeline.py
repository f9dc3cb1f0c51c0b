#!/usr/bin/env python3
"""Prints the last kernels of a rocprofv3 --kernel-trace CSV as a timeline (start / end in us relative to the
first of them, queue, name): shows whether launches on different streams overlap."""
import csv
import glob
import sys

rows = []
for f in glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True):
    rows += list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
last = rows[-int(sys.argv[2]) if len(sys.argv) > 2 else -12:]
t0 = int(last[0]["Start_Timestamp"])
for r in last:
    print("%9.1f %9.1f  q%-3s %s" % ((int(r["Start_Timestamp"]) - t0) / 1e3, (int(r["End_Timestamp"]) - t0) / 1e3,
                                      r.get("Queue_Id", "?"), r["Kernel_Name"][:70]))
