#!/usr/bin/env python3
"""Averages rocprofv3 --pmc counter_collection.csv rows per kernel and counter."""
import collections, csv, glob, sys
for d in sys.argv[1:]:
    for f in glob.glob(d + "/*/*counter_collection.csv"):
        acc = collections.defaultdict(lambda: collections.defaultdict(list))
        for r in csv.DictReader(open(f)):
            acc[r["Kernel_Name"][:70]][r["Counter_Name"]].append(float(r["Counter_Value"]))
        for k, v in acc.items():
            if "cost_" in k or "lap_" in k:
                print(k)
                for c, vals in v.items():
                    print("    %-28s n=%d mean=%.1f" % (c, len(vals), sum(vals) / len(vals)))
