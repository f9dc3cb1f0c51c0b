#!/usr/bin/env python3
"""HBM-side traffic per launch of the cost kernels from two rocprofv3 counter passes.

    rocprofv3 --pmc FETCH_SIZE --kernel-trace -d gpurun_out/pmc_fetch -- python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline
    rocprofv3 --pmc WRITE_SIZE --kernel-trace -d gpurun_out/pmc_write -- python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline
    python tools/pmc_traffic.py gpurun_out/pmc_fetch gpurun_out/pmc_write \
        cost_small_kernel=cfg2 cost_tile_kernel=cfg4 lap_block_kernel=cfg2 > profiles/<name>.json

The trailing kernel=workload pairs record which workload each kernel ran in that command.

FETCH_SIZE and WRITE_SIZE are reported in units of 1024 bytes; on gfx950 FETCH_SIZE counts the
128-byte requests of wide streaming reads as 64 bytes, so it is doubled here.  The JSON maps a
kernel-name prefix to {"fetch_bytes", "write_bytes", "traffic_bytes", "launches"} (means per launch).
"""
import collections
import csv
import glob
import json
import sys


def collect(directory, counter):
    acc = collections.defaultdict(list)
    for f in glob.glob(directory + "/**/*counter_collection.csv", recursive=True):
        per_dispatch = collections.defaultdict(float)
        names = {}
        for r in csv.DictReader(open(f)):
            if r["Counter_Name"] != counter:
                continue
            key = r.get("Dispatch_Id") or r.get("Correlation_Id")
            per_dispatch[key] += float(r["Counter_Value"])     # one row per XCD / dimension instance
            names[key] = r["Kernel_Name"]
        for key, v in per_dispatch.items():
            acc[names[key]].append(v)
    return acc


def short(name):
    for tag in ("cost_small_kernel", "cost_tile_kernel", "lap_block_kernel", "lap_wave_kernel", "lap_generic_kernel"):
        if tag in name:
            return tag
    return None


def main():
    workloads = dict(a.split("=", 1) for a in sys.argv[3:])
    fetch = collect(sys.argv[1], "FETCH_SIZE")
    write = collect(sys.argv[2], "WRITE_SIZE")
    out = {}
    for name in sorted(set(fetch) | set(write)):
        tag = short(name)
        if tag is None:
            continue
        f, w = fetch.get(name, []), write.get(name, [])
        fb = 2.0 * 1024.0 * sum(f) / max(len(f), 1)
        wb = 1024.0 * sum(w) / max(len(w), 1)
        out[tag] = {"kernel": name[:120], "fetch_bytes": fb, "write_bytes": wb, "traffic_bytes": fb + wb,
                    "launches": max(len(f), len(w)), "workload": workloads.get(tag),
                    "note": "means per launch; FETCH_SIZE x2 (gfx950 tallies 128-B read requests as 64 B)"}
    json.dump(out, sys.stdout, indent=1)
    print()


if __name__ == "__main__":
    main()
