#!/usr/bin/env python3
"""Summarise rocprofv3 outputs (kernel stats + pmc counter CSVs) for the mining kernels."""
import csv, glob, os, sys, collections
root = sys.argv[1]
for f in glob.glob(os.path.join(root, "trace", "**", "*kernel_stats.csv"), recursive=True):
    print("== kernel stats", f)
    for i, row in enumerate(csv.reader(open(f))):
        if i < 12: print(",".join(row))
for d in sorted(glob.glob(os.path.join(root, "pmc_*"))):
    if not os.path.isdir(d): continue
    for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        agg = collections.defaultdict(lambda: [0.0, 0])
        for row in csv.DictReader(open(f)):
            k = row.get("Kernel_Name", "")
            if "mine_kernel" not in k: continue
            c = row.get("Counter_Name"); v = float(row.get("Counter_Value", 0))
            agg[(k[:60], c)][0] += v; agg[(k[:60], c)][1] += 1
        for (k, c), (s, n) in sorted(agg.items()):
            print(f"{k:60s} {c:28s} per-launch {s/n:18.1f}  launches {n}")
