#!/usr/bin/env python3
"""Summarise rocprofv3 --pmc counter_collection.csv files: per kernel, per counter, summed over dispatches."""
import collections
import csv
import sys

for f in sys.argv[1:]:
    rows = list(csv.DictReader(open(f)))
    agg = collections.defaultdict(lambda: collections.defaultdict(float))
    ndisp = collections.defaultdict(set)
    for r in rows:
        k = r["Kernel_Name"]
        agg[k][r["Counter_Name"]] += float(r["Counter_Value"])
        ndisp[k].add(r["Dispatch_Id"])
    print("==", f)
    for k, d in agg.items():
        if "greb" not in k:
            continue
        print(k[:90], " dispatches:", len(ndisp[k]))
        for c, v in sorted(d.items()):
            print(f"    {c:28s} {v:.5g}   per-dispatch {v / len(ndisp[k]):.5g}")
