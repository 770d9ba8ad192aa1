#!/usr/bin/env python3
"""Kernel durations and the gaps between consecutive kernels of a rocprofv3 --kernel-trace csv (the 384x192 engine's
launch train): python tools/trace_gaps.py <run_kernel_trace.csv>"""
import csv, sys, statistics as st
rows = sorted(csv.DictReader(open(sys.argv[1])), key=lambda r: int(r["Start_Timestamp"]))
prev = None
dur, gap = {}, {}
for r in rows:
    k = r["Kernel_Name"].split("(")[0].split("::")[-1][:40]
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    dur.setdefault(k, []).append(e - s)
    if prev is not None:
        gap.setdefault(k, []).append(s - prev)
    prev = e
for k in dur:
    d, g = dur[k], gap.get(k, [0])
    print(f"{k:42s} n={len(d):5d} duration median {st.median(d) / 1e3:7.2f} us (min {min(d) / 1e3:.2f})   gap before it: median {st.median(g) / 1e3:6.2f} us (min {min(g) / 1e3:.2f})")
