"""Kernel durations and the gaps between them inside one single-frame submission, from a rocprofv3 --kernel-trace CSV of
tools/latency_probe.py: usage graph_gaps.py <kernel_trace.csv>"""
import csv
import sys
from collections import defaultdict

import numpy as np

rows = []
with open(sys.argv[1]) as f:
    for r in csv.DictReader(f):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].split("(")[0].split("<")[0]))
rows.sort()
rows = rows[len(rows) // 3:]  # (skip the warm-up)
dur = defaultdict(list)
gap = defaultdict(list)
span = []
frame_start = None
prev = None
for s, e, n in rows:
    dur[n].append(e - s)
    if prev is not None:
        g = s - prev[1]
        if g < 60000:  # same frame
            gap[prev[2] + " -> " + n].append(g)
        else:
            if frame_start is not None:
                span.append(prev[1] - frame_start)
            frame_start = s
    else:
        frame_start = s
    prev = (s, e, n)
print("kernel durations (median ns):")
for n, v in dur.items():
    print(f"  {n:28s} {np.median(v):9.0f}   x{len(v)}")
print("gaps inside a frame (median ns):")
for n, v in gap.items():
    print(f"  {n:50s} {np.median(v):9.0f}   x{len(v)}")
print("first kernel start -> last kernel end, median ns:", np.median(span) if span else None)
