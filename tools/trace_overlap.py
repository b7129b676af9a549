"""Overlap picture of the bench's kernels from a rocprofv3 --kernel-trace CSV: for the last 40 % of the trace (steady state) the
share of wall time in which 0, 1, 2, 3+ kernels were in flight, and per kernel its mean duration and the mean number of OTHER
kernels in flight while it ran.  usage: trace_overlap.py <kernel_trace.csv>"""
import csv
import sys
from collections import defaultdict

rows = list(csv.DictReader(open(sys.argv[1])))
ev = []
for r in rows:
    n = r["Kernel_Name"].split("(")[0].replace("void ", "").split("<")[0]
    if not n.startswith("k_"):
        continue
    ev.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), n))
ev.sort()
t0, t1 = ev[0][0], max(e[1] for e in ev)
cut = t0 + (t1 - t0) * 6 // 10
ev = [e for e in ev if e[0] >= cut]
t0, t1 = ev[0][0], max(e[1] for e in ev)
pts = []
for s, e, n in ev:
    pts.append((s, 1))
    pts.append((e, -1))
pts.sort()
depth, last, hist = 0, t0, defaultdict(int)
for t, d in pts:
    hist[min(depth, 4)] += t - last
    last = t
    depth += d
tot = sum(hist.values())
print("kernels in flight -> share of wall time:", {k: round(v / tot, 3) for k, v in sorted(hist.items())})
per = defaultdict(lambda: [0, 0, 0.0])
for i, (s, e, n) in enumerate(ev):
    ov = 0
    for s2, e2, n2 in ev:
        if s2 >= e:
            break
        if e2 > s and (s2, e2, n2) != (s, e, n):
            ov += min(e, e2) - max(s, s2)
    p = per[n]
    p[0] += 1
    p[1] += e - s
    p[2] += ov / max(1, e - s)
for n, (c, d, o) in per.items():
    print(f"{n:18s} launches {c:4d}  mean {d / c / 1e3:8.1f} us  others in flight {o / c:4.2f}")
print("wall", (t1 - t0) / 1e6, "ms")
