"""Summarise a rocprofv3 kernel trace CSV: span, GPU busy time (union of kernel intervals), summed kernel time."""
import csv
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
ev = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"][:40]) for r in rows)
lo = float(sys.argv[2]) if len(sys.argv) > 2 else 0.4
t0 = ev[int(len(ev) * lo)][0]
sel = [e for e in ev if e[0] >= t0]
span = sel[-1][1] - t0
busy = 0
cs, ce = sel[0][0], sel[0][1]
for s, e, _ in sel[1:]:
    if s > ce:
        busy += ce - cs
        cs, ce = s, e
    else:
        ce = max(ce, e)
busy += ce - cs
tot = sum(e - s for s, e, _ in sel)
per = {}
for s, e, n in sel:
    per[n] = per.get(n, 0) + (e - s)
print(f"span {span/1e6:.2f} ms  busy {busy/1e6:.2f} ms ({busy/span:.0%})  sum {tot/1e6:.2f} ms  concurrency {tot/busy:.2f}")
for n, v in sorted(per.items(), key=lambda kv: -kv[1])[:14]:
    print(f"  {v/tot:6.1%}  {n}")
