"""Per-kernel means of the counters in a rocprofv3 --pmc counter_collection.csv.  Usage: pmc_sq_summary.py <csv>..."""
import csv
import sys
from collections import defaultdict

acc = defaultdict(lambda: defaultdict(lambda: [0.0, 0]))
for path in sys.argv[1:]:
    for r in csv.DictReader(open(path)):
        k = r["Kernel_Name"].split("(")[0].replace("void ", "")
        if not k.startswith("k_"):
            continue
        a = acc[k][r["Counter_Name"]]
        a[0] += float(r["Counter_Value"])
        a[1] += 1
for k, cs in acc.items():
    print(k, {c: round(v[0] / v[1], 1) for c, v in sorted(cs.items())}, "launches", max(v[1] for v in cs.values()))
