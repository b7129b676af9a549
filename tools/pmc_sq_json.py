"""profiles/<round>_pmc_sq.json from a rocprofv3 --pmc pass on the SQ counters over tools/pmc_batch.py (3 batches of 16
frames).  Usage: pmc_sq_json.py <counter_collection.csv>... <out.json>
Per kernel and per frame: wave instructions by kind, SQ_WAVE_CYCLES / SQ_WAIT_ANY / SQ_BUSY_CYCLES (quad-cycles, see
MI355X_MICROARCH.md) and the share of its wave time a kernel's waves spend parked (SQ_WAIT_ANY / SQ_WAVE_CYCLES)."""
import csv
import json
import sys
from collections import defaultdict

acc = defaultdict(lambda: defaultdict(float))
launches = defaultdict(lambda: defaultdict(int))
for path in sys.argv[1:-1]:
    for r in csv.DictReader(open(path)):
        k = r["Kernel_Name"].split("(")[0].replace("void ", "").split("<")[0]
        if not k.startswith("k_"):
            continue
        acc[k][r["Counter_Name"]] += float(r["Counter_Value"])
        launches[k][r["Counter_Name"]] += 1
n_frames = 48.0  # tools/pmc_batch.py: three 16-frame batches
out = {"source": "rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM SQ_WAVE_CYCLES SQ_BUSY_CYCLES "
                 "SQ_WAIT_ANY --kernel-trace -- python3 tools/pmc_batch.py",
       "unit": "per frame (48 frames of the bench workload in three batches)", "kernels": {}}
for k, cs in acc.items():
    d = {c: v / n_frames for c, v in sorted(cs.items())}
    if d.get("SQ_WAVE_CYCLES"):
        d["wait_any_over_wave_cycles"] = d.get("SQ_WAIT_ANY", 0.0) / d["SQ_WAVE_CYCLES"]
    d["launches"] = max(launches[k].values())
    out["kernels"][k] = d
json.dump(out, open(sys.argv[-1], "w"), indent=1)
for k, d in out["kernels"].items():
    print(k, {c: (round(v, 3) if isinstance(v, float) else v) for c, v in d.items()})
