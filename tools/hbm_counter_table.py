"""Calibration table of the HBM-side PMC counters from two rocprofv3 passes over tools/probes/hbm_counter_probe (every kernel moves a
known byte count): counter / true bytes per kernel, i.e. the factor bench.py's roofline.traffic has to apply per access width.
Usage: hbm_counter_table.py <FETCH_SIZE counter_collection.csv> <WRITE_SIZE counter_collection.csv> [out.json]   (counter unit: KiB)"""
import csv
import json
import sys
from collections import defaultdict

GIB = float(1 << 30)
TRUE = {  # kernel-name fragment -> (bytes loaded, bytes stored)
    "k_store<HIP_vector_type<unsigned int, 4": (0, GIB), "k_store<HIP_vector_type<unsigned int, 2": (0, GIB), "k_store<unsigned int>": (0, GIB),
    "k_store<unsigned short>": (0, GIB), "k_store<unsigned char>": (0, GIB / 4), "k_store_sparse": (0, GIB / 16), "k_atomic_sparse": (0, GIB / 16),
    "k_load<HIP_vector_type<unsigned int, 4": (GIB, 0), "k_load<HIP_vector_type<unsigned int, 2": (GIB, 0), "k_load<unsigned int>": (GIB, 0),
    "k_load<unsigned short>": (GIB, 0), "k_load_sparse": (GIB / 16, 0),
}


def load(path, name):
    acc = defaultdict(lambda: [0.0, 0])
    for r in csv.DictReader(open(path)):
        if r["Counter_Name"] == name:
            acc[r["Kernel_Name"]][0] += float(r["Counter_Value"]) * 1024.0
            acc[r["Kernel_Name"]][1] += 1
    return acc


f, w = load(sys.argv[1], "FETCH_SIZE"), load(sys.argv[2], "WRITE_SIZE")
out = {}
for frag, (lb, sb) in TRUE.items():
    fk = [k for k in f if frag in k]
    wk = [k for k in w if frag in k]
    fetch = sum(f[k][0] / f[k][1] for k in fk) if fk else 0.0
    write = sum(w[k][0] / w[k][1] for k in wk) if wk else 0.0
    out[frag.replace("HIP_vector_type<unsigned int, ", "uint").replace("k_", "")] = {
        "true_load_bytes": lb, "true_store_bytes": sb, "FETCH_SIZE_bytes": fetch, "WRITE_SIZE_bytes": write,
        "fetch_over_true": round(fetch / lb, 3) if lb else None, "write_over_true": round(write / sb, 3) if sb else None,
        "fetch_per_store_byte": round(fetch / sb, 3) if sb else None}
print(json.dumps(out, indent=1))
if len(sys.argv) > 3:
    json.dump({"source": "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (separate passes) over tools/probes/hbm_counter_probe; counter KiB x 1024 per launch",
               "kernels": out}, open(sys.argv[3], "w"), indent=1)
