"""profiles/<round>_pmc_traffic.json from two rocprofv3 --pmc passes over tools/pmc_workload.py (one launch = one frame), or —
with a fourth argument frames=N — profiles/<round>_pmc_traffic_batch[_cfg3].json from passes over tools/pmc_batch64.py (the
batched submissions bench.py times: a Stage A launch covers a whole batch, k_apply_tiles applies it; bytes per frame =
counter sum over all launches / N).
Usage: pmc_traffic_json.py <FETCH_SIZE counter_collection.csv> <WRITE_SIZE counter_collection.csv> <out.json> [frames=N]
Counter unit = KiB.  gfx950 correction (MI355X_MICROARCH.md, HBM section): FETCH_SIZE reports half the bytes of wide
coalesced reads -> read side = 2 x FETCH_SIZE; WRITE_SIZE is taken raw (device-scope atomics are booked as writes)."""
import csv
import json
import sys
from collections import defaultdict


def load(path, name):
    acc = defaultdict(lambda: [0.0, 0])
    rows = list(csv.DictReader(open(path)))
    # a marker kernel (tools/pmc_batch64.py warm=W): only the dispatches after it count
    marks = [int(r["Dispatch_Id"]) for r in rows if "k_probe_seeds" in r["Kernel_Name"]]
    first = max(marks) if marks else -1
    for r in rows:
        if r["Counter_Name"] == name and int(r["Dispatch_Id"]) > first:
            k = r["Kernel_Name"].split("(")[0].replace("void ", "").split("<")[0]
            acc[k][0] += float(r["Counter_Value"])
            acc[k][1] += 1
    return acc


f = load(sys.argv[1], "FETCH_SIZE")
w = load(sys.argv[2], "WRITE_SIZE")
batch = [a for a in sys.argv[4:] if a.startswith("frames=")]
n_frames = int(batch[0].split("=")[1]) if batch else max(v[1] for k, v in f.items() if k in ("k_bin_points", "k_bin_sectors"))
out = {"source": ("rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes) over tools/pmc_batch64.py: the asynchronous "
                  "batched device submissions bench.py times (%d frames)" % n_frames) if batch else
                 "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes) over tools/pmc_workload.py "
                 "(one launch = one VGA frame of the bench workload)",
       "unit": "bytes per frame" + ("" if batch else " (Stage A kernels are launched once per batch in bench.py: multiply by frames_per_launch)"),
       "correction": "read side = 2 x FETCH_SIZE x 1024 (gfx950: FETCH_SIZE reports half of wide coalesced reads, "
                     "MI355X_MICROARCH.md HBM section); write side = WRITE_SIZE x 1024 (uncalibrated; device-scope "
                     "atomics are counted as writes)",
       "kernels": {}}
tot = 0.0
for k in f:
    if not k.startswith("k_"):
        continue
    fb = 2 * f[k][0] * 1024 / n_frames
    wb = (w[k][0] * 1024 / n_frames) if k in w else 0.0
    out["kernels"][k] = {"fetch_bytes": fb, "write_bytes": wb, "total_bytes": fb + wb,
                         "launches_per_frame": f[k][1] / n_frames}
    tot += fb + wb
out["total_bytes_per_frame"] = tot
json.dump(out, open(sys.argv[3], "w"), indent=1)
print(json.dumps({k: round(v["total_bytes"] / 1e6, 2) for k, v in out["kernels"].items()}), round(tot / 1e6, 1), "MB/frame")
