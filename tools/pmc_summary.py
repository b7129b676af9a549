"""Per-kernel HBM traffic from two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE; counter unit = KiB).
gfx950 correction (MI355X_MICROARCH.md §HBM): FETCH_SIZE reports half the bytes of wide coalesced reads, so the read
side is given both raw and doubled; WRITE_SIZE is uncalibrated and given raw."""
import csv
import sys
from collections import defaultdict


def load(path, name):
    acc = defaultdict(lambda: [0.0, 0])
    for r in csv.DictReader(open(path)):
        if r["Counter_Name"] == name:
            k = r["Kernel_Name"].split("(")[0].replace("void ", "")
            acc[k][0] += float(r["Counter_Value"])
            acc[k][1] += 1
    return acc


f = load(sys.argv[1], "FETCH_SIZE")
w = load(sys.argv[2], "WRITE_SIZE")
print(f"{'kernel':40s} {'launches':>8s} {'fetch KiB/launch':>18s} {'x2 (gfx950)':>12s} {'write KiB/launch':>18s}")
for k in sorted(f, key=lambda k: -f[k][0]):
    if not k.startswith("k_"):
        continue
    fl = f[k][0] / f[k][1]
    wl = w[k][0] / w[k][1] if k in w and w[k][1] else float("nan")
    print(f"{k:40s} {f[k][1]:8d} {fl:18.1f} {2*fl:12.1f} {wl:18.1f}")
