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
seen_in, dup = {}, set()
args = [a for a in sys.argv[1:] if not a.startswith("frames=")]
frames_arg = [a for a in sys.argv[1:] if a.startswith("frames=")]
for path in args[:-1]:
    for c, p0 in list(seen_in.items()):
        if p0 != path:
            dup.add((path, c))
    for r in csv.DictReader(open(path)):
        k = r["Kernel_Name"].split("(")[0].replace("void ", "").split("<")[0]
        if not k.startswith("k_"):
            continue
        if (path, r["Counter_Name"]) in dup:
            continue  # (a counter collected by two passes: the first pass counts)
        acc[k][r["Counter_Name"]] += float(r["Counter_Value"])
        launches[k][r["Counter_Name"]] += 1
        seen_in[r["Counter_Name"]] = path
n_frames = float(frames_arg[0].split("=")[1]) if frames_arg else 48.0  # tools/pmc_batch.py: three 16-frame batches; frames=N: tools/pmc_batch64.py
out = {"source": "two rocprofv3 --pmc passes over python3 tools/pmc_batch.py or tools/pmc_batch64.py (--kernel-trace only): SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS "
                 "SQ_INSTS_VMEM SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY; SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY "
                 "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_BUSY_CU_CYCLES.  WAIT_ANY (parked on s_waitcnt / barrier) + WAIT_INST_ANY (issue stall) + "
                 "ACTIVE_INST_ANY (issuing) ~ WAVE_CYCLES (MI355X_MICROARCH.md, rocprofv3 PMC slots)",
       "unit": "per frame (%d frames of the bench workload%s)" % (n_frames, ": asynchronous 64-frame device batches as bench.py submits them" if frames_arg else " in three batches"), "kernels": {}}
for k, cs in acc.items():
    d = {c: v / n_frames for c, v in sorted(cs.items())}
    if d.get("SQ_WAVE_CYCLES"):
        d["wait_any_over_wave_cycles"] = d.get("SQ_WAIT_ANY", 0.0) / d["SQ_WAVE_CYCLES"]
        for c, name in (("SQ_WAIT_INST_ANY", "wait_inst_any_over_wave_cycles"), ("SQ_ACTIVE_INST_ANY", "active_inst_any_over_wave_cycles"),
                        ("SQ_ACTIVE_INST_VALU", "active_inst_valu_over_wave_cycles"), ("SQ_ACTIVE_INST_LDS", "active_inst_lds_over_wave_cycles")):
            if c in d:
                d[name] = d[c] / d["SQ_WAVE_CYCLES"]
    d["launches"] = max(launches[k].values())
    out["kernels"][k] = d
json.dump(out, open(args[-1], "w"), indent=1)
for k, d in out["kernels"].items():
    print(k, {c: (round(v, 3) if isinstance(v, float) else v) for c, v in d.items()})
