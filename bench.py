#!/usr/bin/env python3
"""bench.py — depth frames/s integrated into the map on MI355X (BASELINE.json metric).

A *step* is `--batches-per-step` (27: a multiple of the three slot sets) batches of `--batch` (64) synthetic depth frames of ONE
stream pushed through the hot path (awareness raycast + log-odds block-map update) in order: 1 728 frames, ~17 ms, so that the
driver's 20 steps time a third of a second — one 64-frame batch (0.7 ms) is too small a unit to be robust to a single hiccup of
the host.  `value` = frames / total time of the K steps; `value_p50` = the same
from the median step.  Inputs (uint16 depth frames + poses) are resident in
HBM before the timed region starts.  Workload at N=1 = BASELINE config 2 (640x480 stream, 0.1 m local map,
S1 parameters); `--workload cfg3` selects config 3 (1280x720, 0.05 m).  At N=1 the same run also times a short
config-3 stream (`extra.cfg3`, with its own `roofline` and one-core `cpu_baseline`), the per-call latency of single frames
(`extra.single_frame_us`, the reference's own call pattern, with the CPU oracle's rate beside it) and compares the map of
the first >= 128 frames of the stream — submitted exactly like the timed loop: asynchronous 64-frame device batches on a
handle with three slot sets — with the CPU oracle's (`parity_check`).

Multi-GPU = BASELINE config 4: N independent config-2 streams, pose seeds 42 + rank, one rank per GPU, no data-path
collective (the path shards by stream: SURVEY.md §8e); the barrier and the max-over-ranks reduction of the elapsed time go
through torch.distributed (RCCL), and after the stream ONE global-map merge over RCCL is timed (`merge`).
`python bench.py --gpus N` with no rank environment starts the N ranks itself (a child `python -m torch.distributed.run
--nproc-per-node N bench.py --gpus N ...`, started before this process touches the GPU); under the driver's own
`torch.distributed.run` it is simply rank RANK of WORLD_SIZE.  "scaling": "weak".

Rank 0 prints ONE JSON line.  `roofline` is the HBM roofline of the kernel with the largest summed device time
(measured live with start/stop events of the kernel's own launches on the stream it runs on: pipelined inside the timed
region, and once more alone on the GPU after it); `roofline.atomics` is the bound that actually limits this path —
device-scope atomics, executed at the memory side — with the atomics counted by the kernels themselves.
`cpu_baseline` is the CPU oracle on the box's host cores (one thread; one process per physical core).
"""
from __future__ import annotations

import argparse
import gc
import json
import os
import re
import socket
import subprocess
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # MI355X HBM3E spec peak, /opt/skills/guides/MI355X_MICROARCH.md
XGMI_PEAK_GBS_PER_GPU = 7 * 153.0  # 7 point-to-point links per GPU, one direction
# Device-scope atomics are executed at the memory side whatever their scope; chip-wide rate measured with
# tools/probes/atomic_probe.hip on MI355X (distinct cache lines, all CUs): 34 G atomics/s
ATOMICS_PEAK_PER_S = 34e9


def make_inputs(cfg, n_distinct: int, n_total: int, seed: int):
    """n_distinct jittered room frames (cycled) + n_total random SE(3) poses (north_star: synthetic VGA depth +
    random SE(3) pose)."""
    from mlmapping_amd import synthetic as syn

    base = syn.room_depth(cfg)
    frames = np.stack([syn.jitter_depth(base, k, seed=seed) for k in range(n_distinct)])
    poses = syn.random_poses(n_total, seed=seed)
    q = np.stack([p[0] for p in poses])
    t = np.stack([p[1] for p in poses])
    return frames, q, t


# ---- CPU baseline ------------------------------------------------------------------------------------------------
def physical_core_cpus():
    """One logical CPU of every physical core this process may run on."""
    allowed = sorted(os.sched_getaffinity(0))
    seen, out = set(), []
    for c in allowed:
        try:
            sib = open(f"/sys/devices/system/cpu/cpu{c}/topology/thread_siblings_list").read().strip()
        except OSError:
            sib = str(c)
        if sib not in seen:
            seen.add(sib)
            out.append(c)
    return out


def cpu_worker(argv):
    """`bench.py --cpu-worker CPU WORKLOAD SEED T_START BUDGET`: one oracle map on one pinned core — a warm frame, then the
    stream for BUDGET seconds and at least 10 frames.  Prints `frames seconds`.  (No GPU, no torch.)"""
    cpu, workload, seed, t_start, budget = int(argv[0]), argv[1], int(argv[2]), float(argv[3]), float(argv[4])
    try:
        os.sched_setaffinity(0, {cpu})
    except OSError:
        pass
    from mlmapping_amd.config import S1, S3
    from oracle.binding import OracleMap

    cfg = S1 if workload == "cfg2" else S3
    frames, q, t = make_inputs(cfg, 8, 4096, seed)
    m = OracleMap(cfg)
    m.update_depth(frames[0], q[0], t[0])  # warm: tables paged in, containers grown, first blocks allocated
    while time.time() < t_start:
        time.sleep(0.001)
    t0 = time.perf_counter()
    n = 0
    while n < 10 or time.perf_counter() - t0 < budget:
        k = n + 1
        m.update_depth(frames[k % frames.shape[0]], q[k], t[k])
        n += 1
    print(n, time.perf_counter() - t0)


def cpu_baseline(cfg, workload, frames, q, t, budget_s: float, gpu_check=None, min_frames: int = 3, per_core: bool = True):
    """The CPU oracle (a port of the reference's map_awareness + map_local path: std::unordered_map/set, one thread per
    map — the reference is single-threaded per map) timed on the same stream, bounded to ~budget_s seconds per leg:
    (i) one stream on one core — its map is then compared with the GPU path's map of the same frames (`gpu_check`);
    (ii) one independent stream per PHYSICAL core, one process each (mirrors "one stream per GPU"), every
    process pinned, one warm frame, at least 10 timed frames."""
    from oracle.binding import OracleMap

    m = OracleMap(cfg)
    t0 = time.perf_counter()
    n1 = 0
    while n1 < q.shape[0] and (time.perf_counter() - t0 < budget_s or n1 < min_frames):
        m.update_depth(frames[n1 % frames.shape[0]], q[n1], t[n1])
        n1 += 1
    dt1 = time.perf_counter() - t0
    parity = gpu_check(m, n1) if gpu_check is not None else None
    m.close()
    if not per_core:
        return {"one_core": n1 / dt1, "unit": "frames/s", "cores": 1, "kind": "port",
                "sample": f"first {n1} frames of the same stream on 1 thread in {dt1:.1f} s; oracle/libmlmap_oracle.so"}, parity
    cpus = physical_core_cpus()
    t_start = time.time() + 2.0 + 0.02 * len(cpus)
    procs = [subprocess.Popen([sys.executable, os.path.abspath(__file__), "--cpu-worker", str(c), workload, str(1000 + i), repr(t_start),
                               repr(budget_s)], stdout=subprocess.PIPE, text=True, cwd=ROOT) for i, c in enumerate(cpus)]
    res = []
    for p in procs:
        out, _ = p.communicate()
        if p.returncode == 0 and out.strip():
            a, b = out.split()
            res.append((int(a), float(b)))
    nn = sum(a for a, _ in res)
    dtn = max(b for _, b in res) if res else float("nan")
    out = {"value": nn / dtn if res else None, "unit": "frames/s", "cores": len(res), "kind": "port",
           "one_core": n1 / dt1,
           "sample": f"(i) first {n1} frames of the same stream on 1 thread in {dt1:.1f} s; (ii) {len(res)} processes, one per physical core "
                     f"(pinned, own stream and map, one warm frame): {nn} frames, slowest process {dtn:.1f} s; oracle/libmlmap_oracle.so"}
    return out, parity


# ---- helpers of the GPU legs --------------------------------------------------------------------------------------
def measure_stream(m, cfg, d_frames, q, t, B, BPS, K, W, distinct, barrier, kernel_timing=True, n_settle=24, n_instr=6):
    """The measurement protocol of one stream on handle `m` (asynchronous mode): W untimed steps, a few fully instrumented
    batches to find the kernel with the largest summed device time, `n_settle` untimed batches back in the pipelined regime,
    then K timed steps of BPS batches of B frames between barriers (only the dominant kernel's launches are bracketed there),
    then two synchronous batches with every launch bracketed (each kernel alone on the GPU).  Returns a dict."""
    fsz = cfg.width * cfg.height

    def run_batch(j):
        k0 = j * B
        f0 = k0 % distinct
        if f0 + B <= distinct:
            m.update_map_batch_dev(d_frames.data_ptr() + f0 * fsz * 2, B, cfg.width, cfg.height, q[k0:k0 + B], t[k0:k0 + B])
        else:
            for b in range(B):
                k = k0 + b
                m.update_map_dev(d_frames.data_ptr() + (k % distinct) * fsz * 2, cfg.width, cfg.height, q[k], t[k])

    def acc_times(into):
        for name, ms in m.kernel_times():
            a = into.setdefault(name, [0.0, 0])
            a[0] += ms
            a[1] += 1

    for j in range(W * BPS):
        run_batch(j)
    barrier()
    # which kernel dominates?  A few fully instrumented batches (start/stop events of every launch, on the streams the kernels
    # run on) BEFORE the timed region.  Bracketing every kernel costs throughput (the chain is latency bound), so in the timed
    # region only that kernel's launches are bracketed (timing mode 3), every 8th of them when it is launched per frame.
    ktime_c, ktime, iso = {}, {}, {}
    timed_kernel, timed_every = None, 1
    if kernel_timing:
        m.enable_kernel_timing(2)
        for j in range(n_instr):
            run_batch(j)
        m.sync()
        acc_times(ktime_c)
        timed_kernel = max(ktime_c.items(), key=lambda kv: kv[1][0])[0]
        timed_every = 8 if ktime_c[timed_kernel][1] >= n_instr * B else 1
        m.set_timed_kernel(timed_kernel, timed_every)
        m.enable_kernel_timing(3)
    # settle back into the pipelined regime (their launches are bracketed as well).  Two dozen batches: the HIP runtime grows
    # its pools of signals / kernel-argument buffers while the first few dozen asynchronous batches are in flight (several
    # ms each time, seen at the 3rd, 5th, 9th and ~18th submission of a process)
    for j in range(n_settle):
        run_batch(j % max(1, W * BPS))
    barrier()
    stats, step_end = [], []
    # (the interpreter's cyclic garbage collector stays out of the timed region: a generation-2 pass over the modules loaded
    # here takes ~40 ms — more than many a timed region — at a point that depends on the allocation count)
    gc.collect()
    gc.disable()
    t0 = time.perf_counter()
    for s in range(W, W + K):
        for b in range(BPS):
            run_batch(s * BPS + b)
        stats.append(m.frame_stats())
        step_end.append(time.perf_counter())
    barrier()
    dt = time.perf_counter() - t0
    gc.enable()
    # per-step durations: a step's submissions return once the batch three sets back is confirmed, so in steady state the host
    # clock after a step trails the GPU by a constant two batches; the last step also takes the final drain
    step_dt = np.diff(np.array([t0] + step_end[:-1] + [t0 + dt]))
    if kernel_timing:
        acc_times(ktime)
        # the same kernels alone on the GPU: synchronous batches (a batch's Stage A is complete before its apply launch starts,
        # nothing else is in flight), every launch bracketed
        m.set_async(False)
        m.enable_kernel_timing(2)
        for j in range(2):
            run_batch(j)
        m.sync()
        acc_times(iso)
        m.enable_kernel_timing(0)
        m.set_async(True)
    return {"dt": dt, "step_dt": step_dt, "stats": stats, "ktime_c": ktime_c, "ktime": ktime, "iso": iso, "timed_kernel": timed_kernel,
            "timed_every": timed_every, "n_settle": n_settle, "n_instr": n_instr, "run_batch": run_batch}


def roofline_of(res, cfg, B, BPS, K, fps_one_gpu, pmc, pmc_file, pmc_batch, pmc_batch_file):
    """The `roofline` object from a measure_stream result: HBM roofline of the kernel with the largest summed device time."""
    fsz = cfg.width * cfg.height
    stats, ktime, ktime_c, iso = res["stats"], res["ktime"], res["ktime_c"], res["iso"]
    timed_every, n_settle, n_c = res["timed_every"], res["n_settle"], res["n_instr"]
    mean_bytes = float(np.mean([2 * fsz + 10 * (st["n_hit_cells"] + st["n_miss_cells"]) for st in stats])) if stats else 0.0
    mean_atomics = float(np.mean([st["n_device_atomics"] for st in stats])) if stats else 0.0
    if not ktime:
        return None, mean_bytes
    dom = max(ktime.items(), key=lambda kv: kv[1][0])
    avg_ms = dom[1][0] / dom[1][1]              # average duration of one launch of the dominant kernel
    n_inst = (K * BPS + n_settle) * B           # frames whose launches of that kernel were candidates for bracketing
    frames_per_launch = n_inst / (dom[1][1] * timed_every)  # Stage A kernels: one launch per batch of B frames
    ach = mean_bytes * frames_per_launch / (avg_ms * 1e-3) / 1e9
    a_ach = mean_atomics * fps_one_gpu          # device-scope atomics per second of one GPU's stream
    iso_ms = iso[dom[0]][0] / iso[dom[0]][1] if dom[0] in iso else None
    # HBM-side bytes from the PMC passes committed under profiles/ (rocprofv3 cannot be driven from inside this process;
    # tools/prof_round.sh regenerates them): per launch of the dominant kernel and for the whole path, both from passes over
    # the SAME 64-frame batched submissions that are timed here (pmc_traffic_batch); the older single-frame passes beside them
    traffic, traffic_src, whole, ratio = None, None, None, None
    if pmc_batch and dom[0] in pmc_batch.get("kernels", {}):
        traffic = pmc_batch["kernels"][dom[0]]["total_bytes"] * frames_per_launch
        traffic_src = pmc_batch_file
    elif pmc and dom[0] in pmc.get("kernels", {}):
        traffic = pmc["kernels"][dom[0]]["total_bytes"] * frames_per_launch
        traffic_src = pmc_file
    if pmc_batch:
        whole = pmc_batch.get("total_bytes_per_frame")
        ratio = whole / mean_bytes if whole and mean_bytes else None
    roof = {"bound": "hbm", "kernel": dom[0], "achieved": ach, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": ach / HBM_PEAK_GBS,
            "frames_per_launch": frames_per_launch, "launches_bracketed": f"1 of {timed_every}",
            "traffic": traffic, "traffic_source": traffic_src,
            "traffic_whole_path": whole, "traffic_whole_path_unit": "HBM-side bytes per frame, all kernels of the batched path (PMC)",
            "traffic_over_algorithmic": ratio, "traffic_whole_path_source": pmc_batch_file,
            "traffic_calibration": "read side = 2 x FETCH_SIZE, write side = WRITE_SIZE: both checked against kernels of known byte counts "
                                   "(tools/probes/hbm_counter_probe.hip, profiles/r6_hbm_counter_table.json: coalesced loads of every width report half their "
                                   "bytes, coalesced stores of every width their bytes; a lone 4-byte access or atomic per line counts 32 bytes a side)",
            "avg_launch_us": avg_ms * 1e3, "avg_launch_us_pipelined": avg_ms * 1e3,
            "avg_launch_us_isolated": iso_ms * 1e3 if iso_ms else None,
            "frac_isolated": (mean_bytes * frames_per_launch / (iso_ms * 1e-3) / 1e9 / HBM_PEAK_GBS) if iso_ms else None,
            "algorithmic_bytes_per_frame": mean_bytes,
            "issue": issue_bound(fps_one_gpu, "" if cfg.width == 640 else "_cfg3"),
            "by_rocprof_rule": rocprof_dominant(mean_bytes, frames_per_launch, "" if cfg.width == 640 else "_cfg3"),
            "atomics": {"bound": "device_atomics", "achieved": a_ach, "peak": ATOMICS_PEAK_PER_S, "unit": "atomics/s",
                        "frac": a_ach / ATOMICS_PEAK_PER_S, "atomics_per_frame": mean_atomics,
                        "counted_by": "the Stage A kernels themselves (chunk descriptors, list reservations, per-voxel counts)",
                        "peak_source": "tools/probes/atomic_probe.hip, measured on MI355X"},
            "kernels_us_per_frame": {**{k + " (timed region)": v[0] * 1e3 * timed_every / max(1, n_inst) for k, v in ktime.items()},
                                     **{k + " (instrumented batches before the region)": v[0] * 1e3 / max(1, n_c * B)
                                        for k, v in ktime_c.items()},
                                     **{k + " (alone on the GPU)": v[0] * 1e3 / max(1, 2 * B) for k, v in iso.items()}}}
    used = [os.path.join(ROOT, "profiles", f) for f in (pmc_batch_file, roof["issue"] and roof["issue"]["counters_source"],
                                                          roof["by_rocprof_rule"] and roof["by_rocprof_rule"]["source"]) if f]
    roof["missing_profiles"] = [n for n, v in (("pmc_traffic_batch", pmc_batch), ("pmc_sq", roof["issue"]), ("kernel_stats", roof["by_rocprof_rule"])) if not v]
    roof["stale_profiles"] = bool(roof["missing_profiles"]) or any(profile_is_stale(f) for f in used)
    roof["stale_profiles_note"] = ("true: a summary under profiles/ is missing or was taken on other kernel sources than this tree's "
                                   "(profiles/<tag>_meta.json, tools/prof_round.sh) — traffic / issue / by_rocprof_rule then describe older code")
    return roof, mean_bytes


SIMDS = 256 * 4        # 256 CUs x 4 SIMDs (MI355X_MICROARCH.md)
ENGINE_CLOCK_HZ = 2.4e9  # peak engine clock; a wave's vector instruction takes an issue slot of one quad-cycle
# What one SIMD really issues, measured on this part with every wave slot filled (tools/probes/valu_issue_probe.hip, profiles/r6_valu_issue_probe.txt):
# full-rate operations (v_mul_f32 / v_fma_f32 / v_add_u32: 0.78 - 1.05) and the half-rate class (FP64, v_mul_lo_u32, v_mad_u64_u32,
# v_bfe_u32, v_lshl_or_b32, packed FP32: 0.57) in G wave-instructions per second per SIMD.  The path's kernels mix both classes, so the
# pipe's utilisation lies between instructions / full-rate peak and instructions / half-rate peak.
VALU_FULL_RATE_PER_SIMD = 1.05e9
VALU_HALF_RATE_PER_SIMD = 0.573e9


def csrc_sha16():
    """sha1 (16 hex digits) over the kernel and host sources the library is built from: what a profile set was taken on
    (tools/prof_round.sh writes it to profiles/<tag>_meta.json)."""
    import glob
    import hashlib

    hh = hashlib.sha1()
    for f in sorted(glob.glob(os.path.join(ROOT, "mlmapping_amd", "csrc", "*.h")) + glob.glob(os.path.join(ROOT, "mlmapping_amd", "csrc", "*.hip"))):
        hh.update(os.path.basename(f).encode())
        hh.update(open(f, "rb").read())
    return hh.hexdigest()[:16]


def profile_is_stale(path):
    """True if the profile file's round tag has no profiles/<tag>_meta.json or that names other sources than the tree's."""
    if not path:
        return True
    mm = re.match(r"(r\d+[a-z]*)_", os.path.basename(path))
    meta = os.path.join(ROOT, "profiles", (mm.group(1) if mm else "none") + "_meta.json")
    try:
        return json.load(open(meta)).get("csrc_sha16") != csrc_sha16()
    except Exception:
        return True


def issue_bound(fps_one_gpu: float, tag: str = ""):
    """The bound that actually limits this path: vector-instruction issue.  SQ_ACTIVE_INST_VALU counts, in quad-cycles
    and summed over the waves, the issue slots the kernels' vector instructions take; a SIMD has ONE vector pipe however many wave
    slots are filled.  frac = slots per frame (newest profiles/*_pmc_sq*.json, rocprofv3 --pmc passes over the batched submissions
    timed here) / slots the chip's 1 024 SIMDs offer in the time this run takes per frame."""
    f = newest_profile(f"pmc_sq{tag}.json")
    if not f:
        return None
    try:
        d = json.load(open(f))
        ker = d.get("kernels", {})
        valu_q = sum(k.get("SQ_ACTIVE_INST_VALU", 0.0) for k in ker.values())
        valu_n = sum(k.get("SQ_INSTS_VALU", 0.0) for k in ker.values())
        salu_n = sum(k.get("SQ_INSTS_SALU", 0.0) for k in ker.values())
    except Exception:
        return None
    if not valu_q or not fps_one_gpu:
        return None
    avail = SIMDS * ENGINE_CLOCK_HZ / 4.0 / fps_one_gpu
    full, half = SIMDS * VALU_FULL_RATE_PER_SIMD / fps_one_gpu, SIMDS * VALU_HALF_RATE_PER_SIMD / fps_one_gpu
    return {"bound": "valu_issue", "achieved": valu_n, "unit": "vector wave-instructions per frame",
            "peak": {"if_all_full_rate": full, "if_all_half_rate": half},
            "frac": {"lower": valu_n / full, "upper": valu_n / half},
            "frac_note": "the kernels mix full-rate (2 cycles per wave64) and half-rate (FP64, 32-bit multiply, bit-field: 4 cycles) operations: "
                         "the vector pipes' utilisation lies between the two",
            "valu_wave_instructions_per_frame": valu_n, "salu_wave_instructions_per_frame": salu_n,
            "active_inst_valu_quad_cycles_per_frame": valu_q, "quad_cycle_slots_per_frame_at_2.4GHz": avail,
            "peak_source": "tools/probes/valu_issue_probe.hip on MI355X (profiles/r6_valu_issue_probe.txt): 1.05 / 0.573 G wave-instructions/s per SIMD "
                           "with eight waves resident, x 1 024 SIMDs x the measured time per frame",
            "counters_source": os.path.basename(f), "counters_stale": profile_is_stale(f),
            "per_kernel_valu": {k: v.get("SQ_INSTS_VALU", 0.0) for k, v in ker.items()}}


def rocprof_dominant(mean_bytes: float, frames_per_launch: float, tag: str = ""):
    """The dominant kernel by the rule of the rocprof summary (largest TotalDurationNs of the newest profiles/*_kernel_stats*.csv,
    `rocprofv3 --kernel-trace --stats` of this command) with the same algorithmic bytes per launch priced against its average
    duration there — beside `roofline.kernel`, which the live event pairs of this run pick."""
    import csv

    f = newest_profile(f"kernel_stats{tag}.csv")
    if not f:
        return None
    try:
        rows = list(csv.DictReader(open(f)))
        top = max(rows, key=lambda r: float(r["TotalDurationNs"]))
        name = re.sub(r"^void ", "", top["Name"]).split("(")[0].split("<")[0]
        avg_us = float(top["AverageNs"]) / 1e3
        ach = mean_bytes * frames_per_launch / (avg_us * 1e-6) / 1e9
        return {"kernel": name, "avg_launch_us": avg_us, "achieved": ach, "frac": ach / HBM_PEAK_GBS, "share_of_gpu_time_pct": float(top["Percentage"]),
                "source": os.path.basename(f), "source_stale": profile_is_stale(f)}
    except Exception:
        return None


def newest_profile(pattern: str):
    """profiles/<round tag>_<pattern>: the file of the newest round tag (r3b > r3a > r2f; natural order)."""
    import glob

    def key(p):
        mm = re.match(r"r(\d+)([a-z]*)_", os.path.basename(p))
        return (int(mm.group(1)), mm.group(2)) if mm else (-1, "")

    files = sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_" + pattern)), key=key)
    return files[-1] if files else None


def launch_ranks(args) -> int:
    """`bench.py --gpus N` without a rank environment: start the N ranks as a child torch.distributed.run — this process
    has not touched the GPU (torch.cuda.device_count() does not initialise it) and only waits for the child."""
    import torch

    n_dev = torch.cuda.device_count()
    if n_dev < args.gpus and os.environ.get("MLM_BENCH_DIST_BACKEND", "nccl") == "nccl":
        print(f"bench.py --gpus {args.gpus}: this node shows {n_dev} GPU(s); one rank per GPU is the measured configuration "
              "(MLM_BENCH_DIST_BACKEND=gloo lets ranks share devices to exercise the launcher contract only)", file=sys.stderr)
        return 2
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(args.gpus), "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    return subprocess.call(cmd, env=env, cwd=ROOT)


def single_frame_latency(MLMap, cfg, frames, q, t, d_frames, n_calls=120, cpu=True):
    """Per-call latency of the reference's own call pattern (ONE frame per depth_odom_input_callback, mlmap.cpp:463-507) in
    synchronous mode: median microseconds of mlm_integrate_depth_u16 (host buffer), mlm_integrate_depth_u16_dev (HBM
    resident) and mlm_integrate_callback with the default 500-pixel sampler — and, beside the latter, the CPU oracle's
    callback on one core (`cpu_baseline_sampled500`, a bounded sample of the same calls)."""
    m = MLMap(cfg, max_blocks=32768, max_points=cfg.width * cfg.height, max_batch=2)
    fsz = cfg.width * cfg.height
    out = {}

    def med(fn):
        for k in range(8):
            fn(k)
        m.sync()
        ts = []
        for k in range(8, 8 + n_calls):
            a = time.perf_counter()
            fn(k)
            ts.append(time.perf_counter() - a)
        return float(np.median(ts) * 1e6)

    gc.collect()
    gc.disable()
    out["dense_host_buffer"] = med(lambda k: m.update_map(frames[k % frames.shape[0]], q[k], t[k]))
    out["dense_device_buffer"] = med(lambda k: m.update_map_dev(d_frames.data_ptr() + (k % frames.shape[0]) * fsz * 2, cfg.width, cfg.height, q[k], t[k]))
    zero3 = np.zeros(3)
    out["callback_sampled500"] = med(lambda k: m.depth_odom_callback(frames[k % frames.shape[0]], 0.0, t[k], q[k], zero3, 0.0, zero3, 0.0, 0.0, sampled=True))
    gc.enable()
    m.close()
    if cpu:
        from oracle.binding import OracleMap

        o = OracleMap(cfg)
        for k in range(8):
            o.depth_odom_callback(frames[k % frames.shape[0]], 0.0, t[k], q[k], zero3, 0.0, zero3, 0.0, 0.0, sampled=True)
        ts = []
        t_end = time.perf_counter() + 3.0
        k = 8
        while k < q.shape[0] and (time.perf_counter() < t_end or len(ts) < 50):
            a = time.perf_counter()
            o.depth_odom_callback(frames[k % frames.shape[0]], 0.0, t[k], q[k], zero3, 0.0, zero3, 0.0, 0.0, sampled=True)
            ts.append(time.perf_counter() - a)
            k += 1
        o.close()
        out["cpu_baseline_sampled500"] = {"value": float(np.median(ts) * 1e6), "unit": "us per call (median)", "cores": 1, "kind": "port",
                                          "sample": f"{len(ts)} sampled-500 callbacks of the same stream on 1 thread; oracle/libmlmap_oracle.so"}
    # ... and the reference's shipped real-data configuration VERBATIM (launch/config/config2.yaml: 0.2 m map, frontier mode +
    # inflation, 424x240 32FC1 depth stream, 500 rand() samples per callback, inflate_map every 3rd frame like the 10 Hz timer)
    from mlmapping_amd import synthetic as syn
    from mlmapping_amd.config import CONFIG2_YAML as C2

    m = MLMap(C2, max_blocks=16384, max_points=C2.width * C2.height, max_batch=2)
    depth = [(syn.jitter_depth(syn.room_depth(C2), k).astype(np.float32) / 1000.0) for k in range(8)]
    traj = syn.smooth_trajectory(8 + n_calls, 7)

    def cb(h, k):
        qq, tt = traj[k]
        T = h.depth_odom_callback(depth[k % 8], 0.0, tt, qq, zero3, 0.0, zero3, 0.0, C2.camera2odom_latency, sampled=True)
        if k % 3 == 2:
            h.inflate_map(T[4:])

    gc.collect()
    gc.disable()
    for k in range(8):
        cb(m, k)
    ts = []
    for k in range(8, 8 + n_calls):
        a = time.perf_counter()
        cb(m, k)
        ts.append(time.perf_counter() - a)
    gc.enable()
    m.close()
    row = {"value": float(np.median(ts) * 1e6), "unit": "us per callback (median; every 3rd also runs inflate_map)",
           "workload": "config2.yaml verbatim: frontier mode + inflation, 424x240 32FC1, 500 samples"}
    try:  # the same calls from a C++ process (tools/query_latency.cpp --callback)
        from tools.query_latency import measure_callback

        row["cpp_client"] = measure_callback(C2, depth, [traj[k] for k in range(len(traj))], latency=C2.camera2odom_latency, calls=n_calls - 12, inflate_every=3)
    except Exception as e:  # (the client is a convenience row: its absence does not fail the bench line)
        row["cpp_client"] = {"error": str(e)[:200]}
    if cpu:
        from oracle.binding import OracleMap

        o = OracleMap(C2)
        for k in range(8):
            cb(o, k)
        tc = []
        for k in range(8, 8 + min(n_calls, 60)):
            a = time.perf_counter()
            cb(o, k)
            tc.append(time.perf_counter() - a)
        o.close()
        row["cpu_baseline"] = {"value": float(np.median(tc) * 1e6), "unit": "us per callback (median)", "cores": 1, "kind": "port",
                               "sample": f"{len(tc)} callbacks of the same stream on 1 thread; oracle/libmlmap_oracle.so"}
    out["callback_config2_yaml"] = row
    return out


def other_scenes(MLMap, cfg, frames, q, t, d_frames, device):
    """Rows that are NOT the headline, measured by the same run so that they are on the driver's record: frontier mode
    (use_exploration_frontiers, the shipped config2.yaml's mode) on the bench stream — asynchronous 32-frame batches from HBM as the
    headline is measured, and one frame per synchronous call — and the worst-case "scatter" scene (every pixel in a cell of its own)
    in the default mode.  A few hundred frames each; a row that fails reports its error instead of a number."""
    import torch

    from mlmapping_amd import synthetic as syn

    out = {}
    B = 32
    fsz = cfg.width * cfg.height
    try:
        ex = cfg.with_(use_exploration_frontiers=True)
        m = MLMap(ex, device=device, max_blocks=32768, max_points=fsz, max_batch=B)
        m.set_async(True)
        for j in range(4):  # (the pool and the emulated containers grow to their working size)
            m.update_map_batch_dev(d_frames.data_ptr(), B, cfg.width, cfg.height, q[j * B:(j + 1) * B], t[j * B:(j + 1) * B])
        m.sync()
        t0 = time.perf_counter()
        for j in range(4, 16):
            m.update_map_batch_dev(d_frames.data_ptr(), B, cfg.width, cfg.height, q[j * B:(j + 1) * B], t[j * B:(j + 1) * B])
        m.sync()
        out["frontier_mode_async_batches"] = {"value": 12 * B / (time.perf_counter() - t0), "unit": "frames/s", "frames": 12 * B, "batch": B}
        m.set_async(False)
        ts = []
        for k in range(16 * B, 16 * B + 72):
            a = time.perf_counter()
            m.update_map_dev(d_frames.data_ptr() + (k % B) * fsz * 2, cfg.width, cfg.height, q[k], t[k])
            ts.append(time.perf_counter() - a)
        out["frontier_mode_frame_by_frame"] = {"value": 1.0 / float(np.mean(ts[8:])), "unit": "frames/s", "us_per_frame_median": float(np.median(ts[8:]) * 1e6),
                                               "frames": len(ts) - 8}
        m.close()
    except Exception as e:  # noqa: BLE001 (the headline must not depend on these rows)
        out["frontier_mode_error"] = str(e)[:300]
    try:
        sc = list(syn.stream(cfg, "scatter", "smooth", B))
        d = torch.from_numpy(np.stack([f for f, _ in sc]).view(np.int16)).cuda(device)
        qs, ts_ = np.stack([p[0] for _, p in sc]), np.stack([p[1] for _, p in sc])
        torch.cuda.synchronize()
        m = MLMap(cfg, device=device, max_blocks=32768, max_points=fsz, max_batch=B)
        m.set_async(True)
        for _ in range(6):  # (slots grow, the large-table pass arms, the column table widens)
            m.update_map_batch_dev(d.data_ptr(), B, cfg.width, cfg.height, qs, ts_)
        m.sync()
        t0 = time.perf_counter()
        for _ in range(10):
            m.update_map_batch_dev(d.data_ptr(), B, cfg.width, cfg.height, qs, ts_)
        m.sync()
        st = m.frame_stats()
        out["scatter_scene_async_batches"] = {"value": 10 * B / (time.perf_counter() - t0), "unit": "frames/s", "frames": 10 * B, "batch": B,
                                              "hit_cells_per_frame": st["n_hit_cells"], "sector_fallbacks": st["n_sector_fallbacks"]}
        m.close()
        del d
    except Exception as e:  # noqa: BLE001
        out["scatter_scene_error"] = str(e)[:300]
    return out


def main():
    if len(sys.argv) > 1 and sys.argv[1] == "--cpu-worker":
        return cpu_worker(sys.argv[2:])
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--batch", type=int, default=64, help="frames per batched submission (one batched Stage A launch sequence)")
    ap.add_argument("--batches-per-step", type=int, default=27, help="batched submissions per step (a step = this x --batch frames)")
    ap.add_argument("--workload", default="cfg2", choices=["cfg2", "cfg3"])
    ap.add_argument("--distinct", type=int, default=64, help="distinct depth frames kept in HBM (cycled; rounded up to a multiple of --batch)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-kernel-timing", action="store_true", help="do not record HIP events around the kernels")
    ap.add_argument("--no-extra", action="store_true", help="skip the short config-3 measurement and the latency rows")
    ap.add_argument("--cpu-budget", type=float, default=10.0)
    args = ap.parse_args()

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(launch_ranks(args))  # (nothing above has touched the GPU)

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        raise SystemExit(f"bench.py: --gpus {args.gpus} but WORLD_SIZE={world}")

    import torch

    from mlmapping_amd.config import S1, S3
    from mlmapping_amd.mlmap import MLMap

    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: there is no CPU fallback")
    # MLM_BENCH_DIST_BACKEND=gloo (test hook, tests/test_gpu_boundary.py): the launcher contract — rank environment,
    # barrier, max over ranks, merge leg, one JSON line from rank 0 — exercised on a box with fewer GPUs than ranks (ranks
    # share devices, the collectives run on the CPU).  The measured configuration is always one rank per GPU over RCCL.
    backend = os.environ.get("MLM_BENCH_DIST_BACKEND", "nccl")
    if backend != "nccl":
        local_rank = local_rank % max(1, torch.cuda.device_count())
    elif world > torch.cuda.device_count():
        raise SystemExit(f"bench.py: {world} ranks but {torch.cuda.device_count()} GPU(s) (one rank per GPU over RCCL)")
    torch.cuda.set_device(local_rank)
    dist = None
    if world > 1:
        import torch.distributed as dist_mod

        dist = dist_mod
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group(backend)
        world = dist.get_world_size()  # n_gpus = the ranks the communicator reports

    cfg = S1 if args.workload == "cfg2" else S3
    if os.environ.get("MLM_BENCH_NO_RAYCAST"):  # diagnostic only (not the BASELINE workload): hits without rays
        import dataclasses
        cfg = dataclasses.replace(cfg, use_raycasting=False)
    B, BPS, K, W = args.batch, max(1, args.batches_per_step), args.steps, args.warmup
    args.distinct = max(B, (args.distinct + B - 1) // B * B)  # whole batches: a batch is always ONE batched submission
    n_total = max((K + W) * BPS * B, 8 * B, 256)
    frames, q, t = make_inputs(cfg, args.distinct, n_total, seed=42 + rank)  # config 4: pose seeds 42 .. 42 + N - 1
    # inputs resident in HBM: torch owns the buffer (uint16 payload viewed as int16 storage)
    d_frames = torch.from_numpy(frames.view(np.int16)).cuda(local_rank)
    torch.cuda.synchronize()

    m = MLMap(cfg, device=local_rank, max_blocks=32768, max_points=cfg.width * cfg.height, max_batch=B)
    m.set_async(True)  # batches are submitted back to back; barrier() below waits for the map to be complete

    def barrier():
        m.sync()
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()

    res = measure_stream(m, cfg, d_frames, q, t, B, BPS, K, W, args.distinct, barrier, kernel_timing=not args.no_kernel_timing)
    dt, stats = res["dt"], res["stats"]
    if os.environ.get("MLM_BENCH_STEP_TIMES"):  # diagnostic: host time of every step of the timed region
        print("step ms:", [round(float(x) * 1e3, 2) for x in res["step_dt"]], file=sys.stderr)
    coll_dev = f"cuda:{local_rank}" if backend == "nccl" else "cpu"
    if dist is not None:
        tt = torch.tensor([dt], dtype=torch.float64, device=coll_dev)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        dt = float(tt.item())
    fsz = cfg.width * cfg.height

    # ---- config 4's exchange step: ONE global-map merge over the communicator after the streams (no reference counterpart)
    merge = None
    if dist is not None:
        from mlmapping_amd.merge import merge_device_maps

        barrier()
        merge_device_maps(m, load_back=False)  # (untimed: first use of the all-to-all / all-gather paths of the communicator)
        barrier()
        tm = time.perf_counter()
        merged = merge_device_maps(m, load_back=False)
        torch.cuda.synchronize()
        dist.barrier()
        merge_s = time.perf_counter() - tm
        tt = torch.tensor([merge_s], dtype=torch.float64, device=coll_dev)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        merge_s = float(tt.item())
        n_u = int(merged["keys"].shape[0])
        n_pad = (n_u + world - 1) // world * world
        # per rank, one direction: keys all-gather + reduce-scatter (all-to-all) and all-gather of 5 B per voxel each
        fabric = (world - 1) / world * (2 * 5 * n_pad * m.cells) + (world - 1) * 8 * n_u
        merge = {"merge_ms": merge_s * 1e3, "union_blocks": n_u, "own_blocks": m.block_count(), "merge_bytes_per_rank": fabric,
                 "achieved_xgmi_gbs_per_gpu": fabric / merge_s / 1e9, "xgmi_peak_gbs_per_gpu": XGMI_PEAK_GBS_PER_GPU,
                 "backend": "rccl" if backend == "nccl" else backend + " (host staged: launcher-contract test, not a measurement)",
                 "includes": "key union, mlm_merge_pack, all-to-all reduce-scatter, mlm_merge_finish, all-gather (no load-back)"}
        del merged

    # PCIe-inclusive rate: the same batches handed over as HOST buffers (uploads overlap with compute); reported, never
    # `value`
    n_host = 16
    host_batch = np.ascontiguousarray(frames[[b % args.distinct for b in range(B)]])
    m.host_register(host_batch)  # (a replay tool pins its frame buffer once: DMA straight from it, mlm_host_register)
    for s in range(4):  # the first host-buffer submission of every slot set allocates its image buffers
        m.update_map_batch(host_batch, q[s * B:s * B + B], t[s * B:s * B + B])
    m.sync()
    th = time.perf_counter()
    for s in range(n_host):
        k0 = (s * B) % (n_total - B + 1)
        m.update_map_batch(host_batch, q[k0:k0 + B], t[k0:k0 + B])
    m.sync()
    pcie_fps = n_host * B / (time.perf_counter() - th)
    last_stats = m.frame_stats()
    m.host_unregister(host_batch)
    m.close()

    if rank == 0:
        F_STEP = B * BPS
        fps = world * K * F_STEP / dt
        step_p50 = float(np.median(res["step_dt"]))
        pmc, pmc_file, pmc_b, pmc_b_file = None, None, None, None
        try:
            f = newest_profile("pmc_traffic.json")
            if f and args.workload == "cfg2":
                pmc_file, pmc = os.path.basename(f), json.load(open(f))
            f = newest_profile("pmc_traffic_batch.json" if args.workload == "cfg2" else "pmc_traffic_batch_cfg3.json")
            if f:
                pmc_b_file, pmc_b = os.path.basename(f), json.load(open(f))
        except Exception:
            pass
        roof, mean_bytes = roofline_of(res, cfg, B, BPS, K, fps / world, pmc, pmc_file, pmc_b, pmc_b_file)
        out = {
            "metric": "depth frames/s into local map", "value": fps, "unit": "frames/s", "n_gpus": world,
            "steps": K, "warmup": W, "ms_per_step": dt / K * 1e3, "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "f64 index / f32 log-odds", "data": "synthetic",
            "config": {"workload": ("BASELINE cfg2: 640x480 room+jitter stream, random SE(3) poses, S1 0.1 m map"
                                    if world == 1 else f"BASELINE cfg4: {world} independent cfg2 streams (pose seeds 42..{41 + world}), one per GPU, "
                                                       "then one RCCL global-map merge (timed separately: merge)")
                       if args.workload == "cfg2" else "BASELINE cfg3: 1280x720 room+jitter stream, S3 0.05 m map",
                       "frames_per_step": F_STEP, "batch": B, "batches_per_step": BPS, "streams": world,
                       "parallelism": f"{world} independent streams"},
            "value_p50": world * F_STEP / step_p50, "ms_per_step_p50": step_p50 * 1e3,
            "ms_per_step_min_max": [float(res["step_dt"].min() * 1e3), float(res["step_dt"].max() * 1e3)],
            "achieved_hbm_gbs_whole_path": fps * mean_bytes / 1e9,
            "pcie_inclusive_frames_per_s": pcie_fps * world,
            "path": {"sector_fallbacks": last_stats["n_sector_fallbacks"], "spec_replays": last_stats["n_spec_replays"],
                     "logit_bit_exact": last_stats["logit_bit_exact"]},
            "roofline": roof,
        }
        if merge is not None:
            out["merge"] = merge
        if world == 1 and args.workload == "cfg2" and not args.no_extra:
            # config 3 (1280x720, 0.05 m) in the same invocation: a short stream, same protocol (its own roofline; its CPU
            # baseline below)
            # (96 frame slots of 2 GB each: 190 GB of the 288 GB; a step = as many batches as there are slot sets, so that the host
            # blocks the same way in every step and the per-step times behind value_p50 mean something)
            B3, BPS3, K3, W3, D3 = 32, 3, 10, 2, 32
            f3, q3, t3 = make_inputs(S3, D3, (K3 + W3) * B3 * BPS3, seed=42)
            d3 = torch.from_numpy(f3.view(np.int16)).cuda(local_rank)
            m3 = MLMap(S3, device=local_rank, max_blocks=65536, max_points=S3.width * S3.height, max_batch=B3)
            m3.set_async(True)

            def sync3():
                m3.sync()
                torch.cuda.synchronize()

            r3 = measure_stream(m3, S3, d3, q3, t3, B3, BPS3, K3, W3, D3, sync3, kernel_timing=not args.no_kernel_timing, n_settle=10, n_instr=2)
            fps3 = K3 * B3 * BPS3 / r3["dt"]
            pmc3, pmc3_file = None, None
            try:
                f = newest_profile("pmc_traffic_batch_cfg3.json")
                if f:
                    pmc3_file, pmc3 = os.path.basename(f), json.load(open(f))
            except Exception:
                pass
            roof3, b3 = roofline_of(r3, S3, B3, BPS3, K3, fps3, None, None, pmc3, pmc3_file)
            out["extra"] = {"cfg3": {"workload": "BASELINE cfg3: 1280x720 room+jitter stream, S3 0.05 m map", "value": fps3,
                                     "unit": "frames/s", "steps": K3, "frames_per_step": B3 * BPS3, "batch": B3,
                                     "value_p50": B3 * BPS3 / float(np.median(r3["step_dt"])),
                                     "achieved_hbm_gbs_whole_path": fps3 * b3 / 1e9,
                                     "sector_fallbacks": r3["stats"][-1]["n_sector_fallbacks"], "roofline": roof3}}
            m3.close()
            del d3
            out["extra"]["single_frame_us"] = single_frame_latency(MLMap, cfg, frames, q, t, d_frames, cpu=not args.no_cpu_baseline)
            # the drop-in query interface the way a planner calls it: ONE position per call from a C++ client of the facade
            # (tools/query_latency.cpp, a child process), the CPU oracle's per-position cost beside it
            try:
                from tools.query_latency import measure as query_latency

                out["extra"]["single_query_us"] = query_latency(cfg, with_oracle=not args.no_cpu_baseline)
                cb = out["extra"]["single_query_us"].pop("callback_sampled500_cpp", None)
                if cb:  # (the same 500-sample callback as single_frame_us.callback_sampled500, called from a C++ process instead of through ctypes)
                    out["extra"]["single_frame_us"]["callback_sampled500_cpp_client"] = cb
            except Exception as e:  # noqa: BLE001 (the headline must not depend on this row)
                out["extra"]["single_query_us"] = {"error": str(e)[:300]}
            out["extra"]["other_scenes"] = other_scenes(MLMap, cfg, frames, q, t, d_frames, local_rank)
            if not args.no_cpu_baseline:
                out["extra"]["cfg3"]["cpu_baseline"], _ = cpu_baseline(S3, "cfg3", f3, q3, t3, args.cpu_budget, None, min_frames=3, per_core=False)
        if not args.no_cpu_baseline and world == 1:  # rank 0 at N=1 only
            def gpu_check(oracle_map, n):
                """A fresh GPU handle fed the frames leg (i) integrated — submitted EXACTLY like the timed loop: asynchronous
                B-frame contiguous device batches (mlm_integrate_depth_batch_dev) on a handle with max_batch = B and three slot
                sets — compared with the oracle's map (outside every timed region): block keys, occupancy classes and the
                float bits of every voxel's log-odds."""
                from tests.util import compare_maps

                g = MLMap(cfg, device=local_rank, max_blocks=32768, max_points=cfg.width * cfg.height, max_batch=B)
                g.set_async(True)
                nb = n // B
                for j in range(nb):
                    k0 = j * B
                    f0 = k0 % args.distinct
                    g.update_map_batch_dev(d_frames.data_ptr() + f0 * fsz * 2, B, cfg.width, cfg.height, q[k0:k0 + B], t[k0:k0 + B])
                for k in range(nb * B, n):  # (a remainder shorter than a batch: frame by frame)
                    g.update_map_dev(d_frames.data_ptr() + (k % args.distinct) * fsz * 2, cfg.width, cfg.height, q[k], t[k])
                gb, cb = g.export_blocks(), oracle_map.export_blocks()
                gs = g.frame_stats()
                g.close()
                info = {"frames": n, "batch": B, "batches": nb, "submission": "async mlm_integrate_depth_batch_dev, max_batch = batch, 3 slot sets (as timed)",
                        "spec_replays": gs["n_spec_replays"], "sector_fallbacks": gs["n_sector_fallbacks"], "pool_grows": gs["n_pool_grows"]}
                try:
                    d = compare_maps(gb, cb, "bench parity check", exact=False)
                    return {**info, "blocks": d["blocks"], "voxels": d["cells"], "classes_equal": True, "max_dodd": d["max_dodd"],
                            "bit_mismatch": d["bit_mismatch"]}
                except AssertionError as e:
                    return {**info, "classes_equal": False, "error": str(e)[:300]}

            # (leg (i) runs at least two whole batches of the stream — 128 frames, ~20 s of one core — so that the parity check
            # covers the 64-frame k_apply_tiles chain at its real size)
            out["cpu_baseline"], out["parity_check"] = cpu_baseline(cfg, args.workload, frames, q, t, args.cpu_budget, gpu_check,
                                                                    min_frames=2 * B if B <= 64 else B)
        print(json.dumps(out))
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
