#!/usr/bin/env python3
"""bench.py — depth frames/s integrated into the map on MI355X (BASELINE.json metric).

A *step* is one batch of `--batch` synthetic depth frames of ONE stream pushed through the hot path
(awareness raycast + log-odds block-map update) in order.  Inputs (uint16 depth frames + poses) are resident in
HBM before the timed region starts.  Workload at N=1 = BASELINE config 2 (640x480 stream, 0.1 m local map,
S1 parameters); `--workload cfg3` selects config 3 (1280x720, 0.05 m).  At N=1 the same run also times a short
config-3 stream (`extra.cfg3`), the per-call latency of single frames (`extra.single_frame_us`) and compares the map of
the first frames of the stream with the CPU oracle's (`parity_check`).

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


def cpu_baseline(cfg, workload, frames, q, t, budget_s: float, gpu_check=None):
    """The CPU oracle (a port of the reference's map_awareness + map_local path: std::unordered_map/set, one thread per
    map — the reference is single-threaded per map) timed on the same stream, bounded to ~budget_s seconds per leg:
    (i) one stream on one core — its map is then compared with the GPU path's map of the same frames (`gpu_check`);
    (ii) one independent stream per PHYSICAL core, one process each (mirrors "one stream per GPU"), every
    process pinned, one warm frame, at least 10 timed frames."""
    from oracle.binding import OracleMap

    m = OracleMap(cfg)
    t0 = time.perf_counter()
    n1 = 0
    while n1 < q.shape[0] and (time.perf_counter() - t0 < budget_s or n1 < 3):
        m.update_depth(frames[n1 % frames.shape[0]], q[n1], t[n1])
        n1 += 1
    dt1 = time.perf_counter() - t0
    parity = gpu_check(m, n1) if gpu_check is not None else None
    m.close()
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
def time_stream(m, cfg, d_frames, q, t, B, K, W, distinct, sync, collect=None, settle=0):
    """W untimed (+ `settle` more untimed repeats of them) + K timed steps of B frames; returns seconds for the K steps."""
    fsz = cfg.width * cfg.height

    def run_step(s, timed):
        k0 = s * B
        f0 = k0 % distinct
        if f0 + B <= distinct:
            m.update_map_batch_dev(d_frames.data_ptr() + f0 * fsz * 2, B, cfg.width, cfg.height, q[k0:k0 + B], t[k0:k0 + B])
        else:
            for b in range(B):
                k = k0 + b
                m.update_map_dev(d_frames.data_ptr() + (k % distinct) * fsz * 2, cfg.width, cfg.height, q[k], t[k])
        if timed and collect is not None:
            collect(m.frame_stats())

    for s in range(W):
        run_step(s, False)
    for s in range(settle):
        run_step(s % max(1, W), False)
    sync()
    gc.collect()
    gc.disable()
    t0 = time.perf_counter()
    for s in range(W, W + K):
        run_step(s, True)
    sync()
    dt = time.perf_counter() - t0
    gc.enable()
    return dt, run_step


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


def single_frame_latency(MLMap, cfg, frames, q, t, d_frames, n_calls=120):
    """Per-call latency of the reference's own call pattern (ONE frame per depth_odom_input_callback, mlmap.cpp:463-507) in
    synchronous mode: median microseconds of mlm_integrate_depth_u16 (host buffer), mlm_integrate_depth_u16_dev (HBM
    resident) and mlm_integrate_callback with the default 500-pixel sampler."""
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
    return out


def main():
    if len(sys.argv) > 1 and sys.argv[1] == "--cpu-worker":
        return cpu_worker(sys.argv[2:])
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=100)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--batch", type=int, default=64, help="frames per step (one batched Stage A launch sequence)")
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
    B, K, W = args.batch, args.steps, args.warmup
    args.distinct = max(B, (args.distinct + B - 1) // B * B)  # whole batches: a step is always ONE batched submission
    n_total = (K + W) * B
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

    # warm-up (untimed), then: which kernel dominates?  A few fully instrumented batches (start/stop events of every
    # launch, on the streams the kernels run on) BEFORE the timed region: the kernel with the largest summed device time is
    # the one the roofline is about.  Bracketing every kernel costs throughput (the per-frame chain is latency bound), so
    # in the timed region only that kernel's launches are bracketed (timing mode 3), every 8th of them when it is
    # launched per frame.
    _, run_step = time_stream(m, cfg, d_frames, q, t, B, 0, W, args.distinct, barrier)
    ktime_c, ktime = {}, {}
    n_c = min(K, 6)
    n_settle = 24  # untimed batches between the instrumented ones and the timed region (see below)
    timed_kernel, timed_every = None, 1
    if not args.no_kernel_timing:
        m.enable_kernel_timing(2)
        for s in range(n_c):
            run_step(s, False)
        m.sync()
        for name, ms in m.kernel_times():
            a = ktime_c.setdefault(name, [0.0, 0])
            a[0] += ms
            a[1] += 1
        timed_kernel = max(ktime_c.items(), key=lambda kv: kv[1][0])[0]
        per_frame = ktime_c[timed_kernel][1] >= n_c * B
        timed_every = 8 if per_frame else 1
        m.set_timed_kernel(timed_kernel, timed_every)
        m.enable_kernel_timing(3)
    # settle back into the pipelined regime (their launches are bracketed as well).  Two dozen batches: the HIP runtime
    # grows its pools of signals / kernel-argument buffers while the first few dozen asynchronous batches are in flight
    # (several ms each time, seen at the 3rd, 5th, 9th and ~18th submission of a process)
    for s in range(n_settle):
        run_step(s % max(1, W), False)
    barrier()
    stats = []
    step_t = []
    # (the interpreter's cyclic garbage collector stays out of the timed region: a generation-2 pass over the modules
    # loaded here takes ~40 ms — more than many a timed region — at a point that depends on the allocation count)
    gc.collect()
    gc.disable()
    t0 = time.perf_counter()
    for s in range(W, W + K):
        run_step(s, True)
        stats.append(m.frame_stats())
        step_t.append(time.perf_counter())
    barrier()
    dt = time.perf_counter() - t0
    gc.enable()
    if os.environ.get("MLM_BENCH_STEP_TIMES"):  # diagnostic: host time of every submission call of the timed region
        d = np.diff(np.array([t0] + step_t)) * 1e3
        print("step ms:", [(i, round(float(x), 2)) for i, x in enumerate(d) if x > 1.5], "final barrier", round((t0 + dt - step_t[-1]) * 1e3, 2),
              file=sys.stderr)
    coll_dev = f"cuda:{local_rank}" if backend == "nccl" else "cpu"
    if dist is not None:
        tt = torch.tensor([dt], dtype=torch.float64, device=coll_dev)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        dt = float(tt.item())
    iso = {}
    if not args.no_kernel_timing:
        for name, ms in m.kernel_times():
            a = ktime.setdefault(name, [0.0, 0])
            a[0] += ms
            a[1] += 1
        # the same kernels alone on the GPU: synchronous batches (a batch's Stage A is complete before its per-frame launches
        # start, nothing else is in flight), every launch bracketed
        m.set_async(False)
        m.enable_kernel_timing(2)
        for s in range(2):
            run_step(s, False)
        m.sync()
        for name, ms in m.kernel_times():
            a = iso.setdefault(name, [0.0, 0])
            a[0] += ms
            a[1] += 1
        m.enable_kernel_timing(0)
        m.set_async(True)
    fsz = cfg.width * cfg.height
    algo_bytes = [2 * fsz + 10 * (st["n_hit_cells"] + st["n_miss_cells"]) for st in stats]
    atomics = [st["n_device_atomics"] for st in stats]

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
    n_host = min(K, 8)
    host_batch = np.ascontiguousarray(frames[[b % args.distinct for b in range(B)]])
    for s in range(4):  # the first host-buffer submission of every slot set allocates its image buffers
        m.update_map_batch(host_batch, q[s * B:s * B + B], t[s * B:s * B + B])
    m.sync()
    th = time.perf_counter()
    for s in range(n_host):
        k0 = s * B
        m.update_map_batch(host_batch, q[k0:k0 + B], t[k0:k0 + B])
    m.sync()
    pcie_fps = n_host * B / (time.perf_counter() - th)
    last_stats = m.frame_stats()
    m.close()

    if rank == 0:
        fps = world * K * B / dt
        mean_bytes = float(np.mean(algo_bytes)) if algo_bytes else 0.0
        mean_atomics = float(np.mean(atomics)) if atomics else 0.0
        roof = None
        # HBM-side bytes per launch of the dominant kernel from the PMC passes committed under profiles/ (rocprofv3
        # cannot be driven from inside this process; tools/pmc_workload.py + tools/pmc_traffic_json.py regenerate it)
        pmc, pmc_file = None, None
        try:
            f = newest_profile("pmc_traffic.json")
            if f and args.workload == "cfg2":
                pmc_file = os.path.basename(f)
                pmc = json.load(open(f))
        except Exception:
            pmc = None
        if ktime:
            dom = max(ktime.items(), key=lambda kv: kv[1][0])
            avg_ms = dom[1][0] / dom[1][1]          # average duration of one launch of the dominant kernel
            n_inst = (K + n_settle) * B             # frames whose launches of that kernel were candidates for bracketing
            frames_per_launch = n_inst / (dom[1][1] * timed_every)  # Stage A kernels: one launch per batch of B frames
            ach = mean_bytes * frames_per_launch / (avg_ms * 1e-3) / 1e9
            a_ach = mean_atomics * fps / world      # device-scope atomics per second of one GPU's stream
            iso_ms = iso[dom[0]][0] / iso[dom[0]][1] if dom[0] in iso else None
            roof = {"bound": "hbm", "kernel": dom[0], "achieved": ach, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                    "frac": ach / HBM_PEAK_GBS,
                    "frames_per_launch": frames_per_launch, "launches_bracketed": f"1 of {timed_every}",
                    "traffic": (pmc["kernels"][dom[0]]["total_bytes"] * frames_per_launch
                                if pmc and dom[0] in pmc.get("kernels", {}) else None),
                    "traffic_source": pmc_file, "avg_launch_us": avg_ms * 1e3,
                    "avg_launch_us_pipelined": avg_ms * 1e3,
                    "avg_launch_us_isolated": iso_ms * 1e3 if iso_ms else None,
                    "frac_isolated": (mean_bytes * frames_per_launch / (iso_ms * 1e-3) / 1e9 / HBM_PEAK_GBS) if iso_ms else None,
                    "algorithmic_bytes_per_frame": mean_bytes,
                    "atomics": {"bound": "device_atomics", "achieved": a_ach, "peak": ATOMICS_PEAK_PER_S, "unit": "atomics/s",
                                "frac": a_ach / ATOMICS_PEAK_PER_S, "atomics_per_frame": mean_atomics,
                                "counted_by": "the Stage A kernels themselves (chunk descriptors, list reservations, per-voxel counts)",
                                "peak_source": "tools/probes/atomic_probe.hip, measured on MI355X"},
                    "kernels_us_per_frame": {**{k + " (timed region)": v[0] * 1e3 * timed_every / max(1, n_inst) for k, v in ktime.items()},
                                             **{k + " (instrumented batches before the region)": v[0] * 1e3 / max(1, n_c * B)
                                                for k, v in ktime_c.items()},
                                             **{k + " (alone on the GPU)": v[0] * 1e3 / max(1, 2 * B) for k, v in iso.items()}}}
        out = {
            "metric": "depth frames/s into local map", "value": fps, "unit": "frames/s", "n_gpus": world,
            "steps": K, "warmup": W, "ms_per_step": dt / K * 1e3, "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "f64 index / f32 log-odds", "data": "synthetic",
            "config": {"workload": ("BASELINE cfg2: 640x480 room+jitter stream, random SE(3) poses, S1 0.1 m map"
                                    if world == 1 else f"BASELINE cfg4: {world} independent cfg2 streams (pose seeds 42..{41 + world}), one per GPU, "
                                                       "then one RCCL global-map merge (timed separately: merge)")
                       if args.workload == "cfg2" else "BASELINE cfg3: 1280x720 room+jitter stream, S3 0.05 m map",
                       "frames_per_step": B, "streams": world, "parallelism": f"{world} independent streams"},
            "achieved_hbm_gbs_whole_path": fps * mean_bytes / 1e9,
            "pcie_inclusive_frames_per_s": pcie_fps * world,
            "path": {"sector_fallbacks": last_stats["n_sector_fallbacks"], "spec_replays": last_stats["n_spec_replays"],
                     "logit_bit_exact": last_stats["logit_bit_exact"]},
            "roofline": roof,
        }
        if merge is not None:
            out["merge"] = merge
        if world == 1 and args.workload == "cfg2" and not args.no_extra:
            # config 3 (1280x720, 0.05 m) in the same invocation: a short stream, same protocol
            B3, K3, W3, D3 = 16, 12, 2, 16
            f3, q3, t3 = make_inputs(S3, D3, (K3 + W3) * B3, seed=42)
            d3 = torch.from_numpy(f3.view(np.int16)).cuda(local_rank)
            m3 = MLMap(S3, device=local_rank, max_blocks=65536, max_points=S3.width * S3.height, max_batch=B3)
            m3.set_async(True)

            def sync3():
                m3.sync()
                torch.cuda.synchronize()

            st3 = []
            dt3, _ = time_stream(m3, S3, d3, q3, t3, B3, K3, W3, D3, sync3, st3.append, settle=10)
            b3 = float(np.mean([2 * S3.width * S3.height + 10 * (s["n_hit_cells"] + s["n_miss_cells"]) for s in st3]))
            out["extra"] = {"cfg3": {"workload": "BASELINE cfg3: 1280x720 room+jitter stream, S3 0.05 m map", "value": K3 * B3 / dt3,
                                     "unit": "frames/s", "steps": K3, "frames_per_step": B3,
                                     "achieved_hbm_gbs_whole_path": K3 * B3 / dt3 * b3 / 1e9,
                                     "sector_fallbacks": st3[-1]["n_sector_fallbacks"]}}
            m3.close()
            del d3
            out["extra"]["single_frame_us"] = single_frame_latency(MLMap, cfg, frames, q, t, d_frames)
        if not args.no_cpu_baseline and world == 1:  # rank 0 at N=1 only
            def gpu_check(oracle_map, n):
                """A fresh GPU handle fed the frames leg (i) integrated, compared with the oracle's map (outside every timed
                region): block keys, occupancy classes and the float bits of every voxel's log-odds."""
                from tests.util import compare_maps

                g = MLMap(cfg, device=local_rank, max_blocks=32768, max_points=cfg.width * cfg.height, max_batch=16)
                g.set_async(True)
                for k in range(n):
                    g.update_map_dev(d_frames.data_ptr() + (k % args.distinct) * fsz * 2, cfg.width, cfg.height, q[k], t[k])
                gb, cb = g.export_blocks(), oracle_map.export_blocks()
                g.close()
                try:
                    d = compare_maps(gb, cb, "bench parity check", exact=False)
                    return {"frames": n, "blocks": d["blocks"], "voxels": d["cells"], "classes_equal": True, "max_dodd": d["max_dodd"],
                            "bit_mismatch": d["bit_mismatch"]}
                except AssertionError as e:
                    return {"frames": n, "classes_equal": False, "error": str(e)[:300]}

            out["cpu_baseline"], out["parity_check"] = cpu_baseline(cfg, args.workload, frames, q, t, args.cpu_budget, gpu_check)
        print(json.dumps(out))
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
