#!/usr/bin/env python3
"""bench.py — depth frames/s integrated into the map on MI355X (BASELINE.json metric).

A *step* is one batch of `--batch` synthetic depth frames of ONE stream pushed through the hot path
(awareness raycast + log-odds block-map update) in order.  Inputs (uint16 depth frames + poses) are resident in
HBM before the timed region starts.  Workload at N=1 = BASELINE config 2 (640x480 stream, 0.1 m local map,
S1 parameters); `--workload cfg3` selects config 3 (1280x720, 0.05 m).  At N=1 the same run also times a short
config-3 stream (`extra.cfg3`) so that both configurations are measured by the driver's own invocation.

Multi-GPU (`python -m torch.distributed.run --nproc-per-node N bench.py --gpus N ...`): one independent depth
stream + map per rank/GPU, no data-path collective (the path shards by stream: SURVEY.md §8e); the barrier and the
max-over-ranks reduction of the elapsed time go through torch.distributed (RCCL).  "scaling": "weak".

Rank 0 prints ONE JSON line.  `roofline` is the HBM roofline of the kernel with the largest summed device time
(measured live with start/stop events of the kernel's own launches on the stream it runs on); `roofline.atomics`
is the bound that actually limits this path — device-scope atomics, executed at the memory side — with the
atomics counted by the kernels themselves.  `cpu_baseline` is the CPU oracle on the box's host cores.
"""
from __future__ import annotations

import argparse
import gc
import json
import os
import sys
import threading
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # MI355X HBM3E spec peak, /opt/skills/guides/MI355X_MICROARCH.md
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


def cpu_baseline(cfg, frames, q, t, budget_s: float = 10.0):
    """The CPU oracle (a port of the reference's map_awareness + map_local path: std::unordered_map/set, one thread per
    map — the reference is single-threaded per map) timed on the same frames/poses, bounded to ~budget_s seconds per leg:
    (i) one stream on one core, (ii) one independent stream per host core (mirrors "one stream per GPU")."""
    from oracle.binding import OracleMap

    def run(n_threads: int, budget: float):
        counts = [0] * n_threads
        maps = [OracleMap(cfg) for _ in range(n_threads)]
        t_end = time.perf_counter() + budget

        def work(i):
            n = 0
            while n < q.shape[0] and (time.perf_counter() < t_end or n < 3):
                maps[i].update_depth(frames[n % frames.shape[0]], q[n], t[n])  # (ctypes releases the GIL)
                n += 1
            counts[i] = n

        t0 = time.perf_counter()
        th = [threading.Thread(target=work, args=(i,)) for i in range(n_threads)]
        for x in th:
            x.start()
        for x in th:
            x.join()
        dt = time.perf_counter() - t0
        return sum(counts), dt

    cores = os.cpu_count() or 1
    n1, dt1 = run(1, budget_s)
    nn, dtn = run(cores, budget_s) if cores > 1 else (n1, dt1)
    return {"value": nn / dtn, "unit": "frames/s", "cores": cores, "kind": "port",
            "one_core": n1 / dt1,
            "sample": f"first frames of the same stream: 1 thread {n1} frames in {dt1:.1f} s; {cores} threads (one map each) "
                      f"{nn} frames in {dtn:.1f} s; oracle/libmlmap_oracle.so"}


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


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=100)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--batch", type=int, default=64, help="frames per step (one batched Stage A launch sequence)")
    ap.add_argument("--workload", default="cfg2", choices=["cfg2", "cfg3"])
    ap.add_argument("--distinct", type=int, default=64, help="distinct depth frames kept in HBM (cycled; rounded up to a multiple of --batch)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-kernel-timing", action="store_true", help="do not record HIP events around the kernels")
    ap.add_argument("--no-extra", action="store_true", help="skip the short config-3 measurement")
    ap.add_argument("--cpu-budget", type=float, default=10.0)
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))

    import torch

    from mlmapping_amd.config import S1, S3
    from mlmapping_amd.mlmap import MLMap

    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: there is no CPU fallback")
    # MLM_BENCH_DIST_BACKEND=gloo (test hook, tests/test_gpu_boundary.py): the launcher contract — rank environment,
    # barrier, max over ranks, one JSON line from rank 0 — exercised on a box with fewer GPUs than ranks (ranks share
    # devices, the collectives run on the CPU).  The measured configuration is always one rank per GPU over RCCL.
    backend = os.environ.get("MLM_BENCH_DIST_BACKEND", "nccl")
    if backend != "nccl":
        local_rank = local_rank % max(1, torch.cuda.device_count())
    torch.cuda.set_device(local_rank)
    dist = None
    if world > 1:
        import torch.distributed as dist_mod

        dist = dist_mod
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group(backend)

    cfg = S1 if args.workload == "cfg2" else S3
    if os.environ.get("MLM_BENCH_NO_RAYCAST"):  # diagnostic only (not the BASELINE workload): hits without rays
        import dataclasses
        cfg = dataclasses.replace(cfg, use_raycasting=False)
    B, K, W = args.batch, args.steps, args.warmup
    args.distinct = max(B, (args.distinct + B - 1) // B * B)  # whole batches: a step is always ONE batched submission
    n_total = (K + W) * B
    frames, q, t = make_inputs(cfg, args.distinct, n_total, seed=42 + rank)
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
    if dist is not None:
        tt = torch.tensor([dt], dtype=torch.float64, device=f"cuda:{local_rank}" if backend == "nccl" else "cpu")
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        dt = float(tt.item())
    if not args.no_kernel_timing:
        for name, ms in m.kernel_times():
            a = ktime.setdefault(name, [0.0, 0])
            a[0] += ms
            a[1] += 1
        m.enable_kernel_timing(0)
    fsz = cfg.width * cfg.height
    algo_bytes = [2 * fsz + 10 * (st["n_hit_cells"] + st["n_miss_cells"]) for st in stats]
    atomics = [st["n_device_atomics"] for st in stats]
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
            import glob
            files = sorted(glob.glob(os.path.join(ROOT, "profiles", "r2*_pmc_traffic.json")))
            if files and args.workload == "cfg2":
                pmc_file = os.path.basename(files[-1])
                pmc = json.load(open(files[-1]))
        except Exception:
            pmc = None
        if ktime:
            dom = max(ktime.items(), key=lambda kv: kv[1][0])
            avg_ms = dom[1][0] / dom[1][1]          # average duration of one launch of the dominant kernel
            n_inst = (K + n_settle) * B             # frames whose launches of that kernel were candidates for bracketing
            frames_per_launch = n_inst / (dom[1][1] * timed_every)  # Stage A kernels: one launch per batch of B frames
            ach = mean_bytes * frames_per_launch / (avg_ms * 1e-3) / 1e9
            a_ach = mean_atomics * fps / world      # device-scope atomics per second of one GPU's stream
            roof = {"bound": "hbm", "kernel": dom[0], "achieved": ach, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                    "frac": ach / HBM_PEAK_GBS,
                    "frames_per_launch": frames_per_launch, "launches_bracketed": f"1 of {timed_every}",
                    "traffic": (pmc["kernels"][dom[0]]["total_bytes"] * frames_per_launch
                                if pmc and dom[0] in pmc.get("kernels", {}) else None),
                    "traffic_source": pmc_file, "avg_launch_us": avg_ms * 1e3,
                    "algorithmic_bytes_per_frame": mean_bytes,
                    "atomics": {"bound": "device_atomics", "achieved": a_ach, "peak": ATOMICS_PEAK_PER_S, "unit": "atomics/s",
                                "frac": a_ach / ATOMICS_PEAK_PER_S, "atomics_per_frame": mean_atomics,
                                "counted_by": "k_sector (chunk descriptors, list reservations, one bucket-min + one push per "
                                              "unique hit, one count per unique miss cell)",
                                "peak_source": "tools/probes/atomic_probe.hip, measured on MI355X"},
                    "kernels_us_per_frame": {**{k + " (timed region)": v[0] * 1e3 * timed_every / max(1, n_inst) for k, v in ktime.items()},
                                             **{k + " (instrumented batches before the region)": v[0] * 1e3 / max(1, n_c * B)
                                                for k, v in ktime_c.items()}}}
        out = {
            "metric": "depth frames/s into local map", "value": fps, "unit": "frames/s", "n_gpus": world,
            "steps": K, "warmup": W, "ms_per_step": dt / K * 1e3, "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "f64 index / f32 log-odds", "data": "synthetic",
            "config": {"workload": "BASELINE cfg2: 640x480 room+jitter stream, random SE(3) poses, S1 0.1 m map"
                       if args.workload == "cfg2" else "BASELINE cfg3: 1280x720 room+jitter stream, S3 0.05 m map",
                       "frames_per_step": B, "streams": world, "parallelism": f"{world} independent streams"},
            "achieved_hbm_gbs_whole_path": fps * mean_bytes / 1e9,
            "pcie_inclusive_frames_per_s": pcie_fps * world,
            "path": {"sector_fallbacks": last_stats["n_sector_fallbacks"], "spec_replays": last_stats["n_spec_replays"]},
            "roofline": roof,
        }
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
        if not args.no_cpu_baseline and world == 1:  # rank 0 at N=1 only
            out["cpu_baseline"] = cpu_baseline(cfg, frames, q, t, args.cpu_budget)
        print(json.dumps(out))
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
