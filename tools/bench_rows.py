"""Throughput of the widened rows of the scope table on the GPU, with the CPU oracle timed beside them on the
same inputs: frontier mode, the ROS-free depth/odom callback (reference-default 500-sample path and dense), obstacle
inflation, the planner queries, setFree_map_in_bound and the /global_map payload.  One JSON object on stdout.
The oracle is used here as the CPU baseline only (like bench.py's cpu_baseline leg)."""
import gc
import json
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mlmapping_amd import synthetic as syn  # noqa: E402
from mlmapping_amd.config import S1, SDEF  # noqa: E402
from mlmapping_amd.mlmap import MLMap  # noqa: E402
from oracle.binding import OracleMap  # noqa: E402

gc.disable()  # (a generation-2 collection is a 40 ms pause: it would land in one of the short timed loops below)


def timeit(f, reps):
    f()
    t0 = time.perf_counter()
    for _ in range(reps):
        f()
    return (time.perf_counter() - t0) / reps


out = {}

# ---- frontier mode (use_exploration_frontiers: true), VGA, S1 map
cfg = S1.with_(use_exploration_frontiers=True)
frames = list(syn.stream(cfg, "room_jitter", "smooth", 24))
gpu, cpu = MLMap(cfg, max_blocks=32768), OracleMap(cfg)
for img, (q, t) in frames[:4]:
    gpu.update_map(img, q, t)
t0 = time.perf_counter()
for img, (q, t) in frames[4:]:
    gpu.update_map(img, q, t)
gpu.sync()
g = (len(frames) - 4) / (time.perf_counter() - t0)
t0 = time.perf_counter()
for img, (q, t) in frames[:6]:
    cpu.update_depth(img, q, t)
c = 6 / (time.perf_counter() - t0)
out["frontier_mode_frames_per_s"] = {"gpu": g, "cpu_oracle_1_thread": c, "workload": "640x480 room+jitter, S1, frame by frame"}
gpu.close()
# the same through the batch entry point: Stage A of 16 frames in one launch sequence
gpu = MLMap(cfg, max_blocks=32768, max_batch=16)
fb = np.stack([f[0] for f in frames[:16]])
qb = np.stack([f[1][0] for f in frames[:16]])
tb = np.stack([f[1][1] for f in frames[:16]])
gpu.update_map_batch(fb, qb, tb)
t0 = time.perf_counter()
for _ in range(6):
    gpu.update_map_batch(fb, qb, tb)
gpu.sync()
out["frontier_mode_batch16_frames_per_s"] = {"gpu": 6 * 16 / (time.perf_counter() - t0), "cpu_oracle_1_thread": c}
# ... and with asynchronous submission (Stage A of a batch overlaps the map-dependent part of the batch before it)
gpu.set_async(True)
gpu.update_map_batch(fb, qb, tb)
gpu.sync()
t0 = time.perf_counter()
for _ in range(12):
    gpu.update_map_batch(fb, qb, tb)
gpu.sync()
out["frontier_mode_batch16_async_frames_per_s"] = {"gpu": 12 * 16 / (time.perf_counter() - t0), "cpu_oracle_1_thread": c}
gpu.close()
# ... and as bench.py measures the default mode: 32-frame batches, frames resident in HBM, asynchronous submission
import torch  # noqa: E402

gpu = MLMap(cfg, max_blocks=32768, max_batch=32)
f32 = np.stack([frames[k % len(frames)][0] for k in range(32)])
q32 = np.stack([frames[k % len(frames)][1][0] for k in range(32)])
t32 = np.stack([frames[k % len(frames)][1][1] for k in range(32)])
d32 = torch.from_numpy(f32.view(np.int16)).cuda()
torch.cuda.synchronize()
gpu.set_async(True)
for _ in range(4):  # (the block pool grows to its working size in the first calls: growth is not what this row measures)
    gpu.update_map_batch_dev(d32.data_ptr(), 32, cfg.width, cfg.height, q32, t32)
gpu.sync()
t0 = time.perf_counter()
for _ in range(24):
    gpu.update_map_batch_dev(d32.data_ptr(), 32, cfg.width, cfg.height, q32, t32)
gpu.sync()
out["frontier_mode_batch32_async_resident_frames_per_s"] = {"gpu": 24 * 32 / (time.perf_counter() - t0), "cpu_oracle_1_thread": c,
                                                            "workload": "as bench.py: frames resident in HBM"}
gpu.close()
del d32

# ---- map for the query rows: 40 frames of the bench stream
cfg = S1
gpu, cpu = MLMap(cfg, max_blocks=32768, max_batch=8), OracleMap(cfg)
fr = list(syn.stream(cfg, "room_jitter", "random", 16))
for img, (q, t) in fr:
    gpu.update_map(img, q, t)
    cpu.update_depth(img, q, t)
rng = np.random.default_rng(3)
pos = rng.uniform(-6, 6, size=(1_000_000, 3))
pos_c = pos[:100_000]
for name, fg, fc in (
        ("getOccupancy", lambda: gpu.getOccupancy(pos), lambda: cpu.getOccupancy(pos_c)),
        ("getOccupancy_inflate", lambda: gpu.getOccupancy(pos, 0.1), lambda: cpu.getOccupancy(pos_c, 0.1)),
        ("getOdd", lambda: gpu.getOdd(pos), lambda: cpu.getOdd(pos_c)),
        ("getOddGrad", lambda: gpu.getOddGrad(pos), lambda: cpu.getOddGrad(pos_c))):
    tg, tc = timeit(fg, 3), timeit(fc, 1)
    out[name + "_Mpos_per_s"] = {"gpu_incl_pcie": len(pos) / tg / 1e6, "cpu_oracle_1_thread": len(pos_c) / tc / 1e6}
ct = fr[-1][1][1]
tg, tc = timeit(lambda: gpu.inflate_map(ct), 5), timeit(lambda: cpu.inflate_map(ct), 2)
out["inflate_map_ms"] = {"gpu": tg * 1e3, "cpu_oracle_1_thread": tc * 1e3, "blocks": gpu.block_count()}
tg, tc = timeit(lambda: gpu.getInflateOccupancy(pos), 3), timeit(lambda: cpu.getInflateOccupancy(pos_c), 1)
out["getInflateOccupancy_Mpos_per_s"] = {"gpu_incl_pcie": len(pos) / tg / 1e6, "cpu_oracle_1_thread": len(pos_c) / tc / 1e6}
tg, tc = timeit(lambda: gpu.global_map_points(), 5), timeit(lambda: cpu.global_map_points(), 2)
out["global_map_export_ms"] = {"gpu_incl_pcie": tg * 1e3, "cpu_oracle_1_thread": tc * 1e3, "points": int(len(gpu.global_map_points()))}
lo, hi = np.array([-1.0, -1.0, -0.5]), np.array([1.0, 1.0, 0.5])
tg, tc = timeit(lambda: gpu.setFree_map_in_bound(lo, hi), 5), timeit(lambda: cpu.setFree_map_in_bound(lo, hi), 2)
out["setFree_map_in_bound_ms"] = {"gpu": tg * 1e3, "cpu_oracle_1_thread": tc * 1e3, "box_m": [2, 2, 1]}
gpu.close()

# ---- ROS-free callback, reference default configuration (500 rand() samples of a 32FC1 frame) and dense
cfg = SDEF
base = syn.room_depth(cfg).astype(np.float32) / 1000.0
for sampled in (True, False):
    gpu, cpu = MLMap(cfg, max_blocks=4096), OracleMap(cfg)
    traj = syn.smooth_trajectory(40, 5)

    def call(m, k):
        q, t = traj[k % 40]
        return m.depth_odom_callback(base, t_img=10.0 + k / 30.0, odom_p=t, odom_q=q, odom_v=[0.3, -0.1, 0.02],
                                     t_odom=10.0 + k / 30.0 - 0.004, imu_w=[0.05, -0.2, 0.4], t_imu=10.0 + k / 30.0 - 0.002,
                                     latency=0.085, sampled=sampled)
    for k in range(3):
        call(gpu, k)
    t0 = time.perf_counter()
    per = []
    for k in range(3, 33):
        t1 = time.perf_counter()
        call(gpu, k)
        per.append((time.perf_counter() - t1) * 1e3)
    g = 30 / (time.perf_counter() - t0)
    if os.environ.get("MLM_ROWS_DEBUG"):
        print("callback ms per call:", " ".join(f"{x:.2f}" for x in per), file=sys.stderr)
    n_c = 30 if sampled else 6
    t0 = time.perf_counter()
    for k in range(n_c):
        call(cpu, k)
    c = n_c / (time.perf_counter() - t0)
    out["callback_%s_frames_per_s" % ("sampled500" if sampled else "dense")] = {
        "gpu_incl_pcie": g, "cpu_oracle_1_thread": c, "workload": "%dx%d 32FC1 frame, reference default map" % (cfg.width, cfg.height)}
    gpu.close()

print(json.dumps(out, indent=1))
