"""Per-call time of the ROS-free callback in the reference's default mode (500 rand() samples of a 32FC1 frame)."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mlmapping_amd import synthetic as syn
from mlmapping_amd.config import SDEF
from mlmapping_amd.mlmap import MLMap

cfg = SDEF
base = syn.room_depth(cfg).astype(np.float32) / 1000.0
gpu = MLMap(cfg, max_blocks=4096)
traj = syn.smooth_trajectory(40, 5)
def call(k):
    q, t = traj[k % 40]
    return gpu.depth_odom_callback(base, t_img=10.0 + k / 30.0, odom_p=t, odom_q=q, odom_v=[0.3, -0.1, 0.02], t_odom=10.0 + k / 30.0 - 0.004,
                                   imu_w=[0.05, -0.2, 0.4], t_imu=10.0 + k / 30.0 - 0.002, latency=0.085, sampled=True)
for k in range(3):
    call(k)
ts = []
for k in range(3, 43):
    t0 = time.perf_counter(); call(k); ts.append((time.perf_counter() - t0) * 1e3)
print("ms per call:", " ".join(f"{x:.2f}" for x in ts))
print(gpu.frame_stats())
