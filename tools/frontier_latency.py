"""Frontier mode, one frame per synchronous call: dense VGA frames at S1 and the shipped config2.yaml callback (500 samples), frames/s and
microseconds per call.  Knob ex_defer via MLM_KNOBS."""
import gc, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from mlmapping_amd import synthetic as syn
from mlmapping_amd.config import CONFIG2_YAML as C2, S1
from mlmapping_amd.mlmap import MLMap
gc.disable()
cfg = S1.with_(use_exploration_frontiers=True)
frames = list(syn.stream(cfg, "room_jitter", "smooth", 64))
m = MLMap(cfg, max_blocks=32768, max_points=cfg.width * cfg.height, max_batch=2)
for img, (q, t) in frames[:8]: m.update_map(img, q, t)
ts = []
for img, (q, t) in frames[8:]:
    a = time.perf_counter(); m.update_map(img, q, t); ts.append(time.perf_counter() - a)
m.sync(); m.close()
print("dense VGA frontier frame by frame: %.1f us median, %.0f frames/s" % (np.median(ts) * 1e6, 1 / np.mean(ts)))
m = MLMap(C2, max_blocks=16384, max_points=C2.width * C2.height, max_batch=2)
depth = [(syn.jitter_depth(syn.room_depth(C2), k).astype(np.float32) / 1000.0) for k in range(8)]
traj = syn.smooth_trajectory(400, 7); z = np.zeros(3)
ts = []
for k in range(300):
    q, t = traj[k]
    a = time.perf_counter(); m.depth_odom_callback(depth[k % 8], 0.0, t, q, z, 0.0, z, 0.0, 0.085, sampled=True); ts.append(time.perf_counter() - a)
a = np.array(ts[20:]) * 1e6
print("config2.yaml callback (no inflate): %.1f us median, mean %.1f, p10 %.1f, p90 %.1f, p99 %.1f" % (np.median(a), a.mean(), np.percentile(a, 10), np.percentile(a, 90), np.percentile(a, 99)))
print("  first 12 calls after warm-up:", np.round(a[:12], 0))
