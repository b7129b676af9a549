"""Frontier mode as bench.py measures the default mode: 32-frame batches of HBM-resident frames, asynchronous submission."""
import gc, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from mlmapping_amd import synthetic as syn
from mlmapping_amd.config import S1
from mlmapping_amd.mlmap import MLMap
gc.disable()
cfg = S1.with_(use_exploration_frontiers=True)
frames = list(syn.stream(cfg, "room_jitter", "smooth", 32))
gpu = MLMap(cfg, max_blocks=32768, max_batch=32)
f32 = np.stack([f[0] for f in frames]); q32 = np.stack([f[1][0] for f in frames]); t32 = np.stack([f[1][1] for f in frames])
d = torch.from_numpy(f32.view(np.int16)).cuda(); torch.cuda.synchronize()
gpu.set_async(True)
for _ in range(6): gpu.update_map_batch_dev(d.data_ptr(), 32, cfg.width, cfg.height, q32, t32)
gpu.sync()
t0 = time.perf_counter()
for _ in range(20): gpu.update_map_batch_dev(d.data_ptr(), 32, cfg.width, cfg.height, q32, t32)
gpu.sync()
print("frontier resident async batch 32:", round(20 * 32 / (time.perf_counter() - t0)), "frames/s")
