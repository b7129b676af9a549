"""Worst-case scene (every pixel its own cell: the sector tables overflow, the handle backs off to the cell-table path):
throughput of asynchronous 32-frame batches of HBM-resident frames."""
import gc, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from mlmapping_amd import synthetic as syn
from mlmapping_amd.config import S1
from mlmapping_amd.mlmap import MLMap
gc.disable()
cfg = S1
frames = list(syn.stream(cfg, "scatter", "smooth", 32))
gpu = MLMap(cfg, max_blocks=65536, max_batch=32)
f32 = np.stack([f[0] for f in frames]); q32 = np.stack([f[1][0] for f in frames]); t32 = np.stack([f[1][1] for f in frames])
d = torch.from_numpy(f32.view(np.int16)).cuda(); torch.cuda.synchronize()
gpu.set_async(True)
for _ in range(4): gpu.update_map_batch_dev(d.data_ptr(), 32, cfg.width, cfg.height, q32, t32)
gpu.sync()
t0 = time.perf_counter()
for _ in range(10): gpu.update_map_batch_dev(d.data_ptr(), 32, cfg.width, cfg.height, q32, t32)
gpu.sync()
st = gpu.frame_stats()
print("scatter scene, async batch 32:", round(10 * 32 / (time.perf_counter() - t0)), "frames/s; fall-backs", st["n_sector_fallbacks"], "hit cells", st["n_hit_cells"])
