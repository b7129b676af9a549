"""Frames/s on the worst-case "scatter" scene (every pixel in a cell of its own): asynchronous 32-frame batches from HBM."""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mlmapping_amd import synthetic as syn  # noqa: E402
from mlmapping_amd.config import S1  # noqa: E402
from mlmapping_amd.mlmap import MLMap  # noqa: E402

cfg, B = S1, 32
frames = np.stack([f for f, _ in syn.stream(cfg, "scatter", "smooth", B)])
poses = [p for _, p in syn.stream(cfg, "scatter", "smooth", B)]
q = np.stack([p[0] for p in poses])
t = np.stack([p[1] for p in poses])
d = torch.from_numpy(frames.view(np.int16)).cuda()
torch.cuda.synchronize()
m = MLMap(cfg, max_blocks=32768, max_points=cfg.width * cfg.height, max_batch=B)
m.set_async(True)
for _ in range(6):
    m.update_map_batch_dev(d.data_ptr(), B, cfg.width, cfg.height, q, t)
m.sync()
for rep in range(3):
    t0 = time.perf_counter()
    for _ in range(12):
        m.update_map_batch_dev(d.data_ptr(), B, cfg.width, cfg.height, q, t)
    m.sync()
    print(f"scatter scene: {12 * B / (time.perf_counter() - t0):.0f} frames/s", {k: m.frame_stats()[k] for k in ("n_hit_cells", "n_sector_fallbacks")})
