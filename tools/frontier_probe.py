"""Frontier mode as bench.py measures the default mode: 32-frame batches resident in HBM, asynchronous submission."""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mlmapping_amd import synthetic as syn  # noqa: E402
from mlmapping_amd.config import S1  # noqa: E402
from mlmapping_amd.mlmap import MLMap  # noqa: E402

cfg = S1.with_(use_exploration_frontiers=True)
frames = list(syn.stream(cfg, "room_jitter", "smooth", 24))
B = int(os.environ.get("MLM_FP_BATCH", "32"))
gpu = MLMap(cfg, max_blocks=32768, max_batch=B)
f32 = np.stack([frames[k % len(frames)][0] for k in range(B)])
q32 = np.stack([frames[k % len(frames)][1][0] for k in range(B)])
t32 = np.stack([frames[k % len(frames)][1][1] for k in range(B)])
d32 = torch.from_numpy(f32.view(np.int16)).cuda()
torch.cuda.synchronize()
gpu.set_async(True)
for _ in range(4):
    gpu.update_map_batch_dev(d32.data_ptr(), B, cfg.width, cfg.height, q32, t32)
gpu.sync()
for rep in range(3):
    t0 = time.perf_counter()
    for _ in range(24):
        gpu.update_map_batch_dev(d32.data_ptr(), B, cfg.width, cfg.height, q32, t32)
    gpu.sync()
    print(f"frontier mode, {B}-frame async resident batches: {24 * B / (time.perf_counter() - t0):.0f} frames/s", gpu.frame_stats()["block_capacity"], gpu.frame_stats()["n_pool_grows"])
