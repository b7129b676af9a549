"""Per-kernel device time of ONE sampled frame (500 listed pixels, the reference's call pattern) with nothing else on the GPU:
synchronous single-frame calls with every launch bracketed (timing mode 2 takes the general submission, not the graph; the
kernels are the same).  usage: small_frame_times.py [sdef] [n_points]"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bench import make_inputs  # noqa: E402
from mlmapping_amd.config import S1, SDEF  # noqa: E402
from mlmapping_amd.mlmap import MLMap  # noqa: E402

cfg = SDEF if "sdef" in sys.argv else S1
npts = int([a for a in sys.argv[1:] if a.isdigit()][0]) if any(a.isdigit() for a in sys.argv[1:]) else 500
frames, q, t = make_inputs(cfg, 8, 80, 42)
m = MLMap(cfg, max_blocks=32768, max_points=cfg.width * cfg.height, max_batch=2)
rng = np.random.default_rng(1)
pix = [rng.choice(cfg.width * cfg.height, npts, replace=False).astype(np.int32) for _ in range(80)]
for k in range(16):
    m.update_map(frames[k % 8], q[k], t[k], pixel_idx=pix[k])
m.sync()
m.enable_kernel_timing(2)
n = 48
for k in range(16, 16 + n):
    m.update_map(frames[k % 8], q[k], t[k], pixel_idx=pix[k])
m.sync()
acc = {}
for name, ms in m.kernel_times():
    a = acc.setdefault(name, [0.0, 0])
    a[0] += ms
    a[1] += 1
tot = 0.0
for name, (ms, c) in acc.items():
    print(f"{name:18s} {ms * 1e3 / n:8.2f} us/frame  ({c / n:.2f} launches per frame)")
    tot += ms * 1e3 / n
print(f"{'sum':18s} {tot:8.2f} us/frame   ", {k: m.frame_stats()[k] for k in ("n_points", "n_hit_cells", "n_miss_cells", "n_multi_cells")})
