"""Per-kernel times of frontier mode (use_exploration_frontiers: true): frame by frame and through the batch entry point."""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mlmapping_amd import synthetic as syn  # noqa: E402
from mlmapping_amd.config import S1  # noqa: E402
from mlmapping_amd.mlmap import MLMap  # noqa: E402

cfg = S1.with_(use_exploration_frontiers=True)
frames = list(syn.stream(cfg, "room_jitter", "smooth", 20))
for batch in (1, 16):
    m = MLMap(cfg, max_blocks=32768, max_batch=batch)
    fb = np.stack([f[0] for f in frames[4:20]])
    qb = np.stack([f[1][0] for f in frames[4:20]])
    tb = np.stack([f[1][1] for f in frames[4:20]])
    for img, (q, t) in frames[:4]:
        m.update_map(img, q, t)
    m.enable_kernel_timing(2)
    t0 = time.perf_counter()
    if batch == 1:
        for k in range(16):
            m.update_map(fb[k], qb[k], tb[k])
    else:
        m.update_map_batch(fb, qb, tb)
    m.sync()
    wall = time.perf_counter() - t0
    acc = {}
    for name, ms in m.kernel_times():
        a = acc.setdefault(name, [0.0, 0])
        a[0] += ms
        a[1] += 1
    tot = 0.0
    print(f"batch {batch}: wall {wall / 16 * 1e6:.0f} us/frame")
    for k, (ms, n) in acc.items():
        print(f"  {k:22s} {ms * 1e3 / 16:8.1f} us/frame ({n / 16:.2f} launches/frame)")
        tot += ms * 1e3 / 16
    print(f"  kernel sum {tot:.1f} us/frame")
    m.close()
