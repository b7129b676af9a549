"""Host-buffer (PCIe-inclusive) rate of mlm_integrate_depth_batch on the bench workload: 64-frame batches from pageable memory and
from a buffer pinned with mlm_host_register, several repetitions."""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bench import make_inputs  # noqa: E402
from mlmapping_amd.config import S1  # noqa: E402
from mlmapping_amd.mlmap import MLMap  # noqa: E402

B = 64
frames, q, t = make_inputs(S1, B, B * 40, seed=42)
frames = np.ascontiguousarray(frames)
m = MLMap(S1, max_blocks=32768, max_points=S1.width * S1.height, max_batch=B)
m.set_async(True)
for pinned in (False, True):
    if pinned:
        m.host_register(frames)
    for rep in range(4):
        t0 = time.perf_counter()
        for s in range(8):
            k0 = ((rep * 8 + s) * B) % (B * 39)
            m.update_map_batch(frames, q[k0:k0 + B], t[k0:k0 + B])
        m.sync()
        dt = time.perf_counter() - t0
        print(f"{'pinned  ' if pinned else 'pageable'} rep {rep}: {8 * B / dt:8.0f} frames/s  ({dt / 8 * 1e3:.2f} ms per {B}-frame batch, "
              f"{8 * B * frames[0].nbytes / dt / 1e9:.1f} GB/s over the link)")
m.host_unregister(frames)
