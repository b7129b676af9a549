"""Diagnostic: host time of every batch submission of the bench stream and the longest kernel launches around a slow one."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

from bench import make_inputs
from mlmapping_amd.config import S1
from mlmapping_amd.mlmap import MLMap

K = int(sys.argv[1]) if len(sys.argv) > 1 else 20
W, B = 5, 32
frames, q, t = make_inputs(S1, 32, (K + W) * B, 42)
d = torch.from_numpy(frames.view(np.int16)).cuda()
torch.cuda.synchronize()
m = MLMap(S1, max_blocks=32768, max_points=640 * 480, max_batch=B)
m.set_async(True)
import gc
gc.disable()
ts = []
m.enable_kernel_timing(2)
for s in range(W + K):
    t0 = time.perf_counter()
    m.update_map_batch_dev(d.data_ptr(), B, 640, 480, q[s * B:s * B + B], t[s * B:s * B + B])
    ts.append(time.perf_counter() - t0)
m.sync()
kt = m.kernel_times()
ts = np.array(ts) * 1e3
print("slow submissions (ms):", [(i, round(float(x), 1)) for i, x in enumerate(ts) if x > 2.0])
big = [(i, n, round(ms, 2)) for i, (n, ms) in enumerate(kt) if ms > 0.5]
print("launches longer than 0.5 ms (index, kernel, ms):", big[:40])
st = m.frame_stats()
print({k: st[k] for k in ("n_blocks", "n_spec_replays", "hit_bucket_count", "n_sector_fallbacks")})
