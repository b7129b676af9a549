"""Stage A / Stage B+C spans per 16-frame batch in the pipelined (async) bench configuration."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bench import make_inputs  # noqa: E402
from mlmapping_amd.config import S1  # noqa: E402
from mlmapping_amd.mlmap import MLMap  # noqa: E402

B = 16
frames, q, t = make_inputs(S1, B, B * 60, seed=42)
m = MLMap(S1, max_blocks=32768, max_points=S1.width * S1.height, max_batch=B)
m.set_async(True)
for k in range(5):
    m.update_map_batch(frames, q[B * k:B * k + B], t[B * k:B * k + B])
m.sync()
m.enable_kernel_timing(4)
for k in range(5, 45):
    m.update_map_batch(frames, q[B * k:B * k + B], t[B * k:B * k + B])
m.sync()
acc = {}
for name, ms in m.kernel_times():
    acc.setdefault(name, []).append(ms)
for name, v in acc.items():
    v = sorted(v)
    print(f"{name:16s} median {v[len(v)//2]*1e3/B:7.2f} us/frame  mean {sum(v)/len(v)*1e3/B:7.2f}  ({len(v)} batches)")
