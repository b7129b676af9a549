"""Host time of one batched submission (Stage A launches + one apply launch per frame) with the GPU idle: how far the
submitting thread is from being the bound."""
import gc, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from bench import make_inputs
from mlmapping_amd.config import S1
from mlmapping_amd.mlmap import MLMap

B = int(sys.argv[1]) if len(sys.argv) > 1 else 64
frames, q, t = make_inputs(S1, B, 40 * B, 42)
d = torch.from_numpy(frames.view(np.int16)).cuda()
torch.cuda.synchronize()
m = MLMap(S1, max_blocks=32768, max_points=640 * 480, max_batch=B)
m.set_async(True)
gc.disable()
ts = []
for s in range(40):
    m.sync()
    t0 = time.perf_counter()
    m.update_map_batch_dev(d.data_ptr(), B, 640, 480, q[s * B:s * B + B], t[s * B:s * B + B])
    ts.append((time.perf_counter() - t0) * 1e3)
m.sync()
ts = np.array(ts[8:])
print(f"batch {B}: submission {np.median(ts):.3f} ms median ({np.median(ts) / B * 1e3:.1f} us per frame), min {ts.min():.3f}, max {ts.max():.3f}")
