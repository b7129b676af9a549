"""The bench workload exactly as bench.py submits it — asynchronous 64-frame contiguous device batches on a handle with three
slot sets (config 3: 16-frame batches) — for the rocprofv3 --pmc passes that feed profiles/*_pmc_traffic_batch[_cfg3].json and
the SQ passes.  Every kernel launch of the batched path (k_apply_tiles included) is in the trace; bytes per frame = counter sum
/ frames.  Usage: rocprofv3 --pmc <counters> --kernel-trace -d <dir> -- python3 tools/pmc_batch64.py [cfg3] [n_batches]
Prints `frames N` (what tools/pmc_traffic_json.py takes as its frames= argument).  With warm=W the measured batches are preceded by
W untimed ones and a marker kernel (k_probe_seeds): pmc_traffic_json.py then counts only what follows the marker — the steady
state bench.py times (the first frames of a stream replay the emulated container's rehashes and grow the pool)."""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bench import make_inputs  # noqa: E402
from mlmapping_amd.config import S1, S3  # noqa: E402
from mlmapping_amd.mlmap import MLMap  # noqa: E402

cfg = S3 if "cfg3" in sys.argv else S1
B = 16 if cfg is S3 else 64
nb = int([a for a in sys.argv[1:] if a.isdigit()][0]) if any(a.isdigit() for a in sys.argv[1:]) else 6
warm = int([a for a in sys.argv[1:] if a.startswith("warm=")][0].split("=")[1]) if any(a.startswith("warm=") for a in sys.argv[1:]) else 0
frames, q, t = make_inputs(cfg, B, B * (nb + warm), seed=42)
d_frames = torch.from_numpy(frames.view(np.int16)).cuda()
torch.cuda.synchronize()
m = MLMap(cfg, max_blocks=65536 if cfg is S3 else 32768, max_points=cfg.width * cfg.height, max_batch=B)
m.set_async(True)
for j in range(warm):
    m.update_map_batch_dev(d_frames.data_ptr(), B, cfg.width, cfg.height, q[j * B:(j + 1) * B], t[j * B:(j + 1) * B])
if warm:
    m.sync()
    m.debug_probe_seeds()  # (the marker)
q, t = q[warm * B:], t[warm * B:]
for j in range(nb):
    m.update_map_batch_dev(d_frames.data_ptr(), B, cfg.width, cfg.height, q[j * B:(j + 1) * B], t[j * B:(j + 1) * B])
m.sync()
print("frames", nb * B, m.frame_stats())
