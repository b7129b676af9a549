"""Single-frame integrate calls of the bench workload (so that one kernel launch == one frame) for the rocprofv3 --pmc
passes that feed profiles/*_pmc_traffic.json.  Usage: python tools/pmc_workload.py [n_frames]"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bench import make_inputs  # noqa: E402
from mlmapping_amd.config import S1  # noqa: E402
from mlmapping_amd.mlmap import MLMap  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 48
frames, q, t = make_inputs(S1, 16, n, seed=42)
m = MLMap(S1, max_blocks=32768, max_points=S1.width * S1.height, max_batch=1)
for k in range(n):
    m.update_map(frames[k % 16], q[k], t[k])
print(m.frame_stats())
