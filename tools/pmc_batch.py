"""Three 16-frame batches of the bench workload (batched Stage A launches, as in bench.py) for rocprofv3 --pmc passes on
the SQ counters.  Usage: rocprofv3 --pmc <counters> --kernel-trace -d <dir> -- python3 tools/pmc_batch.py [cfg3]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bench import make_inputs  # noqa: E402
from mlmapping_amd.config import S1, S3  # noqa: E402
from mlmapping_amd.mlmap import MLMap  # noqa: E402

cfg = S3 if "cfg3" in sys.argv else S1
frames, q, t = make_inputs(cfg, 16, 48, seed=42)
m = MLMap(cfg, max_blocks=65536 if cfg is S3 else 32768, max_points=cfg.width * cfg.height, max_batch=16)
for k in range(3):
    m.update_map_batch(frames, q[16 * k:16 * k + 16], t[16 * k:16 * k + 16])
print(m.frame_stats())
