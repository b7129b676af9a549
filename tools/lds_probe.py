"""What mlm_create sets up for the standard configurations: LDS per column of k_sector, frame-local grid, device memory."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ["MLM_DEBUG_CREATE"] = "1"
from mlmapping_amd.config import S1, S3, SDEF
from mlmapping_amd.mlmap import MLMap
for name, c, kw in (("S1 batch 64", S1, dict(max_blocks=32768, max_batch=64)), ("S1 batch 1", S1, dict(max_blocks=32768, max_batch=1)),
                    ("S3 batch 16", S3, dict(max_blocks=65536, max_batch=16)), ("SDEF batch 1", SDEF, dict(max_blocks=4096, max_batch=1)),
                    ("S1 frontier batch 32", S1.with_(use_exploration_frontiers=True), dict(max_blocks=32768, max_batch=32))):
    print(name, file=sys.stderr)
    m = MLMap(c, **kw)
    m.close()
