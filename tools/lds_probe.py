import sys; sys.path.insert(0,'.')
from mlmapping_amd.config import S1,S3
from mlmapping_amd.mlmap import MLMap
for c in (S1,S3, S1.with_(use_exploration_frontiers=True)):
    m=MLMap(c,max_blocks=4096,max_batch=2); m.close()
