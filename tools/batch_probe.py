import sys, time, numpy as np
sys.path.insert(0,'.')
from bench import make_inputs
from mlmapping_amd.config import S1
from mlmapping_amd.mlmap import MLMap
B=int(sys.argv[1])
frames,q,t=make_inputs(S1,32,B*12,seed=42)
idx=[b%32 for b in range(B)]
fb=np.ascontiguousarray(frames[idx])
m=MLMap(S1,max_blocks=32768,max_points=S1.width*S1.height,max_batch=B)
m.set_async(True)
for s in range(12):
    t0=time.perf_counter(); m.update_map_batch(fb,q[s*B:s*B+B],t[s*B:s*B+B]); m.sync(); dt=time.perf_counter()-t0
    st=m.frame_stats()
    print(B,s,round(dt*1e3,2),'ms', {k:st[k] for k in ('n_spec_replays','n_rehash_epochs','n_blocks')})
