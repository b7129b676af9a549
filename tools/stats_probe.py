import sys, numpy as np, time
sys.path.insert(0,'.')
from mlmapping_amd import synthetic as syn
from mlmapping_amd.config import S1, S3
from mlmapping_amd.mlmap import MLMap
for name,cfg,scene,poses in [("cfg2-bench",S1,"room_jitter","random"),("cfg1-room-static",S1,"room","static"),("scatter",S1,"scatter","static"),("cfg3",S3,"room_jitter","random")]:
    m = MLMap(cfg, max_blocks=32768, max_batch=2)
    acc=[]
    for img,(q,t) in syn.stream(cfg, scene, poses, 6):
        t0=time.perf_counter(); m.update_map(img,q,t); dt=time.perf_counter()-t0
        st=m.frame_stats(); st['ms']=round(dt*1e3,3); acc.append(st)
    print(name, acc[-1])
    m.close()
