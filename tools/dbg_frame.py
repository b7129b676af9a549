"""debug: one dense frame (config 1) against the oracle"""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mlmapping_amd.config import S1, S3
from mlmapping_amd import synthetic as syn
from mlmapping_amd.mlmap import MLMap
from oracle.binding import OracleMap
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
from util import compare_maps
cfg = S3 if "cfg3" in sys.argv else S1
gpu, cpu = MLMap(cfg, max_blocks=8192, record_awareness=True), OracleMap(cfg)
img = syn.room_depth(cfg)
q, t = syn.static_pose()
print("integrate", flush=True)
gpu.update_map(img, q, t)
print("done", gpu.frame_stats(), flush=True)
cpu.update_depth(img, q, t)
gc, go, _ = gpu.awareness_hits()
cc, co = cpu.hit_cells_sorted()
print("hit cells", len(gc), len(cc), "sets equal", np.array_equal(gc, cc))
if np.array_equal(gc, cc):
    bad = np.nonzero(go.view(np.uint32) != co.view(np.uint32))[0]
    print("odd mismatches", len(bad), [(int(gc[k]), float(go[k]), float(co[k])) for k in bad[:8]])
print(compare_maps(gpu.export_blocks(), cpu.export_blocks(), "frame 0"))
