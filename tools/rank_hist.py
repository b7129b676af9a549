"""How many contributions the cells that go through k_rank have (diagnostic build, `make -C mlmapping_amd/csrc prof`): a histogram
over a few frames of the bench workload.  usage: rank_hist.py [cfg3] [scatter]"""
import ctypes
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mlmapping_amd import mlmap as mm
from mlmapping_amd import synthetic as syn
from mlmapping_amd.config import S1, S3

L = mm.load_library(os.path.join(os.path.dirname(mm.LIB_PATH), "libmlmap_hip_prof.so"))
L.mlm_debug_spans.argtypes = [ctypes.c_void_p]
mm._lib = L
cfg = S3 if "cfg3" in sys.argv else S1
scene = "scatter" if "scatter" in sys.argv else "room_jitter"
n = 6
m = mm.MLMap(cfg, max_blocks=32768, max_batch=2)
buf = (ctypes.c_ulonglong * 32)()
frames = list(syn.stream(cfg, scene, "static" if scene == "scatter" else "random", n))
m.update_map(frames[0][0], *frames[0][1])
L.mlm_debug_spans(buf)
for img, (q, t) in frames[1:]:
    m.update_map(img, q, t)
L.mlm_debug_spans(buf)
h = np.array([buf[9 + 2 * b] for b in range(9)], dtype=np.float64) / (n - 1)
names = ["3", "4", "5-8", "9-16", "17-32", "33-64", "65-128", "129-256", ">256"]
print(("cfg3" if cfg is S3 else "cfg2"), scene, "ranked cells per frame: %.0f, references per cell %.1f" % (h.sum(), buf[27] / (n - 1) / max(1.0, h.sum())))
print("  cells that would not fit one 64-pixel word per row: %.0f (%.1f %%); cells taller than 32 rows: %.0f (%.1f %%)" % (buf[29] / (n - 1), 100.0 * buf[29] / (n - 1) / max(1.0, h.sum()), buf[31] / (n - 1), 100.0 * buf[31] / (n - 1) / max(1.0, h.sum())))
for nm, v in zip(names, h):
    print("  n = %-8s %8.0f cells  %5.1f %%" % (nm, v, 100.0 * v / max(1.0, h.sum())))
