"""Where k_bin_points spends its cycles: runs the bench workload through the diagnostic build (`make -C
mlmapping_amd/csrc prof`, per-phase shader-clock sums) and prints each phase's share of the summed wave time."""
import ctypes
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mlmapping_amd import mlmap as mm
from mlmapping_amd import synthetic as syn
from mlmapping_amd.config import S1, S3

NAMES = ["0 LDS init", "1 load+bin (FP64)", "2 A: wave grouping / node post", "3 A: outer rays+queue", "4 sync",
         "5 B: LDS cell aggregation", "6 sync", "7 -", "8 -", "9 C: write-out (nodes, pairs)", "10 C: queued rays", "11 -"]
L = mm.load_library(os.path.join(os.path.dirname(mm.LIB_PATH), "libmlmap_hip_prof.so"))
L.mlm_debug_phases.argtypes = [ctypes.c_void_p]
mm._lib = L  # MLMap() below binds to the diagnostic build
cfg = S3 if "cfg3" in sys.argv else S1
m = mm.MLMap(cfg, max_blocks=32768, max_batch=16)
frames = list(syn.stream(cfg, "room_jitter", "random", 16))
imgs = np.stack([f[0] for f in frames])
q = np.stack([f[1][0] for f in frames])
t = np.stack([f[1][1] for f in frames])
buf = (ctypes.c_ulonglong * 16)()
m.update_map_batch(imgs, q, t)
L.mlm_debug_phases(buf)
for rep in range(2):
    m.update_map_batch(imgs, q, t)
    L.mlm_debug_phases(buf)
    tot = sum(buf[:12])
    print(f"rep {rep}: total wave-cycles {tot/1e6:.1f} M")
    for n, v in zip(NAMES, buf[:12]):
        print(f"  {n:32s} {100.0 * v / tot:5.1f} %")
    print(f"  (block,cell) pairs per frame {buf[12] / 16:.0f}, groups per frame {buf[13] / 16:.0f}")
