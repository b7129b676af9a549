"""Where k_tile spends a lone frame's time: single frames of the bench workload through the diagnostic build
(`make -C mlmapping_amd/csrc prof`): thread 0's clock per phase, averaged over the touched tiles."""
import ctypes
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mlmapping_amd import mlmap as mm
from mlmapping_amd import synthetic as syn
from mlmapping_amd.config import S1

NAMES = ["0 parameters, count + first descriptors, LDS clear", "1 counting pass", "2 block lookups arrive", "3 touched blocks + creation",
         "4 compaction scan + reservations", "5 records", "6 placing pass", "7 -"]
L = mm.load_library(os.path.join(os.path.dirname(mm.LIB_PATH), "libmlmap_hip_prof.so"))
L.mlm_debug_tile_phases.argtypes = [ctypes.c_void_p]
mm._lib = L
cfg, n = S1, 16
m = mm.MLMap(cfg, max_blocks=32768, max_batch=2)
frames = list(syn.stream(cfg, "room_jitter", "random", n))
buf = (ctypes.c_ulonglong * 8)()
rng = np.random.default_rng(1)
pix = rng.choice(cfg.width * cfg.height, 500, replace=False).astype(np.int32) if "sampled" in sys.argv else None
for img, (q, t) in frames[:4]:
    m.update_map(img, q, t, pixel_idx=pix)
L.mlm_debug_tile_phases(buf)
tiles = 0
for img, (q, t) in frames:
    m.update_map(img, q, t, pixel_idx=pix)
L.mlm_debug_tile_phases(buf)
tot = sum(buf)
print(f"k_tile, single frames: {tot / n / 100.0:.0f} tile-microseconds per frame at 100 MHz clock64 (sum over touched tiles)")
for nm, v in zip(NAMES, buf):
    print(f"  {nm:52s} {100.0 * v / max(1, tot):5.1f} %")
