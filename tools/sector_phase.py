"""Where k_sector spends its cycles: the bench workload through the diagnostic build (`make -C mlmapping_amd/csrc prof`,
per-phase shader-clock sums over the workgroups' waves) — each phase's share of the summed wave time."""
import ctypes
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mlmapping_amd import mlmap as mm
from mlmapping_amd import synthetic as syn
from mlmapping_amd.config import S1, S3

NAMES = ["0 set-up: column count, LDS tables, voxel axes, tile runs", "1 pass 0: book records on cells (after the records arrived)", "2 lists + reservations + multi-kind descriptors", "3 rays into the LDS mask (DDA)",
         "4 hit list, single-kind odds, voxel pushes", "5 pass 1: references", "6 miss cells: voxel counts + queue", "7 pass 0: chunk descriptors staged", "8 pass 0: records arrive", "9 function entry -> column count known, first descriptors requested"]
L = mm.load_library(os.path.join(os.path.dirname(mm.LIB_PATH), "libmlmap_hip_prof.so"))
L.mlm_debug_phases.argtypes = [ctypes.c_void_p]
mm._lib = L
cfg = S3 if "cfg3" in sys.argv else S1
n = int(os.environ.get("MLM_PHASE_BATCH", "16"))
m = mm.MLMap(cfg, max_blocks=32768, max_batch=n)
frames = list(syn.stream(cfg, "scatter", "smooth", n)) if "scatter" in sys.argv else list(syn.stream(cfg, "room_jitter", "random", n))
imgs = np.stack([f[0] for f in frames])
q = np.stack([f[1][0] for f in frames])
t = np.stack([f[1][1] for f in frames])
buf = (ctypes.c_ulonglong * 16)()
m.update_map_batch(imgs, q, t)
L.mlm_debug_phases(buf)
if "single" in sys.argv:  # frame by frame: a phase's cycles per wave = its share of a column's latency (80 columns x 8 waves at S1)
    m.set_async(False)
    rng = np.random.default_rng(1)
    pix = [rng.choice(cfg.width * cfg.height, 500, replace=False).astype(np.int32) if "sampled" in sys.argv else None for _ in range(n)]
    for k in range(n):
        m.update_map(imgs[k], q[k], t[k], pixel_idx=pix[k])
    L.mlm_debug_phases(buf)
    acc = np.zeros(10)
    for k in range(n):  # (a frame overwrites the previous frame's per-workgroup sums: read after every frame)
        m.update_map(imgs[k], q[k], t[k], pixel_idx=pix[k])
        L.mlm_debug_phases(buf)
        acc += np.array(buf[:10], dtype=np.float64)
    st = m.frame_stats()
    tot = acc.sum()
    print(f"single frames: {tot / n / 2.4e3:.0f} wave-microseconds per frame in k_sector (2.4 GHz shader clock)", {k: st[k] for k in ("n_points", "n_hit_cells", "n_groups")})
    for nm, v in zip(NAMES, acc):
        print(f"  {nm:72s} {100.0 * v / tot:5.1f} %   {v / n / 2.4e3:8.1f} wave-us per frame")
    sys.exit(0)
for rep in range(4 if "scatter" in sys.argv else 2):
    m.update_map_batch(imgs, q, t)
    L.mlm_debug_phases(buf)
    tot = sum(buf[:10])
    print(f"rep {rep}: total wave-cycles {tot/1e6:.1f} M  ({tot / n / 8 / 1e3:.1f} k cycles per frame per wave slot)")
    for nm, v in zip(NAMES, buf[:10]):
        print(f"  {nm:40s} {100.0 * v / tot:5.1f} %   {v / n / 1e6:.3f} M wave-cycles per frame")
print(m.frame_stats())
