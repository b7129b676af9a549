"""What of a lone frame's k_sector / k_tile time is the workgroups' own: first workgroup start -> last workgroup end and the longest
workgroup (diagnostic build, `make -C mlmapping_amd/csrc prof`), beside the kernel's duration between HIP events.
usage: kernel_spans.py [sampled] [sdef]"""
import ctypes
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mlmapping_amd import mlmap as mm
from mlmapping_amd import synthetic as syn
from mlmapping_amd.config import S1, SDEF

L = mm.load_library(os.path.join(os.path.dirname(mm.LIB_PATH), "libmlmap_hip_prof.so"))
L.mlm_debug_spans.argtypes = [ctypes.c_void_p]
mm._lib = L
cfg = SDEF if "sdef" in sys.argv else S1
n = 24
m = mm.MLMap(cfg, max_blocks=32768, max_batch=2)
frames = list(syn.stream(cfg, "room_jitter", "random", n))
rng = np.random.default_rng(1)
pix = rng.choice(cfg.width * cfg.height, 500, replace=False).astype(np.int32) if "sampled" in sys.argv else None
buf = (ctypes.c_ulonglong * 32)()
for img, (q, t) in frames[:8]:
    m.update_map(img, q, t, pixel_idx=pix)
L.mlm_debug_spans(buf)
acc = np.zeros((2, 3))
for img, (q, t) in frames[8:]:
    m.update_map(img, q, t, pixel_idx=pix)
    L.mlm_debug_spans(buf)
    ratio = (buf[28], buf[29])
    for k in range(2):
        acc[k] += [(buf[4 * k + 1] - buf[4 * k]) / 100.0, buf[4 * k + 2] / 100.0, buf[4 * k + 3]]
acc /= n - 8
for k, nm in enumerate(["k_sector", "k_tile"]):
    print(f"{nm:10s} first workgroup start -> last end {acc[k, 0]:6.2f} us, longest workgroup {acc[k, 1]:6.2f} us, {acc[k, 2]:.0f} workgroups reached the end")
L.mlm_debug_wg_times.argtypes = [ctypes.c_void_p]
wg = np.zeros((2, 2048, 2), np.uint64)
L.mlm_debug_wg_times(wg.ctypes.data)
for k, (nm, nwg) in enumerate([("k_sector", cfg.am_n_phi if hasattr(cfg, "am_n_phi") else 360), ("k_tile", 361)]):
    a = wg[k, :nwg].astype(np.int64)
    t0 = a[:, 0].min()
    life = (a[:, 1] - a[:, 0]) / 100.0
    start = (a[:, 0] - t0) / 100.0
    order = np.argsort(-life)[:12]
    print(nm, "last frame: workgroup lifetimes us: median %.2f p90 %.2f max %.2f; starts: median %.2f max %.2f" % (np.median(life), np.percentile(life, 90), life.max(), np.median(start), start.max()))
    print("   longest:", " ".join(f"wg{int(i)}:{life[i]:.1f}(start {start[i]:.1f})" for i in order))
print("clock64 ticks per 1000 ticks of the 100 MHz clock (workgroup 70):", ratio)
m.enable_kernel_timing(2)
for img, (q, t) in frames[8:]:
    m.update_map(img, q, t, pixel_idx=pix)
m.sync()
acc = {}
for name, ms in m.kernel_times():
    a = acc.setdefault(name, [0.0, 0])
    a[0] += ms
    a[1] += 1
for name, (ms, c) in acc.items():
    print(f"{name:18s} {ms * 1e3 / c:8.2f} us between its events")
