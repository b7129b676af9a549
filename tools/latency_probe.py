"""Per-call latency of single frames in synchronous mode (the reference's call pattern).  usage: latency_probe.py dense|sampled [n]"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bench import make_inputs  # noqa: E402
from mlmapping_amd.config import S1, SDEF  # noqa: E402
from mlmapping_amd.mlmap import MLMap  # noqa: E402

what = sys.argv[1]
n = int(sys.argv[2]) if len(sys.argv) > 2 else 300
cfg = SDEF if "sdef" in sys.argv else S1
frames, q, t = make_inputs(cfg, 16, n + 16, 42)
m = MLMap(cfg, max_blocks=32768, max_points=cfg.width * cfg.height, max_batch=2)
z3 = np.zeros(3)
ts = []
for k in range(n + 16):
    if k == 16:
        m.debug_clocks()
    a = time.perf_counter()
    if what == "dense":
        m.update_map(frames[k % 16], q[k], t[k])
    else:
        m.depth_odom_callback(frames[k % 16], 0.0, t[k], q[k], z3, 0.0, z3, 0.0, 0.0, sampled=True)
    ts.append(time.perf_counter() - a)
ts = np.array(ts[16:]) * 1e6
st = m.frame_stats()
print(what, "median %.1f us, p10 %.1f, p90 %.1f" % (np.median(ts), np.percentile(ts, 10), np.percentile(ts, 90)), "graph launches", st["n_graph_launches"], "points", st["n_points"],
      "hits", st["n_hit_cells"], "miss", st["n_miss_cells"])
clk = m.debug_clocks() / n
print("host clocks, us per call: sample %.1f | set-up %.1f | launch call %.1f | to the wait %.1f | wait %.1f | rest %.1f  (sum %.1f)" % (*clk[:6], clk[:6].sum()))
