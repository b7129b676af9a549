"""Frames/s on depth images above 2^20 pixels (1024x1024, 1920x1080) into the 0.1 m map: asynchronous 16-frame batches from HBM."""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bench import make_inputs
from mlmapping_amd.config import S1
from mlmapping_amd.mlmap import MLMap
for w, h in ((1024, 1024), (1920, 1080)):
    cfg = S1.with_(width=w, height=h, cam_cx=w / 2.0, cam_cy=h / 2.0, cam_fx=0.6 * w, cam_fy=0.6 * w)
    B = 16
    frames, q, t = make_inputs(cfg, B, 40 * B, 42)
    d = torch.from_numpy(frames.view(np.int16)).cuda(); torch.cuda.synchronize()
    m = MLMap(cfg, max_blocks=32768, max_points=w * h, max_batch=B)
    m.set_async(True)
    for j in range(8): m.update_map_batch_dev(d.data_ptr(), B, w, h, q[j * B:(j + 1) * B], t[j * B:(j + 1) * B])
    m.sync()
    t0 = time.perf_counter()
    for j in range(8, 32): m.update_map_batch_dev(d.data_ptr(), B, w, h, q[j * B:(j + 1) * B], t[j * B:(j + 1) * B])
    m.sync()
    st = m.frame_stats()
    print(f"{w}x{h}: {24 * B / (time.perf_counter() - t0):.0f} frames/s", {k: st[k] for k in ("n_hit_cells", "n_miss_cells", "n_sector_fallbacks", "n_slot_grows")})
    m.close(); del d
