"""Which frames take the sector path / fall back, and kernel times per frame (diagnostic)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from mlmapping_amd import synthetic as syn
from mlmapping_amd.config import S1, S3, SDEF
from mlmapping_amd.mlmap import MLMap

for name, cfg, scene in (("S1 room_jitter", S1, "room_jitter"), ("S1 scatter", S1, "scatter"), ("S1 corridor", S1, "corridor"),
                         ("S3 room_jitter", S3, "room_jitter"), ("SDEF room", SDEF, "room_jitter")):
    n = 8
    frames = np.stack([img for img, _ in syn.stream(cfg, scene, "random", n)])
    poses = syn.random_poses(n, 42)
    q = np.stack([p[0] for p in poses]); t = np.stack([p[1] for p in poses])
    m = MLMap(cfg, max_blocks=32768, max_points=cfg.width * cfg.height, max_batch=8)
    m.update_map_batch(frames, q, t)
    m.enable_kernel_timing(2)
    m.update_map_batch(frames, q, t)
    kt = {}
    for k, ms in m.kernel_times():
        kt[k] = kt.get(k, 0.0) + ms
    st = m.frame_stats()
    print(name, {k: st[k] for k in ("n_hit_cells", "n_miss_cells", "n_multi_cells", "n_groups", "n_rays", "n_spec_replays", "n_sector_fallbacks")})
    print("   us/frame:", {k: round(v * 1e3 / n, 2) for k, v in kt.items()})
    m.close()
