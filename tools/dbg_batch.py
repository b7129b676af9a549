"""Repro hunt: synchronous host batches of 2-8 frames on a handle with max_batch 4, compared with the oracle after every call."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from mlmapping_amd import synthetic as syn
from mlmapping_amd.config import SDEF
from mlmapping_amd.mlmap import MLMap
from oracle.binding import OracleMap
from tests.util import compare_maps

ops_allowed = sys.argv[1].split(",")
max_blocks = int(sys.argv[2])
wander = float(sys.argv[3])
for seed in range(1, 7):
    cfg = SDEF.with_(depth_noise_coe=0.00375, lm_occupied_sh=2.0)
    gpu, cpu = MLMap(cfg, max_blocks=max_blocks, max_points=cfg.width * cfg.height, max_batch=4), OracleMap(cfg)
    rng = np.random.default_rng(seed)
    base = syn.room_depth(cfg)
    traj = syn.smooth_trajectory(400, seed)
    k = 0
    def frame():
        global k
        img = syn.jitter_depth(base, k, seed=seed)
        q, t = traj[k]
        t = t + np.array([wander * k, -0.75 * wander * k, 0.0])
        k += 1
        return img, q, t
    bad = None
    for step in range(40):
        op = rng.choice(ops_allowed)
        if op == "dense":
            img, q, t = frame(); gpu.update_map(img, q, t); cpu.update_depth(img, q, t)
        elif op == "points":
            _, q, t = frame()
            pts = rng.uniform([-2, -1.5, 0.3], [2, 1.5, 5.0], size=(int(rng.integers(0, 400)), 3))
            gpu.update_map_points(pts, q, t); cpu.update_points(pts, q, t)
        elif op == "sampled":
            img, q, t = frame()
            pix = (rng.integers(0, cfg.height, 500) * cfg.width + rng.integers(0, cfg.width, 500)).astype(np.int32)
            gpu.update_map(img, q, t, pixel_idx=pix); cpu.update_depth_indexed(img, pix, q, t)
        else:
            fr = [frame() for _ in range(int(rng.integers(2, 9)))]
            gpu.update_map_batch(np.stack([f[0] for f in fr]), np.stack([f[1] for f in fr]), np.stack([f[2] for f in fr]))
            for img, q, t in fr: cpu.update_depth(img, q, t)
        try:
            compare_maps(gpu.export_blocks(), cpu.export_blocks(), "step")
        except AssertionError as e:
            bad = (step, op, k, str(e)[:80]); break
    st = gpu.frame_stats()
    print("seed", seed, "->", bad, {x: st[x] for x in ("n_pool_grows", "n_slot_grows", "n_spec_replays", "n_sector_fallbacks", "n_graph_launches")})
    gpu.close()
