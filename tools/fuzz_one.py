"""Debug helper: replay one trial of tests/test_gpu_parity.py::test_random_configurations with extra diagnostics.
usage: python tools/fuzz_one.py TRIAL [max_blocks]"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mlmapping_amd import synthetic as syn  # noqa: E402
from mlmapping_amd.config import S1  # noqa: E402
from mlmapping_amd.mlmap import MLMap  # noqa: E402
from oracle.binding import OracleMap  # noqa: E402

want = int(sys.argv[1])
max_blocks = int(sys.argv[2]) if len(sys.argv) > 2 else 4096
rng = np.random.default_rng(2024)
for trial in range(want + 1):
    d = float(rng.choice([0.05, 0.1, 0.15, 0.2, 0.25]))
    cfg = S1.with_(
        am_d_Rho=d, am_d_Phi_deg=float(rng.choice([0.5, 1.0, 2.0, 3.0, 5.0])), am_d_Z=float(rng.choice([d, 2 * d, 0.5 * d])),
        am_n_Rho=int(rng.integers(20, 100)), am_n_Z_below=int(rng.integers(5, 30)), am_n_Z_over=int(rng.integers(5, 30)),
        depth_noise_coe=float(rng.choice([1e-6, 0.001, 0.00375, 0.008])),
        subbox_d_xyz=float(rng.choice([d, 2 * d, 0.5 * d])), subbox_n=int(rng.choice([4, 5, 8, 10, 16])),
        lm_log_odds_min=float(rng.uniform(-3, -1)), lm_log_odds_max=float(rng.uniform(3, 5)),
        lm_measurement_miss=float(rng.uniform(-1.2, -0.3)), lm_occupied_sh=float(rng.uniform(1.0, 3.0)),
        use_exploration_frontiers=bool(trial % 3 == 2),
        cam_fx=float(rng.uniform(150, 400)), cam_fy=float(rng.uniform(150, 400)), cam_cx=163.3, cam_cy=117.9,
        width=320, height=240)
    if 3 * cfg.depth_noise_coe * (cfg.am_n_Rho * cfg.am_d_Rho) ** 2 / cfg.am_d_Rho > 10:
        cfg = cfg.with_(depth_noise_coe=1e-6)
    frames = []
    for k in range(3):
        depth = rng.integers(300, int(1000 * cfg.am_n_Rho * cfg.am_d_Rho * 1.3), size=(240, 320)).astype(np.uint16)
        depth[rng.random((240, 320)) < 0.02] = 0
        if k == 1:
            depth[:] = (1000 * 0.6 * cfg.am_n_Rho * cfg.am_d_Rho + 200 * np.sin(np.arange(320) / 25.0)[None, :]).astype(np.uint16)
        frames.append(depth)
    if trial != want:
        if not cfg.use_exploration_frontiers:
            rng.uniform(-8, 8, size=(20000, 3))
        else:
            rng.uniform(-8, 8, size=(20000, 3))
        continue
    print(cfg)
    gpu, cpu = MLMap(cfg, max_blocks=max_blocks, max_points=320 * 240, record_awareness=True), OracleMap(cfg)
    for k in range(3):
        q, t = syn.random_poses(3, seed=trial)[k]
        cpu.update_depth(frames[k], q, t)
        gpu.update_map(frames[k], q, t)
        g, c = gpu.export_blocks(), cpu.export_blocks()
        st = gpu.frame_stats()
        print("frame", k, {x: st[x] for x in ("n_hit_cells", "n_miss_cells", "n_blocks", "n_rehash_epochs", "n_spec_replays", "n_sector_fallbacks", "n_pool_grows", "block_capacity")})
        print("  blocks", g["keys"].shape, c["keys"].shape, "keys equal", g["keys"].shape == c["keys"].shape and np.array_equal(g["keys"], c["keys"]))
        if g["keys"].shape == c["keys"].shape and np.array_equal(g["keys"], c["keys"]):
            bad = g["occ"] != c["occ"]
            print("  occ differ", int(bad.sum()), "lo bits differ", int((g["log_odds"].view(np.uint32) != c["log_odds"].view(np.uint32)).sum()))
            if bad.any():
                rows = np.unique(np.nonzero(bad)[0])
                print("  blocks with differences:", rows.size, "first keys", g["keys"][rows[:5]].tolist())
                r0 = rows[0]
                cols = np.nonzero(bad[r0])[0][:8]
                print("  sample cells", cols.tolist(), "gpu L", g["log_odds"][r0, cols].tolist(), "cpu L", c["log_odds"][r0, cols].tolist(),
                      "gpu occ", g["occ"][r0, cols].tolist(), "cpu occ", c["occ"][r0, cols].tolist())
