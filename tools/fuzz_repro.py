"""Replay tests/test_gpu_parity.py::test_random_configurations for one seed (same draws: tests/util.py fuzz_trial) and report the
first trial and frame whose hit cells or odds differ from the oracle's, with the cells:  python tools/fuzz_repro.py SEED TRIALS [-v]
(MLM_KNOBS=name=value,... sets experiment knobs of the library: mlmapping_amd/mlmap.py)"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np

from mlmapping_amd import synthetic as syn
from mlmapping_amd.mlmap import MLMap
from oracle.binding import OracleMap
from tests.util import fuzz_trial

seed, trials = int(sys.argv[1]), int(sys.argv[2])
rng = np.random.default_rng(seed)
for trial in range(trials):
    cfg, depths, _ = fuzz_trial(rng, trial)
    if "-v" in sys.argv:
        print("trial", trial, "explore", cfg.use_exploration_frontiers, flush=True)
    gpu, cpu = MLMap(cfg, max_blocks=4096, max_points=320 * 240, record_awareness=True), OracleMap(cfg)
    for k, depth in enumerate(depths):
        q, t = syn.random_poses(3, seed=trial)[k]
        cpu.update_depth(depth, q, t)
        gpu.update_map(depth, q, t)
        gc, go, _ = gpu.awareness_hits()
        cc, co = cpu.hit_cells_sorted()
        if not np.array_equal(gc, cc) or not np.array_equal(go.view(np.uint32), co.view(np.uint32)):
            bad = np.nonzero(go.view(np.uint32) != co.view(np.uint32))[0] if gc.shape == cc.shape else []
            st = gpu.frame_stats()
            print("trial", trial, "frame", k, cfg)
            print("cells equal", np.array_equal(gc, cc), "n hits", gc.shape[0], "odds differing", len(bad),
                  {x: st[x] for x in ("n_multi_cells", "n_sector_fallbacks", "n_slot_grows", "n_device_atomics")})
            nR, nP = cfg.am_n_Rho, cfg.n_phi
            for b in bad[:12]:
                c = int(gc[b])
                z, r = divmod(c, nR * nP)
                ph, rho = divmod(r, nR)
                print("  cell rho", rho, "phi", ph, "z", z, "gpu", float(go[b]), hex(int(go.view(np.uint32)[b])), "cpu", float(co[b]), hex(int(co.view(np.uint32)[b])))
            sys.exit(1)
    gpu.close()
print("no difference in", trials, "trials")
