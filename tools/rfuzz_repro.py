"""Replay ONE trial of tests/test_gpu_fuzz_recovery.py (same draws):  python tools/rfuzz_repro.py SEED TRIAL [sync]
(MLM_DEBUG_DRAIN=1 prints the drain's decisions)."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np

from mlmapping_amd import mlmap, synthetic as syn
from mlmapping_amd.mlmap import MLMap
from oracle.binding import OracleMap
from tests.util import compare_maps, fuzz_trial

seed, target = int(sys.argv[1]), int(sys.argv[2])
rng, krng, srng = np.random.default_rng(seed), np.random.default_rng(seed + 1), np.random.default_rng(seed + 2)
for trial in range(target + 1):
    cfg, depths, pos = fuzz_trial(rng, trial)
    kn = {}
    if krng.random() < 0.5:
        kn["sec_tab"] = int(krng.choice([256, 512]))
    if krng.random() < 0.3:
        kn["sec_fail_every"] = int(krng.choice([2, 3]))
    if krng.random() < 0.2:
        kn["sec_tab_big"] = 0
    if krng.random() < 0.3:
        kn["big_arm"] = int(krng.choice([0, 1, 2]))
    if krng.random() < 0.2:
        kn["graph"] = 0
    if krng.random() < 0.3:
        kn["tile_sh"] = int(krng.choice([1, 2, 3]))
    if krng.random() < 0.2:
        kn["slot_sets"] = 2
    max_blocks = int(krng.choice([16, 64, 4096]))
    max_points = int(krng.choice([320 * 240, 320 * 240, 4 * 320 * 240]))
    max_batch = int(krng.choice([1, 2, 3, 4]))
    pattern = str(krng.choice(["single", "single_async", "batch", "batch_async", "twice"]))
    poses = syn.random_poses(3, seed=trial)
    frames = [(depths[k], poses[k][0], poses[k][1]) for k in range(3)]
    if pattern == "twice":
        frames = frames + [(depths[2 - k], poses[k][0], poses[k][1] + np.array([0.3, -0.2, 0.1])) for k in range(3)]
    batch = pattern.startswith("batch") and max_batch >= 2
    small = False if batch else srng.random() < 0.3
    lists = []
    for k, (img, q, t) in enumerate(frames):
        lists.append(srng.integers(0, img.size, int(srng.choice([300, 1500, 4000]))).astype(np.int32) if (small and k != 1 and not batch) else None)
    if trial != target:
        continue
    print("trial", trial, kn, "max_blocks", max_blocks, "max_points", max_points, "max_batch", max_batch, pattern, "small", small, flush=True)
    for name, v in kn.items():
        mlmap.debug_set(name, v)
    gpu, cpu = MLMap(cfg, max_blocks=max_blocks, max_points=max_points, max_batch=max_batch), OracleMap(cfg)
    mlmap.debug_reset()
    gpu.set_async(pattern.endswith("async") and "sync" not in sys.argv)
    for k, (img, q, t) in enumerate(frames):
        if lists[k] is not None:
            gpu.update_map(img, q, t, pixel_idx=lists[k])
            cpu.update_depth_indexed(img, lists[k], q, t)
        else:
            gpu.update_map(img, q, t)
            cpu.update_depth(img, q, t)
        print("frame", k, "submitted; oracle blocks", cpu.export_blocks()[0].shape[0] if isinstance(cpu.export_blocks(), tuple) else "?", flush=True)
        if "each" in sys.argv:
            compare_maps(gpu.export_blocks(), cpu.export_blocks(), f"frame {k}")
            print("  equal; stats", {x: gpu.frame_stats()[x] for x in ("n_blocks", "n_pool_grows", "n_sector_fallbacks", "n_spec_replays", "block_capacity")}, flush=True)
    compare_maps(gpu.export_blocks(), cpu.export_blocks(), "end")
    print("equal at the end", gpu.frame_stats())
