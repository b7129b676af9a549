"""Fuzz of the RECOVERY paths: the random map geometries of test_random_configurations (tests/util.py fuzz_trial) on handles whose
limits and knobs force what a camera stream rarely does — column tables too small for the scene (large-table pass, its arming and
the redo at drain), forced sector fall-backs every 2nd / 3rd frame, a pool of 16 blocks and lists sized for a quarter of the frame
(growth of pool and slots while frames are in flight), no graph — fed frame by frame, in batches, synchronously and asynchronously.
Every combination must leave the oracle's map, bit for bit.  (Fuzz seed 4242 found a rerun of Stage A counting a column's points
twice exactly where two of these paths met: test_gpu_slots.py::test_rerun_with_a_column_waiting_for_the_large_table.)
MLM_RFUZZ_SEED / MLM_RFUZZ_TRIALS: other or longer runs."""
import os

import numpy as np
import pytest

from mlmapping_amd import synthetic as syn
from tests.util import compare_maps, fuzz_trial

pytestmark = pytest.mark.gpu


def test_random_configurations_on_the_recovery_paths(knobs):
    _recovery_fuzz(knobs, int(os.environ.get("MLM_RFUZZ_SEED", "31")), int(os.environ.get("MLM_RFUZZ_TRIALS", "40")), "MLM_RFUZZ_SEED" not in os.environ)


def test_recovery_paths_fresh_seed(knobs):
    """the same with a seed derived from the kernel sources (a new one with every change of the code; printed, and in every message:
    MLM_RFUZZ_SEED=<seed> MLM_RFUZZ_TRIALS=12 replays it in the test above)"""
    from bench import csrc_sha16

    seed = int(csrc_sha16(), 16) % (1 << 31) + 1
    print("fresh recovery fuzz seed", seed)
    _recovery_fuzz(knobs, seed, 12, False)


def _recovery_fuzz(knobs, seed, trials, expect_all_paths, only_trial=None):
    from mlmapping_amd.mlmap import MLMap
    from oracle.binding import OracleMap

    rng = np.random.default_rng(seed)
    krng = np.random.default_rng(seed + 1)  # (limits, knobs and call pattern: a stream of its own, so the inputs are fuzz_trial's)
    srng = np.random.default_rng(seed + 2)  # (which frame-by-frame trials integrate pixel lists: small frames start without the prologue kernel)
    seen = {"n_sector_fallbacks": 0, "n_slot_grows": 0, "n_pool_grows": 0, "n_spec_replays": 0}
    for trial in range(trials):
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
        if only_trial is not None and trial != only_trial:  # (a replay of one trial: the others only take their draws)
            if not (pattern.startswith("batch") and max_batch >= 2):
                small = srng.random() < 0.3
                for k in range(6 if pattern == "twice" else 3):
                    if small and k != 1:
                        srng.integers(0, depths[0].size, int(srng.choice([300, 1500, 4000])))
            continue
        for name, v in kn.items():
            knobs.set(name, v)
        what = f"recovery fuzz seed {seed} trial {trial}: knobs {kn} max_blocks {max_blocks} max_points {max_points} max_batch {max_batch} {pattern} cfg {cfg}"
        gpu, cpu = MLMap(cfg, max_blocks=max_blocks, max_points=max_points, max_batch=max_batch), OracleMap(cfg)
        from mlmapping_amd import mlmap

        mlmap.debug_reset()  # (the knobs are read by mlm_create)
        poses = syn.random_poses(3, seed=trial)
        frames = [(depths[k], poses[k][0], poses[k][1]) for k in range(3)]
        if pattern == "twice":  # six frames: the slots come round again after whatever the first three left in them
            frames = frames + [(depths[2 - k], poses[k][0], poses[k][1] + np.array([0.3, -0.2, 0.1])) for k in range(3)]
        try:
            gpu.set_async(pattern.endswith("async"))
            if pattern.startswith("batch") and max_batch >= 2:
                for k0 in range(0, len(frames), max_batch):
                    fr = frames[k0:k0 + max_batch]
                    gpu.update_map_batch(np.stack([f[0] for f in fr]), np.stack([f[1] for f in fr]), np.stack([f[2] for f in fr]))
                for img, q, t in frames:
                    cpu.update_depth(img, q, t)
                compare_maps(gpu.export_blocks(), cpu.export_blocks(), what)
            else:
                small = srng.random() < 0.3
                for k, (img, q, t) in enumerate(frames):
                    if small and k != 1:  # (the middle frame stays dense: the slot's state changes hands both ways)
                        pix = srng.integers(0, img.size, int(srng.choice([300, 1500, 4000]))).astype(np.int32)
                        gpu.update_map(img, q, t, pixel_idx=pix)
                        cpu.update_depth_indexed(img, pix, q, t)
                    else:
                        gpu.update_map(img, q, t)
                        cpu.update_depth(img, q, t)
                    if not pattern.endswith("async") or k == len(frames) - 1:
                        compare_maps(gpu.export_blocks(), cpu.export_blocks(), what + f" frame {k} pixel lists {small}")
            if cfg.use_exploration_frontiers:
                assert np.array_equal(gpu.export_frontier(), cpu.export_frontier()), what + ": frontier"
            assert np.array_equal(gpu.getOccupancy(pos[:2000]), cpu.getOccupancy(pos[:2000])), what
        except AssertionError:
            raise
        except Exception as e:  # (an error return of the library: say which trial)
            raise AssertionError(what + f": {e}") from e
        st = gpu.frame_stats()
        for name in seen:
            seen[name] += int(st[name])
        gpu.close()
    print("recovery paths taken:", seen)
    if expect_all_paths:
        assert all(v > 0 for v in seen.values()), seen  # (the default run does reach every one of them)


def test_pool_fills_behind_a_frame_that_is_replayed(knobs):
    """Seed 6303, trial 276 of the fuzzer above (asynchronous frames, drained together): the second frame does not fit the emulated hit
    container without a rehash and is replayed with exact keys while the third, submitted behind it, has already found the block pool
    full — the sticky pool-full flag on the device must not fail the second frame's replay; the pool grows when the third frame's turn
    comes.  Reference: map_local.h:215-231 (allocate_ram never fails)."""
    _recovery_fuzz(knobs, 6303, 277, False, only_trial=276)
