"""Repeat the scenario of tests/test_gpu_small_frames.py::test_small_frames_between_everything_else in one process with fresh handles, the map
compared with the oracle after every frame:  python tools/small_frames_stress.py stage_a_gives_up[,plain,...] REPS   (MB=<max_blocks>, NOASYNC=1:
variations).  How the race of an asynchronous frame on the shared cell-table path with its own upload was found (HISTORY.md 12.5)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from mlmapping_amd import mlmap, synthetic as syn
from mlmapping_amd.config import S1
from mlmapping_amd.mlmap import MLMap
from oracle.binding import OracleMap
from tests.util import compare_maps

variants = sys.argv[1].split(",")
reps = int(sys.argv[2])
FRONTIER = bool(os.environ.get("FRONTIER"))  # use_exploration_frontiers: true (the frontier sets are compared as well)
cfg = S1.with_(use_exploration_frontiers=True) if FRONTIER else S1
frames = list(syn.stream(cfg, "room_jitter", "random", 30, seed=4))
def _pix(rng, n):
    return (rng.integers(0, cfg.height, n) * cfg.width + rng.integers(0, cfg.width, n)).astype(np.int32)
# the oracle's maps after every frame, once per variant-independent input (the inputs do not depend on the variant)
rng = np.random.default_rng(31)
ops = []
k = 0; n_small = 0
while k < len(frames):
    if k % 9 == 4: ops.append(("dense", k, None))
    elif k % 9 == 7 and k + 1 < len(frames): ops.append(("batch", k, None)); k += 1
    elif k % 9 == 2: ops.append(("async", k, _pix(rng, 700)))
    else: ops.append(("small", k, _pix(rng, (500, 3000, 4096, 1)[n_small % 4]))); n_small += 1
    k += 1
cpu = OracleMap(cfg)
ref = []
for kind, k, pix in ops:
    img, (q, t) = frames[k]
    if kind == "dense": cpu.update_depth(img, q, t)
    elif kind == "batch":
        cpu.update_depth(img, q, t); img2, (q2, t2) = frames[k + 1]; cpu.update_depth(img2, q2, t2)
    else: cpu.update_depth_indexed(img, pix, q, t)
    ref.append((cpu.export_blocks(), cpu.export_frontier() if FRONTIER else None))
fails = 0
for rep in range(reps):
    for variant in variants:
        mlmap.debug_reset()
        if variant == "stage_a_gives_up": mlmap.debug_set("sec_fail_every", 3)
        if variant == "no_graph": mlmap.debug_set("graph", 0)
        gpu = MLMap(cfg, max_blocks=16 if variant == "pool_grows" else int(os.environ.get("MB", "4096")), max_points=cfg.width * cfg.height, max_batch=2)
        mlmap.debug_reset()
        try:
            for j, (kind, k, pix) in enumerate(ops):
                img, (q, t) = frames[k]
                if kind == "dense": gpu.update_map(img, q, t)
                elif kind == "batch":
                    img2, (q2, t2) = frames[k + 1]
                    gpu.update_map_batch(np.stack([img, img2]), np.stack([q, q2]), np.stack([t, t2]))
                elif kind == "async":
                    if os.environ.get("NOASYNC"): gpu.update_map(img, q, t, pixel_idx=pix)
                    else:
                        gpu.set_async(True); gpu.update_map(img, q, t, pixel_idx=pix); gpu.sync(); gpu.set_async(False)
                else: gpu.update_map(img, q, t, pixel_idx=pix)
                st_before = gpu.frame_stats()
                compare_maps(gpu.export_blocks(), ref[j][0], f"rep {rep} {variant} op {j} {kind} frame {k}")
                if FRONTIER: assert np.array_equal(gpu.export_frontier(), ref[j][1]), f"rep {rep} {variant} op {j} {kind} frame {k}: frontier sets differ"
                if rep == 0 and j >= 24: print("ok op", j, kind, {x: st_before[x] for x in ("n_sector_fallbacks", "n_spec_replays", "n_graph_launches", "n_pool_grows", "n_blocks", "block_capacity")}, flush=True)
        except AssertionError as e:
            fails += 1
            try:
                compare_maps(gpu.export_blocks(), ref[j - 1][0], "vs the map BEFORE this frame")
                print("  -> equal to the map before this frame: the frame was not applied", flush=True)
            except AssertionError as e2:
                print("  -> also differs from the map before:", str(e2)[:160], flush=True)
            print("  stats now:", gpu.frame_stats(), flush=True)
            print("FAIL", str(e)[:300], {x: st_before[x] for x in ("n_sector_fallbacks", "n_spec_replays", "n_graph_launches", "n_pool_grows", "n_hit_cells", "n_miss_cells", "n_blocks")}, flush=True)
        gpu.close()
print("failures:", fails, "of", reps * len(variants))
