"""Small frames that start without the prologue kernel (a slot whose last frame handed its counters back and left them clear:
k_bin_sectors_hostf, mlm_hand_back, submit_single_graph / launch_stage_a_sector) between everything that must withdraw that state:
dense frames, batches, frames whose Stage A gives up and is replayed, a pool that has to grow, asynchronous stretches — default and
frontier mode, against the oracle after every frame.  Reference: src/mlmap.cpp:463-507 (one small frame per callback),
src/map_awareness.cpp:173-282, src/map_local.cpp:143-237."""
import ctypes

import numpy as np
import pytest

from mlmapping_amd import synthetic as syn
from mlmapping_amd.config import CONFIG2_YAML, S1, SDEF
from tests.util import compare_maps

pytestmark = pytest.mark.gpu


def _pix(rng, cfg, n):
    return (rng.integers(0, cfg.height, n) * cfg.width + rng.integers(0, cfg.width, n)).astype(np.int32)


@pytest.mark.parametrize("frontier", [False, True], ids=["default", "frontier"])
@pytest.mark.parametrize("variant", ["plain", "stage_a_gives_up", "pool_grows", "no_graph"])
def test_small_frames_between_everything_else(knobs, variant, frontier):
    from mlmapping_amd.mlmap import MLMap
    from oracle.binding import OracleMap

    if variant == "stage_a_gives_up":
        knobs.set("sec_fail_every", "3")
    if variant == "no_graph":
        knobs.set("graph", "0")
    cfg = S1.with_(use_exploration_frontiers=True) if frontier else S1
    gpu = MLMap(cfg, max_blocks=16 if variant == "pool_grows" else 4096, max_points=cfg.width * cfg.height, max_batch=2)
    cpu = OracleMap(cfg)
    rng = np.random.default_rng(31)
    frames = list(syn.stream(cfg, "room_jitter", "random", 30, seed=4))
    k = 0
    n_small = 0
    while k < len(frames):
        img, (q, t) = frames[k]
        what = f"{variant}, frontier {frontier}, frame {k}"
        if k % 9 == 4:  # a dense frame (its graph keeps the prologue)
            gpu.update_map(img, q, t)
            cpu.update_depth(img, q, t)
        elif k % 9 == 7 and k + 1 < len(frames):  # a batch of two on the same slot set
            img2, (q2, t2) = frames[k + 1]
            gpu.update_map_batch(np.stack([img, img2]), np.stack([q, q2]), np.stack([t, t2]))
            cpu.update_depth(img, q, t)
            cpu.update_depth(img2, q2, t2)
            k += 1
        elif k % 9 == 2:  # an asynchronous small frame, drained by the export below
            gpu.set_async(True)
            pix = _pix(rng, cfg, 700)
            gpu.update_map(img, q, t, pixel_idx=pix)
            cpu.update_depth_indexed(img, pix, q, t)
            gpu.sync()
            gpu.set_async(False)
        else:  # small frames: 500 samples (2 strips), 3 000 (12 strips), 4 096 (16: the largest that goes without the prologue)
            pix = _pix(rng, cfg, (500, 3000, 4096, 1)[n_small % 4])
            gpu.update_map(img, q, t, pixel_idx=pix)
            cpu.update_depth_indexed(img, pix, q, t)
            n_small += 1
        compare_maps(gpu.export_blocks(), cpu.export_blocks(), what)
        if frontier:
            assert np.array_equal(gpu.export_frontier(), cpu.export_frontier()), what
        k += 1
    st = gpu.frame_stats()
    if variant == "stage_a_gives_up":
        assert st["n_sector_fallbacks"] > 0, st
    if variant == "pool_grows":
        assert st["n_pool_grows"] > 0, st
    gpu.close()


def test_callback_stream_config2_yaml_with_inflation(knobs):
    """The shipped config2.yaml callback (frontier mode, 500 rand() samples, inflate_map every third call) for 40 calls: from the second
    call on every frame starts without the prologue kernel; map, frontier and inflated occupancy equal the oracle's."""
    from mlmapping_amd.mlmap import MLMap
    from oracle.binding import OracleMap

    libc = ctypes.CDLL("libc.so.6")
    cfg = CONFIG2_YAML
    gpu, cpu = MLMap(cfg, max_blocks=256, max_points=cfg.width * cfg.height, max_batch=2), OracleMap(cfg)
    base = syn.room_depth(cfg)
    n = 40
    traj = syn.smooth_trajectory(n, 5)
    z = np.zeros(3)
    for k in range(n):
        depth = syn.jitter_depth(base, k, seed=3).astype(np.float32) / 1000.0
        q, t = traj[k]
        libc.srand(500 + k)
        tg = gpu.depth_odom_callback(depth, 0.0, t, q, z, 0.0, z, 0.0, cfg.camera2odom_latency, sampled=True)
        libc.srand(500 + k)
        tc = cpu.depth_odom_callback(depth, 0.0, t, q, z, 0.0, z, 0.0, cfg.camera2odom_latency, sampled=True)
        assert np.array_equal(tg, tc)
        if k % 3 == 2:
            gpu.inflate_map(tg[4:])
            cpu.inflate_map(tc[4:])
        if k % 5 == 4 or k == n - 1:
            compare_maps(gpu.export_blocks(), cpu.export_blocks(), f"config2.yaml callback {k}")
            assert np.array_equal(gpu.export_frontier(), cpu.export_frontier()), k
    gpu.close()


def test_default_callback_stream_sdef(knobs):
    """The reference-default map (config_sim.yaml geometry) through the sampled callback for 60 calls, a dense call every tenth."""
    from mlmapping_amd.mlmap import MLMap
    from oracle.binding import OracleMap

    libc = ctypes.CDLL("libc.so.6")
    cfg = SDEF
    gpu, cpu = MLMap(cfg, max_blocks=2048, max_batch=2), OracleMap(cfg)
    z = np.zeros(3)
    for k, (img, (q, t)) in enumerate(syn.stream(cfg, "room_jitter", "smooth", 60, seed=8)):
        libc.srand(900 + k)
        gpu.depth_odom_callback(img, 0.0, t, q, z, 0.0, z, 0.0, 0.0, sampled=(k % 10 != 9))
        libc.srand(900 + k)
        cpu.depth_odom_callback(img, 0.0, t, q, z, 0.0, z, 0.0, 0.0, sampled=(k % 10 != 9))
        if k % 6 == 5 or k == 59:
            compare_maps(gpu.export_blocks(), cpu.export_blocks(), f"callback {k}")
    st = gpu.frame_stats()
    assert st["n_graph_launches"] >= 55, st
    gpu.close()



def test_launches_of_a_lone_frontier_callback(knobs):
    """What a synchronous frontier-mode callback enqueues once the stream is under way (per-call kernel timing lists the launches):
    Stage A without a prologue kernel (the previous frame left the slot's counters clear), the bucket-first pass inside k_rank (no
    k_ex_order_min), the map-dependent launches — k_ex_apply_misses hands the counters back, the release scan runs behind the ticket —:
    nine kernels, no copies between them.
    Reference call: src/mlmap.cpp:463-507."""
    from mlmapping_amd.mlmap import MLMap

    cfg = CONFIG2_YAML
    gpu = MLMap(cfg, max_blocks=1024, max_points=cfg.width * cfg.height, max_batch=2)
    base = syn.room_depth(cfg)
    traj = syn.smooth_trajectory(24, 5)
    z = np.zeros(3)
    gpu.enable_kernel_timing(1)
    names = None
    for k in range(24):
        depth = syn.jitter_depth(base, k, seed=3).astype(np.float32) / 1000.0
        q, t = traj[k]
        gpu.depth_odom_callback(depth, 0.0, t, q, z, 0.0, z, 0.0, cfg.camera2odom_latency, sampled=True)
        names = [n for n, _ in gpu.kernel_times()]
    assert names == ["k_bin_sectors", "k_sector", "k_rank", "k_ex_order_keys", "k_ex_register", "k_apply", "k_ex_observe", "k_ex_apply_misses",
                     "k_ex_release"], names
    assert gpu.frame_stats()["n_spec_replays"] <= 6  # (the stream's first frames, while the emulated containers grow)
    gpu.close()


def test_async_frame_on_the_shared_cell_table_path_waits_for_its_inputs(knobs):
    """An asynchronous call while the handle backs off from the sector path (a frame's Stage A gave up shortly before) runs its Stage A
    on the MAIN stream (the cell-table path's per-frame state exists once), while its pixel list went up on the slot set's Stage A
    stream: the main stream has to wait for it.  Without that wait the list was read before it had arrived in one run in twenty-five of
    the scenario of test_small_frames_between_everything_else[stage_a_gives_up] (a third of the runs with the round-5 library, late in
    a process) — the scenario again, on thirty fresh handles, the map compared after its asynchronous frames.
    Reference: the callback hands over a complete frame, src/mlmap.cpp:463-507."""
    from mlmapping_amd import mlmap
    from mlmapping_amd.mlmap import MLMap
    from oracle.binding import OracleMap

    cfg = S1
    frames = list(syn.stream(cfg, "room_jitter", "random", 30, seed=4))
    rng = np.random.default_rng(31)
    ops, k, n_small = [], 0, 0
    while k < len(frames):
        if k % 9 == 4:
            ops.append(("dense", k, None))
        elif k % 9 == 7 and k + 1 < len(frames):
            ops.append(("batch", k, None))
            k += 1
        elif k % 9 == 2:
            ops.append(("async", k, _pix(rng, cfg, 700)))
        else:
            ops.append(("small", k, _pix(rng, cfg, (500, 3000, 4096, 1)[n_small % 4])))
            n_small += 1
        k += 1
    cpu, want = OracleMap(cfg), []
    for kind, k, pix in ops:
        img, (q, t) = frames[k]
        if kind == "dense":
            cpu.update_depth(img, q, t)
        elif kind == "batch":
            cpu.update_depth(img, q, t)
            cpu.update_depth(frames[k + 1][0], *frames[k + 1][1])
        else:
            cpu.update_depth_indexed(img, pix, q, t)
        want.append(cpu.export_blocks() if kind == "async" else None)
    for rep in range(30):
        knobs.set("sec_fail_every", "3")
        gpu = MLMap(cfg, max_blocks=4096, max_points=cfg.width * cfg.height, max_batch=2)
        mlmap.debug_reset()
        for j, (kind, k, pix) in enumerate(ops):
            img, (q, t) = frames[k]
            if kind == "dense":
                gpu.update_map(img, q, t)
            elif kind == "batch":
                img2, (q2, t2) = frames[k + 1]
                gpu.update_map_batch(np.stack([img, img2]), np.stack([q, q2]), np.stack([t, t2]))
            elif kind == "async":
                gpu.set_async(True)
                gpu.update_map(img, q, t, pixel_idx=pix)
                gpu.sync()
                gpu.set_async(False)
                compare_maps(gpu.export_blocks(), want[j], f"fresh handle {rep}, asynchronous frame {k}")
            else:
                gpu.update_map(img, q, t, pixel_idx=pix)
        assert gpu.frame_stats()["n_sector_fallbacks"] > 0
        gpu.close()
