"""Random sequences of every map-changing and map-reading entry point on one handle — dense / sampled / point-list frames, asynchronous
batches, setFree_map_in_bound, inflate_map, single-position and bulk queries of every kind, sync, mode switches — mirrored on the
oracle; answers are compared where they are asked and the maps at the end.  What this guards: the bookkeeping BETWEEN the calls (the
host mirror's dirty boxes, drains, frontier mode's deferred tail, slot sets, growth of pool and slots)."""
import numpy as np
import pytest

from mlmapping_amd import synthetic as syn
from mlmapping_amd.config import SDEF
from tests.util import ODDS_TOL, compare_maps, voxel_centres

import os

pytestmark = pytest.mark.gpu
EXTRA = [int(x) for x in os.environ.get("MLM_STRESS_SEEDS", "").split(",") if x]  # more seeds for a longer soak: MLM_STRESS_SEEDS=21,22,...


@pytest.fixture(scope="module")
def mods():
    from mlmapping_amd.mlmap import MLMap
    from oracle.binding import OracleMap

    return MLMap, OracleMap


@pytest.mark.parametrize("explore", [False, True])
@pytest.mark.parametrize("seed", [1, 2, 3] + EXTRA)
def test_random_operation_sequences(mods, explore, seed):
    MLMap, OracleMap = mods
    cfg = SDEF.with_(depth_noise_coe=0.00375, lm_occupied_sh=2.0, use_exploration_frontiers=explore)
    gpu, cpu = MLMap(cfg, max_blocks=256, max_points=cfg.width * cfg.height, max_batch=4), OracleMap(cfg)
    rng = np.random.default_rng(100 * seed + int(explore))
    base = syn.room_depth(cfg)
    traj = syn.smooth_trajectory(400, seed)
    k = 0  # frames integrated so far
    is_async = False
    log = []

    def frame():
        nonlocal k
        img = syn.jitter_depth(base, k, seed=seed)
        q, t = traj[k]
        t = t + np.array([0.02 * k, -0.015 * k, 0.0])  # (the camera wanders: new blocks keep appearing)
        k += 1
        return img, q, t

    def positions(n):
        b = cpu.export_blocks()
        if b["keys"].shape[0] == 0:
            return rng.uniform(-3, 6, size=(n, 3))
        lo, hi = b["keys"].min(0) * cfg.subbox_d_xyz * cfg.subbox_n - 1.0, (b["keys"].max(0) + 1) * cfg.subbox_d_xyz * cfg.subbox_n + 1.0
        return np.concatenate([rng.uniform(lo, hi, size=(n // 2, 3)), voxel_centres(b, cfg, n - n // 2, seed=int(rng.integers(1 << 30)))])

    for step in range(70):
        op = rng.choice(["dense", "sampled", "points", "batch", "setfree", "inflate", "q1", "qbulk", "sync", "mode"],
                        p=[0.16, 0.12, 0.06, 0.14, 0.06, 0.08, 0.2, 0.08, 0.05, 0.05])
        log.append(op)
        if op == "dense":
            img, q, t = frame()
            gpu.update_map(img, q, t)
            cpu.update_depth(img, q, t)
        elif op == "sampled":
            img, q, t = frame()
            pix = (rng.integers(0, cfg.height, 500) * cfg.width + rng.integers(0, cfg.width, 500)).astype(np.int32)
            gpu.update_map(img, q, t, pixel_idx=pix)
            cpu.update_depth_indexed(img, pix, q, t)
        elif op == "points":
            _, q, t = frame()
            pts = rng.uniform([-2, -1.5, 0.3], [2, 1.5, 5.0], size=(int(rng.integers(0, 400)), 3))
            gpu.update_map_points(pts, q, t)
            cpu.update_points(pts, q, t)
        elif op == "batch":
            fr = [frame() for _ in range(int(rng.integers(2, 9)))]
            gpu.update_map_batch(np.stack([f[0] for f in fr]), np.stack([f[1] for f in fr]), np.stack([f[2] for f in fr]))
            for img, q, t in fr:
                cpu.update_depth(img, q, t)
        elif op == "setfree":
            c = positions(2)[0]
            lo_, hi_ = c - rng.uniform(0.1, 0.8, 3), c + rng.uniform(0.1, 0.8, 3)
            gpu.setFree_map_in_bound(lo_, hi_)
            cpu.setFree_map_in_bound(lo_, hi_)
        elif op == "inflate":
            c = traj[max(k - 1, 0)][1]
            gpu.inflate_map(c)
            cpu.inflate_map(c)
        elif op == "q1":  # a planner sampling positions one by one, every query kind
            pos = positions(12)
            for i in range(pos.shape[0]):
                p = pos[i:i + 1]
                kind = int(rng.integers(0, 5))
                if kind == 0:
                    assert gpu.getOccupancy(p)[0] == cpu.getOccupancy(p)[0], (step, log)
                elif kind == 1:
                    assert gpu.getOccupancy(p, inflate=0.25)[0] == cpu.getOccupancy(p, inflate=0.25)[0], (step, log)
                elif kind == 2:
                    assert gpu.getInflateOccupancy(p)[0] == cpu.getInflateOccupancy(p)[0], (step, log)
                elif kind == 3:
                    assert gpu.getOdd(p).view(np.uint32)[0] == cpu.getOdd(p).view(np.uint32)[0], (step, log)
                else:
                    assert np.array_equal(gpu.getOddGrad(p, 4), cpu.getOddGrad(p, 4)), (step, log)
        elif op == "qbulk":
            pos = positions(3000)
            assert np.array_equal(gpu.getOccupancy(pos), cpu.getOccupancy(pos)), (step, log)
            assert np.array_equal(gpu.getInflateOccupancy(pos), cpu.getInflateOccupancy(pos)), (step, log)
            assert np.abs(gpu.getOdd(pos) - cpu.getOdd(pos)).max() <= ODDS_TOL, (step, log)
        elif op == "sync":
            gpu.sync()
        elif op == "mode":
            is_async = not is_async
            gpu.set_async(is_async)
    compare_maps(gpu.export_blocks(), cpu.export_blocks(), f"after {log}")
    if explore:
        gf, cf = gpu.export_frontier(), cpu.export_frontier()
        assert gf.shape == cf.shape and np.array_equal(gf, cf)
    pos = positions(400)
    assert np.array_equal(np.concatenate([gpu.getOccupancy(pos[i:i + 1]) for i in range(400)]), cpu.getOccupancy(pos))
    st = gpu.frame_stats()
    assert st["n_host_queries"] > 0 and st["n_pool_grows"] >= 1, st


@pytest.mark.parametrize("seed", [11, 12, 13, 14] + EXTRA)
def test_random_operation_sequences_s1_with_growth(mods, seed):
    """The same on the 0.1 m map of configs 1/2 from a pool of 64 blocks and frame slots sized for camera frames, with the ROS-free
    callback (float depth, rand() sampler), device-resident batches, empty frames and — now and then — a scatter frame that
    overruns the columns' cell tables and the slots' lists (large-table pass, slot growth) in the middle of everything else."""
    import ctypes

    import torch

    MLMap, OracleMap = mods
    libc = ctypes.CDLL("libc.so.6")
    from mlmapping_amd.config import S1

    cfg = S1
    gpu, cpu = MLMap(cfg, max_blocks=64, max_points=cfg.width * cfg.height, max_batch=4), OracleMap(cfg)
    rng = np.random.default_rng(seed)
    base = syn.room_depth(cfg)
    sc = syn.ScatterScene(cfg, seed=seed)
    traj = syn.random_poses(300, seed)
    k = 0
    is_async = False
    log = []

    def frame(kind="room"):
        nonlocal k
        if kind == "scatter":
            img = sc.next()
        elif kind == "empty":
            img = np.zeros_like(base)
        else:
            img = syn.jitter_depth(base, k, seed=seed)
        q, t = traj[k]
        k += 1
        return img, q, t

    def positions(n):
        b = cpu.export_blocks()
        if b["keys"].shape[0] == 0:
            return rng.uniform(-3, 6, size=(n, 3))
        lo, hi = b["keys"].min(0) * 1.0 - 1.0, b["keys"].max(0) * 1.0 + 2.0
        return np.concatenate([rng.uniform(lo, hi, size=(n // 2, 3)), voxel_centres(b, cfg, n - n // 2, seed=int(rng.integers(1 << 30)))])

    zero3 = np.zeros(3)
    for step in range(45):
        op = rng.choice(["dense", "scatter", "empty", "batch", "batch_dev", "callback_s", "callback_d", "setfree", "inflate", "q1", "qbulk", "sync", "mode"],
                        p=[0.12, 0.06, 0.03, 0.12, 0.1, 0.1, 0.05, 0.05, 0.06, 0.17, 0.06, 0.04, 0.04])
        log.append(str(op))
        if op in ("dense", "scatter", "empty"):
            img, q, t = frame("room" if op == "dense" else op)
            gpu.update_map(img, q, t)
            cpu.update_depth(img, q, t)
        elif op == "batch":
            fr = [frame("scatter" if rng.random() < 0.15 else "room") for _ in range(int(rng.integers(2, 10)))]
            gpu.update_map_batch(np.stack([f[0] for f in fr]), np.stack([f[1] for f in fr]), np.stack([f[2] for f in fr]))
            for img, q, t in fr:
                cpu.update_depth(img, q, t)
        elif op == "batch_dev":
            fr = [frame() for _ in range(int(rng.integers(1, 7)))]
            d = torch.from_numpy(np.stack([f[0] for f in fr]).view(np.int16)).cuda()
            torch.cuda.synchronize()
            gpu.update_map_batch_dev(d.data_ptr(), len(fr), cfg.width, cfg.height, np.stack([f[1] for f in fr]), np.stack([f[2] for f in fr]))
            gpu.sync()  # (device inputs must stay alive until the frames are through)
            del d
            for img, q, t in fr:
                cpu.update_depth(img, q, t)
        elif op in ("callback_s", "callback_d"):
            img, q, t = frame()
            depth = img.astype(np.float32) / 1000.0
            args = dict(t_img=1.0 + k / 30.0, odom_p=t, odom_q=q, odom_v=[0.2, -0.1, 0.0], t_odom=1.0 + k / 30.0 - 0.003, imu_w=[0.01, 0.1, -0.2],
                        t_imu=1.0 + k / 30.0 - 0.001, latency=0.02, sampled=op == "callback_s")
            libc.srand(1000 + k)
            tg = gpu.depth_odom_callback(depth, **args)
            libc.srand(1000 + k)
            tc = cpu.depth_odom_callback(depth, **args)
            assert np.array_equal(tg, tc)
        elif op == "setfree":
            c = positions(2)[0]
            lo_, hi_ = c - rng.uniform(0.1, 0.6, 3), c + rng.uniform(0.1, 0.6, 3)
            gpu.setFree_map_in_bound(lo_, hi_)
            cpu.setFree_map_in_bound(lo_, hi_)
        elif op == "inflate":
            c = traj[max(k - 1, 0)][1]
            gpu.inflate_map(c)
            cpu.inflate_map(c)
        elif op == "q1":
            pos = positions(10)
            for i in range(pos.shape[0]):
                p = pos[i:i + 1]
                kind = int(rng.integers(0, 4))
                if kind == 0:
                    assert gpu.getOccupancy(p)[0] == cpu.getOccupancy(p)[0], (step, log)
                elif kind == 1:
                    assert gpu.getInflateOccupancy(p)[0] == cpu.getInflateOccupancy(p)[0], (step, log)
                elif kind == 2:
                    assert gpu.getOdd(p).view(np.uint32)[0] == cpu.getOdd(p).view(np.uint32)[0], (step, log)
                else:
                    assert np.array_equal(gpu.getOddGrad(p, 3), cpu.getOddGrad(p, 3)), (step, log)
        elif op == "qbulk":
            pos = positions(4000)
            assert np.array_equal(gpu.getOccupancy(pos), cpu.getOccupancy(pos)), (step, log)
            assert np.abs(gpu.getOdd(pos) - cpu.getOdd(pos)).max() <= ODDS_TOL, (step, log)
        elif op == "sync":
            gpu.sync()
        elif op == "mode":
            is_async = not is_async
            gpu.set_async(is_async)
    compare_maps(gpu.export_blocks(), cpu.export_blocks(), f"after {log}")
    st = gpu.frame_stats()
    print(st)
    assert st["n_pool_grows"] >= 1 and st["n_sector_fallbacks"] == 0, (st, log)


def test_two_handles_driven_from_two_threads(mods):
    """Two maps on one GPU, each driven by its own thread — frames, single-position and bulk queries, setFree, inflation — at the same
    time: nothing in the library is shared between handles except the device; both maps must equal their oracles."""
    import threading

    MLMap, OracleMap = mods
    cfgs = [SDEF.with_(depth_noise_coe=0.00375, lm_occupied_sh=2.0), SDEF.with_(use_exploration_frontiers=True, lm_occupied_sh=2.0, depth_noise_coe=0.00375)]
    errors = []

    def drive(i):
        try:
            cfg = cfgs[i]
            gpu, cpu = MLMap(cfg, max_blocks=256, max_points=cfg.width * cfg.height, max_batch=4), OracleMap(cfg)
            rng = np.random.default_rng(70 + i)
            base = syn.room_depth(cfg)
            traj = syn.smooth_trajectory(200, 30 + i)
            k = 0
            for step in range(60):
                op = rng.choice(["dense", "batch", "q1", "qbulk", "setfree", "inflate", "mode"], p=[0.3, 0.15, 0.25, 0.1, 0.08, 0.07, 0.05])
                if op == "dense":
                    img, (q, t) = syn.jitter_depth(base, k, seed=i), traj[k]
                    k += 1
                    gpu.update_map(img, q, t)
                    cpu.update_depth(img, q, t)
                elif op == "batch":
                    n = int(rng.integers(2, 7))
                    fr = [(syn.jitter_depth(base, k + j, seed=i), traj[k + j]) for j in range(n)]
                    k += n
                    gpu.update_map_batch(np.stack([f[0] for f in fr]), np.stack([f[1][0] for f in fr]), np.stack([f[1][1] for f in fr]))
                    for img, (q, t) in fr:
                        cpu.update_depth(img, q, t)
                elif op in ("q1", "qbulk"):
                    pos = rng.uniform([-2, -5, 0], [6, 5, 3], size=(8 if op == "q1" else 2000, 3))
                    if op == "q1":
                        for j in range(8):
                            assert gpu.getOccupancy(pos[j:j + 1])[0] == cpu.getOccupancy(pos[j:j + 1])[0]
                            assert gpu.getOdd(pos[j:j + 1]).view(np.uint32)[0] == cpu.getOdd(pos[j:j + 1]).view(np.uint32)[0]
                    else:
                        assert np.array_equal(gpu.getOccupancy(pos), cpu.getOccupancy(pos))
                elif op == "setfree":
                    c = rng.uniform([0, -2, 0.5], [4, 2, 2.5])
                    gpu.setFree_map_in_bound(c - 0.4, c + 0.4)
                    cpu.setFree_map_in_bound(c - 0.4, c + 0.4)
                elif op == "inflate":
                    gpu.inflate_map(traj[max(k - 1, 0)][1])
                    cpu.inflate_map(traj[max(k - 1, 0)][1])
                else:
                    gpu.set_async(bool(rng.integers(0, 2)))
            compare_maps(gpu.export_blocks(), cpu.export_blocks(), f"handle {i}")
            if cfg.use_exploration_frontiers:
                assert np.array_equal(gpu.export_frontier(), cpu.export_frontier())
            gpu.close()
        except Exception as e:  # noqa: BLE001
            import traceback

            errors.append((i, traceback.format_exc()))

    ths = [threading.Thread(target=drive, args=(i,)) for i in range(2)]
    for th in ths:
        th.start()
    for th in ths:
        th.join()
    assert not errors, errors
