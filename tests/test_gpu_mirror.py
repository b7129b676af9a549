"""The host mirror behind small query batches (mlmapping_amd/csrc/mlm_mirror.h): single-position getOccupancy / getOdd / getOddGrad
/ getOccupancy(pos, inflate) / getInflateOccupancy / getOdd(glb_id, subbox_id) must equal the oracle's (include/mlmap.h:142-295),
equal the kernel path's, follow every kind of map change, and cost one refresh per change — not one per call."""
import threading

import numpy as np
import pytest

from mlmapping_amd import synthetic as syn
from mlmapping_amd.config import S1, SDEF
from tests.util import ODDS_TOL, compare_maps, voxel_centres

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def mods():
    from mlmapping_amd.mlmap import MLMap
    from oracle.binding import OracleMap

    return MLMap, OracleMap


def _one_by_one(fn, pos, *a):
    return np.concatenate([np.atleast_1d(fn(pos[i:i + 1], *a)) for i in range(pos.shape[0])])


def _check_all_kinds(gpu, cpu, pos, what, inflate=0.15):
    """every query kind, one position per call, against the oracle; odds as float BITS (both sides use the host libm's pow)"""
    assert np.array_equal(_one_by_one(gpu.getOccupancy, pos), cpu.getOccupancy(pos)), what
    assert np.array_equal(_one_by_one(gpu.getInflateOccupancy, pos), cpu.getInflateOccupancy(pos)), what
    go, co = _one_by_one(gpu.getOdd, pos), cpu.getOdd(pos)
    assert np.array_equal(go.view(np.uint32), co.view(np.uint32)), f"{what}: getOdd bits"
    k = min(pos.shape[0], 300)
    assert np.array_equal(_one_by_one(lambda p: gpu.getOccupancy(p, inflate=inflate), pos[:k]), cpu.getOccupancy(pos[:k], inflate=inflate)), what
    gg = np.concatenate([gpu.getOddGrad(pos[i:i + 1], 5) for i in range(k)])
    assert np.array_equal(gg, cpu.getOddGrad(pos[:k], 5)), f"{what}: getOddGrad"


def test_single_position_queries_match_oracle_and_kernel_path(mods, knobs):
    MLMap, OracleMap = mods
    cfg = S1
    gpu, cpu = MLMap(cfg, max_blocks=8192), OracleMap(cfg)
    for img, (q, t) in syn.stream(cfg, "room_jitter", "smooth", 5):
        gpu.update_map(img, q, t)
        cpu.update_depth(img, q, t)
    gpu.inflate_map([0.0, 0.0, 1.5])
    cpu.inflate_map([0.0, 0.0, 1.5])
    b = cpu.export_blocks()
    rng = np.random.default_rng(2)
    pos = np.concatenate([rng.uniform(b["keys"].min(0) - 1.0, b["keys"].max(0) + 2.0, size=(700, 3)), voxel_centres(b, cfg, 800, seed=1)])
    s0 = gpu.frame_stats()
    _check_all_kinds(gpu, cpu, pos, "S1")
    s1 = gpu.frame_stats()
    assert s1["n_mirror_refreshes"] - s0["n_mirror_refreshes"] == 1, "the map did not change: one refresh serves every call"
    assert s1["n_host_queries"] - s0["n_host_queries"] >= 3 * pos.shape[0]
    # mid-sized batches (still the host) and the getOdd(glb_id, subbox_id) overload
    for n in (2, 7, 64, 256):
        assert np.array_equal(gpu.getOccupancy(pos[:n]), cpu.getOccupancy(pos[:n]))
        assert np.array_equal(gpu.getOdd(pos[:n]).view(np.uint32), cpu.getOdd(pos[:n]).view(np.uint32))
    sel = rng.integers(0, b["keys"].shape[0], 300)
    glb = np.concatenate([b["keys"][sel], rng.integers(-60, 60, size=(100, 3)).astype(np.int32)])
    sub = rng.integers(0, cfg.cells_per_block, glb.shape[0]).astype(np.int32)
    at = np.concatenate([gpu.getOddAt(glb[i:i + 1], sub[i:i + 1]) for i in range(glb.shape[0])])
    assert np.array_equal(at.view(np.uint32), cpu.getOddAt(glb, sub).view(np.uint32))
    assert gpu.frame_stats()["n_mirror_refreshes"] == s1["n_mirror_refreshes"]
    # the kernel path (mirror off) on the same map: same classes, odds within the tolerance (the device's pow is not the host's)
    knobs.set("mirror", 0)
    dev = MLMap(cfg, max_blocks=8192)
    for img, (q, t) in syn.stream(cfg, "room_jitter", "smooth", 5):
        dev.update_map(img, q, t)
    dev.inflate_map([0.0, 0.0, 1.5])
    assert np.array_equal(_one_by_one(dev.getOccupancy, pos[:200]), cpu.getOccupancy(pos[:200]))
    assert dev.frame_stats()["n_host_queries"] == 0
    d_odd, h_odd = dev.getOdd(pos), _one_by_one(gpu.getOdd, pos)
    assert np.abs(d_odd - h_odd).max() <= 2e-7, "kernel path and host mirror disagree beyond an ulp of the float odd"
    assert np.array_equal(dev.getOccupancy(pos), _one_by_one(gpu.getOccupancy, pos))


def test_mirror_follows_every_kind_of_map_change(mods):
    """integrate (near, then far away: new blocks), setFree_map_in_bound, inflate_map, import_blocks — after each the single-position
    answers equal the oracle's, and a local change refreshes only the blocks around it."""
    MLMap, OracleMap = mods
    cfg = S1
    gpu, cpu = MLMap(cfg, max_blocks=128), OracleMap(cfg)  # (the pool grows under the mirror: 128 -> ...)
    img = syn.room_depth(cfg)
    rng = np.random.default_rng(9)

    def positions():
        b = cpu.export_blocks()
        return np.concatenate([rng.uniform(b["keys"].min(0) - 1.0, b["keys"].max(0) + 2.0, size=(150, 3)), voxel_centres(b, cfg, 250, seed=3)])

    def both(f):
        f(gpu)
        f(cpu)

    for k in range(3):
        q, t = syn.translating_pose(k)
        gpu.update_map(img, q, t)
        cpu.update_depth(img, q, t)
    _check_all_kinds(gpu, cpu, positions(), "first frames")
    # a frame 40 m away: new blocks, the old ones untouched — the refresh copies the new neighbourhood only
    s0 = gpu.frame_stats()
    q, t = syn.translating_pose(0)
    t_far = np.array([40.0, -25.0, 1.5])
    gpu.update_map(img, q, t_far)
    cpu.update_depth(img, q, t_far)
    n_before = s0["n_blocks"]
    _check_all_kinds(gpu, cpu, positions(), "after a far frame")
    s1 = gpu.frame_stats()
    assert s1["n_mirror_refreshes"] == s0["n_mirror_refreshes"] + 1
    assert s1["n_mirror_blocks"] - s0["n_mirror_blocks"] <= s1["n_blocks"] - n_before + 4, "blocks far from the frame were copied again"
    # setFree_map_in_bound
    both(lambda m: m.setFree_map_in_bound([0.5, -1.0, 0.3], [2.5, 1.0, 2.0]))
    _check_all_kinds(gpu, cpu, positions(), "after setFree")
    # inflate_map (changes inflate_occupancy only, creates neighbour blocks)
    both(lambda m: m.inflate_map([0.2, 0.0, 1.5]))
    _check_all_kinds(gpu, cpu, positions(), "after inflate")
    # import_blocks: overwrite one block and create one far from everything
    b = cpu.export_blocks()
    C = cfg.cells_per_block
    keys = np.array([b["keys"][0], [500, 500, 3]], dtype=np.int32)
    lo = rng.uniform(-2, 4, size=(2, C)).astype(np.float32)
    occ = rng.choice(np.frombuffer(b"ufo", dtype=np.uint8), size=(2, C))
    gpu.import_blocks(keys, lo, occ, occ)
    idx = rng.integers(0, 2 * C, 400)  # the imported blocks answer with the imported planes
    centres = voxel_centres({"keys": keys}, cfg, 2 * C)  # all 2 * C voxels, in (block, cell) order
    exp = np.where(occ.reshape(-1) == ord("o"), 0, np.where(occ.reshape(-1) == ord("f"), 1, -1))
    assert np.array_equal(_one_by_one(gpu.getOccupancy, centres[idx]), exp[idx])
    p10 = np.power(10.0, lo.reshape(-1).astype(np.float64))
    assert np.array_equal(_one_by_one(gpu.getOdd, centres[idx]).view(np.uint32), (p10 / (1 + p10)).astype(np.float32)[idx].view(np.uint32))
    # ... and the blocks the import did not touch keep the oracle's answers
    untouched = positions()
    far = np.abs(untouched - (b["keys"][0] + 0.5) * cfg.subbox_d_xyz * cfg.subbox_n).max(axis=1) > 1.5 * cfg.subbox_d_xyz * cfg.subbox_n
    assert np.array_equal(_one_by_one(gpu.getOccupancy, untouched[far]), cpu.getOccupancy(untouched[far]))
    assert gpu.frame_stats()["n_pool_grows"] >= 1


def test_mirror_in_async_mode_and_through_pool_growth(mods):
    """a single-position query observes everything submitted before it (it drains), also while the pool grows under the mirror"""
    MLMap, OracleMap = mods
    cfg = S1
    gpu, cpu = MLMap(cfg, max_blocks=64, max_batch=4), OracleMap(cfg)
    gpu.set_async(True)
    frames = list(syn.stream(cfg, "room_jitter", "random", 16, seed=5))
    rng = np.random.default_rng(4)
    for k in range(0, 16, 4):
        fb = np.stack([f[0] for f in frames[k:k + 4]])
        qb = np.stack([f[1][0] for f in frames[k:k + 4]])
        tb = np.stack([f[1][1] for f in frames[k:k + 4]])
        gpu.update_map_batch(fb, qb, tb)
        for f in frames[k:k + 4]:
            cpu.update_depth(f[0], *f[1])
        b = cpu.export_blocks()
        pos = np.concatenate([rng.uniform(b["keys"].min(0) - 1.0, b["keys"].max(0) + 2.0, size=(100, 3)), voxel_centres(b, cfg, 200, seed=k)])
        # NO sync() here: the query itself waits for what was submitted
        assert np.array_equal(_one_by_one(gpu.getOccupancy, pos), cpu.getOccupancy(pos)), f"batch {k // 4}"
        assert np.array_equal(_one_by_one(gpu.getOdd, pos).view(np.uint32), cpu.getOdd(pos).view(np.uint32))
    st = gpu.frame_stats()
    assert st["n_pool_grows"] >= 1 and st["n_mirror_refreshes"] == 4
    compare_maps(gpu.export_blocks(), cpu.export_blocks(), "async + small queries")


def test_mirror_growth_carries_its_contents_and_respects_the_limit(mods):
    """the planes grow with the map: what they held is copied on the host (only new or changed blocks cross the link), and a map
    that needs more pinned memory than mlm_set_host_mirror_limit allows is queried by kernels — same answers"""
    MLMap, OracleMap = mods
    cfg = S1
    gpu, cpu = MLMap(cfg, max_blocks=2048), OracleMap(cfg)
    rng = np.random.default_rng(9)
    copied = []
    for k in range(6):  # the sensor walks away: new blocks every frame, the first ones are never touched again
        q, t = syn.static_pose()
        t = np.array([20.0 * k, 0.0, 0.0]) + t  # (farther apart than the awareness cylinder is wide: an integrate call marks what it can reach)
        img = syn.room_depth(cfg)
        gpu.update_map(img, q, t)
        cpu.update_depth(img, q, t)
        b = cpu.export_blocks()
        pos = np.concatenate([rng.uniform(b["keys"].min(0) - 1.0, b["keys"].max(0) + 2.0, size=(60, 3)), voxel_centres(b, cfg, 120, seed=k)])
        s0 = gpu.frame_stats()
        assert np.array_equal(_one_by_one(gpu.getOccupancy, pos), cpu.getOccupancy(pos)), f"frame {k}"
        assert np.array_equal(_one_by_one(gpu.getOdd, pos).view(np.uint32), cpu.getOdd(pos).view(np.uint32)), f"frame {k}"
        s1 = gpu.frame_stats()
        copied.append(s1["n_mirror_blocks"] - s0["n_mirror_blocks"])
    n_blocks = gpu.frame_stats()["n_blocks"]
    assert n_blocks > 256, "the planes (256 blocks at first) must have grown in this test"
    assert sum(copied) < 2 * n_blocks, (copied, n_blocks)  # (growing the planes did not copy the whole map again each time)
    # a limit below what the planes hold: the copy is freed, small queries still answer (as kernels), and no host query is counted
    gpu.set_host_mirror_limit(1 << 20)
    h0 = gpu.frame_stats()["n_host_queries"]
    assert np.array_equal(_one_by_one(gpu.getOccupancy, pos[:40]), cpu.getOccupancy(pos[:40]))
    go, co = _one_by_one(gpu.getOdd, pos[:40]), cpu.getOdd(pos[:40])
    assert np.max(np.abs(go - co)) <= ODDS_TOL
    assert gpu.frame_stats()["n_host_queries"] == h0
    gpu.set_host_mirror_limit(1 << 30)  # raised again: the host path returns
    assert np.array_equal(_one_by_one(gpu.getOdd, pos[:40]).view(np.uint32), co.view(np.uint32))
    assert gpu.frame_stats()["n_host_queries"] > h0


def test_mirror_in_frontier_mode_released_blocks(mods):
    """use_exploration_frontiers: released blocks answer with element 0 (mlmap.h:183-184,221-222) on the host path too"""
    MLMap, OracleMap = mods
    cfg = S1.with_(use_exploration_frontiers=True, subbox_n=5)
    gpu, cpu = MLMap(cfg, max_blocks=16384), OracleMap(cfg)
    for img, (q, t) in syn.stream(cfg, "room_jitter", "smooth", 6):
        gpu.update_map(img, q, t)
        cpu.update_depth(img, q, t)
    b = cpu.export_blocks()
    assert b["collapsed"].any(), "the scene should release some blocks"
    col = b["keys"][b["collapsed"].astype(bool)]
    pos = np.concatenate([voxel_centres({"keys": col}, cfg, 400, seed=2), voxel_centres(b, cfg, 400, seed=3)])
    _check_all_kinds(gpu, cpu, pos, "frontier mode")
    compare_maps(gpu.export_blocks(), cpu.export_blocks(), "frontier mode after small queries")


def test_single_position_queries_from_a_second_thread(mods):
    """the planner thread asks position by position while the callback thread integrates: every answer set must be the map at
    SOME frame boundary; the final map equals the oracle's"""
    MLMap, OracleMap = mods
    cfg = SDEF
    n = 10
    frames = list(syn.stream(cfg, "room_jitter", "smooth", n))
    cpu = OracleMap(cfg)
    pos = np.random.default_rng(8).uniform([-2, -5, 0], [6, 5, 3], size=(40, 3))
    states = [cpu.getOccupancy(pos).copy()]
    for img, (q, t) in frames:
        cpu.update_depth(img, q, t)
        states.append(cpu.getOccupancy(pos).copy())
    states = np.stack(states)
    gpu = MLMap(cfg, max_blocks=8192, max_batch=4)
    errors, answers = [], []
    stop = threading.Event()

    def planner():
        try:
            while not stop.is_set():
                answers.append(gpu.getOccupancy(pos))  # (40 positions: one host batch, one lock — a consistent snapshot)
                for i in range(pos.shape[0]):
                    assert gpu.getOccupancy(pos[i:i + 1])[0] in (-1, 0, 1)
        except Exception as e:  # noqa: BLE001
            errors.append(e)

    th = threading.Thread(target=planner)
    th.start()
    import time

    for mode in (False, True):
        gpu.set_async(mode)
        for img, (q, t) in frames[: n // 2] if not mode else frames[n // 2:]:
            gpu.update_map(img, q, t)
            seen, t0 = len(answers), time.time()
            while len(answers) == seen and not errors and time.time() - t0 < 5.0:  # (let the planner get a word in between two frames)
                time.sleep(0.0005)
    gpu.sync()
    stop.set()
    th.join()
    assert not errors, errors
    assert len(answers) >= n // 2
    for a in answers:
        assert (states == a[None, :]).all(axis=1).any(), "a concurrent query saw a map no frame boundary produces"
    assert len({a.tobytes() for a in answers}) >= 3, "the planner should have seen the map at several frame boundaries"
    assert np.array_equal(_one_by_one(gpu.getOccupancy, pos), cpu.getOccupancy(pos))
    compare_maps(gpu.export_blocks(), cpu.export_blocks(), "after concurrent single-position queries")


def test_queries_on_an_empty_map(mods):
    """before the first frame: every query kind, one position per call and in bulk, answers like the reference's empty
    observed_group_map (UNKNOWN, 0.5, zero gradient) — found by tests/test_gpu_random_ops.py: the host mirror of an empty map"""
    MLMap, OracleMap = mods
    for cfg in (S1, S1.with_(use_exploration_frontiers=True)):
        gpu, cpu = MLMap(cfg, max_blocks=256), OracleMap(cfg)
        pos = np.random.default_rng(0).uniform(-5, 5, size=(300, 3))
        _check_all_kinds(gpu, cpu, pos, "empty map")
        assert np.array_equal(gpu.getOccupancy(np.tile(pos, (20, 1))), cpu.getOccupancy(np.tile(pos, (20, 1))))
        assert not gpu.getOddGrad(np.tile(pos, (20, 1))).any()
        assert gpu.getOddAt(np.array([[0, 0, 0]], dtype=np.int32), np.array([3], dtype=np.int32))[0] == 0.5
        gpu.setFree_map_in_bound([0, 0, 0], [1, 1, 1])
        gpu.inflate_map([0.0, 0.0, 1.0])
        cpu.inflate_map([0.0, 0.0, 1.0])
        _check_all_kinds(gpu, cpu, pos[:50], "empty map after setFree + inflate")
        gpu.close()


@pytest.mark.parametrize("explore", [False, True])
def test_planner_that_asks_after_every_frame(mods, explore):
    """queries after every synchronous frame switch the EAGER refresh on (the integrate call launches it as it returns, the first
    query only takes it in); frames without queries in between switch it off again; a pool that outgrows the mirror's planes sends the
    refresh back to the query — the answers equal the oracle's all the way (every kind, one position per call)"""
    MLMap, OracleMap = mods
    cfg = S1.with_(use_exploration_frontiers=explore)
    gpu, cpu = MLMap(cfg, max_blocks=64, max_points=cfg.width * cfg.height, max_batch=2), OracleMap(cfg)
    rng = np.random.default_rng(8)
    frames = list(syn.stream(cfg, "room_jitter", "smooth", 22, seed=4))

    def ask(n, what):
        b = cpu.export_blocks()
        pos = np.concatenate([voxel_centres(b, cfg, n - n // 3, seed=int(rng.integers(1 << 30))), rng.uniform(-3, 6, size=(n // 3, 3))])
        _check_all_kinds(gpu, cpu, pos, what)

    for k, (img, (q, t)) in enumerate(frames):
        t = t + np.array([0.15 * k, -0.1 * k, 0.0])  # (the camera wanders: the pool of 64 blocks and the mirror's planes grow on the way)
        if k % 5 == 4:  # the reference's sampler pattern now and then
            pix = (rng.integers(0, cfg.height, 500) * cfg.width + rng.integers(0, cfg.width, 500)).astype(np.int32)
            gpu.update_map(img, q, t, pixel_idx=pix)
            cpu.update_depth_indexed(img, pix, q, t)
        else:
            gpu.update_map(img, q, t)
            cpu.update_depth(img, q, t)
        if k in (9, 10, 11, 15):  # (frames nobody asks about: the eager refresh stops, and starts again with the next question)
            continue
        if k == 13:  # another kind of change between the frame and the question
            lo_, hi_ = t - 0.4, t + 0.4
            gpu.setFree_map_in_bound(lo_, hi_)
            cpu.setFree_map_in_bound(lo_, hi_)
        if k == 17:
            gpu.inflate_map(t)
            cpu.inflate_map(t)
        ask(24, f"after frame {k}")
    compare_maps(gpu.export_blocks(), cpu.export_blocks(), "planner stream")
    st = gpu.frame_stats()
    assert st["n_mirror_refreshes"] >= 16 and st["n_host_queries"] > 1000 and st["n_pool_grows"] >= 1, st
