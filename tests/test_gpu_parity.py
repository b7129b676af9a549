"""GPU parity: the HIP path (through the C ABI) against the CPU oracle on the same seeded inputs.

Bar (BASELINE.json north_star): hit/miss cell sets, block keys and occupancy classes bit-exact; odds within 1e-4.
"""
import ctypes

import numpy as np
import pytest

from mlmapping_amd import synthetic as syn
from mlmapping_amd.config import S1, S1_SIGMA0, S3, SDEF
from tests.util import ODDS_TOL, compare_maps, fuzz_trial, voxel_centres

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def mods():
    from mlmapping_amd.mlmap import MLMap
    from oracle.binding import OracleMap

    return MLMap, OracleMap


def _awareness_equal(gpu, cpu):
    gc, go, _ = gpu.awareness_hits()
    cc, co = cpu.hit_cells_sorted()
    assert np.array_equal(gc, cc), "hit cell sets differ"
    assert np.array_equal(go.view(np.uint32), co.view(np.uint32)), "hit odds differ (float bits)"
    assert np.array_equal(gpu.awareness_misses(), np.sort(cpu.misses()).astype(np.int64)), "miss cell sets differ"
    st = gpu.frame_stats()
    assert st["n_out_of_range"] == cpu.out_of_range_count()
    assert st["hit_bucket_count"] == cpu.hit_bucket_count()


def test_tables_and_pose(mods):
    MLMap, OracleMap = mods
    for cfg in (S1, SDEF):
        gpu, cpu = MLMap(cfg, max_blocks=1024, record_awareness=True), OracleMap(cfg)
        assert np.array_equal(gpu.odds_table().view(np.uint32), cpu.odds_table().view(np.uint32))
        for q, t in syn.random_poses(5, seed=3):
            gpu.update_map_points(np.array([[0.1, 0.2, 1.0]]), q, t)
            cpu.awareness_points(np.array([[0.1, 0.2, 1.0]]), q, t)
            gq, gt = gpu.T_ls()
            cq, ct = cpu.T_ls()
            assert np.array_equal(gq, cq) and np.array_equal(gt, ct)


def test_config1_kat_and_awareness_sets(mods):
    """BASELINE config 1: one 640x480 room frame, identity pose, 0.1 m — counts of SURVEY §8d and exact sets."""
    MLMap, OracleMap = mods
    gpu, cpu = MLMap(S1, max_blocks=4096, record_awareness=True), OracleMap(S1)
    img = syn.room_depth(S1)
    q, t = syn.static_pose()
    gpu.update_map(img, q, t)
    cpu.update_depth(img, q, t)
    st = gpu.frame_stats()
    assert (st["n_points"], st["n_hit_cells"], st["n_miss_cells"], st["n_blocks"]) == (307200, 14154, 74340, 116)
    _awareness_equal(gpu, cpu)
    d = compare_maps(gpu.export_blocks(), cpu.export_blocks(), "frame 0")
    assert gpu.class_counts() == {"blocks": 116, "o": 7924, "f": 32960}
    print(d)


@pytest.mark.parametrize("scene,poses,cfg,frames", [
    ("room", "static", S1, 20),
    ("room_jitter", "smooth", S1, 12),
    ("room_jitter", "random", S1, 8),
    ("scatter", "static", S1, 4),
    ("room", "translating", S1_SIGMA0, 8),
    ("room", "translating", SDEF, 10),
])
def test_stream_parity(mods, scene, poses, cfg, frames):
    """Multi-frame streams: the map must track the oracle frame after frame (iteration-order effects included)."""
    MLMap, OracleMap = mods
    gpu, cpu = MLMap(cfg, max_blocks=8192, record_awareness=True), OracleMap(cfg)
    worst = {"max_dodd": 0.0, "bit_mismatch": 0}
    for k, (img, (q, t)) in enumerate(syn.stream(cfg, scene, poses, frames)):
        gpu.update_map(img, q, t)
        cpu.update_depth(img, q, t)
        _awareness_equal(gpu, cpu)
        d = compare_maps(gpu.export_blocks(), cpu.export_blocks(), f"{scene}/{poses} frame {k}")
        worst["max_dodd"] = max(worst["max_dodd"], d["max_dodd"])
        worst["bit_mismatch"] = max(worst["bit_mismatch"], d["bit_mismatch"])
    if scene == "scatter":  # every pixel in a cell of its own: the columns overflow the small cell table — the first frame's columns are
        # redone with the large table when the call drains (no frame takes the cell-table path), which also schedules that pass for the rest
        assert gpu.frame_stats()["n_sector_fallbacks"] == 0, gpu.frame_stats()
    if scene == "room" and poses == "static" and cfg is S1:
        c = gpu.class_counts()
        assert (c["o"], c["f"]) == (9435, 32462)  # SURVEY §8d frame-19 KAT (iteration-order dependent)
    print(scene, poses, worst)


@pytest.mark.parametrize("node_lds,agg_lds", [(32, 256), (448, 16), (48, 16)])
def test_forced_buffer_overflows(mods, monkeypatch, knobs, node_lds, agg_lds):
    """The rare paths of k_bin_points / k_book_cells (a block's LDS group buffer or cell table overflows: groups booked one
    by one by k_assign_nodes, rays walked lane by lane) forced by shrinking the buffers; results must not change."""
    MLMap, OracleMap = mods
    knobs.set("node_lds", str(node_lds))
    knobs.set("agg_lds", str(agg_lds))
    for scene, poses, cfg, frames in (("room_jitter", "random", S1, 3), ("scatter", "static", S1, 2)):
        gpu, cpu = MLMap(cfg, max_blocks=8192, record_awareness=True), OracleMap(cfg)
        for k, (img, (q, t)) in enumerate(syn.stream(cfg, scene, poses, frames)):
            gpu.update_map(img, q, t)
            cpu.update_depth(img, q, t)
            _awareness_equal(gpu, cpu)
            compare_maps(gpu.export_blocks(), cpu.export_blocks(), f"overflow {node_lds}/{agg_lds} {scene} frame {k}")
        gpu.close()
    cfg = S1.with_(use_exploration_frontiers=True)
    gpu, cpu = MLMap(cfg, max_blocks=8192, record_awareness=True), OracleMap(cfg)
    for k, (img, (q, t)) in enumerate(syn.stream(cfg, "room_jitter", "smooth", 3)):
        gpu.update_map(img, q, t)
        cpu.update_depth(img, q, t)
        compare_maps(gpu.export_blocks(), cpu.export_blocks(), f"overflow {node_lds}/{agg_lds} frontier frame {k}")
        assert np.array_equal(gpu.export_frontier(), cpu.export_frontier())
    gpu.close()


@pytest.mark.parametrize("w,h", [(333, 77), (5, 3), (33, 9), (640, 1), (1, 240), (2040, 48), (2041, 40), (4000, 24), (9000, 12), (24, 3000)])
def test_odd_image_sizes(mods, w, h):
    """Frame sizes that are not multiples of the 32x8 pixel tiles / the 4x4 tile groups (partial tiles, single rows and
    columns), images wider than the 2 040 pixels a reference of the ranking kernel used to span (its columns are relative to the
    cell's first pixel since round 6: any width stays on the sector path) and one taller than the 2 047 rows it can span must
    give the same map as the oracle — without a frame taking the cell-table path."""
    MLMap, OracleMap = mods
    cfg = S1.with_(width=w, height=h, cam_cx=w / 2.0, cam_cy=h / 2.0)
    gpu, cpu = MLMap(cfg, max_blocks=8192, max_points=max(w * h, 4096), record_awareness=True), OracleMap(cfg)
    rng = np.random.default_rng(w * 1000 + h)
    for k in range(3):
        depth = rng.integers(400, 6000, size=(h, w)).astype(np.uint16)
        depth[rng.random((h, w)) < 0.05] = 0
        q, t = syn.random_poses(3, seed=h)[k]
        gpu.update_map(depth, q, t)
        cpu.update_depth(depth, q, t)
        _awareness_equal(gpu, cpu)
        compare_maps(gpu.export_blocks(), cpu.export_blocks(), f"{w}x{h} frame {k}")
    assert gpu.frame_stats()["n_sector_fallbacks"] == 0, (w, h)
    gpu.close()


def test_frontier_mode_batch(mods):
    """Frontier mode through the batch entry point (Stage A of all frames in one launch sequence, the map-dependent part
    frame by frame): same map, frontier and released blocks as frame-by-frame submission and as the oracle."""
    MLMap, OracleMap = mods
    cfg = S1.with_(use_exploration_frontiers=True)
    n = 10
    frames = np.stack([img for img, _ in syn.stream(cfg, "room_jitter", "smooth", n)])
    poses = [p for _, p in syn.stream(cfg, "room_jitter", "smooth", n)]
    q = np.stack([p[0] for p in poses])
    t = np.stack([p[1] for p in poses])
    a, b, cpu = MLMap(cfg, max_blocks=8192, max_batch=4), MLMap(cfg, max_blocks=8192), OracleMap(cfg)
    a.update_map_batch(frames, q, t)  # 4 + 4 + 2
    for k in range(n):
        b.update_map(frames[k], q[k], t[k])
        cpu.update_depth(frames[k], q[k], t[k])
    ea, eb = a.export_blocks(), b.export_blocks()
    for key in ("keys", "occ", "infl", "collapsed"):
        assert np.array_equal(ea[key], eb[key]), key
    assert np.array_equal(ea["log_odds"], eb["log_odds"])
    assert np.array_equal(a.export_frontier(), b.export_frontier())
    compare_maps(ea, cpu.export_blocks(), "frontier batch vs oracle")
    assert np.array_equal(a.export_frontier(), cpu.export_frontier())
    assert a.frame_stats()["n_miss_cells"] == b.frame_stats()["n_miss_cells"]
    # asynchronous submission: Stage A of a batch overlaps the map-dependent part of the batch before it
    c = MLMap(cfg, max_blocks=8192, max_batch=3)
    c.set_async(True)
    for k0 in (0, 3, 6):
        c.update_map_batch(frames[k0:k0 + 3], q[k0:k0 + 3], t[k0:k0 + 3])
    c.update_map(frames[9], q[9], t[9])
    ec = c.export_blocks()  # (exports wait for everything submitted)
    for key in ("keys", "occ", "infl", "collapsed"):
        assert np.array_equal(ec[key], eb[key]), key
    assert np.array_equal(ec["log_odds"], eb["log_odds"])
    assert np.array_equal(c.export_frontier(), b.export_frontier())


def test_config3_720p(mods):
    """BASELINE config 3: 1280x720, 0.05 m voxels."""
    MLMap, OracleMap = mods
    gpu, cpu = MLMap(S3, max_blocks=8192, record_awareness=True), OracleMap(S3)
    img = syn.room_depth(S3)
    for k in range(2):
        q, t = syn.translating_pose(k)
        gpu.update_map(img, q, t)
        cpu.update_depth(img, q, t)
        _awareness_equal(gpu, cpu)
        compare_maps(gpu.export_blocks(), cpu.export_blocks(), f"S3 frame {k}")
        if k == 0:
            st = gpu.frame_stats()
            assert (st["n_hit_cells"], st["n_miss_cells"]) == (136766, 659944)


def test_config3_stream_random_poses(mods):
    """BASELINE config 3 as a stream: 20 jittered 1280x720 frames with a random SE(3) pose each into the 0.05 m map, through
    the asynchronous batch entry point in batches of 5 (this map's tiles are 4x4 voxel columns of 91 layers; the batched
    apply walks them frame after frame), from a pool of 1 024 blocks that grows on the way — against the oracle fed frame by
    frame."""
    MLMap, OracleMap = mods
    cfg, n, B = S3, 20, 5
    base = syn.room_depth(cfg)
    frames = np.stack([syn.jitter_depth(base, k, seed=7) for k in range(B)])
    poses = syn.random_poses(n, seed=7)
    q = np.stack([p[0] for p in poses])
    t = np.stack([p[1] for p in poses])
    gpu, cpu = MLMap(cfg, max_blocks=1024, max_batch=B), OracleMap(cfg)
    gpu.set_async(True)
    for k0 in range(0, n, B):
        gpu.update_map_batch(frames, q[k0:k0 + B], t[k0:k0 + B])
        for j in range(B):
            cpu.update_depth(frames[j], q[k0 + j], t[k0 + j])
    gpu.sync()
    d = compare_maps(gpu.export_blocks(), cpu.export_blocks(), "cfg3 stream, 20 frames")
    st = gpu.frame_stats()
    assert st["n_sector_fallbacks"] == 0 and st["n_pool_grows"] >= 1, st
    print("cfg3 stream", d, st["block_capacity"])


def test_near_wall_thousands_of_pixels_per_cell(mods):
    """A surface a quarter of a metre in front of the sensor: single cells collect tens of thousands of pixels, and the column
    kernel's 11-bit count of a cell's references wraps.  Such cells have one kind of contribution or saturate — they need no
    references — so the frame must stay on the sector path (no fall-back) and match the oracle; the right half of the image
    is an ordinary room."""
    MLMap, OracleMap = mods
    cfg = S1
    img = syn.room_depth(cfg).copy()
    img[:, : cfg.width // 2] = 250
    rng = np.random.default_rng(5)
    img[:, : cfg.width // 2] += rng.integers(0, 30, size=(cfg.height, cfg.width // 2)).astype(np.uint16)
    gpu, cpu = MLMap(cfg, max_blocks=4096, record_awareness=True), OracleMap(cfg)
    for k, (q, t) in enumerate(syn.random_poses(3, seed=11)):
        gpu.update_map(img, q, t)
        cpu.update_depth(img, q, t)
        _awareness_equal(gpu, cpu)
        compare_maps(gpu.export_blocks(), cpu.export_blocks(), f"near wall frame {k}")
    assert gpu.frame_stats()["n_sector_fallbacks"] == 0, gpu.frame_stats()


def test_column_table_widens_with_the_scene(mods, monkeypatch, capfd):
    """Depth noise of metres spreads a column's hits over several hundred cells: they overflow the 512-entry LDS table the handle
    starts with on this map.  The first such frame falls back once, the overflowed columns of the next frames are redone by the
    large-table pass, and after two batches like that the handle doubles the table (widen_sec_tab) — the map equals the oracle's
    throughout, in single frames and in batches."""
    MLMap, OracleMap = mods
    monkeypatch.setenv("MLM_DEBUG_CREATE", "1")
    cfg = S1
    base = syn.room_depth(cfg)
    frames = np.stack([syn.jitter_depth(base, k, amp_mm=3000, seed=3) for k in range(4)])
    poses = syn.random_poses(12, seed=3)
    q = np.stack([p[0] for p in poses])
    t = np.stack([p[1] for p in poses])
    gpu, cpu = MLMap(cfg, max_blocks=4096, max_batch=4), OracleMap(cfg)
    for k in range(4):  # single frames
        gpu.update_map(frames[k], q[k], t[k])
        cpu.update_depth(frames[k], q[k], t[k])
    compare_maps(gpu.export_blocks(), cpu.export_blocks(), "noisy scene, single frames")
    for k0 in (4, 8):  # batches
        gpu.update_map_batch(frames, q[k0:k0 + 4], t[k0:k0 + 4])
        for j in range(4):
            cpu.update_depth(frames[j], q[k0 + j], t[k0 + j])
    compare_maps(gpu.export_blocks(), cpu.export_blocks(), "noisy scene, batches")
    st = gpu.frame_stats()
    err = capfd.readouterr().err
    assert "cell table widened" in err, err[-2000:]
    assert st["n_sector_fallbacks"] <= 2, st


@pytest.mark.parametrize("threads,tab", [("256", "512"), ("256", "1024"), ("512", "512"), ("512", "2048")])
def test_column_kernel_instantiations(mods, monkeypatch, knobs, threads, tab):
    """The column kernel exists with 256- and 512-thread workgroups and takes tables of 512 to 2 048 entries (1 to 4 entries per
    thread); the handle picks one pair per map.  Every pair must give the oracle's map — batches (where the choice applies) and
    single frames (which always take the 512-thread instantiation) alike."""
    MLMap, OracleMap = mods
    knobs.set("sec_threads", threads)
    knobs.set("sec_tab", tab)
    cfg = S1
    frames = np.stack([img for img, _ in syn.stream(cfg, "room_jitter", "random", 4)])
    poses = syn.random_poses(10, seed=21)
    q = np.stack([p[0] for p in poses])
    t = np.stack([p[1] for p in poses])
    gpu, cpu = MLMap(cfg, max_blocks=4096, max_batch=4), OracleMap(cfg)
    for k0 in (0, 4):
        gpu.update_map_batch(frames, q[k0:k0 + 4], t[k0:k0 + 4])
        for j in range(4):
            cpu.update_depth(frames[j], q[k0 + j], t[k0 + j])
    for k in (8, 9):
        gpu.update_map(frames[k - 8], q[k], t[k])
        cpu.update_depth(frames[k - 8], q[k], t[k])
    compare_maps(gpu.export_blocks(), cpu.export_blocks(), f"column kernel {threads} threads, table {tab}")
    assert gpu.frame_stats()["n_sector_fallbacks"] == 0


def test_single_frames_go_through_the_graph(mods, monkeypatch, knobs):
    """Synchronous single-frame calls (the reference's call pattern: one frame per depth callback, mlmap.cpp:463-507) are
    submitted as one HIP-graph replay; the result is the general submission's (knob graph = 0) bit for bit — dense frames from a
    host buffer, pixel lists, the callback's sampler — and the oracle's."""
    MLMap, OracleMap = mods
    cfg = SDEF
    frames = list(syn.stream(cfg, "room_jitter", "random", 8))
    maps = []
    for graph in ("1", "0"):
        knobs.set("graph", graph)
        gpu = MLMap(cfg, max_blocks=2048, max_batch=2)
        for k, (img, (q, t)) in enumerate(frames):
            if k % 3 == 2:
                gpu.update_map(img, q, t, pixel_idx=np.arange(0, img.size, 7))
            else:
                gpu.update_map(img, q, t)
        st = gpu.frame_stats()
        assert (st["n_graph_launches"] >= len(frames) - 2) if graph == "1" else st["n_graph_launches"] == 0, st
        maps.append(gpu.export_blocks())
        gpu.close()
    cpu = OracleMap(cfg)
    for k, (img, (q, t)) in enumerate(frames):
        if k % 3 == 2:
            cpu.update_depth_indexed(img, np.arange(0, img.size, 7), q, t)
        else:
            cpu.update_depth(img, q, t)
    compare_maps(maps[0], cpu.export_blocks(), "graph submission vs oracle")
    compare_maps(maps[1], cpu.export_blocks(), "general submission vs oracle")


def test_reference_sampler_via_pixel_list(mods):
    """The reference's 500-sample rand() path: the host draws the pixel list, both sides integrate it."""
    MLMap, OracleMap = mods
    gpu, cpu = MLMap(SDEF, max_blocks=2048, record_awareness=True), OracleMap(SDEF)
    img = syn.room_depth(SDEF)
    rng = np.random.default_rng(1)
    for k in range(10):
        q, t = syn.translating_pose(k)
        pix = (rng.integers(0, SDEF.height, 500) * SDEF.width + rng.integers(0, SDEF.width, 500)).astype(np.int32)
        gpu.update_map(img, q, t, pixel_idx=pix)
        cpu.update_depth_indexed(img, pix, q, t)
        _awareness_equal(gpu, cpu)
        compare_maps(gpu.export_blocks(), cpu.export_blocks(), f"sampled frame {k}")


def test_edge_points(mods):
    """SURVEY §8c G5 edge cases through the explicit point interface."""
    MLMap, OracleMap = mods
    cfg = S1
    gpu, cpu = MLMap(cfg, max_blocks=4096, record_awareness=True), OracleMap(cfg)
    q, t = syn.static_pose()
    eps = 4.4e-16
    # sensor frame: x right, y down, z forward (T_B_S of config_sim.yaml)
    pts = np.array([
        [eps, 0.0, 2.0],       # y_l = -eps -> phi_idx == nPhi drop
        [0.0, 0.0, 2.0],       # on the optical axis
        [0.0, 0.0, 30.0],      # rho >= nRho: clamp to the border then cast
        [0.0, -5.0, 3.0],      # z above the map, ray cells in range
        [0.0, 5.0, 3.0],       # z below
        [0.0, 0.0, 0.05],      # cell rho = 1 (0.12 + 0.05 m): no miss cells
        [0.0, 0.0, -0.12],     # x_l == 0 exactly: fast_atan2 quirk
        [-0.12, 0.0, -0.12],   # somewhere behind
        [0.0, 0.0, 0.0],
        [3.0, 0.3, 6.3],
    ])
    gpu.update_map_points(pts, q, t)
    cpu.update_points(pts, q, t)
    _awareness_equal(gpu, cpu)
    compare_maps(gpu.export_blocks(), cpu.export_blocks(), "edge points")
    # empty input
    gpu.update_map_points(np.zeros((0, 3)), q, t)
    cpu.update_points(np.zeros((0, 3)), q, t)
    assert gpu.frame_stats()["n_hit_cells"] == 0
    compare_maps(gpu.export_blocks(), cpu.export_blocks(), "after empty frame")


def test_queries(mods):
    MLMap, OracleMap = mods
    cfg = S1
    gpu, cpu = MLMap(cfg, max_blocks=8192), OracleMap(cfg)
    for img, (q, t) in syn.stream(cfg, "room_jitter", "smooth", 6):
        gpu.update_map(img, q, t)
        cpu.update_depth(img, q, t)
    b = cpu.export_blocks()
    rng = np.random.default_rng(5)
    lo, hi = b["keys"].min(0) * 1.0 - 1.0, b["keys"].max(0) * 1.0 + 2.0
    pos = np.concatenate([rng.uniform(lo, hi, size=(50000, 3)), voxel_centres(b, cfg, 100000)])
    assert np.array_equal(gpu.getOccupancy(pos), cpu.getOccupancy(pos))
    assert np.abs(gpu.getOdd(pos) - cpu.getOdd(pos)).max() <= ODDS_TOL
    assert np.array_equal(gpu.getOccupancy(pos[:20000], inflate=0.15), cpu.getOccupancy(pos[:20000], inflate=0.15))
    gg, cg = gpu.getOddGrad(pos[:30000]), cpu.getOddGrad(pos[:30000])
    assert np.abs(gg - cg).max() <= 1e-4 * max(1.0, np.abs(cg).max())
    # setFree_map_in_bound: idempotent, then everything must still agree
    bmin, bmax = np.array([0.5, -1.0, 0.3]), np.array([2.5, 1.0, 2.0])
    for _ in range(2):
        gpu.setFree_map_in_bound(bmin, bmax)
        cpu.setFree_map_in_bound(bmin, bmax)
    compare_maps(gpu.export_blocks(), cpu.export_blocks(), "after setFree")
    assert np.array_equal(gpu.getOccupancy(pos), cpu.getOccupancy(pos))


def test_batch_pipeline_matches_frame_by_frame(mods):
    """mlm_integrate_depth_batch_dev (Stage A of several frames in flight, speculative Stage B) must give the same
    map as the oracle fed frame by frame — including the first batch, where the emulated hit container rehashes."""
    MLMap, OracleMap = mods
    cfg = S1
    n = 21
    frames = np.stack([img for img, _ in syn.stream(cfg, "room_jitter", "random", n)])
    poses = syn.random_poses(n, 42)
    q = np.stack([p[0] for p in poses])
    t = np.stack([p[1] for p in poses])
    gpu, cpu = MLMap(cfg, max_blocks=16384, max_batch=8), OracleMap(cfg)
    gpu.update_map_batch(frames, q, t)
    for k in range(n):
        cpu.update_depth(frames[k], q[k], t[k])
    info = compare_maps(gpu.export_blocks(), cpu.export_blocks(), "batch of 21")
    st = gpu.frame_stats()
    assert st["hit_bucket_count"] == cpu.hit_bucket_count()
    assert st["n_hit_cells"] == len(cpu.hits()[1]) and st["n_miss_cells"] == len(cpu.misses())
    print(info)


def test_async_submission_matches(mods):
    """Asynchronous mode (two batches in flight, confirmation one call late) must end in the same map."""
    MLMap, OracleMap = mods
    cfg = S1
    n = 30
    frames = np.stack([img for img, _ in syn.stream(cfg, "room_jitter", "smooth", n)])
    poses = syn.smooth_trajectory(n, 42)
    q = np.stack([p[0] for p in poses])
    t = np.stack([p[1] for p in poses])
    gpu, cpu = MLMap(cfg, max_blocks=16384, max_batch=4), OracleMap(cfg)
    gpu.set_async(True)
    for k0 in range(0, n, 5):  # batches of 5 > max_batch 4: also splits inside a call
        gpu.update_map_batch(frames[k0:k0 + 5], q[k0:k0 + 5], t[k0:k0 + 5])
    gpu.sync()
    for k in range(n):
        cpu.update_depth(frames[k], q[k], t[k])
    print(compare_maps(gpu.export_blocks(), cpu.export_blocks(), "async 30 frames"))
    assert gpu.frame_stats()["n_hit_cells"] == len(cpu.hits()[1])


def test_inflation_and_global_map(mods):
    """§8f rank 2/4: inflate_map (incl. the reference's visiting-order wipe rule and block allocation by inflation),
    getInflateOccupancy, getOccupancy(pos, inflate) and the /global_map point payload."""
    MLMap, OracleMap = mods
    for cfg in (S1, SDEF):
        gpu, cpu = MLMap(cfg, max_blocks=8192), OracleMap(cfg)
        pose = None
        for k, (img, (q, t)) in enumerate(syn.stream(cfg, "room_jitter", "smooth", 5)):
            gpu.update_map(img, q, t)
            cpu.update_depth(img, q, t)
            pose = t
            if k in (2, 4):  # the 5 Hz timer fires between frames; repeated inflation accumulates outside the cube
                gpu.inflate_map(pose)
                cpu.inflate_map(pose)
        g, c = gpu.export_blocks(), cpu.export_blocks()
        compare_maps(g, c, "after inflation")
        assert np.array_equal(g["infl"], c["infl"]), f"{int((g['infl'] != c['infl']).sum())} inflate cells differ"
        assert (c["infl"] == ord("o")).sum() > 0
        gp, cp = gpu.global_map_points(), cpu.global_map_points()
        assert gp.shape == cp.shape
        key = lambda a: a[np.lexsort((a[:, 2], a[:, 1], a[:, 0]))]
        assert np.array_equal(key(gp).view(np.uint32), key(cp).view(np.uint32)), "global map points differ"
        rng = np.random.default_rng(9)
        pos = np.concatenate([rng.uniform(c["keys"].min(0) * cfg.subbox_d_xyz * cfg.subbox_n - 1,
                                          (c["keys"].max(0) + 1) * cfg.subbox_d_xyz * cfg.subbox_n + 1, size=(40000, 3)),
                              voxel_centres(c, cfg, 60000)])
        assert np.array_equal(gpu.getInflateOccupancy(pos), cpu.getInflateOccupancy(pos))


@pytest.mark.parametrize("name", ["S1-n5", "SDEF"])
def test_exploration_frontiers(mods, name):
    """§8f rank 1: use_exploration_frontiers: true (config2.yaml:36) — frontier sets (order dependent through the
    iteration order of miss_idx_set), released/frozen blocks, and everything else as before."""
    MLMap, OracleMap = mods
    cfg = (S1.with_(use_exploration_frontiers=True, subbox_n=5) if name == "S1-n5"
           else SDEF.with_(use_exploration_frontiers=True, lm_occupied_sh=2.0, depth_noise_coe=0.00375))
    gpu, cpu = MLMap(cfg, max_blocks=16384, record_awareness=True), OracleMap(cfg)
    for k, (img, (q, t)) in enumerate(syn.stream(cfg, "room_jitter", "smooth", 8)):
        gpu.update_map(img, q, t)
        cpu.update_depth(img, q, t)
        _awareness_equal(gpu, cpu)
        compare_maps(gpu.export_blocks(), cpu.export_blocks(), f"explore {name} frame {k}")
        gf, cf = gpu.export_frontier(), cpu.export_frontier()
        assert gf.shape == cf.shape and np.array_equal(gf, cf), f"frontier sets differ at frame {k}: {gf.shape} vs {cf.shape}"
    b = cpu.export_blocks()
    if name == "S1-n5":
        assert b["collapsed"].sum() > 50  # the release path is really exercised
    pos = np.concatenate([np.random.default_rng(3).uniform(-4, 6, size=(40000, 3)), voxel_centres(b, cfg, 60000)])
    assert np.array_equal(gpu.getOccupancy(pos), cpu.getOccupancy(pos))
    assert np.abs(gpu.getOdd(pos) - cpu.getOdd(pos)).max() <= ODDS_TOL
    gpu.setFree_map_in_bound([0.0, -1.0, 0.5], [2.0, 1.0, 1.5])
    cpu.setFree_map_in_bound([0.0, -1.0, 0.5], [2.0, 1.0, 1.5])
    gpu.inflate_map([0.0, 0.0, 1.5])
    cpu.inflate_map([0.0, 0.0, 1.5])
    g2, c2 = gpu.export_blocks(), cpu.export_blocks()
    compare_maps(g2, c2, "explore after setFree+inflate")
    col = c2["collapsed"].astype(bool)
    assert np.array_equal(g2["infl"][~col], c2["infl"][~col])
    assert np.array_equal(gpu.getInflateOccupancy(pos), cpu.getInflateOccupancy(pos))


def test_ros_free_callback(mods):
    """§8f rank 3: depth_odom_input_callback without ROS — 32FC1 conversion, pose latency compensation
    (SO3 log/exp), the 500-sample rand() sampler and the dense variant."""
    MLMap, OracleMap = mods
    libc = ctypes.CDLL("libc.so.6")
    cfg = SDEF
    base = syn.room_depth(cfg).astype(np.float32) / 1000.0  # metres
    rng = np.random.default_rng(11)
    for sampled in (True, False):
        gpu, cpu = MLMap(cfg, max_blocks=4096, record_awareness=True), OracleMap(cfg)
        for k in range(6):
            depth = base + rng.uniform(0, 0.05, size=base.shape).astype(np.float32)
            depth[rng.integers(0, cfg.height, 50), rng.integers(0, cfg.width, 50)] = 0.0  # invalid pixels
            q, t = syn.smooth_trajectory(6, 5)[k]
            args = dict(t_img=10.0 + k / 30.0, odom_p=t, odom_q=q, odom_v=[0.3, -0.1, 0.02], t_odom=10.0 + k / 30.0 - 0.004,
                        imu_w=[0.05, -0.2, 0.4], t_imu=10.0 + k / 30.0 - 0.002, latency=0.085, sampled=sampled)
            libc.srand(100 + k)
            tg = gpu.depth_odom_callback(depth, **args)
            libc.srand(100 + k)
            tc = cpu.depth_odom_callback(depth, **args)
            assert np.array_equal(tg, tc), "compensated T_wb differs"
            _awareness_equal(gpu, cpu)
            compare_maps(gpu.export_blocks(), cpu.export_blocks(), f"callback sampled={sampled} frame {k}")
        assert gpu.frame_stats()["n_points"] == (500 if sampled else gpu.frame_stats()["n_points"])


def test_callback_mixed_upload_streams_on_one_handle(mods):
    """One handle, every entry point that uploads host inputs, interleaved — dense 16UC1 callback (graph path), sampled callback
    (pinned staging buffer allocated on first use), 32FC1 dense callback, a wide host image through mlm_integrate_depth_u16
    with a pixel list, explicit points — then mlm_destroy.  The uploads go to the stream the call's Stage A runs on
    (upload_stream); where a frame-level veto moves Stage A to another stream an event orders it behind the upload, and that
    event must survive the (re)allocation of the sampler's staging buffer (round-3 advisor finding: it was destroyed there and
    used again by the next cross-stream upload)."""
    MLMap, OracleMap = mods
    libc = ctypes.CDLL("libc.so.6")
    cfg = SDEF
    base_u16 = syn.room_depth(cfg)
    base_f32 = base_u16.astype(np.float32) / 1000.0
    gpu, cpu = MLMap(cfg, max_blocks=4096, record_awareness=True), OracleMap(cfg)
    rng = np.random.default_rng(5)
    poses = syn.smooth_trajectory(12, 3)
    for k in range(12):
        q, t = poses[k]
        args = dict(t_img=1.0 + k / 30.0, odom_p=t, odom_q=q, odom_v=[0.2, 0.0, 0.0], t_odom=1.0 + k / 30.0, imu_w=[0.0, 0.0, 0.2],
                    t_imu=1.0 + k / 30.0, latency=0.01)
        kind = k % 6
        libc.srand(40 + k)
        if kind == 0:    # dense 16UC1 callback: uploads on the main stream (graph path)
            gpu.depth_odom_callback(base_u16, sampled=False, **args)
            libc.srand(40 + k)
            cpu.depth_odom_callback(base_u16, sampled=False, **args)
        elif kind == 1:  # sampled callback: allocates the pinned staging buffer on its first use
            gpu.depth_odom_callback(base_u16, sampled=True, **args)
            libc.srand(40 + k)
            cpu.depth_odom_callback(base_u16, sampled=True, **args)
        elif kind == 2:  # 32FC1 dense: converted on a side stream
            gpu.depth_odom_callback(base_f32, sampled=False, **args)
            libc.srand(40 + k)
            cpu.depth_odom_callback(base_f32, sampled=False, **args)
        elif kind == 3:  # host image + pixel list
            pix = rng.choice(cfg.width * cfg.height, 700, replace=False).astype(np.int32)
            gpu.update_map(base_u16, q, t, pixel_idx=pix)
            cpu.update_depth_indexed(base_u16, pix, q, t)
        elif kind == 4:  # explicit points
            pts = cpu.project_dense(base_u16)[::97]
            gpu.update_map_points(pts, q, t)
            cpu.update_points(pts, q, t)
        else:            # 32FC1 sampled
            gpu.depth_odom_callback(base_f32, sampled=True, **args)
            libc.srand(40 + k)
            cpu.depth_odom_callback(base_f32, sampled=True, **args)
        compare_maps(gpu.export_blocks(), cpu.export_blocks(), f"mixed entry points, call {k} (kind {kind})")
    gpu.close()  # mlm_destroy: every event is destroyed exactly once


def test_gpu_against_golden_digests(mods):
    """The HIP path against the committed golden fixtures (tests/golden/oracle_digests.json): every integer/byte output
    must hash to the recorded digest (log-odds are float and compared by tolerance elsewhere)."""
    import importlib.util
    import json
    import os

    MLMap, _ = mods
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    spec = importlib.util.spec_from_file_location("make_golden", os.path.join(root, "tests", "golden", "make_golden.py"))
    mg = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mg)
    want = json.load(open(os.path.join(root, "tests", "golden", "oracle_digests.json")))
    for name, (cfg, scene, poses, frames) in mg.CASES.items():
        gpu = MLMap(cfg, max_blocks=16384, record_awareness=True)
        for k, (img, (q, t)) in enumerate(syn.stream(cfg, scene, poses, max(frames) + 1)):
            gpu.update_map(img, q, t)
            if k in frames:
                cells, odds, _ = gpu.awareness_hits()
                miss = gpu.awareness_misses().astype(np.uint64)
            if k == max(frames):
                gpu.inflate_map(t)
            if k in frames:
                w = want[f"{name}/{k}"]
                assert mg.h(cells, odds) == w["hits"], f"{name}/{k}: hit cells/odds"
                assert mg.h(miss) == w["misses"], f"{name}/{k}: miss cells"
                b = gpu.export_blocks()
                col = b["collapsed"].astype(bool)
                for key in ("occ", "infl"):
                    b[key][col, 1:] = 0  # a released block keeps element 0 only in the reference
                assert mg.h(b["keys"], b["collapsed"]) == w["blocks"], f"{name}/{k}: block keys"
                assert mg.h(b["occ"]) == w["occ"], f"{name}/{k}: occupancy"
                assert mg.h(b["infl"]) == w["infl"], f"{name}/{k}: inflate occupancy"
                assert mg.h(gpu.export_frontier()) == w["frontier"], f"{name}/{k}: frontier"


def test_determinism_and_properties_full_size(mods):
    """Size-independent properties at BASELINE's full sizes: two handles fed the same stream end bit-identical
    (no dependence on GPU scheduling), batch == frame-by-frame, setFree is idempotent, an empty frame is a no-op."""
    MLMap, _ = mods
    for cfg, n in ((S1, 12), (S3, 3)):
        frames = np.stack([img for img, _ in syn.stream(cfg, "room_jitter", "random", n)])
        poses = syn.random_poses(n, 42)
        q = np.stack([p[0] for p in poses])
        t = np.stack([p[1] for p in poses])
        a = MLMap(cfg, max_blocks=32768, max_batch=4)
        b = MLMap(cfg, max_blocks=32768, max_batch=1)
        a.update_map_batch(frames, q, t)
        for k in range(n):
            b.update_map(frames[k], q[k], t[k])
        ea, eb = a.export_blocks(), b.export_blocks()
        for key in ("keys", "occ", "infl"):
            assert np.array_equal(ea[key], eb[key])
        assert np.array_equal(ea["log_odds"].view(np.uint32), eb["log_odds"].view(np.uint32)), "log-odds bits differ"
        before = a.export_blocks()
        a.update_map_points(np.zeros((0, 3)), q[0], t[0])
        after = a.export_blocks()
        assert all(np.array_equal(before[k], after[k]) for k in ("keys", "occ", "log_odds"))
        a.setFree_map_in_bound([-1, -1, 0.5], [1, 1, 1.5])
        once = a.export_blocks()
        a.setFree_map_in_bound([-1, -1, 0.5], [1, 1, 1.5])
        twice = a.export_blocks()
        assert np.array_equal(once["occ"], twice["occ"]) and np.array_equal(once["log_odds"], twice["log_odds"])
        a.close()
        b.close()


def test_heavy_cell_beyond_lds_window(mods):
    """A multi-kind cell with more contributions than the LDS sort window (2048) takes the from-memory ranking path."""
    MLMap, OracleMap = mods
    cfg = S1
    gpu, cpu = MLMap(cfg, max_blocks=2048, record_awareness=True), OracleMap(cfg)
    q, t = syn.static_pose()
    rng = np.random.default_rng(21)
    # sensor frame z forward: two adjacent range cells around 6 m (3*sigma > 1 there, so they spread into each other)
    n = 6000
    z = np.where(rng.random(n) < 0.5, 5.93, 6.03) + rng.uniform(-0.004, 0.004, n)
    pts = np.stack([rng.uniform(-0.003, 0.003, n), rng.uniform(-0.003, 0.003, n), z], axis=1)
    gpu.update_map_points(pts, q, t)
    cpu.update_points(pts, q, t)
    st = gpu.frame_stats()
    assert st["n_contrib_slots"] > 2 * 2048, st
    _awareness_equal(gpu, cpu)
    compare_maps(gpu.export_blocks(), cpu.export_blocks(), "heavy cell")


@pytest.mark.parametrize("n", [300, 900, 3000, 4096])
def test_heavy_cells_in_a_small_frame(mods, n):
    """A frame of at most 4 096 points on its own runs its cells' float chains inside k_rank<true> (no k_chain_lanes launch): cells with
    a few dozen contributions (half a wave ranks them), several hundred (the whole wave redoes them) and more than 1 024 (the kinds
    go through memory) — the same two adjacent range cells as above, which spread into each other."""
    MLMap, OracleMap = mods
    cfg = S1
    gpu, cpu = MLMap(cfg, max_blocks=2048, record_awareness=True), OracleMap(cfg)
    q, t = syn.static_pose()
    rng = np.random.default_rng(n)
    z = np.where(rng.random(n) < 0.5, 5.93, 6.03) + rng.uniform(-0.004, 0.004, n)
    pts = np.stack([rng.uniform(-0.003, 0.003, n), rng.uniform(-0.003, 0.003, n), z], axis=1)
    for k in range(2):  # (the second call replays the single-frame graph)
        gpu.update_map_points(pts, q, t)
        cpu.update_points(pts, q, t)
        _awareness_equal(gpu, cpu)
        compare_maps(gpu.export_blocks(), cpu.export_blocks(), f"heavy cells, {n} points, call {k}")
    assert gpu.frame_stats()["n_multi_cells"] >= 2


def test_argument_and_capacity_errors(mods, monkeypatch, knobs):
    """Error behaviour of the boundary: statuses, never exceptions or silent corruption."""
    import ctypes as C

    from mlmapping_amd.mlmap import MlmError, load_library

    MLMap, _ = mods
    L = load_library()
    assert L.mlm_create(None, None, 0, C.byref(C.c_void_p())) == -1  # MLM_ERR_INVALID
    bad = S1.with_(am_n_Rho=0)
    with pytest.raises(MlmError):
        MLMap(bad)
    with pytest.raises(MlmError):
        MLMap(S1, device=99)
    with pytest.raises(MlmError, match="UNSUPPORTED"):  # outside the supported envelope: said at creation, not at the first frame
        MLMap(S1.with_(am_n_Rho=1000, depth_noise_coe=1e-6), max_blocks=64, max_points=1000)
    knobs.set("pool_grow", "0")
    m = MLMap(S1, max_blocks=8, max_points=640 * 480)  # block pool far too small and not allowed to grow
    with pytest.raises(MlmError, match="CAPACITY"):
        m.update_map(syn.room_depth(S1), *syn.static_pose())
    knobs.set("pool_grow", "1")
    m2 = MLMap(SDEF, max_blocks=1024, max_points=1000)
    with pytest.raises(MlmError, match="CAPACITY"):
        m2.update_map(syn.room_depth(SDEF), *syn.static_pose())  # 230 400 pixels > max_points
    m2.update_map(syn.room_depth(SDEF), *syn.static_pose(), pixel_idx=np.arange(0, 230400, 400))  # 576 pixels: fine
    assert m2.frame_stats()["n_points"] == 576


def _fuzz_configurations(mods, seed, trials):
    MLMap, OracleMap = mods
    rng = np.random.default_rng(seed)
    for trial in range(trials):
        cfg, depths, pos = fuzz_trial(rng, trial)
        gpu, cpu = MLMap(cfg, max_blocks=4096, max_points=320 * 240, record_awareness=True), OracleMap(cfg)
        for k, depth in enumerate(depths):
            q, t = syn.random_poses(3, seed=trial)[k]
            cpu.update_depth(depth, q, t)
            gpu.update_map(depth, q, t)  # (tiny blocks over a long range need more than the 4096 initial blocks: the pool grows)
            _awareness_equal(gpu, cpu)
            compare_maps(gpu.export_blocks(), cpu.export_blocks(), f"fuzz seed {seed} trial {trial} frame {k} cfg {cfg}")
            if cfg.use_exploration_frontiers:
                assert np.array_equal(gpu.export_frontier(), cpu.export_frontier()), f"fuzz seed {seed} trial {trial}: frontier"
        assert np.array_equal(gpu.getOccupancy(pos), cpu.getOccupancy(pos)), f"fuzz seed {seed} trial {trial}"
        gpu.close()


def test_random_configurations(mods):
    """Fuzz: random map geometries, noise levels, thresholds and camera models — the HIP path must track the oracle on
    all of them (sets and classes exact, log-odds bits).  MLM_FUZZ_SEED / MLM_FUZZ_TRIALS: longer or different runs of the same
    fuzz (default: 30 trials, seed 2024)."""
    import os

    _fuzz_configurations(mods, int(os.environ.get("MLM_FUZZ_SEED", "2024")), int(os.environ.get("MLM_FUZZ_TRIALS", "30")))


def test_random_configurations_fresh_seed(mods):
    """The same fuzz with a seed no earlier run has seen: derived from the kernel sources (bench.csrc_sha16), so every change of the
    code is fuzzed with inputs of its own — the fixed seeds above only say that nothing KNOWN broke.  The seed is printed and in
    every assertion message; MLM_FUZZ_SEED=<seed> MLM_FUZZ_TRIALS=10 replays it in test_random_configurations."""
    from bench import csrc_sha16

    seed = int(csrc_sha16(), 16) % (1 << 31)
    print("fresh fuzz seed", seed)
    _fuzz_configurations(mods, seed, 10)


def test_long_stream_cfg2(mods, oracle_jobs):
    """BASELINE config 2 at the length SURVEY §8d asks for: 1000 frames of the jittered room with a random SE(3) pose per frame
    in asynchronous mode (three slot sets in flight, speculative Stage B), against the oracle fed frame by frame — full map
    comparison (keys, classes, log-odds bits) every 50 frames; the pool starts at 64 blocks and grows on the way.  The oracle's
    two and a half minutes run in a process of their own that starts with the session (tests/oracle_worker.py); this test feeds
    the GPU and compares at every checkpoint as the oracle's maps arrive.  (MLM_LONG_STREAM_FRAMES overrides the length.)"""
    import os

    import torch

    from tests.oracle_worker import long_stream_inputs

    MLMap, OracleMap = mods
    cfg = S1
    n, B = int(os.environ.get("MLM_LONG_STREAM_FRAMES", "1000")) // 50 * 50, 25
    frames, q, t = long_stream_inputs(n)
    distinct = frames.shape[0]
    fetch = oracle_jobs["long_stream"]
    d_frames = torch.from_numpy(frames.view(np.int16)).cuda()
    torch.cuda.synchronize()
    fsz = cfg.width * cfg.height
    gpu = MLMap(cfg, max_blocks=64, max_batch=B)  # a pool of 64 blocks: it grows on the way (allocate_ram never refuses, map_local.h:215-231)
    gpu.set_async(True)
    worst, n_cpu_blocks = 0.0, 0
    for k0 in range(0, n, B):
        for k in range(k0, k0 + B):  # frames are cycled, so a batch is not contiguous in HBM: one call per frame ...
            gpu.update_map_dev(d_frames.data_ptr() + (k % distinct) * fsz * 2, cfg.width, cfg.height, q[k], t[k])
        if (k0 + B) % 50 == 0:
            c = fetch(f"ckpt_{k0 + B}.npz")
            d = compare_maps(gpu.export_blocks(), c, f"cfg2 stream after {k0 + B} frames")
            worst, n_cpu_blocks = max(worst, d["max_dodd"]), c["keys"].shape[0]
    # ... and the contiguous batch entry point on the first 25 frames of a second map
    g2 = MLMap(cfg, max_blocks=32768, max_batch=B)
    g2.set_async(True)
    for rep in range(min(4, n // B)):
        g2.update_map_batch_dev(d_frames.data_ptr(), B, cfg.width, cfg.height, q[rep * B:(rep + 1) * B], t[rep * B:(rep + 1) * B])
    d = compare_maps(g2.export_blocks(), fetch("batch_dev.npz"), "cfg2 batch_dev 100 frames")
    print("long stream worst |d odd|", worst, d)
    st = gpu.frame_stats()
    assert st["n_spec_replays"] + st["n_sector_fallbacks"] >= 1  # the emulated container did rehash on the way
    assert st["n_pool_grows"] >= 1 and st["block_capacity"] >= n_cpu_blocks > 64, st


def test_float_callback_between_packed_host_batches(mods):
    """Packed host batches (one upload into the slot set's batch image buffer), then the first 32FC1 callback (allocates the float
    conversion buffer), then packed host batches again — the batch image buffers must survive the float buffer's allocation
    (round 4: a misplaced loop freed them there and the next batch uploaded into freed memory)."""
    MLMap, OracleMap = mods
    libc = ctypes.CDLL("libc.so.6")
    cfg = SDEF
    frames = np.stack([f[0] for f in syn.stream(cfg, "room_jitter", "smooth", 8)])
    poses = [f[1] for f in syn.stream(cfg, "room_jitter", "smooth", 8)]
    q = np.stack([p[0] for p in poses])
    t = np.stack([p[1] for p in poses])
    gpu, cpu = MLMap(cfg, max_blocks=4096, max_batch=4), OracleMap(cfg)

    def batches():
        for k0 in (0, 4):
            gpu.update_map_batch(frames[k0:k0 + 4], q[k0:k0 + 4], t[k0:k0 + 4])
            for k in range(k0, k0 + 4):
                cpu.update_depth(frames[k], q[k], t[k])

    batches()
    args = dict(t_img=1.0, odom_p=t[0], odom_q=q[0], odom_v=[0.1, 0.0, 0.0], t_odom=1.0, imu_w=[0.0, 0.0, 0.1], t_imu=1.0, latency=0.01)
    f32 = frames[0].astype(np.float32) / 1000.0
    for sampled in (False, True):
        libc.srand(7)
        gpu.depth_odom_callback(f32, sampled=sampled, **args)
        libc.srand(7)
        cpu.depth_odom_callback(f32, sampled=sampled, **args)
    batches()
    compare_maps(gpu.export_blocks(), cpu.export_blocks(), "packed batches around a float callback")
    gpu.close()


@pytest.mark.parametrize("n", [(1 << 20) + 4097, (1 << 21) - 1, (1 << 21) + 4097])
def test_point_count_limits(mods, n):
    """The largest frame the sector path takes (2^21 - 1 points — a full-HD depth image: contribution counts are packed for that), one
    above it (the cell-table path) and one above round 4's limit of 2^20, as explicit points scattered through the awareness cylinder:
    same awareness sets and map."""
    MLMap, OracleMap = mods
    cfg = S1
    rng = np.random.default_rng(n)
    pts = np.stack([rng.uniform(-5.0, 5.0, n), rng.uniform(-1.5, 1.5, n), rng.uniform(0.2, 6.0, n)], axis=1)
    q, t = syn.random_poses(1, seed=9)[0]
    gpu, cpu = MLMap(cfg, max_blocks=8192, max_points=n, max_batch=1, record_awareness=True), OracleMap(cfg)
    gpu.update_map_points(pts, q, t)
    cpu.update_points(pts, q, t)
    _awareness_equal(gpu, cpu)
    compare_maps(gpu.export_blocks(), cpu.export_blocks(), f"{n} points")
    st = gpu.frame_stats()
    print(n, "points:", {k: st[k] for k in ("n_points", "n_hit_cells", "n_miss_cells", "n_sector_fallbacks", "n_device_atomics")})
    assert (st["n_device_atomics"] > 0) == (n < (1 << 21)), st  # (the sector path counts its atomics; the cell-table path reports 0)
    gpu.close()


def test_unusual_poses(mods):
    """Poses a careless caller may hand over: a quaternion that is not normalised (the reference's _transformVector is then a
    scaled, sheared map — so3.cpp:80-84 —; the binning kernel's matrix form must be the same map), a nearly zero one, a huge
    translation, NaN and infinity in the translation (every point leaves the map: static_cast<int>(NaN) semantics).  Awareness sets,
    out-of-range counts and maps equal the oracle's."""
    MLMap, OracleMap = mods
    cfg = S1
    img = syn.room_depth(cfg)
    q0, t0 = syn.random_poses(1, seed=4)[0]
    q0, t0 = np.asarray(q0, np.float64), np.asarray(t0, np.float64)
    cases = [("scaled quaternion", q0 * 1.3, t0), ("shrunk quaternion", q0 * 0.61, t0), ("tiny quaternion", q0 * 1e-9, t0),
             ("far away", q0, t0 + np.array([3.0e5, -2.0e5, 1.0e3])), ("NaN translation", q0, np.array([np.nan, 0.0, 1.0])),
             ("infinite translation", q0, np.array([0.0, np.inf, 1.0])), ("ordinary", q0, t0)]
    gpu, cpu = MLMap(cfg, max_blocks=16384, record_awareness=True), OracleMap(cfg)
    for name, q, t in cases:
        gpu.update_map(img, q, t)
        cpu.update_depth(img, q, t)
        _awareness_equal(gpu, cpu)
        compare_maps(gpu.export_blocks(), cpu.export_blocks(), name)
    gpu.close()


@pytest.mark.parametrize("explore", [False, True])
def test_sampled_callbacks_between_async_batches(mods, explore):
    """The callback's sampler hands its samples to the kernels in a pinned buffer of the handle (no copy): sampled callbacks
    interleaved with asynchronous batches — the buffer is rewritten by the next callback only after the previous frame has read
    it — in the default mode and in frontier mode (nothing speculative there, another submission path)."""
    MLMap, OracleMap = mods
    libc = ctypes.CDLL("libc.so.6")
    cfg = SDEF.with_(use_exploration_frontiers=True) if explore else SDEF
    n = 12
    frames = np.stack([img for img, _ in syn.stream(cfg, "room_jitter", "smooth", n)])
    poses = syn.smooth_trajectory(n, 8)
    q = np.stack([p[0] for p in poses])
    t = np.stack([p[1] for p in poses])
    gpu, cpu = MLMap(cfg, max_blocks=4096, max_batch=3), OracleMap(cfg)
    gpu.set_async(True)
    zero3 = [0.0, 0.0, 0.0]
    for k0 in range(0, n, 4):
        gpu.update_map_batch(frames[k0:k0 + 3], q[k0:k0 + 3], t[k0:k0 + 3])
        for k in range(k0, k0 + 3):
            cpu.update_depth(frames[k], q[k], t[k])
        for rep in range(2):  # two callbacks back to back: the second rewrites the staging buffer
            args = dict(t_img=1.0, odom_p=t[k0 + 3], odom_q=q[k0 + 3], odom_v=zero3, t_odom=1.0, imu_w=zero3, t_imu=1.0, latency=0.0)
            libc.srand(100 + k0 + rep)
            gpu.depth_odom_callback(frames[k0 + 3], sampled=True, **args)
            libc.srand(100 + k0 + rep)
            cpu.depth_odom_callback(frames[k0 + 3], sampled=True, **args)
    gpu.sync()
    compare_maps(gpu.export_blocks(), cpu.export_blocks(), f"callbacks between async batches (frontier mode: {explore})")
    if explore:
        assert np.array_equal(gpu.export_frontier(), cpu.export_frontier())
    gpu.close()


def test_empty_frames_everywhere(mods):
    """Frames with nothing in them — an all-zero depth image, an empty pixel list, a sampler that finds no valid pixel — alone, inside
    asynchronous batches between ordinary frames, and as the first frame of a fresh handle: each is a no-op for the map (it still
    clears the awareness containers, map_awareness.cpp:178), and the frames around it integrate as usual."""
    MLMap, OracleMap = mods
    libc = ctypes.CDLL("libc.so.6")
    cfg = SDEF
    n = 8
    frames = np.stack([img for img, _ in syn.stream(cfg, "room_jitter", "smooth", n)])
    frames[0] = 0
    frames[3] = 0
    frames[4] = 0
    poses = syn.smooth_trajectory(n, 5)
    q = np.stack([p[0] for p in poses])
    t = np.stack([p[1] for p in poses])
    gpu, cpu = MLMap(cfg, max_blocks=2048, max_batch=4, record_awareness=True), OracleMap(cfg)
    gpu.update_map(frames[0], q[0], t[0])                                   # first frame of the handle: empty
    cpu.update_depth(frames[0], q[0], t[0])
    _awareness_equal(gpu, cpu)
    gpu.set_async(True)
    for k0 in (0, 4):                                                       # empty frames inside batches
        gpu.update_map_batch(frames[k0:k0 + 4], q[k0:k0 + 4], t[k0:k0 + 4])
        for k in range(k0, k0 + 4):
            cpu.update_depth(frames[k], q[k], t[k])
    gpu.sync()
    gpu.set_async(False)
    compare_maps(gpu.export_blocks(), cpu.export_blocks(), "batches with empty frames")
    gpu.update_map(frames[1], q[1], t[1], pixel_idx=np.zeros(0, np.int32))  # empty pixel list
    cpu.update_depth_indexed(frames[1], np.zeros(0, np.int32), q[1], t[1])
    _awareness_equal(gpu, cpu)
    args = dict(t_img=1.0, odom_p=t[2], odom_q=q[2], odom_v=[0.0, 0.0, 0.0], t_odom=1.0, imu_w=[0.0, 0.0, 0.0], t_imu=1.0, latency=0.0)
    for img in (frames[3], frames[3].astype(np.float32)):                   # the sampler finds nothing (16UC1 and 32FC1)
        libc.srand(3)
        gpu.depth_odom_callback(img, sampled=True, **args)
        libc.srand(3)
        cpu.depth_odom_callback(img, sampled=True, **args)
        _awareness_equal(gpu, cpu)
    gpu.update_map(frames[5], q[5], t[5])                                   # ... and life goes on
    cpu.update_depth(frames[5], q[5], t[5])
    _awareness_equal(gpu, cpu)
    compare_maps(gpu.export_blocks(), cpu.export_blocks(), "after the empty frames")
    gpu.close()


def test_fast_bin_boundary_points(mods):
    """k_bin_sectors takes its bins from a cheap FP64 evaluation and certifies each with a margin; a wave with a lane nearer to a
    cell boundary than the margin evaluates the reference's own sequence (mlm_bin_point_fast, mlm_device.h).  Points aimed AT the
    boundaries — rho, phi and z of the awareness frame at integer multiples of the cell sizes plus an offset, mapped back into the
    sensor frame — under an identity-like and three random poses, in two rounds: offsets from 0 to 1e-9 of the coordinate (inside
    the margin: the reference's sequence must take over) and offsets from 1e-7 to 1e-4 (outside it, on either side of the
    boundary: the cheap evaluation decides and must be right).  Hit / miss cell sets, odds, out-of-range count and the map must
    equal the oracle's: a bin on the wrong side of a boundary moves a hit cell."""
    MLMap, OracleMap = mods
    cfg = S1
    rng = np.random.default_rng(31)
    poses = [syn.static_pose()] + syn.random_poses(3, seed=17)
    inside = np.array([0.0, 1e-16, -1e-16, 1e-15, -1e-15, 1e-14, -1e-14, 1e-13, -1e-13, 1e-12, -1e-12, 1e-11, -1e-11, 1e-10, -1e-10, 1e-9, -1e-9])
    outside = np.array([1e-7, -1e-7, 1e-6, -1e-6, 1e-5, -1e-5, 1e-4, -1e-4])
    n = 6000
    d_rho, d_phi, d_z = cfg.am_d_Rho, 2 * np.pi / cfg.n_phi, cfg.am_d_Z
    for k, (q, t) in enumerate(poses):
        for offs, name in ((inside, "inside the margin"), (outside, "outside the margin")):
            gpu, cpu = MLMap(cfg, max_blocks=8192, record_awareness=True), OracleMap(cfg)
            gpu.update_map_points(np.array([[0.0, 0.0, 1.0]]), q, t)  # (fixes T_ls for this pose)
            cpu.update_points(np.array([[0.0, 0.0, 1.0]]), q, t)
            q_ls, t_ls = gpu.T_ls()
            w, x, y, z = q_ls
            Rm = np.array([[1 - 2 * (y * y + z * z), 2 * (x * y - w * z), 2 * (x * z + w * y)],
                           [2 * (x * y + w * z), 1 - 2 * (x * x + z * z), 2 * (y * z - w * x)],
                           [2 * (x * z - w * y), 2 * (y * z + w * x), 1 - 2 * (x * x + y * y)]])
            rho = rng.integers(1, cfg.am_n_Rho + 3, n) * d_rho
            phi = rng.integers(0, cfg.n_phi + 1, n) * d_phi
            zz = (rng.integers(-cfg.n_z // 2 - 2, cfg.n_z // 2 + 3, n) + 0.5) * d_z  # (cell boundaries in z lie at (k + 1/2) dZ)
            which = rng.integers(0, 4, n)  # the coordinate(s) put on a boundary; the others land inside a cell
            o = offs[rng.integers(0, len(offs), n)]
            rho = np.where((which == 0) | (which == 3), rho * (1 + o), rho + rng.uniform(0.2, 0.8, n) * d_rho)
            phi = np.where((which == 1) | (which == 3), phi + o, phi + rng.uniform(0.2, 0.8, n) * d_phi)
            zz = np.where(which == 2, zz * (1 + o) + o, zz + rng.uniform(0.2, 0.8, n) * d_z)
            p_l = np.stack([rho * np.cos(phi), rho * np.sin(phi), zz], axis=1)
            p_s = (p_l - np.asarray(t_ls)) @ Rm  # R^T (p_l - t)
            gpu.update_map_points(p_s, q, t)
            cpu.update_points(p_s, q, t)
            _awareness_equal(gpu, cpu)
            compare_maps(gpu.export_blocks(), cpu.export_blocks(), f"boundary points {name}, pose {k}")
            st = gpu.frame_stats()
            waves = (n + 63) // 64
            print(f"boundary points {name}, pose {k}: waves on the reference's sequence: {st['n_bin_exact_waves']} of {waves}")
            if name.startswith("inside"):
                assert st["n_bin_exact_waves"] >= waves // 2, st  # (the round must reach the fall-back)
            else:
                assert st["n_bin_exact_waves"] <= waves // 4, st  # (... and this one the cheap evaluation)
            gpu.close()


def test_fast_bin_seeds(mods):
    """The binning kernel evaluates its FP64 chain with reciprocals and a reciprocal square root refined once, and certifies every
    bin with margins that assume those forms are good to 4e-12 (mlm_bin_point_fast, mlm_device.h).  Measured here on the device over
    2^26 values between 2^-40 and 2^40: the raw seeds and the refined forms."""
    MLMap, _ = mods
    gpu = MLMap(SDEF, max_blocks=256)
    e = gpu.debug_probe_seeds()
    print("seed errors: rcp %.3g -> %.3g, rsq %.3g -> %.3g" % tuple(e))
    assert e[0] < 2.0 ** -20 and e[2] < 2.0 ** -20, e   # what ONE refinement needs
    assert e[1] < 4e-12 and e[3] < 4e-12, e             # what the margins assume
    gpu.close()


def test_bench_batch64_parity(mods, oracle_jobs):
    """The configuration bench.py TIMES, compared with the oracle at its real size: 192 frames of the bench stream (64 distinct
    jittered room frames resident in HBM, random SE(3) poses) submitted as three asynchronous 64-frame contiguous
    mlm_integrate_depth_batch_dev calls on a handle with max_batch = 64 (three slot sets: all three batches in flight), from a
    pool of 64 blocks.  The 64-frame k_apply_tiles chain — one lane per frame bookkeeping, LDS sized by the batch's spread of
    z origins, voxels LDS-resident across 64 frames — is what no smaller batch exercises; the update is order dependent
    (map_local.cpp:147-207), so any slip in the frame order inside a batch shows up in the log-odds bits.  (The oracle's leg runs
    in a process of its own, started with the session: tests/oracle_worker.py.)"""
    import torch

    from bench import make_inputs

    MLMap, OracleMap = mods
    cfg, B, nb = S1, 64, 3
    fetch = oracle_jobs["batch64"]
    frames, q, t = make_inputs(cfg, B, B * nb, seed=42)
    d_frames = torch.from_numpy(frames.view(np.int16)).cuda()
    torch.cuda.synchronize()
    gpu = MLMap(cfg, max_blocks=64, max_points=cfg.width * cfg.height, max_batch=B)
    gpu.set_async(True)
    for j in range(nb):
        gpu.update_map_batch_dev(d_frames.data_ptr(), B, cfg.width, cfg.height, q[j * B:(j + 1) * B], t[j * B:(j + 1) * B])
    d = compare_maps(gpu.export_blocks(), fetch("after_192.npz"), "bench configuration: 3 x 64-frame async batches")
    st = gpu.frame_stats()
    print("batch64 parity", d, {k: st[k] for k in ("n_spec_replays", "n_sector_fallbacks", "n_pool_grows", "block_capacity")})
    assert d["bit_mismatch"] == 0 and st["n_pool_grows"] >= 1
    # the batches after the container has settled run speculatively (no host round trip inside a batch): one more batch on
    # the grown map, still bit-equal
    gpu.update_map_batch_dev(d_frames.data_ptr(), B, cfg.width, cfg.height, q[:B], t[:B])
    compare_maps(gpu.export_blocks(), fetch("after_256.npz"), "bench configuration: fourth batch")
    gpu.close()


def test_corridor_substitute(mods):
    """BASELINE config 5's corridor.bag is absent: the synthetic corridor (2 x 3 x 40 m box) with a recorded
    pose list through the C ABI substitutes.  Full comparison incl. the float log-odds (as odds, 1e-4), frame by frame."""
    MLMap, OracleMap = mods
    for cfg, poses, n in ((S1, "translating", 12), (S1, "smooth", 8), (SDEF, "smooth", 8)):
        gpu, cpu = MLMap(cfg, max_blocks=16384, record_awareness=True), OracleMap(cfg)
        for k, (img, (q, t)) in enumerate(syn.stream(cfg, "corridor", poses, n)):
            gpu.update_map(img, q, t)
            cpu.update_depth(img, q, t)
            _awareness_equal(gpu, cpu)
            d = compare_maps(gpu.export_blocks(), cpu.export_blocks(), f"corridor {poses} frame {k}")
        b = cpu.export_blocks()
        pos = voxel_centres(b, cfg, 50000)
        assert np.array_equal(gpu.getOccupancy(pos), cpu.getOccupancy(pos))
        assert np.abs(gpu.getOdd(pos) - cpu.getOdd(pos)).max() <= ODDS_TOL
        print("corridor", poses, d)
        gpu.close()


def test_no_raycasting(mods):
    """mlmapping_use_raycasting: false (config key a-0): hits only, no miss cells, every point counts as out of range
    (map_awareness.cpp:241,277)."""
    MLMap, OracleMap = mods
    for base in (S1, SDEF):
        cfg = base.with_(use_raycasting=False)
        gpu, cpu = MLMap(cfg, max_blocks=8192, record_awareness=True), OracleMap(cfg)
        for k, (img, (q, t)) in enumerate(syn.stream(cfg, "room_jitter", "smooth", 6)):
            gpu.update_map(img, q, t)
            cpu.update_depth(img, q, t)
            _awareness_equal(gpu, cpu)
            assert gpu.frame_stats()["n_miss_cells"] == 0
            compare_maps(gpu.export_blocks(), cpu.export_blocks(), f"no raycasting frame {k}")
        assert gpu.class_counts()["f"] == 0
        gpu.close()


@pytest.mark.parametrize("max_iter", [0, 1, 3, 5, 8])
def test_odd_grad_iterations(mods, max_iter):
    """getOddGrad(pos, max_iter) for iteration counts other than the default 5 (mlmap.h:237-295)."""
    MLMap, OracleMap = mods
    cfg = S1
    gpu, cpu = MLMap(cfg, max_blocks=8192), OracleMap(cfg)
    for img, (q, t) in syn.stream(cfg, "room_jitter", "smooth", 4):
        gpu.update_map(img, q, t)
        cpu.update_depth(img, q, t)
    b = cpu.export_blocks()
    rng = np.random.default_rng(max_iter)
    pos = np.concatenate([voxel_centres(b, cfg, 30000, seed=max_iter), rng.uniform(-3, 6, size=(10000, 3))])
    gg, cg = gpu.getOddGrad(pos, max_iter), cpu.getOddGrad(pos, max_iter)
    assert np.array_equal(gg == 0, cg == 0), "gradient found / not found differs"
    assert np.abs(gg - cg).max() <= 1e-4 * max(1.0, np.abs(cg).max())
    if max_iter == 0:
        assert not gg.any()


def test_odds_at_block_and_cell(mods):
    """float getOdd(const Vec3I &glb_id, size_t subbox_id), mlmap.h:227-235 — incl. absent blocks (0.5) and, in frontier
    mode, released blocks (element 0 answers)."""
    MLMap, OracleMap = mods
    for cfg in (S1, S1.with_(use_exploration_frontiers=True, subbox_n=5)):
        gpu, cpu = MLMap(cfg, max_blocks=16384), OracleMap(cfg)
        for img, (q, t) in syn.stream(cfg, "room_jitter", "smooth", 5):
            gpu.update_map(img, q, t)
            cpu.update_depth(img, q, t)
        b = cpu.export_blocks()
        rng = np.random.default_rng(1)
        sel = rng.integers(0, b["keys"].shape[0], 20000)
        glb = np.concatenate([b["keys"][sel], rng.integers(-60, 60, size=(5000, 3)).astype(np.int32)])
        sub = rng.integers(0, cfg.cells_per_block, glb.shape[0]).astype(np.int32)
        go, co = gpu.getOddAt(glb, sub), cpu.getOddAt(glb, sub)
        assert np.abs(go - co).max() <= ODDS_TOL
        assert np.array_equal(go == 0.5, co == 0.5)
        from mlmapping_amd.mlmap import MlmError
        with pytest.raises(MlmError, match="INVALID"):
            gpu.getOddAt(glb[:2], np.array([0, cfg.cells_per_block], dtype=np.int32))
        gpu.close()


def test_frontier_points_payload(mods):
    """/frontier PointCloud2 payload (rviz_vis.cpp:267-293): float centres of the frontier cells, bit-equal as a set."""
    MLMap, OracleMap = mods
    cfg = S1.with_(use_exploration_frontiers=True)
    gpu, cpu = MLMap(cfg, max_blocks=16384), OracleMap(cfg)
    for img, (q, t) in syn.stream(cfg, "room_jitter", "smooth", 5):
        gpu.update_map(img, q, t)
        cpu.update_depth(img, q, t)
    gp, cp = gpu.frontier_points(), cpu.frontier_points()
    assert gp.shape == cp.shape and cp.shape[0] > 100
    key = lambda a: a[np.lexsort((a[:, 2], a[:, 1], a[:, 0]))]
    assert np.array_equal(key(gp).view(np.uint32), key(cp).view(np.uint32))
    plain = MLMap(S1, max_blocks=1024)
    plain.update_map(syn.room_depth(S1), *syn.static_pose())
    assert plain.frontier_points().shape == (0, 3)  # no frontier sets without use_exploration_frontiers


def test_callback_nonfinite_depth(mods):
    """32FC1 pixels that are not valid ranges: +Inf (REP-117 "no return"), -Inf, NaN, values whose millimetres do not fit
    an int32 -> 0 (skipped); finite depths beyond 65.535 m saturate (cvRound + saturate_cast<ushort>, mlmap.cpp:482)."""
    MLMap, OracleMap = mods
    libc = ctypes.CDLL("libc.so.6")
    cfg = SDEF
    base = syn.room_depth(cfg).astype(np.float32) / 1000.0
    rng = np.random.default_rng(3)
    for sampled in (False, True):
        gpu, cpu = MLMap(cfg, max_blocks=4096, record_awareness=True), OracleMap(cfg)
        for k in range(4):
            depth = base.copy()
            for val in (np.inf, -np.inf, np.nan, 70.0, 3.0e6, -2.0, 65.5354):
                depth[rng.integers(0, cfg.height, 4000), rng.integers(0, cfg.width, 4000)] = val
            q, t = syn.smooth_trajectory(4, 9)[k]
            args = dict(t_img=5.0 + k / 30.0, odom_p=t, odom_q=q, odom_v=[0.1, 0.0, 0.0], t_odom=5.0 + k / 30.0, imu_w=[0.0, 0.0, 0.1],
                        t_imu=5.0 + k / 30.0, latency=0.0, sampled=sampled)
            libc.srand(7 + k)
            gpu.depth_odom_callback(depth, **args)
            libc.srand(7 + k)
            cpu.depth_odom_callback(depth, **args)
            _awareness_equal(gpu, cpu)
            compare_maps(gpu.export_blocks(), cpu.export_blocks(), f"non-finite depth sampled={sampled} frame {k}")


def test_callback_query_interleaving(mods):
    """Dense 32FC1 callback -> a query large enough to regrow the query buffers -> dense 32FC1 callback -> destroy: the
    handle's float staging buffer must survive the query (it used to be freed there and reused: use after free)."""
    MLMap, OracleMap = mods
    cfg = SDEF
    gpu, cpu = MLMap(cfg, max_blocks=4096), OracleMap(cfg)
    base = syn.room_depth(cfg).astype(np.float32) / 1000.0
    rng = np.random.default_rng(17)
    pos = rng.uniform([-3, -6, -1], [8, 6, 4], size=(2_000_000, 3))
    for k in range(3):
        depth = base + rng.uniform(0, 0.05, size=base.shape).astype(np.float32)
        q, t = syn.smooth_trajectory(3, 2)[k]
        args = dict(t_img=1.0 + k / 30.0, odom_p=t, odom_q=q, odom_v=[0.2, 0.0, 0.0], t_odom=1.0 + k / 30.0 - 0.003,
                    imu_w=[0.0, 0.1, 0.2], t_imu=1.0 + k / 30.0 - 0.001, latency=0.02, sampled=False)
        gpu.depth_odom_callback(depth, **args)
        cpu.depth_odom_callback(depth, **args)
        compare_maps(gpu.export_blocks(), cpu.export_blocks(), f"callback {k}")
        n = pos.shape[0] if k < 2 else 1000
        sel = pos[: n // (k + 1)] if k else pos[:5000]  # grows on the second round: 5 000 -> 1 000 000 positions
        assert np.array_equal(gpu.getOccupancy(sel), cpu.getOccupancy(sel)), f"queries after callback {k}"
        assert np.abs(gpu.getOdd(sel[:200000]) - cpu.getOdd(sel[:200000])).max() <= ODDS_TOL
    gpu.close()


@pytest.mark.parametrize("env", [{"MLM_SEC_FAIL_EVERY": "1", "MLM_SEC_BACKOFF": "0"}, {"MLM_SEC_FAIL_EVERY": "3", "MLM_SEC_TAB": "512", "MLM_SEC_BACKOFF": "0"},
                                 {"MLM_SEC_FAIL_EVERY": "4", "MLM_SEC_BACKOFF": "2"}, {"MLM_SECTORS": "0"},
                                 {"MLM_SEC_FAIL_EVERY": "3", "MLM_SEC_BACKOFF": "1", "MLM_LEAN_SLOTS": "0"},
                                 {"MLM_SEC_TAB": "512", "MLM_SEC_TAB_BIG": "1024", "MLM_SEC_BACKOFF": "0"}])
def test_sector_fallback_and_cell_table_path(mods, monkeypatch, knobs, env):
    """Stage A by azimuth sector falls back to the cell-table path frame by frame when a column overflows its LDS tables
    (forced here by shrinking them); MLM_SECTORS=0 runs the cell-table path alone.  Results must not change."""
    MLMap, OracleMap = mods
    for k, v in env.items():
        knobs.set(k[4:].lower(), v)
    cfg = S1
    n = 10
    frames = np.stack([img for img, _ in syn.stream(cfg, "room_jitter", "random", n)])
    poses = syn.random_poses(n, 42)
    q = np.stack([p[0] for p in poses])
    t = np.stack([p[1] for p in poses])
    gpu, cpu = MLMap(cfg, max_blocks=16384, max_batch=4, record_awareness=True), OracleMap(cfg)
    gpu.update_map_batch(frames[:7], q[:7], t[:7])  # 4 + 3
    for k in range(7):
        cpu.update_depth(frames[k], q[k], t[k])
    compare_maps(gpu.export_blocks(), cpu.export_blocks(), f"{env} batch")
    for k in range(7, n):
        gpu.update_map(frames[k], q[k], t[k])
        cpu.update_depth(frames[k], q[k], t[k])
        _awareness_equal(gpu, cpu)
        compare_maps(gpu.export_blocks(), cpu.export_blocks(), f"{env} frame {k}")
    st = gpu.frame_stats()
    if env.get("MLM_SEC_FAIL_EVERY") == "1":
        assert st["n_sector_fallbacks"] == n, st
    if env.get("MLM_SEC_FAIL_EVERY") == "3" and "MLM_LEAN_SLOTS" not in env:
        assert st["n_sector_fallbacks"] >= n // 3, st
    if "MLM_LEAN_SLOTS" in env:  # (full slots: every frame slot keeps cell-table state of its own)
        assert st["n_sector_fallbacks"] >= 1, st
    if env.get("MLM_SEC_FAIL_EVERY") == "4":  # after a fall-back the next batches skip the sector attempt, then it is retried
        assert 1 <= st["n_sector_fallbacks"] < n // 2, st
    if "MLM_SECTORS" in env:
        assert st["n_sector_fallbacks"] == 0
    gpu.close()


@pytest.mark.parametrize("env", [{}, {"MLM_SEC_FAIL_EVERY": "1", "MLM_SEC_BACKOFF": "0"}, {"MLM_SEC_FAIL_EVERY": "3", "MLM_SEC_BACKOFF": "0"},
                                 {"MLM_SEC_FAIL_EVERY": "4", "MLM_SEC_BACKOFF": "1"}, {"MLM_SECTORS": "0"},
                                 {"MLM_SEC_FAIL_EVERY": "3", "MLM_SEC_BACKOFF": "0", "MLM_LEAN_SLOTS": "0"}])
def test_frontier_mode_sector_path(mods, monkeypatch, knobs, env):
    """Frontier mode runs Stage A by azimuth sector too (insertion times of the miss cells kept in LDS); a frame whose sector
    tables overflow redoes Stage A on the cell-table path before anything that depends on the map is enqueued.  Single
    frames, batches and asynchronous batches against the oracle: map, frontier set, awareness lists."""
    MLMap, OracleMap = mods
    for k, v in env.items():
        knobs.set(k[4:].lower(), v)
    cfg = S1.with_(use_exploration_frontiers=True)
    n = 11
    frames = np.stack([img for img, _ in syn.stream(cfg, "room_jitter", "random", n)])
    poses = syn.random_poses(n, 7)
    q = np.stack([p[0] for p in poses])
    t = np.stack([p[1] for p in poses])
    gpu, cpu = MLMap(cfg, max_blocks=16384, max_batch=4, record_awareness=True), OracleMap(cfg)
    for k in range(3):  # single frames
        gpu.update_map(frames[k], q[k], t[k])
        cpu.update_depth(frames[k], q[k], t[k])
        _awareness_equal(gpu, cpu)
        compare_maps(gpu.export_blocks(), cpu.export_blocks(), f"{env} frame {k}")
        assert np.array_equal(gpu.export_frontier(), cpu.export_frontier()), f"{env} frontier, frame {k}"
    gpu.update_map_batch(frames[3:8], q[3:8], t[3:8])  # 4 + 1
    for k in range(3, 8):
        cpu.update_depth(frames[k], q[k], t[k])
    compare_maps(gpu.export_blocks(), cpu.export_blocks(), f"{env} batch")
    assert np.array_equal(gpu.export_frontier(), cpu.export_frontier()), f"{env} frontier after the batch"
    gpu.set_async(True)
    gpu.update_map_batch(frames[8:10], q[8:10], t[8:10])
    gpu.update_map(frames[10], q[10], t[10])
    for k in range(8, n):
        cpu.update_depth(frames[k], q[k], t[k])
    compare_maps(gpu.export_blocks(), cpu.export_blocks(), f"{env} async")
    assert np.array_equal(gpu.export_frontier(), cpu.export_frontier()), f"{env} frontier after the async batches"
    st = gpu.frame_stats()
    if not env or "MLM_SECTORS" in env:
        assert st["n_sector_fallbacks"] == 0, st
    if env.get("MLM_SEC_FAIL_EVERY") == "1":
        assert st["n_sector_fallbacks"] == n, st
    if env.get("MLM_SEC_FAIL_EVERY") == "3":
        assert st["n_sector_fallbacks"] == 4, st  # frames 0, 3, 6, 9
    if env.get("MLM_SEC_FAIL_EVERY") == "4":
        assert 1 <= st["n_sector_fallbacks"] <= 3, st
    gpu.close()


def test_async_replay_with_mixed_stage_a_paths(mods, monkeypatch, knobs):
    """Asynchronous submission keeps up to three batches in flight.  Here a cell-table batch (submitted while the sector path
    backs off after a forced overflow) is followed by sector batches, and a frame of the OLDER, cell-table batch turns out to
    need a rehash of the emulated hit container (its hit count jumps from ~3 k to ~14 k): the replay must finish every
    pending frame on the path its Stage A took (round-2 advisor finding: the sector frames were resubmitted through
    k_voxelize, which reads per-hit fields k_sector never writes)."""
    MLMap, OracleMap = mods
    knobs.set("sec_fail_every", "7")
    knobs.set("sec_backoff", "1")
    knobs.set("big_arm", "0")  # (a fall-back backs the sector path off right away instead of scheduling the large-table pass)
    cfg = S1
    n = 14
    frames = np.stack([img for img, _ in syn.stream(cfg, "room_jitter", "random", n)])
    frames[:3, 110:, :] = 0  # frames 0-2: only the top rows carry depth -> a small hit container
    poses = syn.random_poses(n, 11)
    q = np.stack([p[0] for p in poses])
    t = np.stack([p[1] for p in poses])
    gpu, cpu = MLMap(cfg, max_blocks=16384, max_batch=2), OracleMap(cfg)
    gpu.set_async(True)
    for k0 in range(0, n, 2):
        gpu.update_map_batch(frames[k0:k0 + 2], q[k0:k0 + 2], t[k0:k0 + 2])
    for k in range(n):
        cpu.update_depth(frames[k], q[k], t[k])
    compare_maps(gpu.export_blocks(), cpu.export_blocks(), "async, cell-table batch followed by sector batches")
    st = gpu.frame_stats()
    assert st["n_spec_replays"] >= 1 and st["n_sector_fallbacks"] >= 2, st
    gpu.close()


@pytest.mark.parametrize("d_sub", [0.3, 0.5])
def test_many_hits_per_voxel(mods, d_sub):
    """Voxels several awareness cells wide collect dozens of hit cells each: more than the seven direct hit slots of a voxel
    in the frame-local grid (list behind the last slot) and more than k_apply_frame orders in registers (selection from
    memory).  The order of a voxel's hits matters through the clamp at log_odds_max and the float additions."""
    MLMap, OracleMap = mods
    cfg = S1.with_(subbox_d_xyz=d_sub, subbox_n=4, lm_log_odds_max=2.5, lm_measurement_hit=0.4)
    gpu, cpu = MLMap(cfg, max_blocks=16384, max_batch=4, record_awareness=True), OracleMap(cfg)
    frames = list(syn.stream(cfg, "room_jitter", "random", 6))
    for k, (img, (q, t)) in enumerate(frames[:2]):
        gpu.update_map(img, q, t)
        cpu.update_depth(img, q, t)
        _awareness_equal(gpu, cpu)
        compare_maps(gpu.export_blocks(), cpu.export_blocks(), f"d_sub {d_sub} frame {k}")
        if k == 0:  # hit cells per voxel that received hits (those are the voxels above 0 after the first frame)
            ratio = gpu.frame_stats()["n_hit_cells"] / max(1, int((cpu.export_blocks()["log_odds"] > 0).sum()))
            print(f"d_sub {d_sub}: {ratio:.1f} hit cells per hit voxel")
            assert ratio > (5 if d_sub < 0.4 else 12), ratio
    imgs = np.stack([f[0] for f in frames[2:]])
    q = np.stack([f[1][0] for f in frames[2:]])
    t = np.stack([f[1][1] for f in frames[2:]])
    gpu.update_map_batch(imgs, q, t)
    for img, (qq, tt) in frames[2:]:
        cpu.update_depth(img, qq, tt)
    compare_maps(gpu.export_blocks(), cpu.export_blocks(), f"d_sub {d_sub} batch")
    st = gpu.frame_stats()
    assert st["n_sector_fallbacks"] == 0, st
    gpu.close()
