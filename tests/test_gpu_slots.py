"""Frame slots sized by need: the per-frame lists start at what camera frames need instead of the worst case
("every awareness cell a multi-kind hit"), are enlarged at a drained point when a frame needs more (sector_overflow 3 ->
grow_slots -> the frame's Stage A runs again on the sector path), and several handles fit one GPU."""
import numpy as np
import pytest

from mlmapping_amd import synthetic as syn
from mlmapping_amd.config import S1, S3
from tests.util import compare_maps, fuzz_trial

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def mods():
    from mlmapping_amd.mlmap import MLMap
    from oracle.binding import OracleMap

    return MLMap, OracleMap


def test_slot_footprint_and_five_handles_on_one_gpu(mods):
    """four S1 handles at batch 16 + one S3 handle alive together, fed interleaved, each equal to its oracle; S1 slot <= 0.1 GB, S3 <= 0.7 GB"""
    MLMap, OracleMap = mods
    hs = [MLMap(S1, max_blocks=2048, max_points=640 * 480, max_batch=16) for _ in range(4)]
    h3 = MLMap(S3, max_blocks=4096, max_points=1280 * 720, max_batch=8)
    per_slot_s1 = hs[0].frame_stats()["device_bytes"] / (3 * 16)  # (an upper bound: the map and its tables are in the numerator too)
    per_slot_s3 = h3.frame_stats()["device_bytes"] / (3 * 8)
    print(f"S1: {per_slot_s1 / 1e9:.3f} GB per frame slot, S3: {per_slot_s3 / 1e9:.3f} GB")
    assert per_slot_s1 <= 0.1e9 and per_slot_s3 <= 0.7e9
    cps = [OracleMap(S1) for _ in range(4)]
    c3 = OracleMap(S3)
    for g in hs:
        g.set_async(True)
    streams = [list(syn.stream(S1, "room_jitter", "random", 32, seed=50 + i)) for i in range(4)]
    f3 = list(syn.stream(S3, "room_jitter", "random", 3, seed=9))
    for k0 in (0, 16):
        for i, g in enumerate(hs):
            fr = streams[i][k0:k0 + 16]
            g.update_map_batch(np.stack([f[0] for f in fr]), np.stack([f[1][0] for f in fr]), np.stack([f[1][1] for f in fr]))
        img, (q, t) = f3[k0 // 16]
        h3.update_map(img, q, t)
        c3.update_depth(img, q, t)
    for i, g in enumerate(hs):
        for img, (q, t) in streams[i]:
            cps[i].update_depth(img, q, t)
        compare_maps(g.export_blocks(), cps[i].export_blocks(), f"S1 handle {i}")
        assert g.frame_stats()["n_sector_fallbacks"] == 0
    compare_maps(h3.export_blocks(), c3.export_blocks(), "S3 handle")
    st3 = h3.frame_stats()
    assert st3["n_sector_fallbacks"] == 0 and st3["n_slot_grows"] == 0, st3  # (a camera frame fits the initial lists)
    for g in hs + [h3]:
        g.close()


@pytest.mark.parametrize("mode", ["single frames", "async batches"])
def test_slots_grow_when_a_scene_needs_more(mods, mode):
    """the scatter scene (every pixel in a cell of its own: 143 k hit cells per frame) overruns the initial lists: they are enlarged,
    the frame's Stage A runs again on the sector path — no cell-table fall-back — and the maps stay equal"""
    MLMap, OracleMap = mods
    cfg = S1
    gpu, cpu = MLMap(cfg, max_blocks=8192, max_points=640 * 480, max_batch=4, record_awareness=True), OracleMap(cfg)
    b0 = gpu.frame_stats()["device_bytes"]
    frames = list(syn.stream(cfg, "scatter", "smooth", 12))
    if mode == "single frames":
        for k, (img, (q, t)) in enumerate(frames[:6]):
            gpu.update_map(img, q, t)
            cpu.update_depth(img, q, t)
            if k in (0, 5):
                gc, go, _ = gpu.awareness_hits()
                cc, co = cpu.hit_cells_sorted()
                assert np.array_equal(gc, cc) and np.array_equal(go.view(np.uint32), co.view(np.uint32))
                assert np.array_equal(gpu.awareness_misses(), np.sort(cpu.misses()).astype(np.int64))
    else:
        gpu.set_async(True)
        room = list(syn.stream(cfg, "room_jitter", "smooth", 4))
        seq = room + frames  # (a first batch that fits, then batches in flight when the lists run out)
        for k0 in range(0, len(seq), 4):
            fr = seq[k0:k0 + 4]
            gpu.update_map_batch(np.stack([f[0] for f in fr]), np.stack([f[1][0] for f in fr]), np.stack([f[1][1] for f in fr]))
        for img, (q, t) in seq:
            cpu.update_depth(img, q, t)
    compare_maps(gpu.export_blocks(), cpu.export_blocks(), f"scatter scene, {mode}")
    st = gpu.frame_stats()
    print(st)
    assert st["n_slot_grows"] >= 1 and st["n_sector_fallbacks"] == 0, st
    assert st["device_bytes"] > b0


def test_worst_case_slots_by_knob(mods, knobs):
    """knob need_slots = 0: every list at its worst case as before round 5 — same maps, three times the memory"""
    MLMap, OracleMap = mods
    cfg = S1
    small = MLMap(cfg, max_blocks=2048, max_points=640 * 480, max_batch=2)
    knobs.set("need_slots", 0)
    big, cpu = MLMap(cfg, max_blocks=2048, max_points=640 * 480, max_batch=2), OracleMap(cfg)
    assert big.frame_stats()["device_bytes"] > 2 * small.frame_stats()["device_bytes"]
    for img, (q, t) in syn.stream(cfg, "scatter", "smooth", 3):
        big.update_map(img, q, t)
        small.update_map(img, q, t)
        cpu.update_depth(img, q, t)
    compare_maps(big.export_blocks(), cpu.export_blocks(), "worst-case slots")
    compare_maps(small.export_blocks(), cpu.export_blocks(), "need-sized slots")
    assert big.frame_stats()["n_slot_grows"] == 0 and small.frame_stats()["n_slot_grows"] >= 1


def test_rerun_with_a_column_waiting_for_the_large_table(mods):
    """A frame of speckle whose 36 k hit cells overrun the initial lists (sector_overflow 3) while one crowded column is still waiting
    for the large-table pass (sector_overflow 1, the pass not scheduled): that column had kept its chunk counts, and the rerun of
    Stage A binned its points on top of them — every point of the column counted twice (fuzz seed 4242, trial 136: 0.999999 where
    the reference has 0.999).  The rerun starts from cleared column lists; the frames after it in the same slot as well."""
    MLMap, OracleMap = mods
    rng = np.random.default_rng(4242)
    for trial in range(137):  # (a trial's inputs follow the earlier trials' draws)
        cfg, depths, _ = fuzz_trial(rng, trial)
    assert cfg.am_n_Rho == 65 and cfg.am_d_Phi_deg == 0.5 and cfg.subbox_n == 4
    gpu, cpu = MLMap(cfg, max_blocks=4096, max_points=320 * 240, record_awareness=True), OracleMap(cfg)
    for k, depth in enumerate(depths):
        q, t = syn.random_poses(3, seed=136)[k]
        cpu.update_depth(depth, q, t)
        gpu.update_map(depth, q, t)
        gc, go, _ = gpu.awareness_hits()
        cc, co = cpu.hit_cells_sorted()
        assert np.array_equal(gc, cc)
        assert np.array_equal(go.view(np.uint32), co.view(np.uint32)), f"frame {k}: hit odds differ"
        compare_maps(gpu.export_blocks(), cpu.export_blocks(), f"speckle frame {k}")
    st = gpu.frame_stats()
    assert st["n_slot_grows"] >= 1 and st["n_sector_fallbacks"] == 0, st


def test_crowded_scene_then_camera_frames_then_crowded_again(mods):
    """a scene that overflows the small cell table in nearly every column (large-table pass armed, the table widened after two
    confirmed batches), then camera frames (the pass disarms), then the crowded scene again — frame by frame and in asynchronous
    batches, camera and scatter frames mixed inside a batch: the map equals the oracle's after every phase"""
    MLMap, OracleMap = mods
    cfg = S1
    gpu, cpu = MLMap(cfg, max_blocks=8192, max_points=640 * 480, max_batch=4, record_awareness=True), OracleMap(cfg)
    scatter = list(syn.stream(cfg, "scatter", "smooth", 14))
    room = list(syn.stream(cfg, "room_jitter", "random", 8, seed=3))

    def single(frames, what):
        for img, (q, t) in frames:
            gpu.update_map(img, q, t)
            cpu.update_depth(img, q, t)
            gc, go, _ = gpu.awareness_hits()
            cc, co = cpu.hit_cells_sorted()
            assert np.array_equal(gc, cc) and np.array_equal(go.view(np.uint32), co.view(np.uint32)), what
        compare_maps(gpu.export_blocks(), cpu.export_blocks(), what)

    def batch(frames, what):
        for k0 in range(0, len(frames), 4):
            fr = frames[k0:k0 + 4]
            gpu.update_map_batch(np.stack([f[0] for f in fr]), np.stack([f[1][0] for f in fr]), np.stack([f[1][1] for f in fr]))
        for img, (q, t) in frames:
            cpu.update_depth(img, q, t)
        compare_maps(gpu.export_blocks(), cpu.export_blocks(), what)

    single(scatter[:4], "scatter, frame by frame")
    single(room[:3], "camera frames after the crowded scene")
    gpu.set_async(True)
    batch(scatter[4:12], "scatter, asynchronous batches")
    batch(room[3:8] + scatter[12:14], "camera frames and scatter frames mixed in the batches")
    assert gpu.frame_stats()["n_sector_fallbacks"] == 0
