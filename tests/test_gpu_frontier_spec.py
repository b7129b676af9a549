"""Frontier mode, one frame per synchronous call: the map-dependent kernels are enqueued before the host has seen the frame's counts
and guard themselves (explore_stage_bc_spec, MlmDev::spec_on).  The three settings of the knob — never, normal, "every frame misses
the speculation" — must leave the same map and frontier as the oracle: the way out (the kernels do nothing) and the way back (the
general path with the counts on the host) are the paths under test."""
import numpy as np
import pytest

from mlmapping_amd import synthetic as syn
from mlmapping_amd.config import CONFIG2_YAML, S1
from tests.util import compare_maps

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("ex_spec", [0, 1, 2])
@pytest.mark.parametrize("which", ["S1 dense + sampled", "config2.yaml callback"])
def test_speculative_frontier_frames(knobs, ex_spec, which):
    from mlmapping_amd.mlmap import MLMap
    from oracle.binding import OracleMap

    knobs.set("ex_spec", ex_spec)
    n = 14
    if which.startswith("S1"):
        cfg = S1.with_(use_exploration_frontiers=True)
        gpu, cpu = MLMap(cfg, max_blocks=64, max_points=cfg.width * cfg.height, max_batch=2), OracleMap(cfg)
        rng = np.random.default_rng(5)
        for k, (img, (q, t)) in enumerate(syn.stream(cfg, "room_jitter", "random", n, seed=21)):
            if k % 3 == 1:  # the reference's 500-sample pattern through the pixel list
                pix = (rng.integers(0, cfg.height, 500) * cfg.width + rng.integers(0, cfg.width, 500)).astype(np.int32)
                gpu.update_map(img, q, t, pixel_idx=pix)
                cpu.update_depth_indexed(img, pix, q, t)
            elif k == 5:  # an empty frame: no hit, no miss cell
                z = np.zeros_like(img)
                gpu.update_map(z, q, t)
                cpu.update_depth(z, q, t)
            else:
                gpu.update_map(img, q, t)
                cpu.update_depth(img, q, t)
            if k % 4 == 3 or k == n - 1:
                compare_maps(gpu.export_blocks(), cpu.export_blocks(), f"{which}, ex_spec {ex_spec}, frame {k}")
                assert np.array_equal(gpu.export_frontier(), cpu.export_frontier())
    else:
        import ctypes

        libc = ctypes.CDLL("libc.so.6")
        cfg = CONFIG2_YAML
        gpu, cpu = MLMap(cfg, max_blocks=256, max_points=cfg.width * cfg.height, max_batch=2), OracleMap(cfg)
        base = syn.room_depth(cfg)
        traj = syn.smooth_trajectory(n, 3)
        for k in range(n):
            depth = syn.jitter_depth(base, k, seed=9).astype(np.float32) / 1000.0
            q, t = traj[k]
            args = dict(t_img=2.0 + k / 30.0, odom_p=t, odom_q=q, odom_v=[0.3, 0.0, -0.1], t_odom=2.0 + k / 30.0 - 0.004, imu_w=[0.0, 0.2, 0.1],
                        t_imu=2.0 + k / 30.0 - 0.002, latency=cfg.camera2odom_latency, sampled=True)
            libc.srand(77 + k)
            tg = gpu.depth_odom_callback(depth, **args)
            libc.srand(77 + k)
            tc = cpu.depth_odom_callback(depth, **args)
            assert np.array_equal(tg, tc)
            if k % 3 == 2:
                gpu.inflate_map(t)
                cpu.inflate_map(t)
        compare_maps(gpu.export_blocks(), cpu.export_blocks(), f"{which}, ex_spec {ex_spec}")
        assert np.array_equal(gpu.export_frontier(), cpu.export_frontier())
    st = gpu.frame_stats()
    if ex_spec == 0:
        assert st["n_spec_replays"] == 0, st
    elif ex_spec == 2:
        assert st["n_spec_replays"] >= n - 3, st  # (every frame with a hit or a miss cell, once the containers have their first buckets)
    else:
        assert st["n_spec_replays"] <= n // 2, st  # (only frames in which an emulated container grows: the first one sizes both here)
    gpu.close()
