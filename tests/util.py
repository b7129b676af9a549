"""Shared comparison helpers for the parity tests."""
import numpy as np

ODDS_TOL = 1e-4  # BASELINE.json north_star: "within 1e-4 on the float odds"


def odds_of(log_odds: np.ndarray) -> np.ndarray:
    """logit_inv (mlmap.h:40) in double, as getOdd computes before narrowing."""
    p = np.power(10.0, log_odds.astype(np.float64))
    return p / (1.0 + p)


def compare_maps(g: dict, c: dict, what: str = "", exact: bool = True) -> dict:
    """GPU block dump vs oracle block dump (both sorted by key).  Integer/byte facts must be identical; the float
    log-odds are compared as odds with the 1e-4 tolerance of the north star AND, with `exact` (the default: the device
    evaluates the host libm's log10f bit for bit, mlm_glibc_log10f, and adds the increments in the reference's order),
    as float bits: not one voxel may differ.  Returns diagnostics."""
    assert g["keys"].shape == c["keys"].shape, f"{what}: block count {g['keys'].shape[0]} vs {c['keys'].shape[0]}"
    assert np.array_equal(g["keys"], c["keys"]), f"{what}: block key sets differ"
    assert np.array_equal(g["collapsed"], c["collapsed"]), f"{what}: released (collapsed) block sets differ"
    if c["collapsed"].any():
        # a released block keeps element 0 only in the reference (vectors resized to 1, map_local.cpp:221-226)
        g = {k: (v.copy() if k in ("occ", "log_odds", "infl") else v) for k, v in g.items()}
        col = c["collapsed"].astype(bool)
        for k in ("occ", "log_odds", "infl"):
            g[k][col, 1:] = c[k][col, 1:]
    occ_bad = int((g["occ"] != c["occ"]).sum())
    assert occ_bad == 0, f"{what}: {occ_bad} cells differ in occupancy class"
    dodd = np.abs(odds_of(g["log_odds"]) - odds_of(c["log_odds"]))
    assert dodd.max() <= ODDS_TOL, f"{what}: max |d odd| = {dodd.max():.3e}"
    dl = np.abs(g["log_odds"].astype(np.float64) - c["log_odds"].astype(np.float64))
    if exact:
        nb = int((g["log_odds"].view(np.uint32) != c["log_odds"].view(np.uint32)).sum())
        assert nb == 0, f"{what}: {nb} of {g['log_odds'].size} voxels differ in log-odds bits (max |dL| {dl.max():.3e})"
    return {"blocks": int(g["keys"].shape[0]), "max_dodd": float(dodd.max()), "max_dL": float(dl.max()),
            "bit_mismatch": int((g["log_odds"].view(np.uint32) != c["log_odds"].view(np.uint32)).sum()),
            "cells": int(g["log_odds"].size)}


def voxel_centres(b: dict, cfg, limit: int = 200000, seed: int = 0) -> np.ndarray:
    """World centres of (a sample of) the allocated voxels: subbox_id2xyz_glb_vec (map_local.h:208-213)."""
    n = cfg.subbox_n
    keys = b["keys"].astype(np.float64)
    ids = np.arange(cfg.cells_per_block)
    cz, cy, cx = ids // (n * n), (ids // n) % n, ids % n
    d = cfg.subbox_d_xyz
    c = np.stack([cx, cy, cz], axis=1).astype(np.float64)
    pos = (keys[:, None, :] * (d * n) + c[None, :, :] * d + d * 0.5).reshape(-1, 3)
    if pos.shape[0] > limit:
        pos = pos[np.random.default_rng(seed).choice(pos.shape[0], limit, replace=False)]
    return pos


def fuzz_trial(rng, trial: int):
    """The inputs of trial `trial` of tests/test_gpu_parity.py::test_random_configurations drawn from `rng` (a trial's draws follow
    the previous trial's: replaying trial k of a seed means drawing trials 0 .. k): a random map geometry / noise level / thresholds /
    camera model, three depth images (speckle, a smooth surface, speckle) and 20 000 query positions."""
    from mlmapping_amd.config import S1

    d = float(rng.choice([0.05, 0.1, 0.15, 0.2, 0.25]))
    cfg = S1.with_(
        am_d_Rho=d, am_d_Phi_deg=float(rng.choice([0.5, 1.0, 2.0, 3.0, 5.0])), am_d_Z=float(rng.choice([d, 2 * d, 0.5 * d])),
        am_n_Rho=int(rng.integers(20, 100)), am_n_Z_below=int(rng.integers(5, 30)), am_n_Z_over=int(rng.integers(5, 30)),
        depth_noise_coe=float(rng.choice([1e-6, 0.001, 0.00375, 0.008])),
        subbox_d_xyz=float(rng.choice([d, 2 * d, 0.5 * d])), subbox_n=int(rng.choice([4, 5, 8, 10, 16])),
        lm_log_odds_min=float(rng.uniform(-3, -1)), lm_log_odds_max=float(rng.uniform(3, 5)),
        lm_measurement_miss=float(rng.uniform(-1.2, -0.3)), lm_occupied_sh=float(rng.uniform(1.0, 3.0)),
        use_exploration_frontiers=bool(trial % 3 == 2),
        cam_fx=float(rng.uniform(150, 400)), cam_fy=float(rng.uniform(150, 400)), cam_cx=163.3, cam_cy=117.9,
        width=320, height=240)
    # keep the noise spread inside the reference's 21-row odds table (3*sigma <= 10)
    if 3 * cfg.depth_noise_coe * (cfg.am_n_Rho * cfg.am_d_Rho) ** 2 / cfg.am_d_Rho > 10:
        cfg = cfg.with_(depth_noise_coe=1e-6)
    depths = []
    for k in range(3):
        depth = rng.integers(300, int(1000 * cfg.am_n_Rho * cfg.am_d_Rho * 1.3), size=(240, 320)).astype(np.uint16)
        depth[rng.random((240, 320)) < 0.02] = 0
        if k == 1:  # a smooth surface as well as speckle
            depth[:] = (1000 * 0.6 * cfg.am_n_Rho * cfg.am_d_Rho + 200 * np.sin(np.arange(320) / 25.0)[None, :]).astype(np.uint16)
        depths.append(depth)
    pos = rng.uniform(-8, 8, size=(20000, 3))
    return cfg, depths, pos
