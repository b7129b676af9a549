"""Map configuration: the YAML keys the reference reads at init (mlmap.cpp:10-33,75-85).

Field names follow the reference's YAML keys with the ``mlmapping_`` prefix dropped
(``launch/config/config_sim.yaml:8-55``).  Presets are the sizes SURVEY.md §8a names:

* ``S1``   — BASELINE configs 1/2/4: 640x480, 0.1 m voxels (awareness values from
  ``launch/config/d435i_mit_flvis.yaml:8-13``, local values from ``config2.yaml:17-23`` with d_xyz 0.1).
* ``S3``   — BASELINE config 3: 1280x720, 0.05 m voxels.
* ``SDEF`` — the reference's shipped default ``config_sim.yaml`` (0.2 m, 500 sampled pixels).
"""
from __future__ import annotations

import ctypes
from dataclasses import dataclass, field, replace
from typing import List

# T_B_S of config_sim.yaml:51-55 (row-major 4x4)
T_BS_SIM = [0.0, 0.0, 1.0, 0.12,
            -1.0, 0.0, 0.0, 0.0,
            0.0, -1.0, 0.0, 0.0,
            0.0, 0.0, 0.0, 1.0]


@dataclass(frozen=True)
class MapConfig:
    # awareness map (map_awareness.cpp:19)
    am_d_Rho: float = 0.2
    am_d_Phi_deg: float = 5.0
    am_d_Z: float = 0.2
    am_n_Rho: int = 40
    am_n_Z_below: int = 20
    am_n_Z_over: int = 20
    use_raycasting: bool = True
    depth_noise_coe: float = 0.000001
    # local map (map_local.cpp:46)
    subbox_d_xyz: float = 0.2
    subbox_n: int = 10
    use_exploration_frontiers: bool = False
    lm_log_odds_min: float = -2.0
    lm_log_odds_max: float = 4.2
    lm_measurement_hit: float = 0.7
    lm_measurement_miss: float = -0.9
    lm_occupied_sh: float = 3.0
    # inflation
    inflate_n: int = 2
    inflate_global_n: int = 2
    apply_inflate: bool = True
    # sampler / camera
    sample_cnt: int = 500
    cam_cx: float = 320.0
    cam_cy: float = 180.0
    cam_fx: float = 347.99755859375
    cam_fy: float = 347.99755859375
    T_B_S: List[float] = field(default_factory=lambda: list(T_BS_SIM))
    # image size the preset is meant for (not a reference key; used by the synthetic harness)
    width: int = 640
    height: int = 360

    # ---- derived sizes (map_awareness.cpp:26-32) ----
    @property
    def n_phi(self) -> int:
        return int(360 / self.am_d_Phi_deg)

    @property
    def n_z(self) -> int:
        return self.am_n_Z_below + self.am_n_Z_over + 1

    @property
    def n_cells(self) -> int:
        return self.am_n_Rho * self.n_phi * self.n_z

    @property
    def cells_per_block(self) -> int:
        return self.subbox_n ** 3

    def with_(self, **kw) -> "MapConfig":
        return replace(self, **kw)


SDEF = MapConfig()

S1 = MapConfig(
    am_d_Rho=0.1, am_d_Phi_deg=1.0, am_d_Z=0.1, am_n_Rho=65, am_n_Z_below=20, am_n_Z_over=20,
    depth_noise_coe=0.00375,
    subbox_d_xyz=0.1, subbox_n=10, lm_log_odds_min=-2.0, lm_log_odds_max=4.2, lm_measurement_hit=0.7,
    lm_measurement_miss=-0.9, lm_occupied_sh=2.0, use_exploration_frontiers=False,
    cam_cx=320.0, cam_cy=240.0, cam_fx=385.0, cam_fy=385.0, width=640, height=480,
)

S3 = MapConfig(
    am_d_Rho=0.05, am_d_Phi_deg=0.5, am_d_Z=0.05, am_n_Rho=130, am_n_Z_below=40, am_n_Z_over=40,
    depth_noise_coe=0.00375,
    subbox_d_xyz=0.05, subbox_n=10, lm_log_odds_min=-2.0, lm_log_odds_max=4.2, lm_measurement_hit=0.7,
    lm_measurement_miss=-0.9, lm_occupied_sh=2.0, use_exploration_frontiers=False,
    cam_cx=640.0, cam_cy=360.0, cam_fx=640.0, cam_fy=640.0, width=1280, height=720,
)

# order-independent variant of S1 named in SURVEY.md §8d config 2 (config_sim.yaml:24,31)
S1_SIGMA0 = S1.with_(depth_noise_coe=0.000001, lm_occupied_sh=3.0)

PRESETS = {"S1": S1, "S3": S3, "SDEF": SDEF, "S1_SIGMA0": S1_SIGMA0}


class CConfig(ctypes.Structure):
    """Binary layout of ``mlm_config`` (include/mlmap_hip.h); the test-side checker uses the same layout."""

    _fields_ = [
        ("am_d_rho", ctypes.c_double),
        ("am_d_phi_deg", ctypes.c_double),
        ("am_d_z", ctypes.c_double),
        ("am_n_rho", ctypes.c_int32),
        ("am_n_z_below", ctypes.c_int32),
        ("am_n_z_over", ctypes.c_int32),
        ("use_raycasting", ctypes.c_int32),
        ("depth_noise_coe", ctypes.c_double),
        ("subbox_d_xyz", ctypes.c_double),
        ("subbox_n", ctypes.c_int32),
        ("use_exploration_frontiers", ctypes.c_int32),
        ("log_odds_min", ctypes.c_double),
        ("log_odds_max", ctypes.c_double),
        ("measurement_hit", ctypes.c_double),
        ("measurement_miss", ctypes.c_double),
        ("occupied_sh", ctypes.c_double),
        ("inflate_n", ctypes.c_int32),
        ("inflate_global_n", ctypes.c_int32),
        ("apply_inflate", ctypes.c_int32),
        ("sample_cnt", ctypes.c_int32),
        ("cam_cx", ctypes.c_double),
        ("cam_cy", ctypes.c_double),
        ("cam_fx", ctypes.c_double),
        ("cam_fy", ctypes.c_double),
        ("T_bs", ctypes.c_double * 16),
    ]


def to_c(cfg: MapConfig) -> CConfig:
    c = CConfig()
    c.am_d_rho = cfg.am_d_Rho
    c.am_d_phi_deg = cfg.am_d_Phi_deg
    c.am_d_z = cfg.am_d_Z
    c.am_n_rho = cfg.am_n_Rho
    c.am_n_z_below = cfg.am_n_Z_below
    c.am_n_z_over = cfg.am_n_Z_over
    c.use_raycasting = int(cfg.use_raycasting)
    c.depth_noise_coe = cfg.depth_noise_coe
    c.subbox_d_xyz = cfg.subbox_d_xyz
    c.subbox_n = cfg.subbox_n
    c.use_exploration_frontiers = int(cfg.use_exploration_frontiers)
    c.log_odds_min = cfg.lm_log_odds_min
    c.log_odds_max = cfg.lm_log_odds_max
    c.measurement_hit = cfg.lm_measurement_hit
    c.measurement_miss = cfg.lm_measurement_miss
    c.occupied_sh = cfg.lm_occupied_sh
    c.inflate_n = cfg.inflate_n
    c.inflate_global_n = cfg.inflate_global_n
    c.apply_inflate = int(cfg.apply_inflate)
    c.sample_cnt = cfg.sample_cnt
    c.cam_cx = cfg.cam_cx
    c.cam_cy = cfg.cam_cy
    c.cam_fx = cfg.cam_fx
    c.cam_fy = cfg.cam_fy
    for i, v in enumerate(cfg.T_B_S):
        c.T_bs[i] = v
    return c


def from_yaml(path: str, width: int = 640, height: int = 480) -> MapConfig:
    """Read the keys the reference reads (``yamlRead.h:7-48``) from one of its config files."""
    import yaml

    with open(path) as f:
        y = yaml.safe_load(f)
    g = lambda k, d=None: y.get("mlmapping_" + k, y.get(k, d))
    return MapConfig(
        am_d_Rho=float(g("am_d_Rho")), am_d_Phi_deg=float(g("am_d_Phi_deg")), am_d_Z=float(g("am_d_Z")),
        am_n_Rho=int(g("am_n_Rho")), am_n_Z_below=int(g("am_n_Z_below")), am_n_Z_over=int(g("am_n_Z_over")),
        use_raycasting=bool(g("use_raycasting", True)), depth_noise_coe=float(g("depth_noise_coe")),
        subbox_d_xyz=float(g("subbox_d_xyz")), subbox_n=int(g("subbox_n")),
        use_exploration_frontiers=bool(g("use_exploration_frontiers", False)),
        lm_log_odds_min=float(g("lm_log_odds_min")), lm_log_odds_max=float(g("lm_log_odds_max")),
        lm_measurement_hit=float(g("lm_measurement_hit")), lm_measurement_miss=float(g("lm_measurement_miss")),
        lm_occupied_sh=float(g("lm_occupied_sh")), inflate_n=int(g("inflate_n", 2)),
        inflate_global_n=int(g("inflate_global_n", 2)), apply_inflate=bool(g("apply_inflate", False)),
        sample_cnt=int(g("sample_cnt", 500)), cam_cx=float(g("cam_cx")), cam_cy=float(g("cam_cy")),
        cam_fx=float(g("cam_fx")), cam_fy=float(g("cam_fy")), T_B_S=[float(v) for v in y["T_B_S"]],
        width=width, height=height,
    )
