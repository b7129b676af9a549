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
    # pose latency compensation of the depth/odom callback (mlmap.cpp:12,485-498); handed to mlm_integrate_callback per call
    camera2odom_latency: float = 0.001
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

# the two shipped config files that carry the current keys, verbatim (tests/test_config_yaml.py holds them to the files):
# launch/config/config_sim.yaml:8-55 is SDEF (640x360 gazebo camera); launch/config/config2.yaml:7-52 is the real-data set-up —
# frontier mode, inflation, a 424x240 depth stream (cx 212.65, cy 117.24, f 213.73), noise 0.00375, occupied above 2.0
CONFIG_SIM_YAML = SDEF
CONFIG2_YAML = MapConfig(
    am_d_Rho=0.20, am_d_Phi_deg=5.0, am_d_Z=0.20, am_n_Rho=40, am_n_Z_below=20, am_n_Z_over=20, use_raycasting=True,
    depth_noise_coe=0.00375, subbox_d_xyz=0.2, subbox_n=10, use_exploration_frontiers=True, lm_log_odds_min=-2.0,
    lm_log_odds_max=4.2, lm_measurement_hit=0.7, lm_measurement_miss=-0.9, lm_occupied_sh=2.0, inflate_n=2, inflate_global_n=2,
    apply_inflate=True, sample_cnt=500, cam_cx=212.6516265869, cam_cy=117.238, cam_fx=213.728866577, cam_fy=213.728866577,
    camera2odom_latency=0.085, width=424, height=240,
)

PRESETS = {"S1": S1, "S3": S3, "SDEF": SDEF, "S1_SIGMA0": S1_SIGMA0, "CONFIG_SIM_YAML": CONFIG_SIM_YAML, "CONFIG2_YAML": CONFIG2_YAML}


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


# every key mlmap::init_map reads from the YAML for this path (src/mlmap.cpp:10-33,75-85); the reference dies with a yaml-cpp
# exception when one is missing (yamlRead.h:25-48) — the three older shipped files (config.yaml, d435i_mit_flvis.yaml,
# l515_t265.yaml) lack the subbox / camera / inflate keys and cannot be loaded by the current reference either
YAML_KEYS = {
    "am_d_Rho": ("mlmapping_am_d_Rho", float), "am_d_Phi_deg": ("mlmapping_am_d_Phi_deg", float), "am_d_Z": ("mlmapping_am_d_Z", float),
    "am_n_Rho": ("mlmapping_am_n_Rho", int), "am_n_Z_below": ("mlmapping_am_n_Z_below", int), "am_n_Z_over": ("mlmapping_am_n_Z_over", int),
    "use_raycasting": ("mlmapping_use_raycasting", bool), "depth_noise_coe": ("mlmapping_depth_noise_coe", float),
    "subbox_d_xyz": ("mlmapping_subbox_d_xyz", float), "subbox_n": ("mlmapping_subbox_n", int),
    "use_exploration_frontiers": ("use_exploration_frontiers", bool),
    "lm_log_odds_min": ("mlmapping_lm_log_odds_min", float), "lm_log_odds_max": ("mlmapping_lm_log_odds_max", float),
    "lm_measurement_hit": ("mlmapping_lm_measurement_hit", float), "lm_measurement_miss": ("mlmapping_lm_measurement_miss", float),
    "lm_occupied_sh": ("mlmapping_lm_occupied_sh", float), "inflate_n": ("mlmapping_inflate_n", int),
    "inflate_global_n": ("mlmapping_inflate_global_n", int), "apply_inflate": ("mlmapping_apply_inflate", bool),
    "sample_cnt": ("mlmapping_sample_cnt", int), "cam_cx": ("mlmapping_cam_cx", float), "cam_cy": ("mlmapping_cam_cy", float),
    "cam_fx": ("mlmapping_cam_fx", float), "cam_fy": ("mlmapping_cam_fy", float),
    "camera2odom_latency": ("camera2odom_latency", float),
}


def from_params(y: dict, width: int, height: int) -> MapConfig:
    """A MapConfig from the parsed YAML mapping of one of the reference's config files (all keys required, like the reference)."""
    kw = {}
    for name, (key, typ) in YAML_KEYS.items():
        if key not in y:
            raise KeyError(f"{key}: the reference reads this key at init (src/mlmap.cpp:10-33,75-85) and the file does not have it")
        kw[name] = typ(y[key])
    if "T_B_S" not in y or len(y["T_B_S"]) != 16:
        raise KeyError("T_B_S: 16 row-major values (include/yamlRead.h:16-24)")
    return MapConfig(T_B_S=[float(v) for v in y["T_B_S"]], width=width, height=height, **kw)


def load_reference_yaml(path: str) -> dict:
    """Parse one of the reference's config files.  yaml-cpp 0.6.2 (the reference's loader) accepts the T_B_S matrix written as a
    flow sequence that starts in column 0 of the line AFTER its key (config_sim.yaml:51-55); strict YAML 1.1 loaders (PyYAML) do
    not — a continuation line must be indented deeper than its key.  The rows of an open flow sequence are therefore indented
    before parsing; nothing else is touched."""
    import yaml

    out, depth = [], 0
    with open(path) as f:
        for line in f:
            body = line.split("#", 1)[0]
            if depth > 0 or body.lstrip().startswith("["):
                line = "  " + line
            depth += body.count("[") - body.count("]")
            out.append(line)
    return yaml.safe_load("".join(out))


def from_yaml(path: str, width: int = 0, height: int = 0) -> MapConfig:
    """Read the keys the reference reads (``yamlRead.h:7-48``, ``mlmap.cpp:10-33,75-85``) from one of its config files.  Image size
    (not a YAML key: the reference takes it from the image messages): 2 * round(cx) x 2 * round(cy) unless given."""
    y = load_reference_yaml(path)
    w = width or 2 * int(round(float(y.get("mlmapping_cam_cx", 320.0))))
    h = height or 2 * int(round(float(y.get("mlmapping_cam_cy", 240.0))))
    return from_params(y, w, h)
