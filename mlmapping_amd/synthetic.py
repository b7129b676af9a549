"""Deterministic synthetic depth scenes and pose lists.

These are the benchmark / test inputs; nothing here is part of the map update itself.
"""
from __future__ import annotations

import math
from typing import Iterator, List, Tuple

import numpy as np

from .config import MapConfig

Pose = Tuple[np.ndarray, np.ndarray]  # (q_wb (w,x,y,z), t_wb)


def room_depth(cfg: MapConfig, half_x: float = 4.0, half_y: float = 1.5, far: float = 4.0) -> np.ndarray:
    """"room" scene: d = min(far, half_x/|dx|, half_y/|dy|), raw = lround(1000 d) uint16 mm.

    With the camera looking along body +x (T_B_S of config_sim.yaml) this is a box 2*half_x wide,
    2*half_y tall with a wall `far` metres ahead.  All pixels are valid.
    """
    u = np.arange(cfg.width, dtype=np.float64)
    v = np.arange(cfg.height, dtype=np.float64)
    dx = (u - cfg.cam_cx) / cfg.cam_fx
    dy = (v - cfg.cam_cy) / cfg.cam_fy
    with np.errstate(divide="ignore"):
        tx = np.where(dx == 0.0, 1e9, half_x / np.abs(dx))
        ty = np.where(dy == 0.0, 1e9, half_y / np.abs(dy))
    d = np.minimum(far, np.minimum(tx[None, :], ty[:, None]))
    raw = np.floor(1000.0 * d + 0.5)  # lround for positive values
    return raw.astype(np.uint16)


def corridor_depth(cfg: MapConfig) -> np.ndarray:
    """Synthetic corridor substituting BASELINE config 5's absent corridor.bag: 2 m wide, 3 m tall, 40 m long
    (clipped by the uint16 / awareness range)."""
    return room_depth(cfg, half_x=1.0, half_y=1.5, far=40.0)


class ScatterScene:
    """"scatter" scene: raw = 500 + (mt19937(12345)() % 5500) per pixel, row-major, RNG continuing across frames."""

    def __init__(self, cfg: MapConfig, seed: int = 12345):
        self.cfg = cfg
        self.bg = np.random.MT19937()
        self.bg._legacy_seeding(seed)  # == std::mt19937(seed)

    def next(self) -> np.ndarray:
        n = self.cfg.width * self.cfg.height
        r = self.bg.random_raw(n).astype(np.uint64)
        raw = 500 + (r % 5500)
        return raw.astype(np.uint16).reshape(self.cfg.height, self.cfg.width)


def jitter_depth(base: np.ndarray, frame: int, amp_mm: int = 100, seed: int = 7) -> np.ndarray:
    """base + U[0, amp_mm) mm of per-pixel noise, seeded per frame (order-sensitivity stress input)."""
    rng = np.random.default_rng(seed * 1000003 + frame)
    noise = rng.integers(0, amp_mm, size=base.shape, dtype=np.int64)
    return np.clip(base.astype(np.int64) + noise, 1, 65535).astype(np.uint16)


def quat_from_rpy(roll: float, pitch: float, yaw: float) -> np.ndarray:
    cr, sr = math.cos(roll / 2), math.sin(roll / 2)
    cp, sp = math.cos(pitch / 2), math.sin(pitch / 2)
    cy, sy = math.cos(yaw / 2), math.sin(yaw / 2)
    return np.array([cr * cp * cy + sr * sp * sy, sr * cp * cy - cr * sp * sy,
                     cr * sp * cy + sr * cp * sy, cr * cp * sy - sr * sp * cy], dtype=np.float64)


def static_pose() -> Pose:
    """BASELINE config 1 "identity pose": T_wb = (I, (0,0,1.5))."""
    return np.array([1.0, 0.0, 0.0, 0.0]), np.array([0.0, 0.0, 1.5])


def translating_pose(k: int) -> Pose:
    return np.array([1.0, 0.0, 0.0, 0.0]), np.array([0.01 * k, 0.0, 1.5])


def random_poses(n: int, seed: int = 42) -> List[Pose]:
    """yaw U(-pi,pi), pitch/roll U(+-10 deg), t in U([-2,2]^2 x [0.5,2.5]); recorded in fixtures where parity
    depends on them."""
    rng = np.random.default_rng(seed)
    out = []
    for _ in range(n):
        yaw = rng.uniform(-math.pi, math.pi)
        pitch = rng.uniform(-math.radians(10), math.radians(10))
        roll = rng.uniform(-math.radians(10), math.radians(10))
        t = np.array([rng.uniform(-2, 2), rng.uniform(-2, 2), rng.uniform(0.5, 2.5)])
        out.append((quat_from_rpy(roll, pitch, yaw), t))
    return out


def smooth_trajectory(n: int, seed: int = 42) -> List[Pose]:
    """A slowly turning, translating camera (what a 30 Hz stream looks like): used by the stream workloads."""
    rng = np.random.default_rng(seed)
    yaw0 = rng.uniform(-math.pi, math.pi)
    out = []
    for k in range(n):
        yaw = yaw0 + 0.01 * k
        pitch = math.radians(3.0) * math.sin(0.05 * k)
        roll = math.radians(2.0) * math.cos(0.03 * k)
        t = np.array([0.01 * k * math.cos(yaw0), 0.01 * k * math.sin(yaw0), 1.5 + 0.1 * math.sin(0.02 * k)])
        out.append((quat_from_rpy(roll, pitch, yaw), t))
    return out


def stream(cfg: MapConfig, scene: str, poses: str, n: int, seed: int = 42) -> Iterator[Tuple[np.ndarray, Pose]]:
    """Yield (depth uint16 HxW, pose) for n frames of a named workload."""
    if scene == "room":
        base = room_depth(cfg)
        frames = (base for _ in range(n))
    elif scene == "room_jitter":
        base = room_depth(cfg)
        frames = (jitter_depth(base, k) for k in range(n))
    elif scene == "corridor":
        base = corridor_depth(cfg)
        frames = (base for _ in range(n))
    elif scene == "scatter":
        sc = ScatterScene(cfg)
        frames = (sc.next() for _ in range(n))
    else:
        raise ValueError(scene)
    if poses == "static":
        pl = [static_pose() for _ in range(n)]
    elif poses == "translating":
        pl = [translating_pose(k) for k in range(n)]
    elif poses == "random":
        pl = random_poses(n, seed)
    elif poses == "smooth":
        pl = smooth_trajectory(n, seed)
    else:
        raise ValueError(poses)
    for k, d in enumerate(frames):
        yield d, pl[k]
