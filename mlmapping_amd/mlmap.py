"""Host-side mirror of the reference's ``mlmap`` class (include/mlmap.h:42-140) over the C ABI of
``include/mlmap_hip.h``.

Only the map-update path and its queries are mirrored: ``update_map``, ``getOccupancy``, ``getOdd``,
``getOddGrad``, ``setFree_map_in_bound`` (+ ``getInflateOccupancy``, ``inflate_map``).  ROS plumbing
(subscriptions, TF, RViz) is out of scope.

This module is pure plumbing: ctypes calls into ``libmlmap_hip.so``.  There is no CPU fallback — if the
library is missing or no MI355X is visible, construction raises.
"""
from __future__ import annotations

import ctypes
import importlib.util
import os
import sys
from typing import Dict, List, Optional, Tuple

import numpy as np

from .config import CConfig, MapConfig, to_c

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("MLMAP_HIP_LIB") or os.path.join(_HERE, "lib", "libmlmap_hip.so")  # env: development builds

MLM_OK = 0
STATUS = {0: "MLM_OK", -1: "MLM_ERR_INVALID", -2: "MLM_ERR_HIP", -3: "MLM_ERR_CAPACITY", -4: "MLM_ERR_UNSUPPORTED"}

# every symbol include/mlmap_hip.h declares
ABI_SYMBOLS = [
    "mlm_create", "mlm_destroy", "mlm_last_error", "mlm_abi_version", "mlm_set_stream",
    "mlm_integrate_depth_u16", "mlm_integrate_depth_u16_dev", "mlm_integrate_depth_batch_dev",
    "mlm_integrate_depth_batch", "mlm_integrate_callback",
    "mlm_integrate_points", "mlm_query_occupancy", "mlm_query_occupancy_inflate", "mlm_query_inflate_occupancy",
    "mlm_query_odds", "mlm_query_odd_grad", "mlm_query_odds_at", "mlm_export_frontier_points", "mlm_import_blocks",
    "mlm_merge_pack", "mlm_merge_finish",
    "mlm_set_free_in_bound", "mlm_inflate_map", "mlm_block_count",
    "mlm_export_blocks", "mlm_export_block_flags", "mlm_export_frontier", "mlm_export_global_map", "mlm_sync", "mlm_set_async", "mlm_set_host_mirror_limit", "mlm_get_frame_stats",
    "mlm_get_awareness_hits",
    "mlm_get_awareness_misses", "mlm_get_T_ls", "mlm_get_odds_table", "mlm_get_kernel_times",
    "mlm_enable_kernel_timing", "mlm_set_timed_kernel", "mlm_host_register", "mlm_host_unregister", "mlm_debug_set", "mlm_debug_reset",
    "mlm_debug_clocks", "mlm_debug_probe_seeds",
]


class MlmError(RuntimeError):
    pass


class Limits(ctypes.Structure):
    _fields_ = [("max_blocks", ctypes.c_int32), ("max_points", ctypes.c_int32), ("max_batch", ctypes.c_int32),
                ("record_awareness", ctypes.c_int32)]


class FrameStats(ctypes.Structure):
    _fields_ = [("n_points", ctypes.c_int64), ("n_hit_cells", ctypes.c_int64), ("n_miss_cells", ctypes.c_int64),
                ("n_out_of_range", ctypes.c_int64), ("n_blocks", ctypes.c_int64), ("n_rehash_epochs", ctypes.c_int64),
                ("hit_bucket_count", ctypes.c_int64), ("n_multi_cells", ctypes.c_int64),
                ("n_contrib_slots", ctypes.c_int64), ("n_groups", ctypes.c_int64), ("n_rays", ctypes.c_int64),
                ("n_spec_replays", ctypes.c_int64), ("n_device_atomics", ctypes.c_int64), ("n_sector_fallbacks", ctypes.c_int64),
                ("logit_bit_exact", ctypes.c_int64), ("n_pool_grows", ctypes.c_int64), ("block_capacity", ctypes.c_int64),
                ("n_graph_launches", ctypes.c_int64), ("n_bin_exact_waves", ctypes.c_int64),
                ("n_host_queries", ctypes.c_int64), ("n_mirror_refreshes", ctypes.c_int64), ("n_mirror_blocks", ctypes.c_int64),
                ("device_bytes", ctypes.c_int64), ("n_slot_grows", ctypes.c_int64)]

    def as_dict(self) -> Dict[str, int]:
        return {k: int(getattr(self, k)) for k, _ in self._fields_}


_lib = None


def _share_hip_runtime_with_torch():
    """A process can drive the GPU through ONE HIP/HSA runtime only.  libmlmap_hip.so asks for `libamdhip64.so.7`
    (the system ROCm); PyTorch-ROCm wheels bundle their own copy and ask for it by another name, so importing torch
    AFTER this library would bring a second runtime that finds "no ROCm-capable device".  When a torch installation is
    present (it owns streams / RCCL in bench.py and in the merge), load ITS runtime first: the library then binds to it
    through the shared soname, whichever of the two is imported first.  MLMAP_HIP_RUNTIME=system keeps the system one
    (hosts without torch, e.g. the C++ facade, are not affected either way)."""
    if "torch" in sys.modules or os.environ.get("MLMAP_HIP_RUNTIME") == "system":
        return
    try:
        spec = importlib.util.find_spec("torch")
    except (ImportError, ValueError):
        spec = None
    if spec is None or not spec.origin:
        return
    cand = os.path.join(os.path.dirname(spec.origin), "lib", "libamdhip64.so")
    if os.path.exists(cand):
        ctypes.CDLL(cand, mode=ctypes.RTLD_GLOBAL)


def load_library(path: Optional[str] = None):
    """dlopen libmlmap_hip.so and declare the prototypes.  Raises if the library is not built."""
    global _lib
    if _lib is not None and path is None:
        return _lib
    p = path or LIB_PATH
    if not os.path.exists(p):
        raise MlmError(f"{p} not found: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
                       "(hipcc --offload-arch=gfx950); there is no CPU fallback")
    _share_hip_runtime_with_torch()
    L = ctypes.CDLL(p)
    vp, i32 = ctypes.c_void_p, ctypes.c_int32
    L.mlm_create.argtypes = [vp, vp, i32, ctypes.POINTER(vp)]
    L.mlm_destroy.argtypes = [vp]
    L.mlm_last_error.argtypes = [vp]
    L.mlm_last_error.restype = ctypes.c_char_p
    L.mlm_set_stream.argtypes = [vp, vp]
    L.mlm_integrate_depth_u16.argtypes = [vp, vp, i32, i32, i32, vp, i32, vp, vp]
    L.mlm_integrate_depth_u16_dev.argtypes = [vp, vp, i32, i32, i32, vp, i32, vp, vp]
    L.mlm_integrate_depth_batch_dev.argtypes = [vp, vp, i32, ctypes.c_size_t, i32, i32, i32, vp, vp]
    L.mlm_integrate_depth_batch.argtypes = [vp, vp, i32, ctypes.c_size_t, i32, i32, i32, vp, vp]
    L.mlm_integrate_points.argtypes = [vp, vp, i32, vp, vp]
    L.mlm_integrate_callback.argtypes = [vp, vp, i32, i32, i32, ctypes.c_double, vp, vp, vp, ctypes.c_double, vp,
                                         ctypes.c_double, ctypes.c_double, i32, vp]
    L.mlm_query_occupancy.argtypes = [vp, vp, i32, vp]
    L.mlm_query_occupancy_inflate.argtypes = [vp, vp, i32, ctypes.c_float, vp]
    L.mlm_query_inflate_occupancy.argtypes = [vp, vp, i32, vp]
    L.mlm_query_odds.argtypes = [vp, vp, i32, vp]
    L.mlm_query_odd_grad.argtypes = [vp, vp, i32, i32, vp]
    L.mlm_query_odds_at.argtypes = [vp, vp, vp, i32, vp]
    L.mlm_export_frontier_points.argtypes = [vp, i32, vp, vp]
    L.mlm_import_blocks.argtypes = [vp, i32, vp, vp, vp, vp, vp]
    L.mlm_merge_pack.argtypes = [vp, vp, i32, vp, vp]
    L.mlm_merge_finish.argtypes = [vp, vp, vp, ctypes.c_size_t, vp]
    L.mlm_set_free_in_bound.argtypes = [vp, vp, vp]
    L.mlm_inflate_map.argtypes = [vp, vp]
    L.mlm_block_count.argtypes = [vp, vp]
    L.mlm_export_blocks.argtypes = [vp, i32, vp, vp, vp, vp, vp]
    L.mlm_export_global_map.argtypes = [vp, i32, vp, vp]
    L.mlm_export_block_flags.argtypes = [vp, i32, vp, vp]
    L.mlm_export_frontier.argtypes = [vp, i32, vp, vp]
    L.mlm_sync.argtypes = [vp]
    L.mlm_set_async.argtypes = [vp, i32]
    L.mlm_set_host_mirror_limit.argtypes = [vp, ctypes.c_size_t]
    L.mlm_get_frame_stats.argtypes = [vp, vp]
    L.mlm_get_awareness_hits.argtypes = [vp, i32, vp, vp, vp, vp]
    L.mlm_get_awareness_misses.argtypes = [vp, i32, vp, vp]
    L.mlm_get_T_ls.argtypes = [vp, vp, vp]
    L.mlm_get_odds_table.argtypes = [vp, vp]
    L.mlm_get_kernel_times.argtypes = [vp, i32, vp, vp, vp]
    L.mlm_enable_kernel_timing.argtypes = [vp, i32]
    L.mlm_set_timed_kernel.argtypes = [vp, ctypes.c_char_p, i32]
    L.mlm_host_register.argtypes = [vp, vp, ctypes.c_size_t]
    L.mlm_host_unregister.argtypes = [vp, vp]
    L.mlm_debug_set.argtypes = [ctypes.c_char_p, ctypes.c_longlong]
    L.mlm_debug_reset.argtypes = []
    L.mlm_debug_clocks.argtypes = [vp, vp, i32]
    L.mlm_debug_probe_seeds.argtypes = [vp, vp]
    if path is None and os.environ.get("MLM_KNOBS"):
        # tooling convenience (tools/*.py, experiments): MLM_KNOBS="rank_grid=64,sec_tab=1024" -> mlm_debug_set before the first create
        for kv in os.environ["MLM_KNOBS"].split(","):
            k, _, v = kv.partition("=")
            if k.strip() and L.mlm_debug_set(k.strip().encode(), int(v)) != MLM_OK:
                raise MlmError(f"MLM_KNOBS: unknown knob {k.strip()!r}")
    if path is None:
        _lib = L
    return L


def _p(a: np.ndarray):
    return a.ctypes.data_as(ctypes.c_void_p)


def _f64(a) -> np.ndarray:
    return np.ascontiguousarray(a, dtype=np.float64)


def debug_set(name: str, value: int):
    """Test / experiment knob read by the next mlm_create of this process (mlm_debug_set; not part of the drop-in contract)."""
    if load_library().mlm_debug_set(name.encode(), int(value)) != MLM_OK:
        raise MlmError(f"mlm_debug_set: unknown knob {name!r}")


def debug_reset():
    load_library().mlm_debug_reset()


class MLMap:
    """One map on one MI355X (one handle = one device + one HIP stream)."""

    FREE, OCCUPIED, UNKNOWN = 1, 0, -1  # mlmap.h:109-114

    def __init__(self, cfg: MapConfig, device: int = 0, max_blocks: int = 0, max_points: int = 0,
                 record_awareness: bool = False, max_batch: int = 0):
        self.cfg = cfg
        self.cells = cfg.cells_per_block
        self._L = load_library()
        self._c = to_c(cfg)
        self._lim = Limits(max_blocks, max_points, max_batch, int(record_awareness))
        self._h = ctypes.c_void_p()
        rc = self._L.mlm_create(ctypes.byref(self._c), ctypes.byref(self._lim), device, ctypes.byref(self._h))
        if rc != MLM_OK:
            msg = self._L.mlm_last_error(self._h).decode() if self._h else ""
            if self._h:
                self._L.mlm_destroy(self._h)
                self._h = ctypes.c_void_p()
            raise MlmError(f"mlm_create failed: {STATUS.get(rc, rc)} {msg}")

    # ---- lifetime -------------------------------------------------------------------------------
    def close(self):
        if getattr(self, "_h", None):
            self._L.mlm_destroy(self._h)
            self._h = ctypes.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _chk(self, rc: int, what: str):
        if rc != MLM_OK:
            raise MlmError(f"{what}: {STATUS.get(rc, rc)}: {self._L.mlm_last_error(self._h).decode()}")

    def set_stream(self, stream_ptr: int):
        self._chk(self._L.mlm_set_stream(self._h, ctypes.c_void_p(stream_ptr)), "mlm_set_stream")

    def sync(self):
        self._chk(self._L.mlm_sync(self._h), "mlm_sync")

    def host_register(self, arr: np.ndarray):
        """Pin a host array the host-buffer entry points will be fed from (mlm_host_register)."""
        self._chk(self._L.mlm_host_register(self._h, _p(arr), arr.nbytes), "mlm_host_register")

    def debug_clocks(self, reset: bool = True) -> np.ndarray:
        """Host clocks of the single-frame callback path, microseconds summed over the calls (mlm_debug_clocks)."""
        out = np.zeros(8)
        self._chk(self._L.mlm_debug_clocks(self._h, _p(out), int(reset)), "mlm_debug_clocks")
        return out

    def debug_probe_seeds(self) -> np.ndarray:
        """Largest relative errors of the reciprocal / reciprocal-square-root seeds and their refined forms (mlm_debug_probe_seeds)."""
        out = np.zeros(4)
        self._chk(self._L.mlm_debug_probe_seeds(self._h, _p(out)), "mlm_debug_probe_seeds")
        return out

    def host_unregister(self, arr: np.ndarray):
        self._chk(self._L.mlm_host_unregister(self._h, _p(arr)), "mlm_host_unregister")

    def set_async(self, on: bool = True):
        """Integrate calls return after submission (two batches in flight); sync()/queries wait for everything."""
        self._chk(self._L.mlm_set_async(self._h, int(on)), "mlm_set_async")

    def set_host_mirror_limit(self, max_bytes: int):
        """Most pinned host memory the mirror of the map (small query batches) may take; 0 = queries always run as kernels."""
        self._chk(self._L.mlm_set_host_mirror_limit(self._h, int(max_bytes)), "mlm_set_host_mirror_limit")

    # ---- update_map (mlmap.cpp:382-386) ---------------------------------------------------------
    def update_map(self, depth_u16: np.ndarray, q_wb, t_wb, pixel_idx=None):
        """project_depth + update_map on a host uint16 depth image (mm).  ``pixel_idx`` (v*W+u) reproduces a
        sampler; None = dense."""
        img = np.ascontiguousarray(depth_u16, dtype=np.uint16)
        hgt, wid = img.shape
        if pixel_idx is None:
            pp, n = None, 0
        else:
            pix = np.ascontiguousarray(pixel_idx, dtype=np.int32)
            pp, n = _p(pix), pix.size
        self._chk(self._L.mlm_integrate_depth_u16(self._h, _p(img), wid, hgt, wid, pp, n, _p(_f64(q_wb)),
                                                  _p(_f64(t_wb))), "mlm_integrate_depth_u16")

    def update_map_dev(self, img_dev_ptr: int, width: int, height: int, q_wb, t_wb, row_stride: int = 0):
        """Same with the image already in HBM (device pointer)."""
        self._chk(self._L.mlm_integrate_depth_u16_dev(self._h, ctypes.c_void_p(img_dev_ptr), width, height,
                                                      row_stride or width, None, 0, _p(_f64(q_wb)), _p(_f64(t_wb))),
                  "mlm_integrate_depth_u16_dev")

    def update_map_batch_dev(self, img_dev_ptr: int, n_frames: int, width: int, height: int, q_wb, t_wb,
                             frame_stride: int = 0, row_stride: int = 0):
        q = _f64(q_wb).reshape(n_frames, 4)
        t = _f64(t_wb).reshape(n_frames, 3)
        self._chk(self._L.mlm_integrate_depth_batch_dev(self._h, ctypes.c_void_p(img_dev_ptr), n_frames,
                                                        frame_stride or width * height, width, height,
                                                        row_stride or width, _p(q), _p(t)),
                  "mlm_integrate_depth_batch_dev")

    def update_map_batch(self, frames_u16: np.ndarray, q_wb, t_wb):
        """K host frames [K,H,W] of one stream, integrated in order (uploads overlap with compute)."""
        fr = np.ascontiguousarray(frames_u16, dtype=np.uint16)
        k, hgt, wid = fr.shape
        q = _f64(q_wb).reshape(k, 4)
        t = _f64(t_wb).reshape(k, 3)
        self._chk(self._L.mlm_integrate_depth_batch(self._h, _p(fr), k, hgt * wid, wid, hgt, wid, _p(q), _p(t)),
                  "mlm_integrate_depth_batch")

    def depth_odom_callback(self, depth, t_img, odom_p, odom_q, odom_v, t_odom, imu_w, t_imu, latency, sampled=True):
        """mlmap::depth_odom_input_callback (mlmap.cpp:463-532) without ROS; depth float32 metres (32FC1) or uint16 mm.
        Returns the latency-compensated T_wb as (q (w,x,y,z), t) in one array of 7."""
        d = np.ascontiguousarray(depth)
        is_f32 = int(d.dtype == np.float32)
        if not is_f32:
            d = np.ascontiguousarray(d, dtype=np.uint16)  # (no copy when it is uint16 already)
        out = np.empty(7)
        self._chk(self._L.mlm_integrate_callback(self._h, _p(d), is_f32, d.shape[1], d.shape[0], float(t_img),
                                                 _p(_f64(odom_p)), _p(_f64(odom_q)), _p(_f64(odom_v)), float(t_odom),
                                                 _p(_f64(imu_w)), float(t_imu), float(latency), int(sampled), _p(out)),
                  "mlm_integrate_callback")
        return out

    def update_map_points(self, xyz_s, q_wb, t_wb):
        """input_pc_pose(PC_s, T_wb) + input_pc_pose_direct on explicit sensor-frame points."""
        xyz = _f64(xyz_s).reshape(-1, 3)
        self._chk(self._L.mlm_integrate_points(self._h, _p(xyz), xyz.shape[0], _p(_f64(q_wb)), _p(_f64(t_wb))),
                  "mlm_integrate_points")

    # ---- queries (mlmap.h:142-295) --------------------------------------------------------------
    def getOccupancy(self, pos_w, inflate: Optional[float] = None) -> np.ndarray:
        pos = _f64(pos_w).reshape(-1, 3)
        out = np.empty(pos.shape[0], dtype=np.int8)
        if inflate is None:
            self._chk(self._L.mlm_query_occupancy(self._h, _p(pos), pos.shape[0], _p(out)), "mlm_query_occupancy")
        else:
            self._chk(self._L.mlm_query_occupancy_inflate(self._h, _p(pos), pos.shape[0], ctypes.c_float(inflate),
                                                          _p(out)), "mlm_query_occupancy_inflate")
        return out.astype(np.int32)

    def getInflateOccupancy(self, pos_w) -> np.ndarray:
        pos = _f64(pos_w).reshape(-1, 3)
        out = np.empty(pos.shape[0], dtype=np.int8)
        self._chk(self._L.mlm_query_inflate_occupancy(self._h, _p(pos), pos.shape[0], _p(out)),
                  "mlm_query_inflate_occupancy")
        return out.astype(np.int32)

    def getOdd(self, pos_w) -> np.ndarray:
        pos = _f64(pos_w).reshape(-1, 3)
        out = np.empty(pos.shape[0], dtype=np.float32)
        self._chk(self._L.mlm_query_odds(self._h, _p(pos), pos.shape[0], _p(out)), "mlm_query_odds")
        return out

    def getOddAt(self, glb_id, subbox_id) -> np.ndarray:
        """float getOdd(const Vec3I &glb_id, size_t subbox_id), mlmap.h:227-235."""
        g = np.ascontiguousarray(glb_id, dtype=np.int32).reshape(-1, 3)
        c = np.ascontiguousarray(subbox_id, dtype=np.int32).reshape(-1)
        out = np.empty(g.shape[0], dtype=np.float32)
        self._chk(self._L.mlm_query_odds_at(self._h, _p(g), _p(c), g.shape[0], _p(out)), "mlm_query_odds_at")
        return out

    def getOddGrad(self, pos_w, max_iter: int = 5) -> np.ndarray:
        pos = _f64(pos_w).reshape(-1, 3)
        out = np.empty((pos.shape[0], 3), dtype=np.float64)
        self._chk(self._L.mlm_query_odd_grad(self._h, _p(pos), pos.shape[0], max_iter, _p(out)), "mlm_query_odd_grad")
        return out

    def setFree_map_in_bound(self, box_min, box_max):
        self._merge_base = None  # (the map no longer is "merged map + own observations": see merge.merge_device_maps)
        self._chk(self._L.mlm_set_free_in_bound(self._h, _p(_f64(box_min)), _p(_f64(box_max))),
                  "mlm_set_free_in_bound")

    def inflate_map(self, ct_pos):
        self._chk(self._L.mlm_inflate_map(self._h, _p(_f64(ct_pos))), "mlm_inflate_map")

    # ---- read-out -------------------------------------------------------------------------------
    def frame_stats(self) -> Dict[str, int]:
        s = FrameStats()
        self._chk(self._L.mlm_get_frame_stats(self._h, ctypes.byref(s)), "mlm_get_frame_stats")
        return s.as_dict()

    def block_count(self) -> int:
        n = ctypes.c_int32()
        self._chk(self._L.mlm_block_count(self._h, ctypes.byref(n)), "mlm_block_count")
        return n.value

    def export_blocks(self) -> Dict[str, np.ndarray]:
        """observed_group_map contents sorted by block key (same dict layout as the oracle binding)."""
        n, C = self.block_count(), self.cells
        keys = np.empty((n, 3), dtype=np.int32)
        lo = np.empty((n, C), dtype=np.float32)
        occ = np.empty((n, C), dtype=np.uint8)
        infl = np.empty((n, C), dtype=np.uint8)
        m = ctypes.c_int32()
        self._chk(self._L.mlm_export_blocks(self._h, n, _p(keys), _p(lo), _p(occ), _p(infl), ctypes.byref(m)),
                  "mlm_export_blocks")
        col = np.zeros(n, dtype=np.uint8)
        self._chk(self._L.mlm_export_block_flags(self._h, n, _p(col), ctypes.byref(m)), "mlm_export_block_flags")
        o = np.lexsort((keys[:, 2], keys[:, 1], keys[:, 0]))
        return {"keys": keys[o], "collapsed": col[o], "log_odds": lo[o], "occ": occ[o], "infl": infl[o]}

    def export_frontier(self) -> np.ndarray:
        """[n,4] int32 (gx,gy,gz,cell id) of the frontier cells, sorted."""
        n = ctypes.c_int32()
        self._chk(self._L.mlm_export_frontier(self._h, 0, None, ctypes.byref(n)), "mlm_export_frontier")
        out = np.empty((n.value, 4), dtype=np.int32)
        if n.value:
            self._chk(self._L.mlm_export_frontier(self._h, n.value, _p(out), ctypes.byref(n)), "mlm_export_frontier")
        return out[np.lexsort((out[:, 3], out[:, 2], out[:, 1], out[:, 0]))]

    def frontier_points(self) -> np.ndarray:
        """float32 [n,3] centres of the frontier cells: the PointCloud2 payload of /frontier (rviz_vis.cpp:267-293)."""
        n = ctypes.c_int32()
        self._chk(self._L.mlm_export_frontier_points(self._h, 0, None, ctypes.byref(n)), "mlm_export_frontier_points")
        out = np.empty((n.value, 3), dtype=np.float32)
        if n.value:
            self._chk(self._L.mlm_export_frontier_points(self._h, n.value, _p(out), ctypes.byref(n)),
                      "mlm_export_frontier_points")
        return out

    def import_blocks(self, keys, log_odds=None, occ=None, infl=None, collapsed=None):
        """Load blocks (layout of export_blocks) into the map; host numpy arrays or device pointers (ints)."""
        def ptr(a, dt):
            if a is None:
                return None, None
            if isinstance(a, int):
                return ctypes.c_void_p(a), None
            arr = np.ascontiguousarray(a, dtype=dt)
            return _p(arr), arr
        if isinstance(keys, tuple):  # (device pointer, n)
            kp, n, keep = ctypes.c_void_p(keys[0]), int(keys[1]), None
        else:
            keep = np.ascontiguousarray(keys, dtype=np.int32).reshape(-1, 3)
            kp, n = _p(keep), keep.shape[0]
        self._merge_base = None  # (a foreign import invalidates the baseline of periodic merges; merge_device_maps sets its own afterwards)
        holds = [ptr(log_odds, np.float32), ptr(occ, np.uint8), ptr(infl, np.uint8), ptr(collapsed, np.uint8)]
        self._chk(self._L.mlm_import_blocks(self._h, n, kp, holds[0][0], holds[1][0], holds[2][0], holds[3][0]),
                  "mlm_import_blocks")

    def export_block_keys_dev(self, keys_dev_ptr: int, cap: int) -> int:
        """Block keys straight into device memory ([cap,3] int32); returns the block count."""
        m = ctypes.c_int32()
        self._chk(self._L.mlm_export_blocks(self._h, cap, ctypes.c_void_p(keys_dev_ptr), None, None, None, ctypes.byref(m)),
                  "mlm_export_blocks")
        return m.value

    def merge_pack(self, keys_dev_ptr: int, n: int, log_odds_dev_ptr: int, seen_dev_ptr: int):
        self._chk(self._L.mlm_merge_pack(self._h, ctypes.c_void_p(keys_dev_ptr), n, ctypes.c_void_p(log_odds_dev_ptr),
                                         ctypes.c_void_p(seen_dev_ptr)), "mlm_merge_pack")

    def merge_finish(self, log_odds_dev_ptr: int, seen_dev_ptr: int, n_cells: int, occ_dev_ptr: int):
        self._chk(self._L.mlm_merge_finish(self._h, ctypes.c_void_p(log_odds_dev_ptr), ctypes.c_void_p(seen_dev_ptr), n_cells,
                                           ctypes.c_void_p(occ_dev_ptr)), "mlm_merge_finish")

    def class_counts(self) -> Dict[str, int]:
        b = self.export_blocks()
        return {"blocks": int(b["keys"].shape[0]), "o": int((b["occ"] == ord("o")).sum()),
                "f": int((b["occ"] == ord("f")).sum())}

    def global_map_points(self) -> np.ndarray:
        """float32 [n,3] centres of the inflated-'o' cells: the PointCloud2 payload of /global_map."""
        n = ctypes.c_int32()
        self._chk(self._L.mlm_export_global_map(self._h, 0, None, ctypes.byref(n)), "mlm_export_global_map")
        out = np.empty((n.value, 3), dtype=np.float32)
        if n.value:
            self._chk(self._L.mlm_export_global_map(self._h, n.value, _p(out), ctypes.byref(n)), "mlm_export_global_map")
        return out

    def awareness_hits(self) -> Tuple[np.ndarray, np.ndarray, np.ndarray]:
        """(cell idx sorted, odds, first-touch time) of the last frame."""
        n = self.frame_stats()["n_hit_cells"]
        cell = np.empty(n, dtype=np.uint32)
        odds = np.empty(n, dtype=np.float32)
        t = np.empty(n, dtype=np.uint32)
        m = ctypes.c_int32()
        self._chk(self._L.mlm_get_awareness_hits(self._h, n, _p(cell), _p(odds), _p(t), ctypes.byref(m)),
                  "mlm_get_awareness_hits")
        o = np.argsort(cell, kind="stable")
        return cell[o].astype(np.int64), odds[o], t[o]

    def awareness_misses(self) -> np.ndarray:
        n = self.frame_stats()["n_miss_cells"]
        cell = np.empty(n, dtype=np.uint32)
        m = ctypes.c_int32()
        self._chk(self._L.mlm_get_awareness_misses(self._h, n, _p(cell), ctypes.byref(m)), "mlm_get_awareness_misses")
        return np.sort(cell).astype(np.int64)

    def T_ls(self):
        q, t = np.empty(4), np.empty(3)
        self._chk(self._L.mlm_get_T_ls(self._h, _p(q), _p(t)), "mlm_get_T_ls")
        return q, t

    def odds_table(self) -> np.ndarray:
        out = np.empty((21, self.cfg.am_n_Rho), dtype=np.float32)
        self._chk(self._L.mlm_get_odds_table(self._h, _p(out)), "mlm_get_odds_table")
        return out

    def enable_kernel_timing(self, on=True):
        """True/1: per call; 2: accumulate over calls until kernel_times() is read; False/0: off."""
        self._chk(self._L.mlm_enable_kernel_timing(self._h, int(on)), "mlm_enable_kernel_timing")

    def set_timed_kernel(self, name: str, every: int = 1):
        """The one kernel whose launches (every `every`-th of them) timing mode 3 brackets."""
        self._chk(self._L.mlm_set_timed_kernel(self._h, name.encode(), int(every)), "mlm_set_timed_kernel")

    def kernel_times(self, cap: int = 1 << 16) -> List[Tuple[str, float]]:
        names = (ctypes.c_char_p * cap)()
        ms = np.empty(cap, dtype=np.float32)
        n = ctypes.c_int32()
        self._chk(self._L.mlm_get_kernel_times(self._h, cap, names, _p(ms), ctypes.byref(n)), "mlm_get_kernel_times")
        return [(names[i].decode(), float(ms[i])) for i in range(min(n.value, cap))]
