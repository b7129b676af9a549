"""Single-position query latency through the C++ facade (tools/query_latency.cpp, built by mlmapping_amd/csrc/Makefile), with the CPU
oracle's per-position cost beside it.  `measure()` is what bench.py puts under extra.single_query_us; run directly it prints the JSON.
The oracle is used as the CPU baseline only."""
import json
import os
import struct
import subprocess
import sys
import tempfile
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
EXE = os.path.join(ROOT, "mlmapping_amd", "lib", "mlm_query_latency")


def measure(cfg=None, n_frames=6, n_pos=20000, with_oracle=True):
    from mlmapping_amd import synthetic as syn
    from mlmapping_amd.config import S1, to_c

    cfg = cfg or S1
    if not os.path.exists(EXE):
        raise RuntimeError(f"{EXE} not built (python -c 'import __graft_entry__ as g; g.build()')")
    frames = list(syn.stream(cfg, "room_jitter", "smooth", n_frames))
    rng = np.random.default_rng(11)
    # positions a planner samples: inside the mapped volume in front of the sensor (most of them in observed blocks)
    pos = rng.uniform([-1.0, -4.0, 0.0], [6.0, 4.0, 3.0], size=(n_pos, 3))
    blob = bytearray(bytes(to_c(cfg)))
    blob += struct.pack("4i", n_frames, cfg.width, cfg.height, n_pos)
    for img, (q, t) in frames:
        blob += np.concatenate([q, t]).astype(np.float64).tobytes() + np.ascontiguousarray(img, dtype=np.uint16).tobytes()
    blob += pos.astype(np.float64).tobytes()
    with tempfile.NamedTemporaryFile(suffix=".bin", delete=False) as f:
        f.write(bytes(blob))
        path = f.name
    try:
        res = json.loads(subprocess.run([EXE, path], check=True, capture_output=True, text=True).stdout)
    finally:
        os.unlink(path)
    res["unit"] = "us per call, one position per call, C++ client of include/mlmap_facade.hpp"
    res["workload"] = f"{cfg.width}x{cfg.height} room+jitter, {n_frames} frames, {n_pos} positions x 3 passes"
    if with_oracle:
        from oracle.binding import OracleMap

        cpu = OracleMap(cfg)
        for img, (q, t) in frames[:-1]:
            cpu.update_depth(img, q, t)
        cpu.getOccupancy(pos[:100])
        rows = {}
        for name, fn in (("getOccupancy", lambda: cpu.getOccupancy(pos)), ("getOdd", lambda: cpu.getOdd(pos)),
                         ("getOddGrad", lambda: cpu.getOddGrad(pos, 5))):
            t0 = time.perf_counter()
            fn()
            rows[name] = (time.perf_counter() - t0) / n_pos * 1e6
        res["cpu_oracle_us_per_position"] = rows  # (the oracle's batch entry points: a loop over the reference's inline query)
    return res


def measure_callback(cfg, depth_frames, poses, latency=0.0, calls=300, inflate_every=0):
    """The reference's callback (500 rand() samples per call) from the C++ client: depth_frames float32 metres or uint16 mm (cycled),
    poses (q, t) per call (cycled; 20 untimed calls first).  Returns {p50, p99, mean, calls} in microseconds per call."""
    from mlmapping_amd.config import to_c

    d0 = np.ascontiguousarray(depth_frames[0])
    is_f32 = int(d0.dtype == np.float32)
    blob = bytearray(bytes(to_c(cfg)))
    blob += struct.pack("7i", len(depth_frames), d0.shape[1], d0.shape[0], is_f32, calls, inflate_every, len(poses))
    blob += struct.pack("d", float(latency))
    for d in depth_frames:
        blob += np.ascontiguousarray(d, dtype=np.float32 if is_f32 else np.uint16).tobytes()
    for q, t in poses:
        blob += np.concatenate([q, t]).astype(np.float64).tobytes()
    with tempfile.NamedTemporaryFile(suffix=".bin", delete=False) as f:
        f.write(bytes(blob))
        path = f.name
    try:
        res = json.loads(subprocess.run([EXE, "--callback", path], check=True, capture_output=True, text=True).stdout)
    finally:
        os.unlink(path)
    res["unit"] = "us per callback, C++ client of the C ABI (no interpreter between the clock and the library)"
    return res


if __name__ == "__main__":
    print(json.dumps(measure()))
