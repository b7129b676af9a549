"""Per-kernel HIP-event times with nothing else on the GPU: B-frame batches (B = 32, or MLM_KT_BATCH) of the bench workload submitted
synchronously (each batch drains before the next is submitted, so Stage A and Stage B+C never overlap).
Usage: python tools/kernel_times.py [cfg3] [batches]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bench import make_inputs  # noqa: E402
from mlmapping_amd.config import S1, S3  # noqa: E402
from mlmapping_amd.mlmap import MLMap  # noqa: E402

cfg = S3 if "cfg3" in sys.argv else S1
if os.environ.get("MLM_BENCH_NO_RAYCAST"):  # diagnostic: hits without rays
    import dataclasses
    cfg = dataclasses.replace(cfg, use_raycasting=False)
if "frontier" in sys.argv:  # use_exploration_frontiers: true
    import dataclasses
    cfg = dataclasses.replace(cfg, use_exploration_frontiers=True)
nb = int([a for a in sys.argv[1:] if a.isdigit()][0]) if any(a.isdigit() for a in sys.argv[1:]) else 8
B = int(os.environ.get("MLM_KT_BATCH", "32"))
frames, q, t = make_inputs(cfg, B, B * (nb + 2), seed=42)
if "scatter" in sys.argv:  # the worst-case scene: every pixel in a cell of its own
    import numpy as np
    from mlmapping_amd import synthetic as syn
    frames = np.stack([f for f, _ in syn.stream(cfg, "scatter", "smooth", B)])
m = MLMap(cfg, max_blocks=65536 if cfg is S3 else 32768, max_points=cfg.width * cfg.height, max_batch=B)
m.set_async(False)
for k in range(2):
    m.update_map_batch(frames, q[B * k:B * k + B], t[B * k:B * k + B])
for k in range(int(os.environ.get("MLM_KT_WARM", "0"))):  # (extra untimed batches: e.g. until the column table has widened)
    m.update_map_batch(frames, q[:B], t[:B])
m.sync()
m.enable_kernel_timing(2)
for k in range(2, nb + 2):
    m.update_map_batch(frames, q[B * k:B * k + B], t[B * k:B * k + B])
m.sync()
acc = {}
all_times = m.kernel_times()
sc = [ms for name, ms in all_times if name == "k_apply_voxelize"]
for name, ms in all_times:
    a = acc.setdefault(name, [0.0, 0])
    a[0] += ms
    a[1] += 1
tot = 0.0
for name, (ms, n) in acc.items():
    per = ms * 1e3 / (nb * B)
    tot += per
    print(f"{name:20s} {per:8.2f} us/frame   ({n} launches, {ms * 1e3 / n:8.1f} us each)")
print(f"{'sum':20s} {tot:8.2f} us/frame")
if sc and len(sc) == nb * (B + 1):
    first = [sc[i * (B + 1)] for i in range(nb)]
    last = [sc[i * (B + 1) + B] for i in range(nb)]
    mid = [sc[i * (B + 1) + j] for i in range(nb) for j in range(1, B)]
    print(f"k_apply_voxelize: voxelize side only {sum(first)/len(first)*1e3:.1f} us, apply side only {sum(last)/len(last)*1e3:.1f} us, "
          f"both {sum(mid)/len(mid)*1e3:.1f} us")
