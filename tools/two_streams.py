"""Diagnostic: aggregate rate of N independent maps fed alternately from one host thread (asynchronous submission) — does one
stream leave the GPU idle?  Usage: python tools/two_streams.py [n_maps]"""
import gc
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

from bench import make_inputs
from mlmapping_amd.config import S1
from mlmapping_amd.mlmap import MLMap

N = int(sys.argv[1]) if len(sys.argv) > 1 else 2
B, K, W = 32, 60, 30
frames, q, t = make_inputs(S1, 32, (K + W) * B, 42)
d = torch.from_numpy(frames.view(np.int16)).cuda()
torch.cuda.synchronize()
maps = [MLMap(S1, max_blocks=32768, max_points=640 * 480, max_batch=B) for _ in range(N)]
for m in maps:
    m.set_async(True)
for s in range(W):
    for m in maps:
        m.update_map_batch_dev(d.data_ptr(), B, 640, 480, q[s * B:s * B + B], t[s * B:s * B + B])
for m in maps:
    m.sync()
gc.disable()
t0 = time.perf_counter()
for s in range(W, W + K):
    for m in maps:
        m.update_map_batch_dev(d.data_ptr(), B, 640, 480, q[s * B:s * B + B], t[s * B:s * B + B])
for m in maps:
    m.sync()
dt = time.perf_counter() - t0
print(N, "maps:", round(N * K * B / dt), "frames/s aggregate")
