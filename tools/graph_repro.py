"""Debug: single-frame graph replay after operations of the merge path.  usage: graph_repro.py VARIANT"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from mlmapping_amd import synthetic as syn  # noqa: E402
from mlmapping_amd.config import SDEF  # noqa: E402
from mlmapping_amd.mlmap import MLMap  # noqa: E402

variant = sys.argv[1]
torch.cuda.set_device(0)
m = MLMap(SDEF, device=0, max_blocks=4096)
frames = list(syn.stream(SDEF, "room_jitter", "random", 6, seed=42))
for i, (img, (q, t)) in enumerate(frames[:3]):
    m.update_map(img, q, t)
print("graph launches", m.frame_stats()["n_graph_launches"], flush=True)
b = m.export_blocks()
n = b["keys"].shape[0]
if variant == "devimport":
    k = torch.from_numpy(b["keys"]).cuda()
    lo = torch.from_numpy(b["log_odds"]).cuda()
    oc = torch.from_numpy(b["occ"]).cuda()
    torch.cuda.synchronize()
    m.import_blocks((k.data_ptr(), n), log_odds=lo.data_ptr(), occ=oc.data_ptr())
elif variant == "keysdev":
    k = torch.empty((n, 3), dtype=torch.int32, device="cuda")
    m.export_block_keys_dev(k.data_ptr(), n)
elif variant == "pack":
    k = torch.from_numpy(b["keys"]).cuda()
    dense = torch.empty((n, m.cells), dtype=torch.float32, device="cuda")
    seen = torch.empty((n, m.cells), dtype=torch.uint8, device="cuda")
    torch.cuda.synchronize()
    m.merge_pack(k.data_ptr(), n, dense.data_ptr(), seen.data_ptr())
elif variant == "torchonly":
    x = torch.zeros(1 << 20, device="cuda")
    x += 1
    torch.cuda.synchronize()
elif variant == "blockcount":
    print(m.block_count())
elif variant == "newblocks":
    # import blocks the map does not hold yet (what a merge with another rank's map does)
    k = b["keys"].copy()
    k[:, 0] += 100
    kd = torch.from_numpy(k).cuda()
    lo = torch.from_numpy(b["log_odds"]).cuda()
    oc = torch.from_numpy(b["occ"]).cuda()
    torch.cuda.synchronize()
    m.import_blocks((kd.data_ptr(), n), log_odds=lo.data_ptr(), occ=oc.data_ptr())
elif variant == "loop":
    for rep in range(60):
        for img, (q, t) in frames:
            m.update_map(img, q, t)
print("variant done", variant, flush=True)
for i, (img, (q, t)) in enumerate(frames[3:]):
    m.update_map(img, q, t)
    print("after", i, m.frame_stats()["n_graph_launches"], flush=True)
print("OK", variant)
