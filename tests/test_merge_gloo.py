"""World-size-2 gloo test of the optional global-map merge (the only collective on the path).
Per-rank maps come from the CPU oracle here; on a GPU node the same function runs over RCCL."""
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _rank_blocks(rank):
    from mlmapping_amd import synthetic as syn
    from mlmapping_amd.config import SDEF
    from oracle.binding import OracleMap

    m = OracleMap(SDEF)
    img = syn.room_depth(SDEF)
    poses = syn.random_poses(3, seed=42 + rank)
    for q, t in poses:
        m.update_depth(img, q, t)
    return m.export_blocks()


def _worker(rank, world, port, out_dir):
    sys.path.insert(0, ROOT)
    import torch
    import torch.distributed as dist

    from mlmapping_amd.config import SDEF
    from mlmapping_amd.merge import merge_global_map

    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    b = _rank_blocks(rank)
    merged = merge_global_map(b, SDEF)
    np.savez(os.path.join(out_dir, f"merged_{rank}.npz"), **{k: v.cpu().numpy() for k, v in merged.items()})
    dist.barrier()
    dist.destroy_process_group()


def test_merge_two_ranks(tmp_path):
    import torch.multiprocessing as mp

    from mlmapping_amd.config import SDEF

    port = 29500 + (os.getpid() % 2000)
    mp.spawn(_worker, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    m0 = np.load(tmp_path / "merged_0.npz")
    m1 = np.load(tmp_path / "merged_1.npz")
    for k in ("keys", "log_odds", "occ"):
        assert np.array_equal(m0[k], m1[k]), f"ranks disagree on {k}"
    # independent expectation
    b0, b1 = _rank_blocks(0), _rank_blocks(1)
    allk = np.unique(np.concatenate([b0["keys"], b1["keys"]]), axis=0)
    order = np.lexsort((allk[:, 2], allk[:, 1], allk[:, 0]))
    allk = allk[order]
    assert np.array_equal(m0["keys"], allk)
    C = SDEF.cells_per_block
    lo = np.zeros((allk.shape[0], C), np.float32)
    seen = np.zeros((allk.shape[0], C), bool)
    idx = {tuple(k): i for i, k in enumerate(allk)}
    for b in (b0, b1):
        for j, k in enumerate(b["keys"]):
            i = idx[tuple(k)]
            lo[i] += b["log_odds"][j]
            seen[i] |= b["occ"][j] != ord("u")
    lo = np.clip(lo, np.float32(SDEF.lm_log_odds_min), np.float32(SDEF.lm_log_odds_max))
    cls = np.full(lo.shape, ord("u"), np.uint8)
    cls[seen] = ord("f")
    cls[lo > np.float32(SDEF.lm_occupied_sh)] = ord("o")
    assert np.allclose(m0["log_odds"], lo, atol=1e-6)
    assert np.array_equal(m0["occ"], cls)
    assert allk.shape[0] > max(b0["keys"].shape[0], b1["keys"].shape[0])  # the union is really larger
