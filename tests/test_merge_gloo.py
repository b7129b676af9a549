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


def _second_round_blocks(rank, M):
    """Rank `rank`'s map some frames after the merged map M was loaded into it: M + its own increments on a few blocks,
    plus one block only this rank has seen."""
    rng = np.random.default_rng(100 + rank)
    keys, lo, occ = M["keys"].copy(), M["log_odds"].copy(), M["occ"].copy()
    rows = rng.choice(keys.shape[0], size=max(1, keys.shape[0] // 3), replace=False)
    lo[rows] += rng.uniform(-0.5, 0.5, size=(rows.size, lo.shape[1])).astype(np.float32)
    occ[rows] = np.where(occ[rows] == ord("u"), ord("f"), occ[rows])
    new_key = np.array([[1000 + rank, 0, 0]], dtype=np.int32)
    new_lo = rng.uniform(-1, 1, size=(1, lo.shape[1])).astype(np.float32)
    return {"keys": np.concatenate([keys, new_key]), "log_odds": np.concatenate([lo, new_lo]),
            "occ": np.concatenate([occ, np.full((1, lo.shape[1]), ord("f"), np.uint8)])}


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
    # second round (periodic merge): every rank's map is now the merged map M plus what it observed since
    nxt = _second_round_blocks(rank, {k: v.cpu().numpy() for k, v in merged.items()})
    merged2 = merge_global_map(nxt, SDEF, baseline=merged)
    np.savez(os.path.join(out_dir, f"merged2_{rank}.npz"), **{k: v.cpu().numpy() for k, v in merged2.items()})
    # third round without new observations: the maps ARE the merged map -> unchanged
    merged3 = merge_global_map(merged2, SDEF, baseline=merged2)
    np.savez(os.path.join(out_dir, f"merged3_{rank}.npz"), **{k: v.cpu().numpy() for k, v in merged3.items()})
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

    # ---- periodic merge: the second round sums the ranks' INCREMENTS on top of the first merged map (summing the maps
    #      themselves would count the shared past twice), the third (no new observations) changes nothing
    M = {k: m0[k] for k in ("keys", "log_odds", "occ")}
    n0, n1 = _second_round_blocks(0, M), _second_round_blocks(1, M)
    m20, m21 = np.load(tmp_path / "merged2_0.npz"), np.load(tmp_path / "merged2_1.npz")
    for k in ("keys", "log_odds", "occ"):
        assert np.array_equal(m20[k], m21[k]), f"second round: ranks disagree on {k}"
    nM = M["keys"].shape[0]
    assert m20["keys"].shape[0] == nM + 2
    exp = M["log_odds"].astype(np.float32) + (n0["log_odds"][:nM] - M["log_odds"]) + (n1["log_odds"][:nM] - M["log_odds"])
    exp = np.clip(exp, np.float32(SDEF.lm_log_odds_min), np.float32(SDEF.lm_log_odds_max))
    pos = {tuple(k): i for i, k in enumerate(m20["keys"])}
    got = m20["log_odds"][[pos[tuple(k)] for k in M["keys"]]]
    assert np.allclose(got, exp, atol=1e-5)
    naive = np.clip(n0["log_odds"][:nM] + n1["log_odds"][:nM], np.float32(SDEF.lm_log_odds_min), np.float32(SDEF.lm_log_odds_max))
    assert np.abs(naive - exp).max() > 0.1  # (what re-summing the maps would have given)
    for r, nb in ((0, n0), (1, n1)):  # each rank's own new block comes through unchanged (clamped)
        assert np.allclose(m20["log_odds"][pos[(1000 + r, 0, 0)]],
                           np.clip(nb["log_odds"][-1], np.float32(SDEF.lm_log_odds_min), np.float32(SDEF.lm_log_odds_max)), atol=1e-6)
    m30 = np.load(tmp_path / "merged3_0.npz")
    assert np.array_equal(m30["keys"], m20["keys"]) and np.array_equal(m30["occ"], m20["occ"])
    assert np.allclose(m30["log_odds"], m20["log_odds"], atol=1e-6)


# ---- world size 4, uneven block counts per rank (one rank with an empty map): the union does not split into equal shards of
#      whole blocks, so it is padded (merge._key_union) and the padding rows travel through the all-to-all / all-gather
def _uneven_blocks(rank, variant):
    b = _rank_blocks(rank % 2 if rank < 3 else 0)
    n = [7, 4, 9, 0][rank]
    o = np.lexsort((b["keys"][:, 2], b["keys"][:, 1], b["keys"][:, 0]))
    sel = o[rank::2][:n] if rank < 3 else o[:0]  # (ranks 0 and 2 share a scene: overlapping and disjoint blocks)
    if variant == 1 and rank == 0:
        sel = sel[1:]  # (its first block is the one only rank 0 holds: the union shrinks by one)
    return {k: b[k][sel] for k in ("keys", "log_odds", "occ")}


def _worker4(rank, world, port, out_dir):
    sys.path.insert(0, ROOT)
    import torch.distributed as dist

    from mlmapping_amd.config import SDEF
    from mlmapping_amd.merge import merge_global_map

    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    for variant in (0, 1):
        merged = merge_global_map(_uneven_blocks(rank, variant), SDEF)
        np.savez(os.path.join(out_dir, f"m4_{variant}_{rank}.npz"), **{k: v.cpu().numpy() for k, v in merged.items()})
    dist.barrier()
    dist.destroy_process_group()


def test_merge_four_ranks_uneven(tmp_path):
    import torch.multiprocessing as mp

    from mlmapping_amd.config import SDEF

    port = 31500 + (os.getpid() % 2000)
    mp.spawn(_worker4, args=(4, port, str(tmp_path)), nprocs=4, join=True)
    padded = 0
    for variant in (0, 1):
        res = [np.load(tmp_path / f"m4_{variant}_{r}.npz") for r in range(4)]
        for r in range(1, 4):
            for k in ("keys", "log_odds", "occ"):
                assert np.array_equal(res[0][k], res[r][k]), f"variant {variant}: rank {r} disagrees on {k}"
        own = [_uneven_blocks(r, variant) for r in range(4)]
        assert own[3]["keys"].shape[0] == 0 and len({b["keys"].shape[0] for b in own}) == 4  # uneven, one empty
        allk = np.unique(np.concatenate([b["keys"] for b in own]), axis=0)
        allk = allk[np.lexsort((allk[:, 2], allk[:, 1], allk[:, 0]))]
        assert np.array_equal(res[0]["keys"], allk)
        padded += int(allk.shape[0] % 4 != 0)
        C = SDEF.cells_per_block
        lo = np.zeros((allk.shape[0], C), np.float32)
        seen = np.zeros((allk.shape[0], C), bool)
        idx = {tuple(k): i for i, k in enumerate(allk)}
        for b in own:  # (rank order = the order the shards are summed in)
            for j, k in enumerate(b["keys"]):
                lo[idx[tuple(k)]] += b["log_odds"][j]
                seen[idx[tuple(k)]] |= b["occ"][j] != ord("u")
        lo = np.clip(lo, np.float32(SDEF.lm_log_odds_min), np.float32(SDEF.lm_log_odds_max))
        cls = np.full(lo.shape, ord("u"), np.uint8)
        cls[seen] = ord("f")
        cls[lo > np.float32(SDEF.lm_occupied_sh)] = ord("o")
        assert np.allclose(res[0]["log_odds"], lo, atol=1e-6)
        assert np.array_equal(res[0]["occ"], cls)
    assert padded >= 1  # (the two variants' unions differ by one block: at least one is not a multiple of the world size)
