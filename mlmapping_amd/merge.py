"""Optional global-map merge across ranks (one map per GPU) over torch.distributed — RCCL ("nccl") on MI355X nodes,
gloo in the CPU tests.

This step has NO counterpart in the reference (single process) — it is the one place the path has a real
exchange, so it is the one place a collective is used; the per-frame path never communicates.

Protocol (all ranks end with the same merged map):
  1. all-gather the block keys (12 B each) -> identical sorted union on every rank;
  2. every rank packs its log-odds of the union's blocks into a dense [n_union, n^3] float tensor (0 where absent) and
     an "observed" flag tensor (occupancy != 'u');
  3. all-reduce(sum) both (xGMI: 7 point-to-point links per GPU; one large message lets RCCL use all of them);
  4. clamp the summed log-odds to [log_odds_min, log_odds_max]; class = 'o' if L > occupied_sh, else 'f' if any rank
     observed the voxel, else 'u'.
"""
from __future__ import annotations

from typing import Dict, Optional

import numpy as np
import torch
import torch.distributed as dist

from .config import MapConfig

_BIAS = 1 << 20


def _pack(keys: torch.Tensor) -> torch.Tensor:
    k = keys.to(torch.int64) + _BIAS
    return (k[:, 0] << 42) | (k[:, 1] << 21) | k[:, 2]


def _unpack(p: torch.Tensor) -> torch.Tensor:
    m = (1 << 21) - 1
    return torch.stack([((p >> 42) & m) - _BIAS, ((p >> 21) & m) - _BIAS, (p & m) - _BIAS], dim=1).to(torch.int32)


def merge_global_map(blocks: Dict[str, "torch.Tensor | np.ndarray"], cfg: MapConfig, group=None,
                     device: Optional[torch.device] = None) -> Dict[str, torch.Tensor]:
    """blocks: {'keys': [n,3] int32, 'log_odds': [n,C] float32, 'occ': [n,C] uint8} of THIS rank (any order).
    Returns the merged map (same dict layout, keys sorted) as tensors on `device`."""
    world = dist.get_world_size(group)
    dev = device or (torch.device("cuda", torch.cuda.current_device()) if dist.get_backend(group) == "nccl"
                     else torch.device("cpu"))
    as_t = lambda a, dt: (a if isinstance(a, torch.Tensor) else torch.from_numpy(np.ascontiguousarray(a))).to(dev, dt)
    keys = as_t(blocks["keys"], torch.int32).reshape(-1, 3)
    lo = as_t(blocks["log_odds"], torch.float32)
    occ = as_t(blocks["occ"], torch.uint8)
    C = cfg.cells_per_block
    n = keys.shape[0]
    # 1. union of block keys
    cnt = torch.tensor([n], dtype=torch.int64, device=dev)
    cnts = [torch.zeros_like(cnt) for _ in range(world)]
    dist.all_gather(cnts, cnt, group=group)
    n_max = int(max(int(c.item()) for c in cnts))
    packed = torch.full((max(n_max, 1),), torch.iinfo(torch.int64).max, dtype=torch.int64, device=dev)
    if n:
        packed[:n] = _pack(keys)
    gathered = [torch.empty_like(packed) for _ in range(world)]
    dist.all_gather(gathered, packed, group=group)
    union = torch.unique(torch.cat(gathered))
    union = union[union != torch.iinfo(torch.int64).max]  # sorted ascending = lexicographic (x,y,z)
    n_u = union.shape[0]
    # 2. dense pack
    dense = torch.zeros((n_u, C), dtype=torch.float32, device=dev)
    seen = torch.zeros((n_u, C), dtype=torch.float32, device=dev)
    if n:
        pos = torch.searchsorted(union, _pack(keys))
        dense[pos] = lo
        seen[pos] = (occ != ord("u")).to(torch.float32)
    # 3. exchange
    if n_u:
        dist.all_reduce(dense, op=dist.ReduceOp.SUM, group=group)
        dist.all_reduce(seen, op=dist.ReduceOp.SUM, group=group)
    # 4. clamp + class
    lo_min = float(np.float32(cfg.lm_log_odds_min))
    lo_max = float(np.float32(cfg.lm_log_odds_max))
    sh = float(np.float32(cfg.lm_occupied_sh))
    dense.clamp_(lo_min, lo_max)
    cls = torch.full((n_u, C), ord("u"), dtype=torch.uint8, device=dev)
    cls[seen > 0] = ord("f")
    cls[dense > sh] = ord("o")
    return {"keys": _unpack(union), "log_odds": dense, "occ": cls}
