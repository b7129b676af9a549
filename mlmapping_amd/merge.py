"""Optional global-map merge across ranks (one map per GPU) over torch.distributed — RCCL ("nccl") on MI355X nodes,
gloo in the CPU tests.

This step has NO counterpart in the reference (single process) — it is the one place the path has a real
exchange, so it is the one place collectives are used; the per-frame path never communicates.

Protocol (all ranks end with the same merged map):
  1. all-gather the block keys (12 B each) -> identical sorted union on every rank, padded to a multiple of the world
     size so that it splits into equal shards of whole blocks;
  2. every rank packs its map into the union's layout: log-odds [n_u, n^3] float32 (0 where it does not hold the block)
     and a "seen" plane [n_u, n^3] uint8 (occupancy != 'u') — 5 bytes per voxel;
  3. DIRECT reduce-scatter: one all-to-all hands shard r of every rank's buffers to rank r, which sums the log-odds and
     ORs the seen flags locally.  xGMI is a point-to-point fabric (7 links per GPU, a fully connected 8-GPU node): the
     all-to-all puts 1/world of the buffer on every link at once, where a ring would be bound by one link;
  4. rank r finishes ITS shard only: clamp to [log_odds_min, log_odds_max]; class 'o' above occupied_sh, else 'f' where
     any rank had seen the voxel, else 'u';
  5. all-gather the finished shards (log-odds + class, 5 bytes per voxel).
Per rank and voxel 2 x 5 x (w-1)/w bytes cross the fabric (two dense fp32 all-reduces would move 16).

Periodic merges: once a merged map M has been loaded back into every rank's handle, each rank's log-odds are M plus what
it observed since.  Summing those maps again would count M `world` times.  So a merge that was loaded back leaves a
BASELINE (the merged log-odds), and the next merge exchanges INCREMENTS: step 2 packs L_r - M, step 3 sums the
increments, and rank r adds M's rows of its shard before step 4 — merged' = clamp(M + sum_r (L_r - M)).  With no
baseline (first merge) this is the plain sum.  Merging twice without new observations returns the same map.

Two front ends share the protocol: `merge_global_map` takes block dumps (numpy / torch, any device — what the gloo test
and a host-side caller use) and `merge_device_maps` works on a live `MLMap` handle: keys, packing and finishing run as
HIP kernels of libmlmap_hip.so straight from / into the device-resident map (mlm_merge_pack / mlm_merge_finish /
mlm_import_blocks), torch only owns the exchange buffers and issues the collectives.
"""
from __future__ import annotations

from typing import Callable, Dict, Optional, Tuple

import numpy as np
import torch
import torch.distributed as dist

from .config import MapConfig

_BIAS = 1 << 20
_PAD = torch.iinfo(torch.int64).max


def _pack(keys: torch.Tensor) -> torch.Tensor:
    k = keys.to(torch.int64) + _BIAS
    return (k[:, 0] << 42) | (k[:, 1] << 21) | k[:, 2]


def _unpack(p: torch.Tensor) -> torch.Tensor:
    m = (1 << 21) - 1
    return torch.stack([((p >> 42) & m) - _BIAS, ((p >> 21) & m) - _BIAS, (p & m) - _BIAS], dim=1).to(torch.int32)


def _host_staged(group) -> bool:
    """gloo has no device collectives for all-gather / all-to-all: tensors that live on a GPU are staged through the host
    (the 2-rank test of the device path on a one-GPU box; the measured configuration is RCCL, device to device)."""
    return dist.get_backend(group) != "nccl"


def _all_gather(out: torch.Tensor, inp: torch.Tensor, group) -> None:
    if inp.is_cuda and _host_staged(group):
        o = torch.empty(out.shape, dtype=out.dtype)
        dist.all_gather_into_tensor(o, inp.cpu(), group=group)
        out.copy_(o)
    else:
        dist.all_gather_into_tensor(out, inp, group=group)


def _all_to_all(out: torch.Tensor, inp: torch.Tensor, group) -> None:
    if inp.is_cuda and _host_staged(group):
        o = torch.empty(out.shape, dtype=out.dtype)
        dist.all_to_all_single(o, inp.cpu(), group=group)
        out.copy_(o)
    else:
        dist.all_to_all_single(out, inp, group=group)


def _key_union(keys: torch.Tensor, world: int, group) -> Tuple[torch.Tensor, int]:
    """Step 1.  Returns (sorted packed union padded with _PAD to a multiple of `world`, number of real blocks)."""
    dev = keys.device
    n = keys.shape[0]
    cnt = torch.tensor([n], dtype=torch.int64, device=dev)
    cnts = torch.zeros(world, dtype=torch.int64, device=dev)
    _all_gather(cnts, cnt, group)
    n_max = max(int(cnts.max().item()), 1)
    packed = torch.full((n_max,), _PAD, dtype=torch.int64, device=dev)
    if n:
        packed[:n] = _pack(keys)
    gathered = torch.empty(world * n_max, dtype=torch.int64, device=dev)
    _all_gather(gathered, packed, group)
    union = torch.unique(gathered)  # sorted ascending = lexicographic (x,y,z); _PAD sorts last
    union = union[union != _PAD]
    n_u = int(union.shape[0])
    n_pad = (n_u + world - 1) // world * world
    if n_pad > n_u:
        union = torch.cat([union, torch.full((n_pad - n_u,), _PAD, dtype=torch.int64, device=dev)])
    return union, n_u


def _exchange(dense: torch.Tensor, seen: torch.Tensor, finish: Callable[[torch.Tensor, torch.Tensor], torch.Tensor], world: int,
              group, add_base: Optional[Callable[[torch.Tensor, int], None]] = None) -> Tuple[torch.Tensor, torch.Tensor]:
    """Steps 3-5 on the packed planes [n_pad, C] (n_pad a multiple of world).  add_base(my_lo, first_row): adds the baseline's
    rows of this rank's shard (periodic merges exchange increments, see the module docstring)."""
    n_pad, C = dense.shape
    shard = n_pad // world
    if world > 1:
        r_lo = torch.empty_like(dense)
        r_seen = torch.empty_like(seen)
        _all_to_all(r_lo, dense, group)    # chunk j of the input goes to rank j
        _all_to_all(r_seen, seen, group)
        my_lo = r_lo.view(world, shard, C).sum(dim=0).contiguous()
        my_seen = r_seen.view(world, shard, C).amax(dim=0).contiguous()
    else:
        my_lo, my_seen = dense.clone(), seen
    if add_base is not None:
        add_base(my_lo, dist.get_rank(group) * shard)
    my_occ = finish(my_lo, my_seen)  # clamps my_lo in place
    out_lo = torch.empty_like(dense)
    out_occ = torch.empty_like(seen)
    _all_gather(out_lo, my_lo.contiguous(), group)
    _all_gather(out_occ, my_occ.contiguous(), group)
    return out_lo, out_occ


def _baseline_ops(base: Optional[Dict[str, torch.Tensor]], union: torch.Tensor, dev):
    """(subtract(dense), add_base) for a baseline {'packed': sorted int64 keys, 'log_odds': [n_b, C]} laid out on `union`
    (which contains every baseline key: blocks never disappear); (None, None) without a baseline."""
    if not base or base["packed"].numel() == 0:
        return None, None
    b_lo = base["log_odds"].to(dev)
    b_keys = base["packed"].to(dev)
    pos = torch.searchsorted(union, b_keys)
    # every baseline block must still be in the union (blocks never disappear): a missing key would silently shift rows
    if bool((pos >= union.shape[0]).any()) or not bool((union[pos.clamp(max=union.shape[0] - 1)] == b_keys).all()):
        raise RuntimeError("merge baseline holds blocks the maps no longer have: the handle's content changed outside the merge "
                           "(recreated handle / foreign import) — drop the baseline (m._merge_base = None) on every rank")

    # (increment and re-add in FP64, rounded once each: a voxel nobody touched since the baseline gets back exactly M)
    def subtract(dense: torch.Tensor) -> None:
        dense[pos] = (dense[pos].double() - b_lo.double()).float()

    def add_base(my_lo: torch.Tensor, first_row: int) -> None:
        sel = (pos >= first_row) & (pos < first_row + my_lo.shape[0])
        rows = pos[sel] - first_row
        my_lo[rows] = (my_lo[rows].double() + b_lo[sel].double()).float()

    return subtract, add_base


def _baseline_tag(base: Optional[Dict[str, torch.Tensor]]) -> int:
    """A cheap fingerprint of a baseline (0: none) that the ranks compare before they exchange increments."""
    if not base or base["packed"].numel() == 0:
        return 0
    p = base["packed"]
    bits = base["log_odds"].contiguous().view(torch.int32).sum(dtype=torch.int64)  # (integer sum: exact, order-free)
    return int((int(p.sum().item()) ^ (int(p.numel()) << 40) ^ int(bits.item())) & 0x7FFFFFFFFFFFFFFF) or 1


def _finish_torch(cfg: MapConfig) -> Callable[[torch.Tensor, torch.Tensor], torch.Tensor]:
    lo_min = float(np.float32(cfg.lm_log_odds_min))
    lo_max = float(np.float32(cfg.lm_log_odds_max))
    sh = float(np.float32(cfg.lm_occupied_sh))

    def finish(lo: torch.Tensor, seen: torch.Tensor) -> torch.Tensor:
        lo.clamp_(lo_min, lo_max)
        occ = torch.where(seen > 0, ord("f"), ord("u")).to(torch.uint8)
        occ[lo > sh] = ord("o")
        return occ

    return finish


def merge_global_map(blocks: Dict[str, "torch.Tensor | np.ndarray"], cfg: MapConfig, group=None,
                     device: Optional[torch.device] = None, baseline: Optional[Dict[str, torch.Tensor]] = None) -> Dict[str, torch.Tensor]:
    """blocks: {'keys': [n,3] int32, 'log_odds': [n,C] float32, 'occ': [n,C] uint8} of THIS rank (any order; the layout
    MLMap.export_blocks and the oracle binding both produce).  Returns the merged map (same layout, keys sorted) as
    tensors on `device`.  baseline: the result of the previous merge if THAT was loaded into every rank's map (the ranks'
    maps then hold it plus their new observations; only the increments are summed)."""
    world = dist.get_world_size(group)
    dev = device or (torch.device("cuda", torch.cuda.current_device()) if dist.get_backend(group) == "nccl"
                     else torch.device("cpu"))
    as_t = lambda a, dt: (a if isinstance(a, torch.Tensor) else torch.from_numpy(np.ascontiguousarray(a))).to(dev, dt)
    keys = as_t(blocks["keys"], torch.int32).reshape(-1, 3)
    lo = as_t(blocks["log_odds"], torch.float32)
    occ = as_t(blocks["occ"], torch.uint8)
    C = cfg.cells_per_block
    union, n_u = _key_union(keys, world, group)
    dense = torch.zeros((union.shape[0], C), dtype=torch.float32, device=dev)
    seen = torch.zeros((union.shape[0], C), dtype=torch.uint8, device=dev)
    if keys.shape[0]:
        pos = torch.searchsorted(union, _pack(keys))
        dense[pos] = lo
        seen[pos] = (occ != ord("u")).to(torch.uint8)
    base = None
    if baseline is not None:
        bk = as_t(baseline["keys"], torch.int32).reshape(-1, 3)
        bp = _pack(bk)
        o = torch.argsort(bp)
        base = {"packed": bp[o], "log_odds": as_t(baseline["log_odds"], torch.float32)[o]}
    subtract, add_base = _baseline_ops(base, union, dev)
    if subtract is not None:
        subtract(dense)
    out_lo, out_occ = _exchange(dense, seen, _finish_torch(cfg), world, group, add_base)
    return {"keys": _unpack(union[:n_u]), "log_odds": out_lo[:n_u], "occ": out_occ[:n_u]}


def merge_device_maps(m, group=None, load_back: bool = True) -> Dict[str, torch.Tensor]:
    """Merge the device-resident maps of all ranks (`m`: this rank's MLMap) over RCCL.  Packing and finishing are HIP
    kernels of the map library working on the handle's own HBM; with `load_back` the merged map replaces the handle's
    content (mlm_import_blocks), so that the usual queries answer from the global map, and is remembered on `m` as the
    baseline of the next merge (which then sums only what the ranks observed since: periodic merges do not count the
    shared past `world` times).  Returns the merged map as device tensors (keys sorted)."""
    world = dist.get_world_size(group)
    dev = torch.device("cuda", torch.cuda.current_device())
    C = m.cells
    n = m.block_count()
    keys = torch.empty((max(n, 1), 3), dtype=torch.int32, device=dev)
    if n:
        m.export_block_keys_dev(keys.data_ptr(), n)
    union, n_u = _key_union(keys[:n], world, group)
    n_pad = int(union.shape[0])
    ukeys = _unpack(union[:n_u]).contiguous()
    dense = torch.empty((n_pad, C), dtype=torch.float32, device=dev)
    seen = torch.empty((n_pad, C), dtype=torch.uint8, device=dev)
    dense[n_u:] = 0  # padding rows (the union is padded to equal shards)
    seen[n_u:] = 0
    torch.cuda.synchronize()
    if n_u:
        m.merge_pack(ukeys.data_ptr(), n_u, dense.data_ptr(), seen.data_ptr())
    base = getattr(m, "_merge_base", None)
    # every rank must merge against the SAME baseline (or none): a rank whose handle was re-imported / cleared / recreated since
    # the last merge has dropped it (MLMap.import_blocks, setFree_map_in_bound), and M would be miscounted
    tag = torch.tensor([_baseline_tag(base)], dtype=torch.int64, device=dev)
    tags = torch.empty(world, dtype=torch.int64, device=dev)
    _all_gather(tags, tag, group)
    if not bool((tags == tags[0]).all()):
        raise RuntimeError("merge_device_maps: the ranks hold different merge baselines (a handle changed outside the merge); "
                           "set m._merge_base = None on every rank and merge maps that do not share a loaded-back past")
    # (one rank: the sum of one map is the map — no increments, so that a periodic merge leaves its log-odds bits alone)
    subtract, add_base = _baseline_ops(base if world > 1 else None, union, dev)
    if subtract is not None:
        subtract(dense)

    def finish(lo: torch.Tensor, sn: torch.Tensor) -> torch.Tensor:
        assert lo.is_contiguous() and sn.is_contiguous()
        occ = torch.empty_like(sn)
        torch.cuda.synchronize()
        m.merge_finish(lo.data_ptr(), sn.data_ptr(), lo.numel(), occ.data_ptr())
        return occ

    out_lo, out_occ = _exchange(dense, seen, finish, world, group, add_base) if n_pad else (dense, seen)
    torch.cuda.synchronize()
    merged = {"keys": ukeys, "log_odds": out_lo[:n_u], "occ": out_occ[:n_u]}
    if load_back and n_u:
        m.import_blocks((merged["keys"].data_ptr(), n_u), log_odds=merged["log_odds"].data_ptr(), occ=merged["occ"].data_ptr())
        m._merge_base = {"packed": union[:n_u].clone(), "log_odds": merged["log_odds"].clone()}
    return merged
