"""Multi-GPU: one process per GPU, `torch.distributed` (backend "nccl" = RCCL over xGMI).

The path shards by independent units -- (charge, precursor-m/z bucket): no neighbour
pair crosses a bucket (reference cluster.py:115-141 processes blocks independently and
only offsets labels afterwards), so every rank runs the whole hot path on its own shard
with NO collective on the data path.  The one exchange step is the all-gatherv of the
results: the sparse neighbour lists (north_star's contract) and/or the labels.

RCCL has no allgatherv; rows per rank differ, so the gather is count exchange
(`all_gather` of one int) + `all_gather` into max-count padded slots, trimmed on
arrival.  On xGMI every rank reaches its 7 peers over dedicated links, so a direct
fan-out all-gather moves each shard once per link.
"""
from __future__ import annotations

from typing import List, Optional, Tuple

import numpy as np


def _dist():
    import torch.distributed as dist
    return dist


def world() -> Tuple[int, int]:
    dist = _dist()
    if dist.is_available() and dist.is_initialized():
        return dist.get_rank(), dist.get_world_size()
    return 0, 1


def shard_units(costs: np.ndarray, world_size: int) -> np.ndarray:
    """Longest-processing-time assignment of independent units (buckets) to ranks
    (SURVEY 8e): returns rank of every unit.  Deterministic."""
    costs = np.asarray(costs, np.float64)
    order = np.argsort(-costs, kind="stable")
    load = np.zeros(world_size)
    owner = np.empty(len(costs), np.int64)
    for u in order:
        r = int(np.argmin(load))
        owner[u] = r
        load[r] += costs[u]
    return owner


def allgather_counts(n_local: int, device) -> List[int]:
    import torch
    dist = _dist()
    rank, ws = world()
    if ws == 1:
        return [int(n_local)]
    t = torch.tensor([int(n_local)], dtype=torch.int64, device=device)
    out = [torch.empty_like(t) for _ in range(ws)]
    dist.all_gather(out, t)
    return [int(x.item()) for x in out]


def allgatherv_rows(t, counts: Optional[List[int]] = None):
    """all-gatherv along dim 0: every rank contributes t[:n_r]; returns the concatenation
    in rank order plus the per-rank counts."""
    import torch
    dist = _dist()
    rank, ws = world()
    if ws == 1:
        return t, [int(t.shape[0])]
    if counts is None:
        counts = allgather_counts(t.shape[0], t.device)
    mx = max(counts)
    pad = t
    if t.shape[0] < mx:
        pad = torch.zeros((mx,) + tuple(t.shape[1:]), dtype=t.dtype, device=t.device)
        pad[: t.shape[0]] = t
    slots = [torch.empty((mx,) + tuple(t.shape[1:]), dtype=t.dtype, device=t.device) for _ in range(ws)]
    dist.all_gather(slots, pad.contiguous())          # one collective (RCCL: a single all-gather)
    return torch.cat([slots[r][: counts[r]] for r in range(ws)], 0), counts


def allgatherv_neighbors(nb_idx, nb_dist, row_offset: int):
    """The contract exchange: every rank's [n_r, k] neighbour lists (ids local to the
    rank's shard, -1 = empty) -> the global sparse graph [sum n_r, k] with ids shifted
    to global rows."""
    import torch
    gi = torch.where(nb_idx >= 0, nb_idx + int(row_offset), nb_idx)
    gi, counts = allgatherv_rows(gi)
    gd, _ = allgatherv_rows(nb_dist, counts)
    return gi, gd, counts


def allgatherv_labels(labels, n_labels_local: int):
    """labels of every shard made globally unique (rank r's labels are offset by the
    number of labels of ranks < r, like the per-charge offset of falcon.py:189-193)."""
    counts = allgather_counts(int(n_labels_local), labels.device)
    rank, ws = world()
    off = int(sum(counts[:rank]))
    g, _ = allgatherv_rows(labels + off)
    return g, counts
