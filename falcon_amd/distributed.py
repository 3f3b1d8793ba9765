"""Multi-GPU: one process per GPU, `torch.distributed` (backend "nccl" = RCCL over xGMI).

The path shards by independent units -- (charge, precursor-m/z bucket): no neighbour
pair crosses a bucket (reference cluster.py:115-141 processes blocks independently and
only offsets labels afterwards), so every rank runs the whole hot path on its own shard
with NO collective on the data path.  The one exchange step is the all-gatherv of the
results: the sparse neighbour lists (north_star's contract) and/or the labels.

RCCL has no allgatherv; rows per rank differ, so the gather is count exchange
(`all_gather` of a few ints) + `all_gather` into max-count padded slots, trimmed on
arrival.  On xGMI every rank reaches its 7 peers over dedicated links, so a direct
fan-out all-gather moves each shard once per link.

What travels is the CSR form of the neighbour lists (`fal_neighbors_to_csr`): after the
precursor filter a row keeps a handful of its <= n_neighbors slots, so the padded ELL
arrays would move ~10x the bytes.  `SparseGraphExchange` issues the collectives
asynchronously (RCCL runs them on its own stream) so that the exchange of one batch of
units overlaps the kernels of the next; `finish()` is the only point that waits.
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


class SparseGraphExchange:
    """One all-gatherv of (row counts, neighbour ids, distances, labels) per call to `start`.

    start(...)  -> handle   enqueue: sizes all_gather (the one host sync), then async all_gathers
    finish(h)   -> dict     wait, trim the padded slots: global CSR + globally unique labels

    Per-rank pieces stay separate views (`counts[r]`, `idx[r]`, `dist[r]`, `labels[r]`): a consumer
    that needs one array concatenates, the bench only needs the data to have arrived.
    """

    def __init__(self, device):
        self.device = device

    def start(self, indptr, idx, dist, labels, n_labels_local: int):
        """indptr i64[n+1] / idx i32[>=nnz] / dist f32[>=nnz] from `Context.neighbors_to_csr` (ids already
        global), labels i32[n] local labels in [0, n_labels_local)."""
        import torch
        d = _dist()
        rank, ws = world()
        n = int(indptr.numel()) - 1
        # sizes: [n, nnz, n_labels] of every rank; nnz lives on the device -> gathered there, read once
        mine = torch.empty(3, dtype=torch.int64, device=indptr.device)
        mine[0] = n
        mine[1] = indptr[n]
        mine[2] = int(n_labels_local)
        if ws > 1:
            sizes_t = torch.empty(3 * ws, dtype=torch.int64, device=indptr.device)
            d.all_gather_into_tensor(sizes_t, mine)
        else:
            sizes_t = mine
        sizes = sizes_t.cpu().numpy().reshape(ws, 3)                      # the one synchronisation
        n_max, nnz_max = int(sizes[:, 0].max()), max(int(sizes[:, 1].max()), 1)
        nnz = int(sizes[rank, 1])
        # payload: one int32 block per rank  [counts n_max | labels n_max | idx nnz_max | dist bits nnz_max]
        width = 2 * n_max + 2 * nnz_max
        send = torch.empty(width, dtype=torch.int32, device=indptr.device)
        send[:n] = (indptr[1:] - indptr[:-1]).to(torch.int32)
        send[n_max:n_max + n] = labels.to(torch.int32)
        send[2 * n_max:2 * n_max + nnz] = idx[:nnz]
        send[2 * n_max + nnz_max:2 * n_max + nnz_max + nnz] = dist[:nnz].view(torch.int32)
        if ws > 1:
            recv = torch.empty(ws * width, dtype=torch.int32, device=indptr.device)
            work = d.all_gather_into_tensor(recv, send, async_op=True)   # ONE collective for the whole payload
        else:
            recv, work = send, None
        return dict(work=work, recv=recv, send=send, sizes=sizes, n_max=n_max, nnz_max=nnz_max, width=width)

    def finish(self, h):
        import torch
        rank, ws = world()
        if h["work"] is not None:
            h["work"].wait()
        sizes, n_max, nnz_max, width = h["sizes"], h["n_max"], h["nnz_max"], h["width"]
        recv = h["recv"].view(ws, width)
        label_off = np.concatenate([[0], np.cumsum(sizes[:, 2])])
        out = dict(counts=[], idx=[], dist=[], labels=[], sizes=sizes, n_labels=int(label_off[-1]))
        for r in range(ws):
            n_r, nnz_r = int(sizes[r, 0]), int(sizes[r, 1])
            blk = recv[r]
            out["counts"].append(blk[:n_r])
            out["labels"].append(blk[n_max:n_max + n_r] + int(label_off[r]))       # falcon.py:189-193 per rank
            out["idx"].append(blk[2 * n_max:2 * n_max + nnz_r])
            out["dist"].append(blk[2 * n_max + nnz_max:2 * n_max + nnz_max + nnz_r].view(torch.float32))
        return out
