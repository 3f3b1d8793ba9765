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
    (SURVEY 8e): returns rank of every unit.  Deterministic (ties: lower rank, earlier unit)."""
    import heapq
    costs = np.asarray(costs, np.float64)
    if world_size <= 1 or len(costs) == 0:
        return np.zeros(len(costs), np.int64)
    order = np.argsort(-costs, kind="stable")
    cl = costs[order].tolist()
    heap = [(0.0, r) for r in range(world_size)]           # (load, rank): the least loaded rank, lowest rank on ties
    seq = []
    for cu in cl:
        load, r = heap[0]
        seq.append(r)
        heapq.heapreplace(heap, (load + cu, r))
    owner = np.empty(len(costs), np.int64)
    owner[order] = seq
    return owner


def deal_units(costs: np.ndarray, world_size: int) -> np.ndarray:
    """Many units (precursor windows: hundreds to thousands per partition) dealt to ranks in one vectorised step: units by
    decreasing cost, ranks in boustrophedon order (0 .. w-1, w-1 .. 0, ...).  Every rank gets the same MIX of units, so an
    error of the cost model (per-spectrum cost drifts with m/z and charge) hits all ranks alike.  Within a fraction of a per
    cent of the LPT deal when costs vary smoothly, and O(n log n) in numpy where the heap loop of `shard_units` costs
    milliseconds of Python per pass.  Deterministic: every rank derives the same deal from the same counts."""
    costs = np.asarray(costs, np.float64)
    if world_size <= 1 or len(costs) == 0:
        return np.zeros(len(costs), np.int32)
    order = np.argsort(-costs, kind="stable")
    i = np.arange(len(costs))
    r = i % (2 * world_size)
    owner = np.empty(len(costs), np.int32)
    owner[order] = np.where(r < world_size, r, 2 * world_size - 1 - r)
    return owner


def deal_job(costs_list, world_size: int, tol: float = 0.03):
    """The deal of a job of several partitions (precursor charges, falcon.py:151-160) to `world_size` GPUs; units = the
    (partition, precursor window) blocks, `costs_list[j]` = the window costs of partition j.
      * whole partitions, by longest-processing-time, when that balances within `tol` of the mean load: a rank then runs its
        partitions exactly as one GPU would, with no per-partition work for the others (many similar partitions);
      * otherwise every window of every partition is dealt on its own (`deal_units` over all of them): every rank gets a
        like mix of windows of every partition -- unless that deal is itself off by more than `tol` (heavy-tailed window
        sizes): then by longest-processing-time (`shard_units`) over all windows.
    Deterministic: every rank derives the same deal from the same counts.  -> [owner int32[n_windows_j] per partition]"""
    sizes = [len(c) for c in costs_list]
    totals = np.array([float(np.sum(c)) for c in costs_list], np.float64)
    if world_size <= 1 or totals.sum() <= 0:
        return [np.zeros(n, np.int32) for n in sizes]
    whole = shard_units(totals, world_size)
    loads = np.bincount(whole, weights=totals, minlength=world_size)
    if loads.max() <= (1.0 + tol) * loads.mean():
        return [np.full(n, whole[j], np.int32) for j, n in enumerate(sizes)]
    allc = np.concatenate([np.asarray(c, np.float64) for c in costs_list])
    owner = deal_units(allc, world_size)
    # a heavy-tailed job (a few windows tens of times the median: real precursor distributions, synth skew=True) leaves the
    # boustrophedon deal unbalanced -- its first `world_size` units go one to each rank whatever they cost (10 M skewed
    # spectra, 8 ranks: worst / mean 1.29 modelled, 1.18 measured, profiles/NOTES.md r5): then longest-processing-time over
    # all units (0.6 ms of heap for 1,600 windows; modelled 1.00)
    loads = np.bincount(owner, weights=allc, minlength=world_size)
    if loads.max() > (1.0 + tol) * loads.mean():
        # longest-processing-time PER PARTITION first: every rank then holds 1 / N of every partition, so its partitions keep
        # the job's proportions and run side by side on the rank's slots like everyone else's (round 6: LPT over all windows at
        # once gave one of 8 ranks most of its load in ONE charge partition -- same total work, 24 -> 31 ms on the two-slot
        # runner, tools/shard_share.py 8 skew); LPT over all windows only where a partition cannot be balanced by itself (a
        # window above 1 / N of its partition).  The better of the candidates is kept.
        per_part = np.concatenate([shard_units(np.asarray(c, np.float64), world_size) for c in costs_list]).astype(np.int32)
        best = loads.max()
        for cand in (per_part, shard_units(allc, world_size).astype(np.int32)):
            m = np.bincount(cand, weights=allc, minlength=world_size).max()
            if m < best:
                owner, best = cand, m
            if best <= (1.0 + tol) * loads.mean():
                break
    out, at = [], 0
    for n in sizes:
        out.append(owner[at:at + n].copy())
        at += n
    return out


NEIGHBOUR_UNITS = 100.0      # cost of one STORED neighbour in units of one scanned (query, candidate) pair: the exact chains of the
                             # kept pairs, the k-th key resolution, DBSCAN, refinement and medoids of the 10 M job take 26 ms for 99 M
                             # stored neighbours against 51 ms of scan + top-k for 10.3 G pairs (profiles/NOTES.md r6)
CLUSTER_NEIGHBOURS = 8.0     # stored neighbours of a row that are members of its own cluster (data-dependent; a constant here)


def window_costs(counts: np.ndarray, batch_size: int, n_probe: int, mz_interval: Optional[float] = None, tol=None,
                 k_ann: int = 128, n_neighbors: int = 64) -> np.ndarray:
    """Estimated cost of every precursor window from its spectrum count alone: the window becomes ceil(count / batch_size)
    buckets (the chunk rule of cluster.py:197-207; gaps inside a window, which would split it further, are not known
    before the window is sorted -- they only make the estimate pessimistic), each costed like `bucket_costs`.

    `mz_interval` + `tol` = (mass, "ppm" | "Da") given (round 6): plus the work that follows the STORED neighbours (exact pair
    chains, k-th key resolution, DBSCAN, refinement, medoids).  A row stores its cluster's members and the share of its k_ann
    best candidates that passes the precursor tolerance: 2 tol / (m/z width of its bucket) -- and a full window is cut into
    `chunks` buckets of 1 / chunks of the window's width, so its rows keep more (measured under skew: 16.3 stored neighbours per
    row on the rank with the fullest windows against 10.5-11.5 elsewhere, the deal 8 % off the model: NOTES r5).  `counts` may
    be 2-D [partitions, windows]: the window index (column) gives the m/z a ppm tolerance is taken at."""
    counts = np.asarray(counts, np.int64)
    chunks = np.maximum(1, -(-counts // max(int(batch_size), 1)))
    size = counts // chunks                                                   # <= batch_size
    if batch_size > (1 << 20):
        from .cluster.cluster import n_list_rule
        cost = chunks * bucket_costs(size, n_list_rule(size, n_probe), n_probe)
    else:
        cost = chunks * _bucket_cost_table(int(batch_size), int(n_probe))[size]
    if mz_interval and mz_interval > 0 and tol is not None:
        w = np.broadcast_to(np.arange(counts.shape[-1], dtype=np.float64), counts.shape)
        tol_mz = float(tol[0]) if tol[1] == "Da" else float(tol[0]) * 1e-6 * (w + 0.5) * float(mz_interval)
        passing = np.minimum(1.0, 2.0 * tol_mz * chunks / float(mz_interval))
        stored = np.minimum(float(n_neighbors), CLUSTER_NEIGHBOURS + float(k_ann) * passing)
        cost = cost + NEIGHBOUR_UNITS * counts * stored
    return cost


_cost_tables = {}


def _bucket_cost_table(batch_size: int, n_probe: int) -> np.ndarray:
    """`bucket_costs` of a bucket of 0 .. batch_size spectra with the lists `n_list_rule` gives it (the plan of a multi-GPU
    step looks tens of thousands of windows up per step)"""
    key = (batch_size, n_probe)
    if key not in _cost_tables:
        from .cluster.cluster import n_list_rule
        size = np.arange(max(batch_size, 1) + 1, dtype=np.int64)
        _cost_tables[key] = bucket_costs(size, n_list_rule(size, n_probe), n_probe)
    return _cost_tables[key]


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

    def start(self, indptr, idx, dist, labels, n_labels_local: int, rows=None):
        """indptr i64[n+1] / idx i32[>=nnz] / dist f32[>=nnz] from `Context.neighbors_to_csr` (ids already
        global), labels i32[n] local labels in [0, n_labels_local).  `rows` (optional, i32/i64[n]): the global
        row of every local row -- a rank that owns a bucket shard of ONE dataset sends them so that every
        receiver can place its labels (`assemble_labels`)."""
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
        # payload: one int32 block per rank  [counts n_max | labels n_max | (rows n_max) | idx nnz_max | dist bits nnz_max]
        nb = 3 if rows is not None else 2
        width = nb * n_max + 2 * nnz_max
        send = torch.empty(width, dtype=torch.int32, device=indptr.device)
        send[:n] = (indptr[1:] - indptr[:-1]).to(torch.int32)
        send[n_max:n_max + n] = labels.to(torch.int32)
        if rows is not None:
            send[2 * n_max:2 * n_max + n] = rows.to(torch.int32)
        send[nb * n_max:nb * n_max + nnz] = idx[:nnz]
        send[nb * n_max + nnz_max:nb * n_max + nnz_max + nnz] = dist[:nnz].view(torch.int32)
        if ws > 1:
            recv = torch.empty(ws * width, dtype=torch.int32, device=indptr.device)
            work = d.all_gather_into_tensor(recv, send, async_op=True)   # ONE collective for the whole payload
        else:
            recv, work = send, None
        return dict(work=work, recv=recv, send=send, sizes=sizes, n_max=n_max, nnz_max=nnz_max, width=width, nb=nb)

    def finish(self, h):
        import torch
        rank, ws = world()
        if h["work"] is not None:
            h["work"].wait()
        sizes, n_max, nnz_max, width, nb = h["sizes"], h["n_max"], h["nnz_max"], h["width"], h["nb"]
        recv = h["recv"].view(ws, width)
        label_off = np.concatenate([[0], np.cumsum(sizes[:, 2])])
        out = dict(counts=[], idx=[], dist=[], labels=[], rows=[], sizes=sizes, n_labels=int(label_off[-1]))
        for r in range(ws):
            n_r, nnz_r = int(sizes[r, 0]), int(sizes[r, 1])
            blk = recv[r]
            out["counts"].append(blk[:n_r])
            out["labels"].append(blk[n_max:n_max + n_r] + int(label_off[r]))       # falcon.py:189-193 per rank
            if nb == 3:
                out["rows"].append(blk[2 * n_max:2 * n_max + n_r])
            out["idx"].append(blk[nb * n_max:nb * n_max + nnz_r])
            out["dist"].append(blk[nb * n_max + nnz_max:nb * n_max + nnz_max + nnz_r].view(torch.float32))
        return out

    @staticmethod
    def assemble_labels(g, n_total: int):
        """labels of the WHOLE dataset from the gathered shards (`rows` sent): labels[row] = that row's globally unique
        label (rank-major offsets, the per-block offset of cluster.py:144-155).  Device tensor i32[n_total]."""
        import torch
        dev = g["labels"][0].device
        out = torch.full((n_total,), -1, dtype=torch.int32, device=dev)
        for rows, lab in zip(g["rows"], g["labels"]):
            out[rows.long()] = lab
        return out


# xGMI on one MI355X node: 7 links x ~153 GB/s per GPU, point to point
XGMI_LINKS, XGMI_LINK_GBS = 7, 153.0


def expected_exchange_bytes(n_total: int, world_size: int, nnz_per_row: float, with_rows: bool = True, imbalance: float = 1.0):
    """What ONE exchange step of a job of `n_total` spectra dealt to `world_size` GPUs moves, from the layout of
    `SparseGraphExchange.start` alone -- a number to hold the first real multi-GPU run against.  Every
    rank sends one int32 block [counts n_max | labels n_max | (rows n_max) | idx nnz_max | dist nnz_max] padded to the largest
    rank's sizes (`imbalance` = largest share / mean share) and receives `world_size` of them.  `nnz_per_row`: stored
    neighbours per spectrum after the precursor filter (CSR, not the [n, n_neighbors] ELL: 9.94 on the synthetic 10 M job,
    measured by tools/shard_share.py).  Times: the payload over all 7 xGMI links at once (direct fan-out) and over one link
    (what a ring all-gather is bound by) -- lower bounds, no latency terms."""
    ws = max(1, int(world_size))
    n_max = int(np.ceil(n_total / ws * imbalance))
    nnz_max = int(np.ceil(n_max * nnz_per_row))
    width = 4 * ((3 if with_rows else 2) * n_max + 2 * nnz_max)
    recv = (ws - 1) * width                                       # from the other ranks
    return {"world_size": ws, "rows_per_rank": n_max, "nnz_per_rank": nnz_max, "block_bytes_per_rank": width,
            "received_bytes_per_rank": recv, "gathered_bytes_total": ws * width,
            "ms_fan_out_7_links": recv / (XGMI_LINKS * XGMI_LINK_GBS * 1e9) * 1e3 if ws > 1 else 0.0,
            "ms_ring_one_link": recv / (XGMI_LINK_GBS * 1e9) * 1e3 if ws > 1 else 0.0}


def _self_check_payload(r: int, n_neighbors: int, scale: int):
    """rank r's synthetic ragged payload (numpy, a function of r alone: every rank can rebuild every other rank's)"""
    rng = np.random.default_rng([2026, r])
    n = scale * (3 + (r * 7) % 5) + 17 * r + (0 if r % 3 else 1)          # different row counts per rank, some odd
    counts = rng.integers(0, n_neighbors + 1, n).astype(np.int64)
    counts[rng.random(n) < 0.2] = 0                                        # rows without neighbours
    indptr = np.concatenate([[0], np.cumsum(counts)]).astype(np.int64)
    nnz = int(indptr[-1])
    idx = rng.integers(0, 2 ** 31 - 1, nnz).astype(np.int32)
    dist = rng.random(nnz).astype(np.float32)
    n_labels = max(1, n // 3)
    labels = rng.integers(0, n_labels, n).astype(np.int32)
    rows = (np.arange(n, dtype=np.int64) * 8 + r).astype(np.int32)         # interleaved global rows
    return indptr, idx, dist, labels, n_labels, rows


def exchange_self_check(device, n_neighbors: int = 64, scale: int = 20000, rounds: int = 3):
    """Collective self-test of the exchange step alone: every rank sends a synthetic ragged CSR payload
    (different sizes per rank, empty rows) through `SparseGraphExchange` -- the calls of the real job: one sizes
    `all_gather_into_tensor`, one padded payload `all_gather_into_tensor(async_op=True)` -- and verifies EVERY rank's received
    piece byte for byte against that rank's payload rebuilt locally.  On the first multi-GPU run this separates collective
    faults (RCCL / xGMI / the padding arithmetic) from pipeline faults.  -> dict (identical on every rank: the verdict is
    all-reduced)."""
    import time
    import torch
    d = _dist()
    rank, ws = world()
    ex = SparseGraphExchange(device)
    mine = _self_check_payload(rank, n_neighbors, scale)
    t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(device)
    indptr, idx, dist_, labels, rows = t(mine[0]), t(mine[1]), t(mine[2]), t(mine[3]), t(mine[5])
    bad, ms = [], []
    for rnd in range(rounds):
        if ws > 1:
            d.barrier()
        t0 = time.perf_counter()
        g = ex.finish(ex.start(indptr, idx, dist_, labels, mine[4], rows=rows))
        if device.type == "cuda":
            torch.cuda.synchronize(device)
        ms.append((time.perf_counter() - t0) * 1e3)
        label_off = 0
        for r in range(ws):
            e = _self_check_payload(r, n_neighbors, scale)
            exp = {"counts": np.diff(e[0]).astype(np.int32), "idx": e[1], "dist": e[2].view(np.int32),
                   "labels": e[3] + label_off, "rows": e[5]}
            label_off += e[4]
            for key, want in exp.items():
                got = g[key][r]
                got = (got.view(torch.int32) if got.dtype == torch.float32 else got).cpu().numpy()
                if got.shape != want.shape or not np.array_equal(got, want):
                    bad.append(f"round {rnd}: rank {rank} received a wrong `{key}` piece of rank {r} "
                               f"({got.shape} vs {want.shape})")
        if g["n_labels"] != label_off:
            bad.append(f"round {rnd}: n_labels {g['n_labels']} != {label_off}")
    ok = torch.tensor([0 if bad else 1], dtype=torch.int32, device=device)
    seen = torch.tensor([rank], dtype=torch.int32, device=device)
    if ws > 1:
        d.all_reduce(ok, op=d.ReduceOp.MIN)
        all_seen = torch.empty(ws, dtype=torch.int32, device=device)
        d.all_gather_into_tensor(all_seen, seen)
        seen = all_seen
    payload = sum(int(a.nbytes) for a in (np.diff(mine[0]).astype(np.int32), mine[1], mine[2], mine[3], mine[5]))
    # what the collective itself moved here (padded blocks), and what the real strong-scaling job is expected to move
    sizes = [(lambda e: (len(e[3]), int(e[0][-1])))(_self_check_payload(r, n_neighbors, scale)) for r in range(ws)]
    width = 4 * (3 * max(s[0] for s in sizes) + 2 * max(max(s[1] for s in sizes), 1))
    return {"ok": bool(int(ok.item()) == 1), "world_size": ws, "ranks_seen": [int(x) for x in seen.cpu().tolist()],
            "local_errors": bad[:5], "payload_bytes_this_rank": payload, "block_bytes_per_rank": width,
            "gathered_bytes_total": ws * width, "ms_per_exchange": [round(x, 3) for x in ms],
            "gbs_per_rank_received": [round((ws - 1) * width / (x * 1e-3) / 1e9, 2) for x in ms] if ws > 1 else None,
            "expected_strong_10M_job": expected_exchange_bytes(10_000_000, ws, 9.94)}


def start_graph_exchange(ctx, exchange: "SparseGraphExchange", outs, lasts, part_off, n_neighbors: int, sharded: bool,
                         with_neighbors: bool = True, csr_buf: Optional[dict] = None):
    """The one exchange step of a pass over a job of several partitions (precursor charges, falcon.py:151-193), as
    `bench.py --gpus N` and the world-size-2 HIP test run it: this rank's neighbour lists of every partition chain into
    ONE CSR on the device (`fal_neighbors_to_csr_mapped`: ids -> dataset rows of the job, partition j's rows start at
    part_off[j]), the labels get the per-partition offsets, and `exchange.start` ships [counts | labels | rows | idx |
    dist].  `outs` / `lasts`: what `ClusterPipeline.run_many(..., shard=...)` returned / left in `.lasts`.
    -> (handle for `exchange.finish`, local labels i32, number of local labels).  `csr_buf`: a dict the caller keeps so
    that the CSR buffers are allocated once."""
    import torch
    dev = ctx.tdev
    labels_all, current = [], 0
    for labels, medoids in outs:
        labels_all.append(labels + current)                  # falcon.py:189-193
        current += int(medoids.numel())
    labels = torch.cat(labels_all) if labels_all else torch.empty(0, dtype=torch.int32, device=dev)
    rows_local = sum(int(o[0].numel()) for o in outs)
    rows_g = None
    if sharded:
        rows_g = torch.cat([last["rows"].to(torch.int32) + int(part_off[j]) for j, last in enumerate(lasts)])
    if with_neighbors:
        csr_buf = csr_buf if csr_buf is not None else {}
        if csr_buf.get("rows", -1) < rows_local:
            cap = max(rows_local, 1) * n_neighbors
            csr_buf["buf"] = (torch.empty(rows_local + 1, dtype=torch.int64, device=dev),
                              torch.empty(cap, dtype=torch.int32, device=dev),
                              torch.empty(cap, dtype=torch.float32, device=dev))
            csr_buf["rows"] = rows_local
        row0 = 0
        csr = (csr_buf["buf"][0][:rows_local + 1], csr_buf["buf"][1], csr_buf["buf"][2])
        if rows_local == 0:
            csr[0].zero_()
        for j, last in enumerate(lasts):                     # charge partitions chain into one CSR on the device
            if not last or "nb_idx" not in last or last["nb_idx"].shape[0] == 0:
                continue
            ctx.neighbors_to_csr(last["nb_idx"], last["nb_dist"], int(part_off[j]), out=csr_buf["buf"], row0=row0,
                                 nb_count=last.get("nb_count"), id_map=last.get("rows"))
            row0 += last["nb_idx"].shape[0]
    else:                                                    # labels only: an empty graph
        csr = (torch.zeros(labels.numel() + 1, dtype=torch.int64, device=dev),
               torch.empty(1, dtype=torch.int32, device=dev), torch.empty(1, dtype=torch.float32, device=dev))
    return exchange.start(csr[0], csr[1], csr[2], labels, current, rows=rows_g), labels, current


# ---------------------------------------------------------------------------------------------
# One dataset, N GPUs: buckets -> ranks by cost (LPT), every rank runs the path on its buckets,
# one all-gatherv of (dataset rows, labels, medoids).  SURVEY 8e; reference analogue: blocks are
# clustered independently and their labels concatenated with offsets (cluster.py:115-155).
# ---------------------------------------------------------------------------------------------
def bucket_costs(sizes: np.ndarray, n_list: np.ndarray, n_probe: int) -> np.ndarray:
    """scan cost of a bucket ~ n_b * candidates per query: n_b for a flat bucket, the probed fraction of
    it for an IVF bucket, plus the k-means passes (11 x n_list centroids per row)."""
    sizes = np.asarray(sizes, np.float64)
    n_list = np.asarray(n_list, np.float64)
    cand = np.where(n_list <= 1, sizes, sizes * np.minimum(1.0, n_probe / np.maximum(n_list, 1.0)) + 12.0 * n_list)
    return sizes * cand


def shard_buckets(splits: np.ndarray, owner: np.ndarray, rank: int):
    """this rank's buckets (whole and in order): -> (first sorted position of each, sizes, the bucket boundaries of the
    subset, bucket ids).  Per-bucket arrays only: the per-row expansion is `shard_rows` (host) or two device ops."""
    mine = np.flatnonzero(owner == rank)
    sizes = (splits[1:] - splits[:-1])[mine].astype(np.int64)
    sub_splits = np.concatenate([[0], np.cumsum(sizes)]).astype(np.int64)
    return np.asarray(splits[:-1], np.int64)[mine], sizes, sub_splits, mine


def shard_rows(splits: np.ndarray, owner: np.ndarray, rank: int):
    """sorted positions of the rows of this rank's buckets (buckets stay whole and in order), the
    bucket boundaries of that subset, and the bucket ids."""
    first, sizes, sub_splits, mine = shard_buckets(splits, owner, rank)
    if len(mine) == 0:
        return np.zeros(0, np.int64), sub_splits, mine
    rows = np.repeat(first - sub_splits[:-1], sizes) + np.arange(sub_splits[-1])
    return rows.astype(np.int64), sub_splits, mine


def merge_shards(n_total: int, shards):
    """shards: per rank (dataset_rows i64[n_r], labels i32[n_r] in [0, n_labels_r), medoid_rows i32[n_labels_r]).
    -> labels i32[n_total] made globally unique rank by rank (the per-block offset of cluster.py:144-155),
    medoids i32[sum n_labels_r] (dataset rows)."""
    labels = np.full(n_total, -1, np.int32)
    medoids, off = [], 0
    for rows, lab, med in shards:
        labels[np.asarray(rows, np.int64)] = np.asarray(lab, np.int32) + off
        medoids.append(np.asarray(med, np.int32))
        off += len(med)
    if (labels < 0).any():
        raise ValueError("merge_shards: some spectra were not assigned to any rank")
    return labels, (np.concatenate(medoids) if medoids else np.zeros(0, np.int32))


def run_sharded(pipe, ds, precursor_tol_mass: float, precursor_tol_mode: str, rt_tol, fragment_tol: float,
                batch_size: int, p, rank: Optional[int] = None, world_size: Optional[int] = None, gather=None,
                local_only: bool = False):
    """Cluster ONE dataset on `world_size` GPUs.  Every rank holds the dataset, sorts it and derives the same
    bucket boundaries (cheap, deterministic); buckets are dealt to ranks by longest-processing-time on
    `bucket_costs`; a rank runs vectorise -> ... -> labels on its buckets only; one all-gatherv assembles the
    result on every rank.  `gather(rows, labels, medoids) -> [(rows, labels, medoids) per rank]` defaults to
    the `torch.distributed` all-gatherv of this module.  -> (labels i32[N], medoids i32[n_labels]) numpy
    (`local_only`: this rank's (dataset rows, labels, medoid rows) without the collective, for `merge_shards`).
    The partition equals the single-GPU one; label ids are rank-major instead of bucket-major."""
    import torch
    r0, w0 = world()
    rank = r0 if rank is None else rank
    world_size = w0 if world_size is None else world_size
    c = pipe.ctx
    n = len(ds)
    st = pipe._front(c, ds, precursor_tol_mass, precursor_tol_mode, rt_tol, batch_size, p)
    sub = pipe._restrict(c, st, p, (rank, world_size))
    order_sub = sub["rows"]
    n_sub = int(order_sub.numel())
    if n_sub:
        pipe._search(ds, sub, precursor_tol_mass, precursor_tol_mode, rt_tol, fragment_tol, p, False)
        sub["order"] = torch.arange(n_sub, dtype=torch.int64, device=c.tdev)      # label the subset locally ...
        labels_sub, medoids_sub, _ = pipe._graph(sub, precursor_tol_mass, precursor_tol_mode, rt_tol, p, False)
        mine_rows = order_sub.cpu().numpy()                                       # ... and map back to dataset rows
        mine_lab = labels_sub.cpu().numpy()
        mine_med = mine_rows[medoids_sub.cpu().numpy()].astype(np.int32)
    else:
        mine_rows, mine_lab, mine_med = np.zeros(0, np.int64), np.zeros(0, np.int32), np.zeros(0, np.int32)
    if local_only:
        return mine_rows, mine_lab, mine_med
    if gather is None:
        gather = lambda a, b, m: _gather_shards(a, b, m, c.tdev)
    return merge_shards(n, gather(mine_rows, mine_lab, mine_med))


def _gather_shards(rows, labels, medoids, device):
    """one all-gatherv round (counts, then max-count padded slots) for the three per-rank arrays"""
    import torch
    rank, ws = world()
    if ws == 1:
        return [(rows, labels, medoids)]
    t_rows = torch.from_numpy(np.ascontiguousarray(rows)).to(device)
    t_lab = torch.from_numpy(np.ascontiguousarray(labels)).to(device)
    t_med = torch.from_numpy(np.ascontiguousarray(medoids)).to(device)
    g_rows, counts = allgatherv_rows(t_rows)
    g_lab, _ = allgatherv_rows(t_lab, counts)
    g_med, mcounts = allgatherv_rows(t_med)
    g_rows, g_lab, g_med = g_rows.cpu().numpy(), g_lab.cpu().numpy(), g_med.cpu().numpy()
    out, a, b = [], 0, 0
    for r in range(ws):
        out.append((g_rows[a:a + counts[r]], g_lab[a:a + counts[r]], g_med[b:b + mcounts[r]]))
        a += counts[r]
        b += mcounts[r]
    return out
