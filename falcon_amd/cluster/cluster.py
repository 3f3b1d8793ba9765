"""Drop-in for `falcon.cluster.cluster.generate_clusters` (reference
falcon/cluster/cluster.py:24-156; call site falcon/falcon.py:178-188) with the
distance + linkage core replaced by the README's vectorise -> ANN -> DBSCAN path,
every stage of which runs in libfalcon_hip.so on one MI355X.

Host logic only: argument handling, the per-bucket `n_list` rule and the order of the
C-ABI calls.  No CPU fallback exists.
"""
from __future__ import annotations

import logging
import math
from dataclasses import dataclass
from typing import Optional, Tuple

import numpy as np

from .. import device as _device

logger = logging.getLogger("falcon")

F16_INDEX_DIMS = (64, 128, 256, 400, 800)   # row widths the float16 index kernels (assign16 / list16) are instantiated for =
                                            # the widths `device.row_width(low_dim)` pads the rows to
F16_FLAT_DIMS = (64, 128, 256, 400)         # ... the on-chip top-k of flat buckets (fused.hip) and the hi/lo split scan
FLAT_MAX = 100           # buckets up to this size use a flat (single list) index
MIN_PTS_PER_LIST = 39    # [SURVEY App. A] Faiss' minimum points per centroid
MAX_N_LIST = 1 << 17


@dataclass
class AnnParams:
    """The README's nearest-neighbour options (README.md:73-79, 107-117) plus the build's."""
    eps: float = 0.1
    low_dim: int = 400            # any integer in [1, 800] (README.md:114-117): the hash modulus; rows are stored
                                  # `device.row_width(low_dim)` columns wide (64 / 128 / 256 / 400 / 800), zero behind low_dim
    n_probe: int = 16
    n_neighbors: int = 64
    n_neighbors_ann: int = 128
    mz_interval: float = 1.0      # fixed precursor windows in m/z (0 = off) [SURVEY 8(d)]
    kmeans_iters: int = 10        # Faiss IVF default (cp.niter = 10)
    hash_seed: int = 0
    min_mz: float = 101.0
    max_mz: float = 1500.0
    dtype: str = "f32"            # "f32", or "f16": float16 vectors (BASELINE config 5).  The similarity of two float16 vectors is
                                  # the float32 fmaf chain over their (exact) float32 images: buckets with an index run the
                                  # float32 IVF path on the images with the float16 rows as prefilter copies -- bit-identical to
                                  # the oracle's "round to float16, then the float32 path"; flat buckets are scanned on the f16
                                  # matrix cores (float32 accumulation, within 2e-6 of the chain)
    f16_index: bool = True        # dtype "f16": buckets with n_list > n_probe get their k-means index (README.md:107-113); off =
                                  # every bucket exhaustively on the f16 matrix cores (the round-2 behaviour)
    scan: str = "f32"             # flat-bucket scan arithmetic for float32 vectors: "f32" (exact fp32 MFMA) or
                                  # "f16x3" (hi/lo float16 split, 3 f16 MFMAs per step, ~3e-7 absolute error)
    prefilter: bool = False       # float32 flat buckets: keep the top-k on chip (f16-MFMA prefilter + exact float32
                                  # refinement, fused.hip); results are bit-identical with and without it.  Off by default:
                                  # at BASELINE configs[1]'s bucket sizes (<= ~1,100 rows) it is on par with the staged
                                  # scan + select, it pays on larger flat buckets
    ivf_prefilter: bool = True    # float32 buckets with an index: fine scan on the f16 matrix cores to 16-bit keys, the k-th best
                                  # key of every query bracketed from them, exact float32 work only inside the precursor
                                  # window and where it decides the k-th key (ivf16.hip); bit-identical neighbour lists
    kmeans_prefilter: bool = True # IVF buckets with <= 2,048 lists: k-means assignment on the f16 matrix cores, rows whose two
                                  # best centroids are closer than the float16 error bound re-evaluated exactly in float32
                                  # (assign16.hip): the index is identical, the build several times faster
    rescore: bool = False         # re-score the ANN neighbours with the reference's matched-peak cosine
                                  # (similarity.py:17-80) before DBSCAN; uses fragment_tol and min_matches
    min_matches: int = 0          # (set from generate_clusters' `min_matches` when rescore is on)
    clustering: str = "dbscan"    # "dbscan" (README.md:143-146) or "hierarchical": the snapshot's linkage + fcluster at
                                  # the distance threshold (cluster.py:283-290) on the re-scored neighbour graph
    linkage: str = "complete"     # (set from generate_clusters' `linkage` when clustering == "hierarchical")


def n_list_rule(sizes: np.ndarray, n_probe: int) -> np.ndarray:
    """lists per bucket: flat for tiny buckets, else 2^floor(log2(n/39)) [SURVEY App. A];
    a bucket whose lists would ALL be probed (n_list <= n_probe) is searched exhaustively
    anyway, so it is kept flat and k-means is skipped (same result)."""
    sizes = np.asarray(sizes, np.int64)
    nl = np.ones(len(sizes), np.int64)
    big = sizes > FLAT_MAX
    if big.any():
        nl[big] = 2 ** np.floor(np.log2(sizes[big] / MIN_PTS_PER_LIST)).astype(np.int64)
    nl = np.minimum(nl, MAX_N_LIST)
    nl[nl <= n_probe] = 1
    return nl.astype(np.int32)


class SpectrumDataset:
    """The five columns `generate_clusters` reads (cluster.py:73-85) in CSR form:
    precursor_mz f32[N], retention_time f32[N], mz f32[nnz], intensity f32[nnz],
    indptr i64[N+1].  Arrays may be numpy (uploaded) or tensors already on the GPU."""

    def __init__(self, precursor_mz, retention_time, mz, intensity, indptr, precursor_charge=None):
        self.precursor_mz, self.retention_time = precursor_mz, retention_time
        self.mz, self.intensity, self.indptr = mz, intensity, indptr
        self.precursor_charge = precursor_charge

    def __len__(self):
        return int(self.precursor_mz.shape[0])

    def columns(self):
        return (self.precursor_mz, self.retention_time, self.mz, self.intensity, self.indptr)

    def on_host(self) -> bool:
        """True when the columns are numpy arrays or CPU tensors (they have to cross PCIe before the path can start)"""
        return not any(t is not None and hasattr(t, "is_cuda") and t.is_cuda for t in self.columns())

    def to_device(self, dev, non_blocking: bool = True) -> "SpectrumDataset":
        """-> the same dataset with its columns on `dev`, copies enqueued on the CURRENT stream (asynchronous from pinned memory)"""
        import torch
        # the dtypes the path expects (float32 columns, int64 offsets); a column that is None (retention_time: legal, _front
        # checks for it) stays None
        want = (torch.float32, torch.float32, torch.float32, torch.float32, torch.int64)
        up = []
        for t, dt in zip(self.columns(), want):
            if t is None:
                up.append(None)
                continue
            if not isinstance(t, torch.Tensor):
                t = torch.from_numpy(np.ascontiguousarray(t, dtype=np.float32 if dt == torch.float32 else np.int64))
            up.append(t.to(dev, dtype=dt, non_blocking=non_blocking))
        return SpectrumDataset(*up, self.precursor_charge)

    def take_rows(self, rows: np.ndarray) -> "SpectrumDataset":
        """Host-side CSR gather: the sub-dataset of `rows` (any order) of a HOST-resident dataset, in pinned memory where torch
        can pin it -- what one GPU of a multi-GPU job uploads instead of the whole partition (a rank touches the peaks
        of its own windows only; reference analogue: `dataset.take(idx)` per block, cluster.py:107-141)."""
        import torch
        as_np = lambda t: t.numpy() if isinstance(t, torch.Tensor) else np.asarray(t)
        rows = np.asarray(rows, np.int64)
        indptr = as_np(self.indptr)
        cnt = indptr[rows + 1] - indptr[rows]
        out = np.zeros(len(rows) + 1, np.int64)
        np.cumsum(cnt, out=out[1:])
        pos = np.repeat(indptr[rows] - out[:-1], cnt) + np.arange(int(out[-1]), dtype=np.int64)

        def pick(col, idx, dt):
            src = as_np(col)
            t = torch.empty(len(idx), dtype=dt, pin_memory=torch.cuda.is_available())
            dst = t.numpy()
            if src.dtype == dst.dtype:
                np.take(src, idx, out=dst)
            else:
                dst[:] = src[idx]
            return t
        rt = None if self.retention_time is None else pick(self.retention_time, rows, torch.float32)
        return SpectrumDataset(pick(self.precursor_mz, rows, torch.float32), rt, pick(self.mz, pos, torch.float32),
                               pick(self.intensity, pos, torch.float32), torch.from_numpy(out), self.precursor_charge)

    @classmethod
    def from_table(cls, table):
        """pyarrow Table / pandas DataFrame with the Lance schema of falcon.py:275-285."""
        if hasattr(table, "to_pandas"):
            table = table.to_pandas()
        mzs = [np.asarray(x, np.float32) for x in table["mz"]]
        its = [np.asarray(x, np.float32) for x in table["intensity"]]
        indptr = np.zeros(len(mzs) + 1, np.int64)
        np.cumsum([len(x) for x in mzs], out=indptr[1:])
        cat = lambda xs: np.concatenate(xs) if xs else np.zeros(0, np.float32)
        return cls(np.asarray(table["precursor_mz"], np.float32), np.asarray(table["retention_time"], np.float32),
                   cat(mzs), cat(its), indptr,
                   np.asarray(table["precursor_charge"]) if "precursor_charge" in table else None)


class ClusterPipeline:
    """vectorise -> sort -> buckets -> IVF -> search -> filter -> DBSCAN -> refine -> medoids,
    device-resident from the first call to the last."""

    def __init__(self, ctx: Optional[_device.Context] = None, device: int = 0):
        self.ctx = ctx or _device.Context(device)
        self.last = {}

    def plan(self, n: int, batch_size: int, p: "AnnParams"):
        """`fal_ctx_plan` for a pass over `n` spectra: the kernels' code objects loaded, the shape-dependent scratch sized.  The
        reference calls generate_clusters ONCE per charge in a fresh process (falcon.py:153-193): the first pass is the only
        one, so it should not stop for either between its kernels.  Cheap when there is nothing left to do (every `run` calls it)."""
        key = (int(n), int(batch_size), p.low_dim, p.n_neighbors_ann, p.n_probe)
        if getattr(self, "_planned", None) is not None and self._planned[0] >= key[0] and self._planned[1:] == key[1:]:
            return
        self.ctx.plan(n, _device.row_width(p.low_dim), p.n_neighbors_ann, p.n_probe, batch_size)
        self._planned = key

    def trim(self):
        """`fal_ctx_trim` + torch's cache: give this pipeline's cached device memory back (between jobs of very different sizes)"""
        import torch
        for c in (self.ctx, getattr(self, "_front_ctx", None)):
            if c is not None:
                c.trim()
        self._planned = None
        torch.cuda.empty_cache()

    # The path in three phases, so that independent partitions (precursor charges, falcon.py:151-160) can be
    # pipelined: `_front` ends with the path's first host synchronisation (bucket boundaries), `_search` only
    # enqueues, `_graph` ends with the second (cluster counts).
    def _front(self, c, ds, precursor_tol_mass, precursor_tol_mode, rt_tol, batch_size, p, pmz=None):
        """sort by precursor m/z (cluster.py:73-85) + bucket boundaries (a5), on context `c`."""
        import torch
        if pmz is None:
            pmz = c.to_dev(ds.precursor_mz, torch.float32)
        order, mzs = c.sort_by_precursor(pmz)
        rts = c.gather_f32(ds.retention_time, order) if (rt_tol is not None and ds.retention_time is not None) else None
        splits = c.precursor_splits(mzs, precursor_tol_mass, precursor_tol_mode, batch_size, p.mz_interval)
        n_list = n_list_rule(np.diff(splits), p.n_probe)
        if p.dtype == "f16" and not p.f16_index:
            # float16 vectors without an index (AnnParams.f16_index off): every bucket is searched exhaustively on the f16
            # matrix cores -- a superset of what n_probe lists find
            n_list[:] = 1
        return dict(order=order, mzs=mzs, rts=rts, splits=splits, n_list=n_list)

    def plan_shards(self, c, datasets, batch_size, p, world, tol=None):
        """The deal of a multi-partition job to `world` GPUs (`distributed.deal_job`): -> [owner int32[windows] per dataset]
        (an empty array for an empty dataset).  One pass over every partition's precursor column, one wait."""
        import torch
        from .. import distributed as fdist
        # a host-resident partition crosses PCIe with its precursor column only (4 bytes per spectrum) before the deal; the
        # device copy stays with the dataset for the front end of the rank's own windows (`_front_windows`)
        for ds in datasets:
            if ds.on_host():
                ds._pmz_dev = c.to_dev(ds.precursor_mz, torch.float32)
        counts = c.window_counts([getattr(ds, "_pmz_dev", None) if ds.on_host() else ds.precursor_mz for ds in datasets],
                                 p.mz_interval)
        # (`tol` = (precursor tolerance, mode): the cost model then prices the stored neighbours too)
        costs = fdist.window_costs(counts, batch_size, p.n_probe, p.mz_interval, tol, p.n_neighbors_ann, p.n_neighbors)
        return fdist.deal_job(list(costs), world)

    def _front_windows(self, c, ds, precursor_tol_mass, precursor_tol_mode, rt_tol, batch_size, p, shard, owner=None):
        """The front end of ONE rank of a job that shares a dataset among `world` GPUs, without the replicated
        sort: buckets never cross a precursor window floor(mz / mz_interval) and a window's buckets depend on its own spectra
        only (`fal_precursor_splits`), so whole WINDOWS are the unit that is dealt out.  Every rank histograms the window of
        every spectrum (one pass over 4 bytes per spectrum), derives the same deal from the counts
        (`plan_shards`: `distributed.window_costs` / `deal_job` over all partitions of the job; `owner` = this partition's
        part of it), and sorts / buckets only its own spectra.  The buckets -- hence neighbour
        lists and clusters -- are exactly those of the single-GPU pass.  Reference analogue: blocks are clustered
        independently and only their labels are offset afterwards (cluster.py:107-155).
        -> the state `_restrict` returns: order = rows = dataset rows of this rank's spectra in precursor order, ..."""
        import torch
        from .. import distributed as fdist
        rank, world = shard[0], shard[1]
        pmz = ds.__dict__.pop("_pmz_dev", None) if ds.on_host() else None      # (uploaded by this pass's plan_shards)
        if pmz is None:
            pmz = c.to_dev(ds.precursor_mz, torch.float32)
        n = int(pmz.numel())
        if owner is None:
            owner = self.plan_shards(c, [ds], batch_size, p, world, tol=(precursor_tol_mass, precursor_tol_mode))[0]
        if len(owner) and (owner == rank).all():
            # the whole partition is this rank's: the single-GPU front end, every dataset row in precursor order
            st = self._front(c, ds, precursor_tol_mass, precursor_tol_mode, rt_tol, batch_size, p, pmz=pmz)
            st.update(rows=st["order"], n_total=n)
            return st
        if not (owner == rank).any():
            rows = torch.zeros(0, dtype=torch.int64, device=c.tdev)
        else:
            rows, mz_sub = c.window_select(pmz, p.mz_interval, owner, rank)         # ascending dataset rows: the sort stays stable
        if rows.numel() == 0:
            e = rows.new_zeros(0)
            return dict(order=e, mzs=pmz[:0], rts=None, splits=np.zeros(1, np.int64), n_list=np.zeros(0, np.int32), rows=e,
                        n_total=n)
        order_sub, mzs = c.sort_by_precursor(mz_sub)
        rows_sorted = rows[order_sub]
        rts = c.gather_f32(ds.retention_time, rows_sorted) if (rt_tol is not None and ds.retention_time is not None) else None
        splits = c.precursor_splits(mzs, precursor_tol_mass, precursor_tol_mode, batch_size, p.mz_interval)
        n_list = n_list_rule(np.diff(splits), p.n_probe)
        if p.dtype == "f16" and not p.f16_index:
            n_list[:] = 1
        return dict(order=rows_sorted, mzs=mzs, rts=rts, splits=splits, n_list=n_list, rows=rows_sorted, n_total=n)

    def _restrict(self, c, st, p, shard):
        """One dataset on several GPUs (reference analogue: blocks are clustered independently and only
        their labels are offset afterwards, cluster.py:115-155).  Every rank derived the SAME buckets in `_front`;
        buckets are dealt to ranks by longest-processing-time on `distributed.bucket_costs`, and this rank keeps the
        rows of its own buckets: st becomes the state of that subset (buckets whole and in order), plus
        `rows` (i64, device: dataset rows of the subset in sorted order) and `n_total`."""
        import torch
        from .. import distributed as fdist
        rank, world = shard
        splits = np.asarray(st["splits"], np.int64)
        owner = fdist.shard_units(fdist.bucket_costs(np.diff(splits), st["n_list"], p.n_probe), world)
        first, sizes, sub_splits, mine = fdist.shard_buckets(splits, owner, rank)
        n_sub = int(sub_splits[-1])
        if n_sub:
            # sorted positions of the subset's rows, expanded on the device from the per-bucket (first, size) pairs
            shift = c.to_dev(np.stack([first - sub_splits[:-1], sizes]), torch.int64)
            pos_d = torch.repeat_interleave(shift[0], shift[1], output_size=n_sub) + torch.arange(n_sub, device=c.tdev)
        order_sub = st["order"][pos_d] if n_sub else st["order"][:0]
        rts = st["rts"]
        return dict(order=order_sub, mzs=st["mzs"][pos_d] if n_sub else st["mzs"][:0],
                    rts=None if rts is None else (rts[pos_d] if n_sub else rts[:0]), splits=sub_splits,
                    n_list=np.asarray(st["n_list"])[mine], rows=order_sub, n_total=int(st["order"].numel()),
                    buckets=mine)

    def _search(self, ds, st, precursor_tol_mass, precursor_tol_mode, rt_tol, fragment_tol, p, keep_intermediates):
        """a2/a3 vectorise, a6 index, a7 search, a8 filter (+ f4 re-scoring): fills st["nb_idx"], st["nb_dist"]."""
        c = self.ctx
        order, mzs, rts, splits, n_list = st["order"], st["mzs"], st["rts"], st["splits"], st["n_list"]
        n_bins, start, _ = _device.get_dim(p.min_mz, p.max_mz, fragment_tol)      # falcon.py:124-126
        all_flat = bool((n_list == 1).all())
        # `--low_dim` is a free integer (README.md:114-117): the hash runs modulo low_dim, the rows are stored W columns wide
        # (the next width the kernels are instantiated for) with zeros behind -- every kernel below sees d = W
        W = _device.row_width(p.low_dim)
        assert W in F16_INDEX_DIMS
        vec = lambda dt: c.vectorize(ds.mz, ds.intensity, ds.indptr, order, start, fragment_tol, n_bins, p.low_dim,
                                     p.hash_seed, True, dt, width=W)
        X = X16 = Xpre = Xkm = None
        which = 0
        if p.dtype not in ("f32", "f16"):
            raise ValueError(f"unknown dtype {p.dtype!r} (f32 or f16)")
        if p.dtype == "f16" and not all_flat:
            # float16 vectors with an index: the exact kernels (k-means close calls, coarse quantiser, pair chains) work on the
            # float32 image of the rounded rows, the float16 rows themselves are every prefilter copy AND the flat buckets' scan
            X, X16 = vec("f16+image")
            Xkm = X16 if p.kmeans_prefilter else None
            if p.ivf_prefilter and not keep_intermediates:
                which, Xpre = 2, X16
        elif p.dtype == "f16":
            X16 = vec("f16")
        elif p.scan == "f16x3":
            if W not in F16_FLAT_DIMS:
                raise ValueError(f"scan='f16x3' needs low_dim <= {F16_FLAT_DIMS[-1]} (got {p.low_dim})")
            X16 = vec("split16")
            if not all_flat:
                X = vec("f32")              # k-means, coarse quantiser and IVF fine scan stay exact fp32
        else:
            if p.prefilter and not keep_intermediates and W in F16_FLAT_DIMS and bool((n_list == 1).any()):
                which |= 1
            if p.ivf_prefilter and not keep_intermediates and bool((n_list > 1).any()):
                which |= 2
            want_km = p.kmeans_prefilter and bool(((n_list > 1) & (n_list <= 2048)).any())      # (scan.h kAssignMergeLists)
            if want_km or which:
                X, x16 = vec("f32+f16")            # the float16 rounding of the same rows, from the same pass over the peaks
                Xkm = x16 if want_km else None
                Xpre = x16 if which else None
            else:
                X = vec("f32")
        index = c.ivf_build(X, splits, n_list, p.kmeans_iters, X16=X16, Xpre=Xpre, Xkm=Xkm, prefilter_which=max(which, 1))
        if keep_intermediates:
            sim, idx = index.search(p.n_probe, p.n_neighbors_ann)
            nb_idx, nb_dist = c.filter_neighbors(sim, idx, mzs, rts, precursor_tol_mass, precursor_tol_mode, rt_tol,
                                                 p.n_neighbors)
            st.update(sim=sim, idx=idx)
        else:
            # production: a7 + a8 in one call, the [n, k_ann] search result never goes to HBM
            nb_idx, nb_dist = index.search_neighbors(p.n_probe, p.n_neighbors_ann, mzs, rts, precursor_tol_mass,
                                                     precursor_tol_mode, rt_tol, p.n_neighbors)
        if p.rescore:
            nb_dist = c.rescore_neighbors(nb_idx, nb_dist, ds.mz, ds.intensity, ds.indptr, order, fragment_tol,
                                          p.min_matches)
        st.update(X=X, X16=X16, Xpre=Xpre, index=index, nb_idx=nb_idx, nb_dist=nb_dist,
                  nb_count=None if keep_intermediates else index.nb_count)

    def _graph(self, st, precursor_tol_mass, precursor_tol_mode, rt_tol, p, keep_intermediates):
        """a9 DBSCAN, a10 refinement, a11/a12 medoids + labels -> (labels, medoids); `last` describes the partition."""
        c = self.ctx
        order, mzs, rts, nb_idx, nb_dist, index = (st[k] for k in ("order", "mzs", "rts", "nb_idx", "nb_dist", "index"))
        hier = p.linkage if p.clustering == "hierarchical" else None      # f4: the snapshot's linkage + cut at the threshold
        if keep_intermediates:
            # staged calls (one C-ABI call per SURVEY 8a row) so that every intermediate can be inspected
            db, n_db = c.linkage_cluster(nb_idx, nb_dist, p.eps, hier) if hier else c.dbscan(nb_idx, nb_dist, p.eps)
            last = dict(order=order, mz_sorted=mzs, rt_sorted=rts, splits=st["splits"], X=st["X"], X16=st["X16"],
                        n_list=st["n_list"], sim=st["sim"], idx=st["idx"], nb_idx=nb_idx, nb_dist=nb_dist, db=db.clone(),
                        n_db=n_db, index=index)
            lab, n_cl = c.refine_clusters(db, n_db, mzs, rts, precursor_tol_mass, precursor_tol_mode, rt_tol)
            last.update(lab_sorted=lab, n_clusters=n_cl)
            labels, medoids = c.finalize(lab, n_cl, order, nb_idx, nb_dist)
        else:
            # production: a9..a12 fused, counts stay on the device, one synchronisation
            # (after the exact re-scoring a stored neighbour may have moved beyond eps, but it is still stored: counts hold)
            labels, medoids, _, _ = c.cluster_graph(nb_idx, nb_dist, p.eps, mzs, rts, precursor_tol_mass,
                                                    precursor_tol_mode, rt_tol, order, linkage=hier, nb_count=st["nb_count"])
            last = dict(nb_idx=nb_idx, nb_dist=nb_dist, nb_count=st["nb_count"], order=order, n_list=st["n_list"],
                        splits=st["splits"])                          # the sparse graph (exchange) + the bucket table
            index.close()
        return labels, medoids, last

    def run(self, ds: SpectrumDataset, precursor_tol_mass: float, precursor_tol_mode: str,
            rt_tol: Optional[float], fragment_tol: float, batch_size: int, p: AnnParams,
            keep_intermediates: bool = False):
        """-> (labels_dev i32[N], medoids_dev i32[n_labels]) as device tensors."""
        import torch
        c = self.ctx
        if len(ds) == 0:
            return c.empty((0,), torch.int32), c.empty((0,), torch.int32)
        self.plan(len(ds), batch_size, p)
        st = self._front(c, ds, precursor_tol_mass, precursor_tol_mode, rt_tol, batch_size, p)
        self._search(ds, st, precursor_tol_mass, precursor_tol_mode, rt_tol, fragment_tol, p, keep_intermediates)
        labels, medoids, self.last = self._graph(st, precursor_tol_mass, precursor_tol_mode, rt_tol, p, keep_intermediates)
        return labels, medoids

    def run_many(self, datasets, precursor_tol_mass: float, precursor_tol_mode: str, rt_tol: Optional[float],
                 fragment_tol: float, batch_size: int, p: AnnParams, shard: Optional[Tuple[int, int]] = None):
        """Several independent partitions, software-pipelined on one GPU: the light front end of partition i + 1
        (sort + bucket boundaries, on a second stream / context) and its host planning run while the scan of
        partition i occupies the matrix cores; all heavy kernels stay on the pipeline's own stream, in order.
        Same results as `run` per partition.  -> [(labels, medoids), ...]; `self.lasts` holds every `last`.

        `shard = (rank, world)`: every partition is ONE dataset shared by `world` GPUs; this rank runs the path on
        its own precursor buckets only (`_restrict`).  Labels / medoids then refer to the rank's rows in sorted
        order (labels[i] belongs to dataset row lasts[j]["rows"][i]; medoids index the same rows) and
        lasts[j]["rows"] (i64, device) maps them back -- the exchange step assembles the global result."""
        import torch
        c = self.ctx
        args = (precursor_tol_mass, precursor_tol_mode, rt_tol)
        live = [i for i, ds in enumerate(datasets) if len(ds) > 0]
        if live:
            self.plan(max(len(datasets[i]) for i in live), batch_size, p)
        if len(live) > 1 and not hasattr(self, "_front_ctx"):
            self._front_stream = torch.cuda.Stream(device=c.tdev)
            with torch.cuda.stream(self._front_stream):
                self._front_ctx = _device.Context(c.device)               # bound to the front stream
        # one partition: nothing to overlap -- its front end runs on the pipeline's own stream (a `PartitionRunner` slot must
        # not spread over more streams than it has to: HIP folds streams onto a few hardware queues, and a front end queued
        # behind another slot's scan kernel waits for it)
        front_stream = self._front_stream if len(live) > 1 else torch.cuda.current_stream(c.tdev)
        front_ctx = self._front_ctx if len(live) > 1 else c
        states = {}
        sharded = shard is not None and shard[1] > 1
        windows = sharded and bool(p.mz_interval and p.mz_interval > 0)
        owners = None
        if windows:
            # the deal covers ALL partitions of the job (a rank owns whole partitions where it can); a caller that runs the
            # partitions of one job through several pipelines (PartitionRunner) plans once and passes every call its part
            owners = shard[2] if len(shard) > 2 else None
            if owners is None:
                front_stream.wait_stream(torch.cuda.current_stream(c.tdev))
                with torch.cuda.stream(front_stream):
                    owners = self.plan_shards(front_ctx, datasets, batch_size, p, shard[1], tol=args[:2])

        def front(i):
            with torch.cuda.stream(front_stream):
                if windows:
                    st = self._front_windows(front_ctx, datasets[i], *args, batch_size, p, shard, owner=owners[i])
                else:
                    st = self._front(front_ctx, datasets[i], *args, batch_size, p)
                    if sharded:                                                # no windows: whole buckets of the sorted dataset
                        st = self._restrict(front_ctx, st, p, shard[:2])
                if sharded:
                    # labels refer to the subset's own rows; `rows` maps them to dataset rows
                    st["order"] = torch.arange(st["rows"].numel(), dtype=torch.int64, device=c.tdev)
                states[i] = st

        if live:
            # inputs may have been produced on the caller's stream
            front_stream.wait_stream(torch.cuda.current_stream(c.tdev))
            front(live[0])
        for pos, i in enumerate(live):
            if states[i]["order"].numel() > 0:
                if sharded and datasets[i].on_host():
                    # a rank touches the peaks of its own windows only -- of a HOST-resident partition it uploads just
                    # those: the rows come back (8 bytes each), the host gathers their CSR in sorted order into pinned memory
                    # (`take_rows`), and the path runs on that sub-dataset as it stands (row i = sorted position i)
                    front_stream.synchronize()
                    sub = datasets[i].take_rows(states[i]["rows"].cpu().numpy())
                    sub = SpectrumDataset(None, None, c.to_dev(sub.mz), c.to_dev(sub.intensity), c.to_dev(sub.indptr))
                    self._search(sub, states[i], *args, fragment_tol, p, False)      # (order = arange: set in front())
                elif sharded:
                    # the subset's rows in sorted order = dataset rows `rows`: vectorise gathers them from the CSR
                    st = dict(states[i], order=states[i]["rows"])
                    torch.cuda.current_stream(c.tdev).wait_stream(front_stream)
                    self._search(datasets[i], st, *args, fragment_tol, p, False)
                    st["order"] = states[i]["order"]
                    states[i] = st
                else:
                    self._search(datasets[i], states[i], *args, fragment_tol, p, False)   # enqueue only
            if pos + 1 < len(live):
                front(live[pos + 1])                                               # overlaps the scan just enqueued
        outs, self.lasts = [], []
        for i, ds in enumerate(datasets):
            if len(ds) == 0 or states[i]["order"].numel() == 0:
                outs.append((c.empty((0,), torch.int32), c.empty((0,), torch.int32)))
                self.lasts.append({"rows": c.empty((0,), torch.int64)} if sharded else {})
                continue
            labels, medoids, last = self._graph(states[i], *args, p, False)
            if sharded:
                last["rows"] = states[i]["rows"]
            outs.append((labels, medoids))
            self.lasts.append(last)
        if self.lasts:
            self.last = self.lasts[-1]
        return outs


    def run_chunked(self, datasets, precursor_tol_mass: float, precursor_tol_mode: str, rt_tol: Optional[float],
                    fragment_tol: float, batch_size: int, p: AnnParams, n_chunks: int, on_chunk=None):
        """Datasets whose working set exceeds one pass (BASELINE configs[3]: 50 M spectra need ~260 GB of vectors, index,
        keys and hand-off buffers at once): the precursor buckets are dealt into `n_chunks` shares exactly as they are
        dealt to GPUs (`_restrict`: LPT on `distributed.bucket_costs`; no neighbour pair crosses a bucket, reference
        cluster.py:107-141) and the shares run one after the other on this GPU, each through `run_many(shard=(c,
        n_chunks))`.  Same partition as one pass; cluster ids are share-major (the reference offsets the labels of its
        blocks the same way, cluster.py:144-155).  `on_chunk(c, outs, lasts)` sees every share's raw results.
        -> [(labels i32[N_j] by dataset row, medoids i32[n_labels_j] dataset rows)] per dataset."""
        import torch
        c = self.ctx
        if n_chunks <= 1:
            return self.run_many(datasets, precursor_tol_mass, precursor_tol_mode, rt_tol, fragment_tol, batch_size, p)
        labels = [torch.full((len(ds),), -1, dtype=torch.int32, device=c.tdev) for ds in datasets]
        medoids = [[] for _ in datasets]
        off = [0] * len(datasets)
        for ch in range(n_chunks):
            outs = self.run_many(datasets, precursor_tol_mass, precursor_tol_mode, rt_tol, fragment_tol, batch_size, p,
                                 shard=(ch, n_chunks))
            _merge_share(labels, medoids, off, outs, self.lasts)
            if on_chunk is not None:
                on_chunk(ch, outs, self.lasts)
            del outs
        return [(labels[j], torch.cat(medoids[j]) if medoids[j] else c.empty((0,), torch.int32)) for j in range(len(datasets))]


def _merge_share(labels, medoids, off, outs, lasts):
    """one bucket share's results into the job's: labels by dataset row with share-major cluster ids (the reference offsets the
    labels of its blocks the same way, cluster.py:144-155), medoids as dataset rows; `off[j]` = clusters of partition j so far"""
    import torch
    for j, ((lab, med), last) in enumerate(zip(outs, lasts)):
        rows = last.get("rows")
        if rows is None or rows.numel() == 0:
            continue
        labels[j][rows] = lab + off[j]
        medoids[j].append(rows[med.long()].to(torch.int32))
        off[j] += int(med.numel())


class PartitionRunner:
    """Runs independent partitions (precursor charges, falcon.py:151-160) concurrently: one host
    thread + one HIP stream + one `fal_ctx` per partition slot, the way the reference clusters its
    blocks on a thread pool (cluster.py:115-136).  The GPU interleaves the streams, so the host
    planning / synchronisation gaps of one partition hide under the kernels of the other."""

    def __init__(self, device: int = 0, n_slots: int = 2):
        import threading
        from concurrent.futures import ThreadPoolExecutor
        self.device, self.n_slots = device, n_slots
        self._tls = threading.local()
        self._pool = ThreadPoolExecutor(max_workers=n_slots, thread_name_prefix="falcon-part")
        self.pipelines = []
        self._lock = threading.Lock()

    def _pipeline(self):
        import torch
        if not hasattr(self._tls, "pipe"):
            torch.cuda.set_device(self.device)
            self._tls.stream = torch.cuda.Stream(device=self.device)
            with torch.cuda.stream(self._tls.stream):
                self._tls.pipe = ClusterPipeline(device=self.device)      # binds the ctx to this stream
            with self._lock:
                self.pipelines.append(self._tls.pipe)
        return self._tls.pipe, self._tls.stream

    def _run_one(self, ds, args, kwargs, shard, arrived=None):
        import torch
        pipe, stream = self._pipeline()
        with torch.cuda.stream(stream):
            if arrived is not None:
                stream.wait_event(arrived)                     # the partition's bytes (uploaded by `run` on the copy stream)
            if shard is not None and shard[1] > 1:
                out = pipe.run_many([ds], *args, shard=shard, **kwargs)[0]       # this rank's windows of the partition
                pipe.last = pipe.lasts[0]
            else:
                out = pipe.run(ds, *args, **kwargs)
            stream.synchronize()
        return out, pipe, dict(pipe.last)        # (a snapshot: the same slot may serve another partition next)

    def run(self, datasets, *args, shard: Optional[Tuple[int, int]] = None, **kwargs):
        """-> [(labels, medoids), ...] in the order of `datasets` (largest partition is started first).  `shard = (rank,
        world)`: every partition is one dataset shared by `world` GPUs, as in `ClusterPipeline.run_many`."""
        return self.collect(self.submit(datasets, *args, shard=shard, **kwargs))

    def collect(self, handle):
        """the results of a `submit`: waits for its partitions (each slot has synchronised its own stream)"""
        futs, n, nothing = handle
        res = [futs[i].result()[::2] if i in futs else nothing() for i in range(n)]
        self.lasts = [r[1] for r in res]                           # every partition's `last`, in the order of `datasets`
        return [r[0] for r in res]

    def submit(self, datasets, *args, shard: Optional[Tuple[int, int]] = None, inputs_ready: bool = False, **kwargs):
        """`run` without the wait: the partitions are queued on the slots (behind whatever the slots are still working on -- a
        stream of jobs keeps the GPU busy across job boundaries) and a handle for `collect` comes back.  `inputs_ready`: the
        datasets were complete on the device before the call (no wait for the caller's stream)."""
        import torch
        if not inputs_ready:
            torch.cuda.current_stream(self.device).synchronize()      # inputs produced on the caller's stream
        order = sorted(range(len(datasets)), key=lambda i: -len(datasets[i]))
        shards = [shard] * len(datasets)
        p = args[5]
        if shard is not None and shard[1] > 1 and p.mz_interval and p.mz_interval > 0:
            # one deal for the whole job (`distributed.deal_job` over every partition's windows), then a slot per partition
            if not hasattr(self, "_planner"):
                self._planner = ClusterPipeline(device=self.device)
            owners = self._planner.plan_shards(self._planner.ctx, datasets, args[4], p, shard[1], tol=args[:2])      # (waits for the counts)
            shards = [(shard[0], shard[1], [owners[i]]) for i in range(len(datasets))]
            order = [i for i in order if (owners[i] == shard[0]).any()]        # partitions this rank has windows of
        dev = torch.device("cuda", self.device)
        # Host-resident partitions (the reference hands numpy columns over: cluster.py:73-85) cross PCIe on ONE copy stream,
        # largest partition first, each followed by an event: a partition's kernels start when ITS bytes have arrived, the
        # next partition's upload travels under them (pinned memory makes the copies asynchronous).
        arrived = {}
        sharded = shard is not None and shard[1] > 1
        if not sharded and any(datasets[i].on_host() for i in order):     # (a sharded rank uploads its own windows only: run_many)
            if not hasattr(self, "_copy_stream"):
                self._copy_stream = torch.cuda.Stream(device=self.device)
            datasets = list(datasets)
            with torch.cuda.stream(self._copy_stream):
                for i in order:
                    if datasets[i].on_host():
                        datasets[i] = datasets[i].to_device(dev)
                        arrived[i] = torch.cuda.Event()
                        arrived[i].record(self._copy_stream)
        futs = {i: self._pool.submit(self._run_one, datasets[i], args, kwargs, shards[i], arrived.get(i)) for i in order}
        nothing = lambda: ((torch.empty(0, dtype=torch.int32, device=dev), torch.empty(0, dtype=torch.int32, device=dev)),
                           {"rows": torch.empty(0, dtype=torch.int64, device=dev)})
        return futs, len(datasets), nothing

    def plan(self, datasets, *args, **kwargs):
        """Before the first pass of a fresh process: create every slot's thread, stream and context and `fal_ctx_plan` each for the
        job's largest partition (code objects loaded once per process, scratch sized).  Optional -- `run` does the same lazily."""
        import threading
        n = max((len(ds) for ds in datasets), default=0)
        batch_size, p = args[4], args[5]
        gate = threading.Barrier(self.n_slots)

        def one():
            pipe, stream = self._pipeline()
            import torch
            with torch.cuda.stream(stream):
                pipe.plan(n, batch_size, p)
            gate.wait()                                  # (every worker thread takes exactly one of the n_slots tasks)
        for f in [self._pool.submit(one) for _ in range(self.n_slots)]:
            f.result()

    def trim(self):
        """every slot's `ClusterPipeline.trim` (the slots are idle between two `run` calls)"""
        with self._lock:
            pipes = list(self.pipelines)
        for pipe in pipes + ([self._planner] if hasattr(self, "_planner") else []):
            pipe.trim()

    def run_chunked(self, datasets, *args, n_chunks: int, on_chunk=None, **kwargs):
        """`ClusterPipeline.run_chunked` with the partitions of every bucket share on concurrent slots: the precursor windows of the
        job are dealt into `n_chunks` shares exactly as they are dealt to GPUs, the shares run one after the other, each
        through `run(shard=(c, n_chunks))`.  Same partition as one pass; cluster ids are share-major.
        -> [(labels i32[N_j] by dataset row, medoids i32[n_labels_j] dataset rows)] per dataset."""
        import torch
        if n_chunks <= 1:
            return self.run(datasets, *args, **kwargs)
        dev = torch.device("cuda", self.device)
        labels = [torch.full((len(ds),), -1, dtype=torch.int32, device=dev) for ds in datasets]
        medoids = [[] for _ in datasets]
        off = [0] * len(datasets)
        for ch in range(n_chunks):
            outs = self.run(datasets, *args, shard=(ch, n_chunks), **kwargs)
            _merge_share(labels, medoids, off, outs, self.lasts)
            if on_chunk is not None:
                on_chunk(ch, outs, self.lasts)
            del outs
        return [(labels[j], torch.cat(medoids[j]) if medoids[j] else torch.empty(0, dtype=torch.int32, device=dev))
                for j in range(len(datasets))]

    def close(self):
        self._pool.shutdown(wait=True)


_default_pipeline: Optional[ClusterPipeline] = None


def generate_clusters(dataset, linkage: str, distance_threshold: float, min_matches: int,
                      precursor_tol_mass: float, precursor_tol_mode: str, rt_tol: Optional[float],
                      fragment_tol: float, batch_size: int, *, ann: Optional[AnnParams] = None,
                      pipeline: Optional[ClusterPipeline] = None) -> Tuple[np.ndarray, np.ndarray]:
    """Same call as the reference (cluster.py:24-34).  `distance_threshold` is the cosine
    distance threshold and plays the role of DBSCAN's eps (README.md:73-79); `linkage` and
    `min_matches` belong to the snapshot's exact-cosine path: `min_matches` is used when `ann.rescore` is on
    (matched-peak re-scoring of the ANN neighbours, similarity.py:17-80); `linkage` selects the hierarchical
    clustering of those exact distances when `ann.clustering == "hierarchical"` and is an error otherwise unless
    it is the reference default "complete".

    Returns (labels int32[N] by dataset row with noise renumbered as singletons --
    cluster.py:144-155 --, medoids int32[n_labels]: medoids[c] = dataset row representing
    cluster c, usable as `dataset.take(medoids)`, falcon.py:198-203)."""
    global _default_pipeline
    if not isinstance(dataset, SpectrumDataset):
        dataset = SpectrumDataset.from_table(
            dataset.to_table(columns=["precursor_mz", "precursor_charge", "retention_time", "mz", "intensity"])
            if hasattr(dataset, "to_table") else dataset)
    import dataclasses
    if linkage not in ("complete", "single", "average"):
        raise ValueError(f"unknown linkage {linkage!r} (choose complete, single or average; cluster.py:283-290)")
    if ann is None:
        p = AnnParams(eps=float(distance_threshold))
    else:
        if abs(float(ann.eps) - float(distance_threshold)) > 1e-12:
            raise ValueError(f"distance_threshold ({distance_threshold}) and ann.eps ({ann.eps}) differ: they name the same "
                             "cosine-distance threshold (README.md:73-79, config.py:104-111)")
        p = dataclasses.replace(ann)                 # never mutate the caller's parameters
    if p.rescore:
        p.min_matches = int(min_matches)
    if p.clustering not in ("dbscan", "hierarchical"):
        raise ValueError(f"unknown clustering {p.clustering!r}")
    if p.clustering == "hierarchical":
        p.linkage = linkage
        p.rescore = True                 # the linkage runs on the exact matched-peak distances (cluster.py:283-290)
        p.min_matches = int(min_matches)
    elif linkage != "complete":
        raise ValueError(f"linkage={linkage!r} only applies to the hierarchical clustering of the exact distances "
                         "(AnnParams(clustering=\"hierarchical\", rescore=True) / --clustering hierarchical); the default "
                         "nearest-neighbour path clusters with DBSCAN and would ignore it")
    pipe = pipeline or _default_pipeline
    if pipe is None:
        pipe = _default_pipeline = ClusterPipeline()
    logger.info("Cluster %d spectra: low_dim=%d n_probe=%d n_neighbors=%d/%d eps=%.3f", len(dataset), p.low_dim,
                p.n_probe, p.n_neighbors, p.n_neighbors_ann, p.eps)
    labels, medoids = pipe.run(dataset, precursor_tol_mass, precursor_tol_mode, rt_tol, fragment_tol, batch_size, p)
    return labels.cpu().numpy(), medoids.cpu().numpy()
