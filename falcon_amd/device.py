"""Thin host wrapper over the C ABI: one `Context` per GPU.

PyTorch is used for plumbing only -- device memory (`torch.empty(..., device=cuda)`),
the stream the library enqueues on, and `torch.distributed` for the multi-GPU
exchange.  Every kernel that touches the data is in libfalcon_hip.so.
"""
from __future__ import annotations

import ctypes as C
from typing import Optional, Sequence

import numpy as np

from . import _lib
from ._lib import FalconHipError, check


def _torch():
    import torch
    return torch


class Context:
    """Owns a `fal_ctx` bound to `cuda:<device>` and torch's current stream on it."""

    def __init__(self, device: int = 0):
        torch = _torch()
        self.lib = _lib.load()
        if not torch.cuda.is_available():
            raise FalconHipError("no HIP device visible to torch; falcon_amd has no CPU fallback")
        self.device = int(device)
        self.tdev = torch.device("cuda", self.device)
        torch.cuda.set_device(self.device)
        stream = torch.cuda.current_stream(self.device).cuda_stream
        h = C.c_void_p()
        check(self.lib.fal_ctx_create(self.device, C.c_void_p(stream), 0, C.byref(h)), "fal_ctx_create")
        self._h = h

    def close(self):
        if getattr(self, "_h", None):
            self.lib.fal_ctx_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # ------------------------------------------------------------------ helpers
    h2d_bytes = 0            # bytes this context has copied host -> device through `to_dev` (the multi-GPU tests count them)

    def to_dev(self, a, dtype=None):
        torch = _torch()
        if isinstance(a, torch.Tensor):
            if not a.is_cuda:
                self.h2d_bytes += a.numel() * a.element_size()
            t = a.to(self.tdev, non_blocking=True)
            return t.to(dtype).contiguous() if dtype is not None else t.contiguous()
        t = torch.from_numpy(np.ascontiguousarray(a))
        if dtype is not None:
            t = t.to(dtype)
        self.h2d_bytes += t.numel() * t.element_size()
        return t.to(self.tdev)

    def empty(self, shape, dtype):
        return _torch().empty(shape, dtype=dtype, device=self.tdev)

    @staticmethod
    def _p(t):
        return C.c_void_p(0) if t is None else C.c_void_p(t.data_ptr())

    def sync(self):
        check(self.lib.fal_ctx_sync(self._h), "fal_ctx_sync")

    def plan(self, n: int, low_dim: int = 400, k_ann: int = 128, n_probe: int = 16, batch_size: int = 2 ** 15):
        """`fal_ctx_plan`: load every kernel's code object and size the shape-dependent scratch before the first pass"""
        check(self.lib.fal_ctx_plan(self._h, int(n), int(low_dim), int(k_ann), int(n_probe), int(batch_size)), "fal_ctx_plan")

    def trim(self):
        """`fal_ctx_trim`: give the context's cached device memory back to the driver (drains the stream)"""
        check(self.lib.fal_ctx_trim(self._h), "fal_ctx_trim")

    def enable_timing(self, on: bool = True):
        check(self.lib.fal_ctx_enable_timing(self._h, int(on)))

    def stage_ms(self, stage: str):
        ms, k = C.c_float(), C.c_int64()
        check(self.lib.fal_ctx_stage_ms(self._h, _lib.STAGES[stage], C.byref(ms), C.byref(k)))
        return ms.value, k.value

    def counter(self, which: int) -> int:
        v = C.c_int64()
        check(self.lib.fal_ctx_counter(self._h, int(which), C.byref(v)))
        return int(v.value)

    # ------------------------------------------------------------------ a2 / a3
    def to_vector_indices(self, mz, min_mz: float, bin_size: float):
        torch = _torch()
        mz = self.to_dev(mz, torch.float32)
        out = self.empty((mz.numel(),), torch.int32)
        check(self.lib.fal_to_vector_indices(self._h, self._p(mz), mz.numel(), min_mz, bin_size, self._p(out)),
              "fal_to_vector_indices")
        return out

    def vectorize(self, mz, intensity, indptr, row_order, min_mz: float, bin_size: float, n_bins: int,
                  low_dim: int, seed: int = 0, normalize: bool = True, dtype: str = "f32", width: Optional[int] = None):
        """a2 + a3 (`fal_vectorize_rows`).  `low_dim` = the hash modulus (any integer >= 1, README.md:114-117); `width` =
        columns of an output row (default: low_dim, which then has to be a multiple of 8; `row_width(low_dim)` = the width
        the path uses): the columns behind low_dim are zero and change no similarity."""
        torch = _torch()
        mz = self.to_dev(mz, torch.float32)
        intensity = self.to_dev(intensity, torch.float32)
        indptr = self.to_dev(indptr, torch.int64)
        row_order = None if row_order is None else self.to_dev(row_order, torch.int64)
        # output row i is spectrum row_order[i] of the CSR: a subset / permutation of the dataset is fine
        n = indptr.numel() - 1 if row_order is None else row_order.numel()
        w = int(low_dim if width is None else width)
        out2 = None
        if dtype in ("f32+f16", "f16+image"):
            # one pass over the peaks, two outputs: float32 rows and their float16 rounding ("f16+image": float16 VECTORS --
            # the float32 output is the image of the rounded values)
            out, out2 = self.empty((n, w), torch.float32), self.empty((n, w), torch.float16)
            code = _lib.FAL_OUT_F32_F16 if dtype == "f32+f16" else _lib.FAL_OUT_F16_IMAGE
        elif dtype == "split16":
            out, code = self.empty((n, 2, w), torch.float16), _lib.FAL_DTYPE_SPLIT16
        elif dtype in ("f16", "float16"):
            out, code = self.empty((n, w), torch.float16), _lib.FAL_DTYPE_F16
        else:
            out, code = self.empty((n, w), torch.float32), _lib.FAL_DTYPE_F32
        check(self.lib.fal_vectorize_rows(self._h, self._p(mz), self._p(intensity), self._p(indptr), self._p(row_order),
                                          n, float(min_mz), float(bin_size), int(n_bins), int(low_dim), w, int(seed),
                                          int(normalize), code, self._p(out), self._p(out2)), "fal_vectorize_rows")
        return out if out2 is None else (out, out2)


    # ------------------------------------------------------------------ a6 / a7
    def ivf_build(self, X, bucket_off: np.ndarray, n_list: np.ndarray, kmeans_iters: int = 10,
                  X16=None, Xpre=None, Xkm=None, prefilter_which: int = 1) -> "IvfIndex":
        """X: float32 [n, d] (may be None when every bucket is flat and X16 is given);
        X16: optional float16 [n, d] (plain rows) or [n, 2, d] (hi/lo split) for the f16 flat scan;
        Xpre: optional float16 [n, d] copy of X used only as the prefilter of `search_neighbors` (exact results);
        prefilter_which: where Xpre is used: 1 = flat buckets (fused.hip), 2 = buckets with an index (ivf16.hip), 3 = both;
        Xkm: optional float16 [n, d] copy of X used as the prefilter of the k-means assignment (identical index)."""
        torch = _torch()
        if X is not None:
            assert X.dtype == torch.float32 and X.is_contiguous() and X.device == self.tdev
            n, d = X.shape
        else:
            assert X16 is not None
            n, d = X16.shape[0], X16.shape[-1]
        bo = np.ascontiguousarray(bucket_off, np.int64)
        nl = np.ascontiguousarray(n_list, np.int32)
        h = C.c_void_p()
        if Xkm is not None:
            assert Xkm.dtype == torch.float16 and Xkm.is_contiguous() and Xkm.device == self.tdev and tuple(Xkm.shape) == (n, d)
        check(self.lib.fal_ivf_build_x16(self._h, self._p(X), self._p(Xkm), n, d, bo.ctypes.data_as(C.c_void_p), len(nl),
                                         nl.ctypes.data_as(C.c_void_p), int(kmeans_iters), C.byref(h)), "fal_ivf_build")
        index = IvfIndex(self, h, X, bo, nl, n, d)
        index.Xkm = Xkm
        if X16 is not None:
            assert X16.dtype == torch.float16 and X16.is_contiguous() and X16.device == self.tdev
            planes = 2 if X16.dim() == 3 else 1
            check(self.lib.fal_ivf_attach_f16(h, self._p(X16), planes), "fal_ivf_attach_f16")
            index.X16 = X16
        if Xpre is not None:
            assert Xpre.dtype == torch.float16 and Xpre.is_contiguous() and Xpre.device == self.tdev and Xpre.shape == (n, d)
            check(self.lib.fal_ivf_attach_prefilter_ex(h, self._p(Xpre), int(prefilter_which)), "fal_ivf_attach_prefilter_ex")
            index.Xpre = Xpre
        return index


    # ------------------------------------------------------------------ sort / a5
    def sort_by_precursor(self, precursor_mz):
        """stable sort (reference cluster.py:73-85) -> order i64[n], mz_sorted f32[n]"""
        torch = _torch()
        pmz = self.to_dev(precursor_mz, torch.float32)
        n = pmz.numel()
        order = self.empty((n,), torch.int64)
        mzs = self.empty((n,), torch.float32)
        check(self.lib.fal_sort_by_precursor(self._h, self._p(pmz), n, self._p(order), self._p(mzs)),
              "fal_sort_by_precursor")
        return order, mzs

    def gather_f32(self, src, order):
        torch = _torch()
        src = self.to_dev(src, torch.float32)
        out = self.empty((order.numel(),), torch.float32)
        check(self.lib.fal_gather_f32(self._h, self._p(src), self._p(order), order.numel(), self._p(out)),
              "fal_gather_f32")
        return out

    N_WINDOWS = 1 << 14           # deal units the multi-GPU front end counts: window w = floor(mz / mz_interval) -> slot w mod 16,384

    def window_counts(self, precursor_mzs, mz_interval: float):
        """spectra per precursor window floor(mz / mz_interval) of every partition of a job (`fal_window_counts`: one pair of
        launches, one copy to pinned host memory, one wait) -> int64 counts [n_parts, windows] (host), the all-zero trailing
        windows cut"""
        torch = _torch()
        pmz = [self.to_dev(x, torch.float32) for x in precursor_mzs]
        n_parts = len(pmz)
        if n_parts == 0:
            return np.zeros((0, 0), np.int64)
        counts = self.empty((n_parts, self.N_WINDOWS), torch.int32)
        if getattr(self, "_wc_host", None) is None or self._wc_host.numel() < n_parts * (self.N_WINDOWS + 2):
            self._wc_host = torch.empty(n_parts * (self.N_WINDOWS + 2), dtype=torch.int32, pin_memory=True)
        host = self._wc_host
        ptrs = (C.c_void_p * n_parts)(*[x.data_ptr() if x.numel() else None for x in pmz])
        ns = (C.c_int64 * n_parts)(*[x.numel() for x in pmz])
        check(self.lib.fal_window_counts(self._h, ptrs, ns, n_parts, float(mz_interval), self.N_WINDOWS, self._p(counts),
                                         C.c_void_p(host.data_ptr())), "fal_window_counts")
        self.sync()
        h = host.numpy()
        last = h[n_parts * self.N_WINDOWS: n_parts * (self.N_WINDOWS + 2)].reshape(n_parts, 2)
        last = last[last[:, 0] <= last[:, 1], 1]                                  # (empty partitions: INT32_MAX, 0)
        width = int(last.max()) + 1 if len(last) else 0
        return h[: n_parts * self.N_WINDOWS].reshape(n_parts, self.N_WINDOWS)[:, :width].astype(np.int64)

    def window_select(self, precursor_mz, mz_interval: float, owner: np.ndarray, rank: int):
        """the spectra whose window is dealt to `rank` (`fal_window_select`) -> rows i64[m] (ascending dataset rows), mz f32[m]"""
        torch = _torch()
        pmz = self.to_dev(precursor_mz, torch.float32)
        n = pmz.numel()
        own = np.full(self.N_WINDOWS, -1, np.int32)
        own[:len(owner)] = owner
        if len(owner):
            own[len(owner):] = owner[-1]                  # (slots behind the counted ones hold no spectrum)
        own_d = self.to_dev(own, torch.int32)
        rows = self.empty((max(n, 1),), torch.int64)
        mzs = self.empty((max(n, 1),), torch.float32)
        m = C.c_int64()
        check(self.lib.fal_window_select(self._h, self._p(pmz), n, float(mz_interval), self.N_WINDOWS, self._p(own_d), int(rank),
                                         self._p(rows), self._p(mzs), C.byref(m)), "fal_window_select")
        return rows[: m.value], mzs[: m.value]

    def precursor_splits(self, mz_sorted, tol: float, mode: str, batch_size: int, mz_interval: float = 1.0,
                         chunk_last: bool = True) -> np.ndarray:
        """reference cluster.py:159-209 (+ the build's two extra rules) -> int64 boundaries (host)"""
        torch = _torch()
        mzs = self.to_dev(mz_sorted, torch.float32)
        n = mzs.numel()
        cap = n + 2
        out = np.empty(cap, np.int64)
        k = C.c_int64()
        check(self.lib.fal_precursor_splits(self._h, self._p(mzs), n, float(tol), int(mode == "Da"), int(batch_size),
                                            float(mz_interval or 0.0), int(chunk_last),
                                            out.ctypes.data_as(C.c_void_p), cap, C.byref(k)), "fal_precursor_splits")
        return out[:k.value].copy()

    # ------------------------------------------------------------------ a8 .. a12
    def filter_neighbors(self, sim, idx, mz_sorted, rt_sorted, tol: float, mode: str, rt_tol, n_neighbors: int):
        torch = _torch()
        n, k_ann = idx.shape
        nb_idx = self.empty((n, n_neighbors), torch.int32)
        nb_dist = self.empty((n, n_neighbors), torch.float32)
        check(self.lib.fal_filter_neighbors(self._h, self._p(sim), self._p(idx), n, k_ann, self._p(mz_sorted),
                                            self._p(rt_sorted), float(tol), int(mode == "Da"),
                                            -1.0 if rt_tol is None else float(rt_tol), int(n_neighbors),
                                            self._p(nb_idx), self._p(nb_dist)), "fal_filter_neighbors")
        return nb_idx, nb_dist

    SCALING = {None: 0, "off": 0, "root": 1, "log": 2, "rank": 3}

    def process_spectra(self, mz, intensity, indptr, precursor_mz, precursor_charge, min_peaks: int,
                        min_mz_range: float, mz_min=None, mz_max=None, remove_precursor_tolerance=None,
                        min_intensity=None, max_peaks_used=None, scaling=None):
        """f1 (`fal_process_spectra`): batch `process_spectrum` (reference spectrum.py:73-169) over raw CSR peaks
        (mz float64 sorted per spectrum, intensity float32, charge 0 = unknown).
        -> valid bool[n], out_indptr i64[n+1], out_mz f32[nnz_out], out_intensity f32[nnz_out] (device tensors)."""
        torch = _torch()
        mz = self.to_dev(mz, torch.float64)
        intensity = self.to_dev(intensity, torch.float32)
        indptr = self.to_dev(indptr, torch.int64)
        pmz = self.to_dev(precursor_mz, torch.float64)
        charge = self.to_dev(precursor_charge, torch.int32)
        n, nnz = indptr.numel() - 1, mz.numel()
        if scaling not in self.SCALING:
            raise ValueError(f"unknown scaling {scaling!r}")
        valid = self.empty((n,), torch.int32)
        out_indptr = self.empty((n + 1,), torch.int64)
        out_mz = self.empty((max(nnz, 1),), torch.float32)
        out_it = self.empty((max(nnz, 1),), torch.float32)
        nan = float("nan")
        check(self.lib.fal_process_spectra(
            self._h, self._p(mz), self._p(intensity), self._p(indptr), n, nnz, self._p(pmz), self._p(charge),
            int(min_peaks), float(min_mz_range), nan if mz_min is None else float(mz_min),
            nan if mz_max is None else float(mz_max),
            -1.0 if remove_precursor_tolerance is None else float(remove_precursor_tolerance),
            -1.0 if min_intensity is None else float(min_intensity), 0 if max_peaks_used is None else int(max_peaks_used),
            self.SCALING[scaling], self._p(valid), self._p(out_indptr), self._p(out_mz), self._p(out_it)),
            "fal_process_spectra")
        nnz_out = int(out_indptr[-1].item())
        return valid.bool(), out_indptr, out_mz[:nnz_out], out_it[:nnz_out]

    def rescore_neighbors(self, nb_idx, nb_dist, mz, intensity, indptr, order, fragment_tol: float, min_matches: int):
        """f4 (`fal_rescore_neighbors`): nb_dist <- 1 - matched-peak cosine (reference similarity.py:17-80), in place."""
        torch = _torch()
        n, k = nb_idx.shape
        mz = self.to_dev(mz, torch.float32)
        intensity = self.to_dev(intensity, torch.float32)
        indptr = self.to_dev(indptr, torch.int64)
        order = self.to_dev(order, torch.int64)
        check(self.lib.fal_rescore_neighbors(self._h, self._p(nb_idx), self._p(nb_dist), n, k, self._p(mz), self._p(intensity),
                                             self._p(indptr), self._p(order), float(fragment_tol), int(min_matches)),
              "fal_rescore_neighbors")
        return nb_dist

    def neighbors_to_csr(self, nb_idx, nb_dist, id_offset: int = 0, out=None, row0: int = 0, nb_count=None, id_map=None):
        """ELL neighbour lists -> CSR (indptr i64[rows+1], idx i32[cap], dist f32[cap]); entries beyond
        indptr[-1] are unspecified.  `out` = (indptr, idx, dist) buffers to fill; with `row0` > 0 the call
        appends a further segment (rows row0.. of `out`, ids shifted by its own id_offset).  `id_map` (i64):
        stored id -> id_map[id] + id_offset (a bucket shard's positions -> dataset rows).  No sync."""
        torch = _torch()
        n, k = nb_idx.shape
        if out is None:
            out = (self.empty((row0 + n + 1,), torch.int64), self.empty((max((row0 + n) * k, 1),), torch.int32),
                   self.empty((max((row0 + n) * k, 1),), torch.float32))
        indptr, idx, dist = out
        if indptr.numel() < row0 + n + 1:
            raise FalconHipError("neighbors_to_csr: indptr buffer too small")
        check(self.lib.fal_neighbors_to_csr_mapped(self._h, self._p(nb_idx), self._p(nb_dist), self._p(nb_count), n, k,
                                                   self._p(id_map), int(id_offset), int(row0),
                                                   self._p(indptr), self._p(idx), self._p(dist)), "fal_neighbors_to_csr")
        return indptr, idx, dist

    def dbscan(self, nb_idx, nb_dist, eps: float):
        torch = _torch()
        n, k = nb_idx.shape
        labels = self.empty((n,), torch.int32)
        nc = C.c_int64()
        check(self.lib.fal_dbscan(self._h, self._p(nb_idx), self._p(nb_dist), n, k, float(eps), self._p(labels),
                                  C.byref(nc)), "fal_dbscan")
        return labels, int(nc.value)

    def refine_clusters(self, labels, n_clusters: int, mz_sorted, rt_sorted, tol: float, mode: str, rt_tol):
        nc = C.c_int64(int(n_clusters))
        check(self.lib.fal_refine_clusters(self._h, self._p(labels), labels.numel(), self._p(mz_sorted),
                                           self._p(rt_sorted), float(tol), int(mode == "Da"),
                                           -1.0 if rt_tol is None else float(rt_tol), C.byref(nc)),
              "fal_refine_clusters")
        return labels, int(nc.value)

    def finalize(self, labels_sorted, n_clusters: int, order, nb_idx, nb_dist):
        torch = _torch()
        n, k = nb_idx.shape
        labels = self.empty((n,), torch.int32)
        medoids = self.empty((n,), torch.int32)
        nl = C.c_int64()
        check(self.lib.fal_finalize(self._h, self._p(labels_sorted), n, int(n_clusters), self._p(order),
                                    self._p(nb_idx), self._p(nb_dist), k, self._p(labels), self._p(medoids),
                                    C.byref(nl)), "fal_finalize")
        return labels, medoids[:int(nl.value)]


    LINKAGE = {"single": 0, "complete": 1, "average": 2}

    def cluster_graph(self, nb_idx, nb_dist, eps: float, mz_sorted, rt_sorted, tol: float, mode: str, rt_tol, order,
                      linkage: Optional[str] = None, nb_count=None):
        """a9..a12 fused: -> labels i32[n] (dataset rows), medoids i32[n_labels], labels_sorted, n_clusters.
        `linkage` = None: DBSCAN(eps); "single" / "complete" / "average": hierarchical clustering cut at `eps` (f4).
        `nb_count` (the search's per-row neighbour counts, rows front-packed): the graph passes read the stored slots only."""
        torch = _torch()
        n, k = nb_idx.shape
        lab_sorted = self.empty((n,), torch.int32)
        labels = self.empty((n,), torch.int32)
        medoids = self.empty((n,), torch.int32)
        nc, nl = C.c_int64(), C.c_int64()
        tail = (self._p(mz_sorted), self._p(rt_sorted), float(tol), int(mode == "Da"),
                -1.0 if rt_tol is None else float(rt_tol), self._p(order), self._p(lab_sorted), self._p(labels),
                self._p(medoids), C.byref(nc), C.byref(nl))
        if linkage is None and nb_count is not None:
            check(self.lib.fal_cluster_graph_counted(self._h, self._p(nb_idx), self._p(nb_dist), self._p(nb_count), n, k,
                                                     float(eps), *tail), "fal_cluster_graph_counted")
        elif linkage is None:
            check(self.lib.fal_cluster_graph(self._h, self._p(nb_idx), self._p(nb_dist), n, k, float(eps), *tail),
                  "fal_cluster_graph")
        else:
            check(self.lib.fal_cluster_graph_linkage(self._h, self._p(nb_idx), self._p(nb_dist), n, k, float(eps),
                                                     self.LINKAGE[linkage], *tail), "fal_cluster_graph_linkage")
        return labels, medoids[:int(nl.value)], lab_sorted, int(nc.value)

    def linkage_cluster(self, nb_idx, nb_dist, threshold: float, linkage: str):
        """f4 staged (`fal_linkage_cluster`): -> labels i32[n] (clusters by lowest row, -1 = groups of one), n_clusters"""
        torch = _torch()
        n, k = nb_idx.shape
        labels = self.empty((n,), torch.int32)
        nc = C.c_int64()
        check(self.lib.fal_linkage_cluster(self._h, self._p(nb_idx), self._p(nb_dist), n, k, float(threshold),
                                           self.LINKAGE[linkage], self._p(labels), C.byref(nc)), "fal_linkage_cluster")
        return labels, int(nc.value)


class IvfIndex:
    """Opaque `fal_ivf` handle (keeps the vectors alive: the index borrows them)."""

    def __init__(self, ctx: Context, handle, X, bucket_off, n_list, n, d):
        self.ctx, self._h, self.X, self.X16, self.Xpre = ctx, handle, X, None, None
        self.bucket_off, self.n_list = bucket_off, n_list
        self.n, self.d = n, d
        t = C.c_int64()
        check(ctx.lib.fal_ivf_total_lists(self._h, C.byref(t)))
        self.total_lists = int(t.value)

    def close(self):
        if getattr(self, "_h", None):
            if getattr(self.ctx, "_h", None):            # (the context owns the pool the index's arrays return to)
                self.ctx.lib.fal_ivf_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def export(self):
        """-> centroids [total_lists, d], assign [n], perm [n], list_off [total_lists+1] (device)."""
        torch = _torch()
        c = self.ctx
        cent = c.empty((self.total_lists, self.d), torch.float32)
        asg = c.empty((self.n,), torch.int32)
        perm = c.empty((self.n,), torch.int32)
        off = c.empty((self.total_lists + 1,), torch.int64)
        check(c.lib.fal_ivf_export(c._h, self._h, c._p(cent), c._p(asg), c._p(perm), c._p(off)), "fal_ivf_export")
        return cent, asg, perm, off

    def search(self, n_probe: int, k_ann: int):
        """-> sim f32[n, k_ann] (pad -inf), idx i32[n, k_ann] (pad -1), rows = sorted rows."""
        torch = _torch()
        c = self.ctx
        sim = c.empty((self.n, k_ann), torch.float32)
        idx = c.empty((self.n, k_ann), torch.int32)
        check(c.lib.fal_ivf_search_topk(c._h, self._h, int(n_probe), int(k_ann), c._p(sim), c._p(idx)),
              "fal_ivf_search_topk")
        return sim, idx

    def search_neighbors(self, n_probe: int, k_ann: int, mz_sorted, rt_sorted, tol: float, mode: str, rt_tol,
                         n_neighbors: int):
        """a7 + a8 fused (`fal_ivf_search_neighbors`): -> nb_idx i32[n, n_neighbors] (pad -1), nb_dist f32 (pad +inf);
        identical to `search` followed by `Context.filter_neighbors`."""
        torch = _torch()
        c = self.ctx
        nb_idx = c.empty((self.n, n_neighbors), torch.int32)
        nb_dist = c.empty((self.n, n_neighbors), torch.float32)
        self.nb_count = c.empty((self.n,), torch.int32)         # stored neighbours per row (rows are front-packed)
        check(c.lib.fal_ivf_search_neighbors(c._h, self._h, int(n_probe), int(k_ann), c._p(mz_sorted), c._p(rt_sorted),
                                             float(tol), int(mode == "Da"), -1.0 if rt_tol is None else float(rt_tol),
                                             int(n_neighbors), c._p(nb_idx), c._p(nb_dist), c._p(self.nb_count)),
              "fal_ivf_search_neighbors")
        return nb_idx, nb_dist


def get_dim(min_mz: float, max_mz: float, bin_size: float):
    """Reference spectrum.py:172-199 (float32 arithmetic), host side of the C ABI."""
    lib = _lib.load()
    dim, s, e = C.c_uint32(), C.c_float(), C.c_float()
    check(lib.fal_get_dim(min_mz, max_mz, bin_size, C.byref(dim), C.byref(s), C.byref(e)), "fal_get_dim")
    return int(dim.value), float(s.value), float(e.value)


def row_width(low_dim: int) -> int:
    """`fal_row_width`: columns of the rows the path stores `low_dim`-dimensional vectors in (64 / 128 / 256 / 400 / 800: the
    widths the cosine kernels are instantiated for; zero columns behind low_dim).  Raises beyond 800."""
    lib = _lib.load()
    w = C.c_uint32()
    if int(low_dim) < 1:
        raise FalconHipError(f"low_dim must be a positive integer (got {low_dim})")
    check(lib.fal_row_width(int(low_dim), C.byref(w)), "fal_row_width")
    return int(w.value)


def hash_lookup(n_bins: int, low_dim: int, seed: int = 0) -> np.ndarray:
    lib = _lib.load()
    out = np.empty(n_bins, np.uint32)
    check(lib.fal_hash_lookup(n_bins, low_dim, seed, out.ctypes.data_as(C.c_void_p)), "fal_hash_lookup")
    return out
