"""SURVEY 8d's second CPU baseline: the same path assembled from LIBRARY routines -- numpy (BLAS sgemm) for the cosine
similarities of a bucket, numpy argpartition for the top-k, scikit-learn's DBSCAN on the precomputed sparse distance graph.

TEST / BENCH INFRASTRUCTURE ONLY (like the rest of oracle/): imported by bench.py's `cpu_baseline` leg, never by the
product.  It is not a parity reference: BLAS sums in its own order (similarities within ~1e-6 of the kernels' chain),
every bucket is searched exhaustively (a superset of what n_probe lists find), and sklearn's DBSCAN visits in index
order.  It answers "what does a numpy + sklearn user get from these host cores" next to the port's figure."""
import time

import numpy as np

from . import falcon_oracle as fo


def cluster_partition(mz, intensity, indptr, precursor_mz, rt, *, eps=0.1, precursor_tol=(20.0, "ppm"), batch_size=2 ** 15,
                      low_dim=400, n_neighbors=64, n_neighbors_ann=128, min_mz=101.0, max_mz=1500.0, fragment_tol=0.05,
                      mz_interval=1.0):
    """one charge partition -> (labels int32[N] in dataset order with noise as singletons, seconds per phase)"""
    from scipy.sparse import csr_matrix
    from sklearn.cluster import DBSCAN
    t = {}
    t0 = time.perf_counter()
    tol, mode = float(precursor_tol[0]), precursor_tol[1]
    pmz = np.asarray(precursor_mz, np.float32)
    order = np.argsort(pmz, kind="stable")
    mzs = pmz[order]
    n_bins, start, _ = fo.get_dim(min_mz, max_mz, fragment_tol)
    X = fo.vectorize(mz, intensity, indptr, start, fragment_tol, n_bins, low_dim, 0, True, order, np.float32)
    splits = fo.bucket_splits(mzs, tol, mode, batch_size, mz_interval)
    t["vectorize"] = time.perf_counter() - t0
    t0 = time.perf_counter()
    N = len(pmz)
    sim = np.full((N, n_neighbors_ann), -np.inf, np.float32)
    idx = np.full((N, n_neighbors_ann), -1, np.int32)
    for a, b in zip(splits[:-1], splits[1:]):
        a, b = int(a), int(b)
        S = X[a:b] @ X[a:b].T                                   # sgemm: all host cores
        k = min(n_neighbors_ann, b - a)
        if k < b - a:
            part = np.argpartition(-S, k - 1, axis=1)[:, :k]
        else:
            part = np.broadcast_to(np.arange(b - a), (b - a, b - a)).copy()
        ps = np.take_along_axis(S, part, 1)
        o = np.argsort(-ps, axis=1, kind="stable")
        sim[a:b, :k] = np.take_along_axis(ps, o, 1)
        idx[a:b, :k] = np.take_along_axis(part, o, 1) + a
    t["gemm_topk"] = time.perf_counter() - t0
    t0 = time.perf_counter()
    nb_idx, nb_dist = fo.filter_neighbors(sim, idx, mzs, None, tol, mode, None, n_neighbors)
    valid = nb_idx >= 0
    indp = np.concatenate([[0], np.cumsum(valid.sum(1))])
    # (a stored zero would read as "no edge" in a sparse precomputed matrix: keep exact duplicates as a tiny distance)
    G = csr_matrix((np.maximum(nb_dist[valid], np.float32(1e-12)), nb_idx[valid], indp), shape=(N, N))
    G = G.maximum(G.T)                                          # sklearn wants a symmetric neighbourhood graph
    from sklearn.neighbors import sort_graph_by_row_values
    G = sort_graph_by_row_values(G.tocsr(), warn_when_not_sorted=False)
    import warnings
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")                         # (sklearn's EfficiencyWarning about row order: timing only)
        db = DBSCAN(eps=eps, min_samples=2, metric="precomputed").fit(G).labels_.astype(np.int32)
    t["filter_dbscan"] = time.perf_counter() - t0
    labels = np.empty(N, np.int32)
    labels[order] = db
    noise = labels == -1
    n_cl = int(labels.max()) + 1 if N else 0
    labels[noise] = np.arange(n_cl, n_cl + int(noise.sum()))
    return labels, t
