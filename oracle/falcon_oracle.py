"""CPU oracle for falcon's vectorise -> ANN -> DBSCAN hot path.

TEST INFRASTRUCTURE -- NOT PRODUCT CODE.  Only `tests/`, `__graft_entry__.smoke()`
and the `cpu_baseline` leg of `bench.py` may import this module; the product
(`falcon_amd/`) never does and fails loudly when its HIP library is missing.

What this is: a plain numpy restatement of every stage in SURVEY.md section 8(a)
(rows a1..a12).  All `file:line` citations are relative to the reference checkout
(bittremieux/falcon @ 2024-12-23).

Pinning status (see DESIGN.md "Oracle"):
  * a1, a2, a5, a10, a11(dense), a12 and the 1-D linkage are pinned against the
    reference's OWN function bodies, run in the build container
    (tests/golden/make_golden.py -> tests/golden/reference_functions.npz).
  * a3 (feature hashing) is pinned against sklearn's MurmurHash3 and against the
    reference's `to_vector` driven with a hash projection matrix (same .npz).
  * a9 (DBSCAN) is pinned against scikit-learn's DBSCAN on sparse precomputed
    graphs (tests/golden/sklearn_dbscan.npz).
  * f1 (`process_spectra`, the preprocessing step in front of the path, spectrum.py:73-169):
    the spectrum_utils 0.3.5 calls it makes are an absent dependency (setup.cfg:20-34);
    restated from SURVEY Appendix B.  PARITY UNPINNED; cross-checked against the host
    implementation `falcon_amd.cluster.spectrum.process_spectrum`.
  * f4 (`cosine_fast`, similarity.py:17-80 -- the matched-peak cosine the snapshot ships) is pinned
    against the reference's own function body (tests/golden/cosine_fast.npz, 320 pairs).
  * a6/a7 (IVF build, n_probe search): the reference snapshot contains NO
    implementation (Faiss is an un-vendored dependency, setup.cfg:25, and the call
    sites are gone -- SURVEY section 0).  PARITY UNPINNED for the index itself; the
    search RESULT is pinned in the exhaustive setting (n_probe = n_list), where it
    must equal a brute-force numpy top-k.
"""
from __future__ import annotations

import math
from typing import List, Optional, Tuple

import numpy as np

f32 = np.float32
f64 = np.float64

FLAT_MAX = 100          # buckets up to this size use a single list (flat index)
MIN_PTS_PER_LIST = 39   # Faiss' min points per centroid [SURVEY App. A, recollection]
MAX_N_LIST = 1 << 17


# --------------------------------------------------------------------------- a1
def get_dim(min_mz: float, max_mz: float, bin_size: float) -> Tuple[int, float, float]:
    """spectrum.py:172-199.  The numba signature is (f4, f4, f4) -> (u4, f4, f4),
    so every operation is float32 (Python's % on floats == fmod for positives)."""
    lo, hi, b = f32(min_mz), f32(max_mz), f32(bin_size)
    start = f32(lo - f32(np.fmod(lo, b)))
    end = f32(f32(hi + b) - f32(np.fmod(hi, b)))
    dim = int(math.ceil(f32(f32(end - start) / b)))
    return dim, float(start), float(end)


# --------------------------------------------------------------------------- a2
def bin_indices(mz: np.ndarray, min_mz: float, bin_size: float) -> np.ndarray:
    """spectrum.py:291: floor((mz - min_mz) / bin_size); mz is float32, min_mz and
    bin_size are Python floats => float64 arithmetic, true division."""
    return np.floor((mz.astype(f64) - f64(min_mz)) / f64(bin_size)).astype(np.int32)


# --------------------------------------------------------------------------- a3
def murmurhash3_32(keys: np.ndarray, seed: int = 0) -> np.ndarray:
    """MurmurHash3_x86_32 of each int32 key (4 little-endian bytes), unsigned result.
    Published algorithm (Appleby, public domain); spec README.md:124-131."""
    k = np.asarray(keys).astype(np.int64).astype(np.uint32).astype(np.uint64)
    M = np.uint64(0xFFFFFFFF)
    c1, c2 = np.uint64(0xCC9E2D51), np.uint64(0x1B873593)

    def rotl(x, r):
        return ((x << np.uint64(r)) | (x >> np.uint64(32 - r))) & M

    k = (k * c1) & M
    k = rotl(k, 15)
    k = (k * c2) & M
    h = np.uint64(seed & 0xFFFFFFFF) ^ k
    h = rotl(h, 13)
    h = (h * np.uint64(5) + np.uint64(0xE6546B64)) & M
    h ^= np.uint64(4)                      # len
    h ^= h >> np.uint64(16)
    h = (h * np.uint64(0x85EBCA6B)) & M
    h ^= h >> np.uint64(13)
    h = (h * np.uint64(0xC2B2AE35)) & M
    h ^= h >> np.uint64(16)
    return h.astype(np.uint32)


def hash_lookup(n_bins: int, low_dim: int, seed: int = 0) -> np.ndarray:
    """hash bin of every m/z bin: murmurhash3_32(bin, seed, unsigned) % low_dim."""
    return (murmurhash3_32(np.arange(n_bins, dtype=np.int64), seed) % np.uint32(low_dim)).astype(np.uint32)


def l2_norm_sq_tree(V: np.ndarray) -> np.ndarray:
    """Sum of squares of each row, float64, in the FIXED order the HIP kernel uses:
    'lane' l (0..63) owns elements {256*p + 4*l + c}; it adds their squares in
    (p, c) order; the 64 partials are combined by an xor-butterfly 32,16,...,1."""
    n, d = V.shape
    P = (d + 255) // 256
    W = np.zeros((n, P * 256), f64)
    W[:, :d] = V.astype(f64)
    W = W.reshape(n, P, 64, 4)
    part = np.zeros((n, 64), f64)
    for p in range(P):
        for c in range(4):
            x = W[:, p, :, c]
            part = part + x * x
    lanes = np.arange(64)
    for off in (32, 16, 8, 4, 2, 1):
        part = part + part[:, lanes ^ off]
    return part[:, 0]


def l2_normalize_rows(V: np.ndarray) -> np.ndarray:
    """Row-wise L2 normalisation in the spirit of faiss.normalize_L2 (spectrum.py:246):
    inv = (float)(1.0 / sqrt(nr)) with nr, sqrt and the divide in float64 (correctly
    rounded on both CPU and gfx950); x *= inv; all-zero rows are left untouched."""
    nr = l2_norm_sq_tree(V)
    out = V.astype(f32).copy()
    nz = nr > 0
    inv = (f64(1.0) / np.sqrt(nr[nz])).astype(f32)
    out[nz] = out[nz] * inv[:, None]
    return out


ROW_WIDTHS = (64, 128, 256, 400, 800)


def row_width(low_dim: int) -> int:
    """[build rule] `--low_dim` is a free integer (README.md:114-117); the path stores the vectors in rows of the next of
    these widths, zero behind column low_dim.  A zero column adds fma(0 * 0, acc) = acc to a chain, so the padding changes no
    value of the low_dim-dimensional inner product -- only the ORDER of its terms, which is that of the row_width-wide chain
    (`sims_f32` with d = row_width)."""
    for w in ROW_WIDTHS:
        if 1 <= low_dim <= w:
            return w
    raise ValueError(f"low_dim must be in [1, {ROW_WIDTHS[-1]}] (got {low_dim})")


def vectorize(mz: np.ndarray, intensity: np.ndarray, indptr: np.ndarray,
              min_mz: float, bin_size: float, n_bins: int, low_dim: int,
              seed: int = 0, norm: bool = True, row_order: Optional[np.ndarray] = None,
              dtype=np.float32, width: Optional[int] = None) -> np.ndarray:
    """a2+a3: CSR peaks -> dense [n, width] vectors (spectrum.py:202-247 with the
    projection realised as feature hashing, README.md:124-131); `width` (default low_dim) >= low_dim
    columns per row, zero behind low_dim.

    Peaks are added in peak order with float32 adds; peaks whose bin falls outside
    [0, n_bins) are ignored.  `row_order[r]` = input spectrum that becomes row r."""
    width = low_dim if width is None else int(width)
    assert width >= low_dim
    n = len(indptr) - 1
    if row_order is None:
        row_order = np.arange(n)
    counts = np.diff(indptr)[row_order]
    rows = np.repeat(np.arange(n), counts)
    # flat peak positions of every output row, in peak order
    starts = np.asarray(indptr[:-1])[row_order]
    pos = np.repeat(starts - np.concatenate([[0], np.cumsum(counts)[:-1]]), counts) + np.arange(counts.sum())
    b = bin_indices(np.asarray(mz)[pos], min_mz, bin_size)
    ok = (b >= 0) & (b < n_bins)
    h = (murmurhash3_32(b[ok], seed) % np.uint32(low_dim)).astype(np.int64)
    V = np.zeros((n, width), f32)
    np.add.at(V, (rows[ok], h), np.asarray(intensity, f32)[pos][ok])   # ordered, unbuffered f32 adds
    if norm:
        V = l2_normalize_rows(V)
    return V.astype(dtype)


# --------------------------------------------------------------------------- a5
def mass_diff(a, b, is_da: bool):
    """spectrum_utils.utils.mass_diff as numba types it for float32 inputs
    (cluster.py:190-195): the difference and the division are float32, the product
    with the int literal 10**6 is float64.  PARITY UNPINNED (numba absent)."""
    a = np.asarray(a, f32)
    b = np.asarray(b, f32)
    diff = (a - b).astype(f32)
    if is_da:
        return diff.astype(f64)
    return (diff / b).astype(f32).astype(f64) * 1e6


def get_precursor_mz_splits(precursor_mzs: np.ndarray, tol: float, mode: str, batch_size: int) -> np.ndarray:
    """cluster.py:159-209, including the quirk that the LAST block is never chunked."""
    mz = np.asarray(precursor_mzs, f32)
    splits = [0]
    n = len(mz)
    if n > 1:
        gap = mass_diff(mz[1:], mz[:-1], mode == "Da") > tol
        for i in (np.flatnonzero(gap) + 1):
            i = int(i)
            block = i - splits[-1]
            if block < batch_size:
                splits.append(i)
            else:
                n_chunks = math.ceil(block / batch_size)
                chunk = block // n_chunks
                for _ in range(block % n_chunks):
                    splits.append(splits[-1] + chunk + 1)
                for _ in range(n_chunks - (block % n_chunks)):
                    splits.append(splits[-1] + chunk)
    splits.append(n)
    return np.asarray(splits, np.int64)


def bucket_splits(precursor_mzs: np.ndarray, tol: float, mode: str, batch_size: int,
                  mz_interval: float = 1.0, chunk_last: bool = True) -> np.ndarray:
    """Bucket boundaries the ANN path uses = reference splits (above)
    + [build rule] the same chunk formula applied to the last block too
    + [build rule, SURVEY 8(d)] fixed precursor windows floor(mz / mz_interval), the two rules above applied inside
      every window (round 3; before, the reference rule ran over the whole array and the window cuts were added to it:
      the two differ only where a gap-free run of >= batch_size spectra crosses a window boundary)."""
    mz = np.asarray(precursor_mzs, f32)
    n = len(mz)
    if mz_interval and mz_interval > 0 and n > 1:
        # [build rule] fixed precursor windows: the reference's rule (+ the chunked last block) runs INSIDE every window, so a
        # window's buckets depend on its own spectra only (the unit the multi-GPU job deals out)
        w = np.floor(mz.astype(f64) / f64(mz_interval))
        edges = np.concatenate([[0], np.flatnonzero(w[1:] != w[:-1]) + 1, [n]])
        if len(edges) > 2:
            cuts = set()
            for a, b in zip(edges[:-1], edges[1:]):
                cuts.update(int(a) + int(x) for x in bucket_splits(mz[a:b], tol, mode, batch_size, 0.0, chunk_last))
            return np.asarray(sorted(cuts), np.int64)
    base = get_precursor_mz_splits(mz, tol, mode, batch_size)
    cuts = set(int(x) for x in base)
    if chunk_last and len(base) >= 2:
        a, b = int(base[-2]), int(base[-1])
        block = b - a
        if block >= batch_size:
            n_chunks = math.ceil(block / batch_size)
            chunk = block // n_chunks
            s = a
            for _ in range(block % n_chunks):
                s += chunk + 1
                cuts.add(s)
            for _ in range(n_chunks - (block % n_chunks)):
                s += chunk
                cuts.add(s)
    if mz_interval and mz_interval > 0 and n > 1:
        w = np.floor(mz.astype(f64) / f64(mz_interval))
        for i in np.flatnonzero(w[1:] != w[:-1]) + 1:
            cuts.add(int(i))
    return np.asarray(sorted(cuts), np.int64)


# ------------------------------------------------------------------------ a6/a7
def n_list_for(n_b: int) -> int:
    """[SURVEY App. A recollection] flat index for tiny buckets, else
    2^floor(log2(n/39)) lists, capped."""
    if n_b <= FLAT_MAX:
        return 1
    return int(min(MAX_N_LIST, 2 ** int(math.floor(math.log2(n_b / MIN_PTS_PER_LIST)))))


# ---- the kernels' summation order (oracle/kordered.c; built by __graft_entry__.build()) ----------
_KLIB = None


def _klib():
    """ctypes handle of oracle/_build/libkordered.so, or None when it has not been built."""
    global _KLIB
    if _KLIB is None:
        import ctypes as C
        import os
        fn = os.path.join(os.path.dirname(os.path.abspath(__file__)), "_build", "libkordered.so")
        if not os.path.isfile(fn):
            _KLIB = False
        else:
            lib = C.CDLL(fn)
            vp, i64, i32 = C.c_void_p, C.c_int64, C.c_int
            lib.fo_sims_kordered.argtypes = [vp, i64, vp, i64, i32, vp]
            lib.fo_argmax_kordered.argtypes = [vp, i64, vp, i64, i32, vp]
            lib.fo_ivf_search.argtypes = [vp, i64, i32, vp, i32, vp, vp, i32, i64, vp, vp]
            lib.fo_topk_rows.argtypes = [vp, i64, i64, i32, vp, vp]
            lib.fo_set_threads.argtypes = [i32]
            _KLIB = lib
    return _KLIB or None


def have_kordered() -> bool:
    """True when the C helper is built: `sims_f32` is then BIT-IDENTICAL to the HIP kernels."""
    return _klib() is not None


def _ptr(a):
    return a.ctypes.data


def sims_f32(A: np.ndarray, B: np.ndarray) -> np.ndarray:
    """Inner products in float32, summed as the HIP kernels sum them (simtile.h): with dh = d/2,
    acc = fma(a[dh+k], b[dh+k], fma(a[k], b[k], acc)) for k = 0..dh-1, one rounding per fma.
    With oracle/_build/libkordered.so (gcc fmaf) the result equals the kernels bit for bit; without it
    the fma is emulated in float64 (exact product, one extra rounding of the sum: differs from a true
    fma only when the float64 sum lands exactly on a float32 rounding boundary)."""
    A = np.ascontiguousarray(A, f32)
    B = np.ascontiguousarray(B, f32)
    na, d = A.shape
    nb = B.shape[0]
    out = np.empty((na, nb), f32)
    lib = _klib()
    if lib is not None and d % 2 == 0:
        rc = lib.fo_sims_kordered(_ptr(A), na, _ptr(B), nb, d, _ptr(out))
        assert rc == 0, rc
        return out
    dh = d // 2
    acc = np.zeros((na, nb), f32)
    A64, B64 = A.astype(f64), B.astype(f64)
    for k in range(dh):
        acc = (A64[:, k, None] * B64[None, :, k] + acc.astype(f64)).astype(f32)
        acc = (A64[:, dh + k, None] * B64[None, :, dh + k] + acc.astype(f64)).astype(f32)
    return acc


def kmeans_assign(X: np.ndarray, C: np.ndarray) -> np.ndarray:
    """argmax inner product, ties -> lowest centroid id."""
    lib = _klib()
    X = np.ascontiguousarray(X, f32)
    C = np.ascontiguousarray(C, f32)
    if lib is not None and X.shape[1] % 2 == 0:
        out = np.empty(len(X), np.int32)
        rc = lib.fo_argmax_kordered(_ptr(X), len(X), _ptr(C), len(C), X.shape[1], _ptr(out))
        assert rc == 0, rc
        return out
    return np.argmax(sims_f32(X, C), axis=1).astype(np.int32)


def kmeans_update(X: np.ndarray, assign: np.ndarray, C_prev: np.ndarray) -> np.ndarray:
    """Spherical update: centroid = normalise(sum of members, float32, member index
    order); empty lists keep their previous centroid."""
    C = np.zeros_like(C_prev, dtype=f32)
    np.add.at(C, assign, X.astype(f32))               # ordered f32 adds
    cnt = np.bincount(assign, minlength=len(C_prev))
    C = l2_normalize_rows(C)
    empty = cnt == 0
    C[empty] = C_prev[empty]
    return C


def ivf_train(X: np.ndarray, n_list: int, n_iter: int = 10) -> np.ndarray:
    """Deterministic k-means: init = rows floor(i*n/n_list); n_iter x (assign, update)."""
    n = len(X)
    init = (np.arange(n_list, dtype=np.int64) * n) // n_list
    C = X[init].astype(f32).copy()
    for _ in range(n_iter):
        C = kmeans_update(X, kmeans_assign(X, C), C)
    return C


def ivf_build(X: np.ndarray, n_list: int, n_iter: int = 10):
    """-> centroids [n_list,d], list id per row, perm (rows in (list, index) order),
    offsets [n_list+1]."""
    if n_list <= 1:
        n = len(X)
        return (np.zeros((1, X.shape[1]), f32), np.zeros(n, np.int32),
                np.arange(n, dtype=np.int64), np.array([0, n], np.int64))
    C = ivf_train(X, n_list, n_iter)
    a = kmeans_assign(X, C)
    perm = np.argsort(a, kind="stable")
    off = np.concatenate([[0], np.cumsum(np.bincount(a, minlength=n_list))]).astype(np.int64)
    return C, a, perm, off


def topk_desc(sim_row: np.ndarray, ids: np.ndarray, k: int):
    """k best by (sim descending, id ascending)."""
    order = np.lexsort((ids, -sim_row.astype(f64)))[:k]
    return sim_row[order], ids[order]


def _topk_rows(S: np.ndarray, k: int):
    """row-wise k best of a dense [n, m] matrix by (sim desc, column asc) -> sim [n,k] (pad -inf), idx [n,k] (pad -1)"""
    S = np.ascontiguousarray(S, f32)
    n, m = S.shape
    sim = np.full((n, k), -np.inf, f32)
    idx = np.full((n, k), -1, np.int32)
    lib = _klib()
    if lib is not None:
        lib.fo_topk_rows(_ptr(S), n, m, k, _ptr(sim), _ptr(idx))
        return sim, idx
    ids = np.arange(m)
    for i in range(n):
        ss, ii = topk_desc(S[i], ids, k)
        sim[i, :len(ss)] = ss
        idx[i, :len(ii)] = ii
    return sim, idx


def coarse_probe(X: np.ndarray, C: np.ndarray, n_probe: int) -> np.ndarray:
    """top-n_probe centroids per query by (sim desc, list id asc)."""
    return _topk_rows(sims_f32(X, C), n_probe)[1]


def ivf_search(X: np.ndarray, C: np.ndarray, assign: np.ndarray, perm: np.ndarray, off: np.ndarray,
               n_probe: int, k_ann: int, base: int = 0, probes: Optional[np.ndarray] = None):
    """n_probe query of every row of the bucket against the bucket's own index.
    -> sim f32[n,k_ann] (pad -inf), idx i32[n,k_ann] (pad -1; ids are base + row)."""
    n = len(X)
    n_list = len(off) - 1
    sim = np.full((n, k_ann), -np.inf, f32)
    idx = np.full((n, k_ann), -1, np.int32)
    if n_list == 1:
        probes = np.zeros((n, 1), np.int32)
    elif probes is None:
        probes = coarse_probe(X, C, min(n_probe, n_list))
    Xf = np.ascontiguousarray(X, f32)
    lib = _klib()
    if lib is not None and Xf.shape[1] % 2 == 0 and n > 0:
        pr = np.ascontiguousarray(probes, np.int32)
        pm = np.ascontiguousarray(perm, np.int64)
        of = np.ascontiguousarray(off, np.int64)
        rc = lib.fo_ivf_search(_ptr(Xf), n, Xf.shape[1], _ptr(pr), pr.shape[1], _ptr(pm), _ptr(of), k_ann, base,
                               _ptr(sim), _ptr(idx))
        assert rc == 0, rc
        return sim, idx
    for i in range(n):
        cand = np.concatenate([perm[off[l]:off[l + 1]] for l in probes[i] if l >= 0])
        s = sims_f32(Xf[cand], Xf[i:i + 1])[:, 0]
        ss, ii = topk_desc(s, cand, k_ann)
        sim[i, :len(ss)] = ss
        idx[i, :len(ii)] = ii + base
    return sim, idx


def exhaustive_topk(X: np.ndarray, k: int, base: int = 0):
    """Ground truth: brute-force cosine top-k inside one bucket, in the kernels' summation order."""
    Xf = np.ascontiguousarray(X, f32)
    sim, idx = _topk_rows(sims_f32(Xf, Xf), k)
    idx[idx >= 0] += base
    return sim, idx


# --------------------------------------------------------------------------- a8
def filter_neighbors(sim: np.ndarray, idx: np.ndarray, precursor_mz: np.ndarray,
                     rt: Optional[np.ndarray], tol: float, mode: str, rt_tol: Optional[float],
                     n_neighbors: int):
    """Drop self / empty slots / neighbours outside the precursor (and RT) tolerance,
    keep the first n_neighbors (rows are sorted by descending sim),
    dist = clip(1 - sim, 0, 1) (cluster.py:626, similarity.py:78).
    mass_diff(query, neighbour) as in cluster.py:190-195."""
    n, ka = idx.shape
    rows = np.arange(n)[:, None]
    j = np.where(idx < 0, 0, idx)
    ok = (idx >= 0) & (idx != rows)
    md = np.abs(mass_diff(np.broadcast_to(precursor_mz[:, None], idx.shape), precursor_mz[j], mode == "Da"))
    ok &= md <= tol
    if rt_tol is not None:
        rd = np.abs((np.asarray(rt, f32)[:, None] - np.asarray(rt, f32)[j]).astype(f32)).astype(f64)
        ok &= rd <= rt_tol
    rank = np.cumsum(ok, axis=1) - 1
    ok &= rank < n_neighbors
    out_idx = np.full((n, n_neighbors), -1, np.int32)
    out_dist = np.full((n, n_neighbors), np.inf, f32)
    r, c = np.nonzero(ok)
    d = np.clip(f32(1.0) - sim[r, c].astype(f32), f32(0), f32(1)).astype(f32)
    out_idx[r, rank[r, c]] = idx[r, c]
    out_dist[r, rank[r, c]] = d
    return out_idx, out_dist


# --------------------------------------------------------------------------- a9
def dbscan_sklearn_order(nb_idx: np.ndarray, nb_dist: np.ndarray, eps: float, min_samples: int = 2) -> np.ndarray:
    """DBSCAN on a sparse precomputed (directed) neighbour graph exactly as
    scikit-learn runs it (README.md:143-146; min_samples = 2 as cluster.py:66):
    a point is its own neighbour; core <=> |N_eps| >= min_samples; clusters grow
    depth-first from unlabelled core points in index order; a border point joins
    the first cluster that reaches it; noise = -1."""
    n = len(nb_idx)
    within = (nb_idx >= 0) & (nb_dist <= f32(eps)) & (nb_idx != np.arange(n)[:, None])
    core = (within.sum(1) + 1) >= min_samples
    # sklearn's neighbourhoods come from a CSR with sorted column indices, self included
    neigh = []
    for i in range(n):
        nb = np.unique(np.concatenate([nb_idx[i][within[i]], [i]]))
        neigh.append(nb)
    labels = np.full(n, -1, np.int32)
    label = 0
    for i in range(n):
        if labels[i] != -1 or not core[i]:
            continue
        stack = []
        while True:
            if labels[i] == -1:
                labels[i] = label
                if core[i]:
                    for v in neigh[i]:
                        if labels[v] == -1:
                            stack.append(v)
            if not stack:
                break
            i = stack.pop()
        label += 1
    return labels


def dbscan_components(nb_idx: np.ndarray, nb_dist: np.ndarray, eps: float) -> np.ndarray:
    """The order-independent DBSCAN(min_samples=2) the HIP path implements:
      core(i)  <=> row i stores a neighbour j != i with dist <= eps;
      clusters =  connected components of core points under core->core eps-edges
                  (taken as undirected);
      border j (non-core, reached by some core i -> j eps-edge) joins the cluster
                  of its LOWEST-index such core i;
      clusters are numbered by their lowest core index; everything else is -1.
    Differs from sklearn only where a directed core->core edge i->j exists whose
    start i is not reachable from j's cluster (sklearn then depends on visiting
    order)."""
    import scipy.sparse as ss
    from scipy.sparse.csgraph import connected_components
    n, k = nb_idx.shape
    rows = np.repeat(np.arange(n), k).reshape(n, k)
    within = (nb_idx >= 0) & (nb_dist <= f32(eps)) & (nb_idx != rows)
    core = within.any(1)
    r, c = rows[within], nb_idx[within]
    cc = core[r] & core[c]
    G = ss.csr_matrix((np.ones(cc.sum(), np.int8), (r[cc], c[cc])), (n, n))
    _, comp = connected_components(G, directed=False)
    labels = np.full(n, -1, np.int64)
    # number components by lowest core index
    core_idx = np.flatnonzero(core)
    first = {}
    for i in core_idx:
        first.setdefault(comp[i], len(first))
    labels[core_idx] = [first[comp[i]] for i in core_idx]
    # border points: lowest-index core in-neighbour
    b = core[r] & ~core[c]
    br, bc = r[b], c[b]
    order = np.lexsort((br, bc))
    br, bc = br[order], bc[order]
    keep = np.concatenate([[True], bc[1:] != bc[:-1]]) if len(bc) else np.zeros(0, bool)
    labels[bc[keep]] = labels[br[keep]]
    return labels.astype(np.int32)



# --------------------------------------------------------------------------- f4
def linkage_clusters(nb_idx: np.ndarray, nb_dist: np.ndarray, t: float, method: str) -> np.ndarray:
    """Hierarchical clustering of the sparse neighbour graph cut at distance t: the snapshot's
    `fcluster(fastcluster.linkage(pdist, linkage), distance_threshold, "distance")` (cluster.py:283-290) with
    "missing pair = distance 1" (cluster.py:621-626); scipy's `linkage` stands in for fastcluster (absent; same
    dendrogram up to the order of equal heights -- PARITY UNPINNED for exact ties).  Only the connected components of
    the edges with d <= t can merge below t < 1, so the dense matrix is built per component.
    d(i, j) = the smaller of the stored directions.  -> labels int32[n]: clusters numbered by lowest row, groups of
    one row = -1 (what _postprocess_cluster makes of them, cluster.py:441-454) -- the contract of `dbscan_components`."""
    from scipy.cluster.hierarchy import fcluster, linkage
    from scipy.sparse import coo_matrix
    from scipy.sparse.csgraph import connected_components
    from scipy.spatial.distance import squareform
    assert t < 1.0 and method in ("single", "complete", "average")
    n, k = nb_idx.shape
    rows = np.repeat(np.arange(n), k)
    cols = nb_idx.ravel().astype(np.int64)
    d = nb_dist.ravel()
    ok = (cols >= 0) & (cols != rows)
    e = ok & (d <= f32(t))
    g = coo_matrix((np.ones(int(e.sum())), (rows[e], cols[e])), shape=(n, n))
    _, comp = connected_components(g, directed=False)
    rep = np.full(n, -1, np.int64)
    order = np.argsort(comp, kind="stable")
    cs = comp[order]
    lidx = np.zeros(n, np.int64)
    for a, b in cluster_group_idx(cs):
        members = order[a:b]                                    # ascending rows (stable sort of arange)
        m = b - a
        if m < 2:
            continue
        if method == "single":
            rep[members] = members[0]
            continue
        lidx[members] = np.arange(m)
        D = np.ones((m, m), f64)
        for la, r in enumerate(members):
            for s in range(k):
                j = int(nb_idx[r, s])
                if j >= 0 and j != r and comp[j] == comp[r]:
                    D[la, lidx[j]] = f64(nb_dist[r, s])
        D = np.minimum(D, D.T)
        np.fill_diagonal(D, 0.0)
        fl = fcluster(linkage(squareform(D, checks=False), method), float(f32(t)), "distance")
        for c in np.unique(fl):
            mm = members[fl == c]
            if len(mm) >= 2:
                rep[mm] = mm.min()
    labels = np.full(n, -1, np.int32)
    reps = np.unique(rep[rep >= 0])
    rank = {int(r): i for i, r in enumerate(reps)}
    for i in np.flatnonzero(rep >= 0):
        labels[i] = rank[int(rep[i])]
    return labels



# -------------------------------------------------------------------------- a10
def linkage_1d(values: np.ndarray, mode: Optional[str]) -> np.ndarray:
    """cluster.py:458-509: complete linkage of a 1-D array by repeatedly merging the
    adjacent pair (in sorted order) with the smallest (max_right - min_left);
    'ppm' divides by the left cluster's minimum.  Arithmetic as mass_diff above."""
    v = np.asarray(values, f32)
    n = len(v)
    Z = np.zeros((n - 1, 4), f64)
    order = np.argsort(v, kind="stable")
    cl = [(v[i], v[i], int(i), 1) for i in order]
    for it in range(n - 1):
        lo = np.array([c[0] for c in cl[:-1]], f32)
        hi = np.array([c[1] for c in cl[1:]], f32)
        dist = (hi - lo).astype(f32)
        if mode == "ppm":
            dist = (dist / lo).astype(f32).astype(f64) * 1e6
        else:
            dist = dist.astype(f64)
        m = int(np.argmin(dist))
        npts = cl[m][3] + cl[m + 1][3]
        Z[it] = [cl[m][2], cl[m + 1][2], dist[m], npts]
        cl[m] = (cl[m][0], cl[m + 1][1], n + it, npts)
        del cl[m + 1]
    return Z


def flat_1d(values: np.ndarray, tol: float, mode: Optional[str]) -> np.ndarray:
    """Partition a 1-D array exactly as fcluster(_linkage(values), tol, 'distance')
    does, returned as contiguous segment ids over the SORTED order, mapped back:
    out[i] = segment of values[i]; segments numbered left to right.
    (Complete linkage is monotone, so the flat clusters are the merges with
    height <= tol.)"""
    v = np.asarray(values, f32)
    n = len(v)
    if n == 1:
        return np.zeros(1, np.int32)
    order = np.argsort(v, kind="stable")
    cl = [(v[i], v[i]) for i in order]          # (min, max) per current cluster
    size = [1] * n
    while len(cl) > 1:
        lo = np.array([c[0] for c in cl[:-1]], f32)
        hi = np.array([c[1] for c in cl[1:]], f32)
        dist = (hi - lo).astype(f32)
        dist = (dist / lo).astype(f32).astype(f64) * 1e6 if mode == "ppm" else dist.astype(f64)
        m = int(np.argmin(dist))
        if dist[m] > tol:
            break
        cl[m] = (cl[m][0], cl[m + 1][1])
        size[m] += size[m + 1]
        del cl[m + 1], size[m + 1]
    seg_sorted = np.repeat(np.arange(len(size)), size)
    out = np.empty(n, np.int32)
    out[order] = seg_sorted
    return out


def fcluster_numbering(Z: np.ndarray, t: float) -> np.ndarray:
    """scipy.cluster.hierarchy.fcluster(Z, t, 'distance') - 1 (cluster.py:412-416).
    scipy is a declared dependency of the reference (setup.cfg:35) and present on
    both boxes; the oracle calls it where the exact NUMBERING matters (the RT
    combination quirk cluster.py:418-429)."""
    import scipy.cluster.hierarchy as sch
    return (sch.fcluster(Z, t, "distance") - 1).astype(np.int32)


def postprocess_cluster(labels: np.ndarray, mzs: np.ndarray, rts: Optional[np.ndarray],
                        tol: float, mode: str, rt_tol: Optional[float], min_samples: int,
                        start_label: int) -> int:
    """cluster.py:362-455, in place on `labels`; returns the number of clusters."""
    m = len(labels)
    if m < min_samples:
        labels[:] = -1
        return 0
    if m == 1:
        assign = np.zeros(1, np.int32)
    else:
        assign = fcluster_numbering(linkage_1d(mzs, mode), tol)
        if rt_tol is not None:
            a_rt = fcluster_numbering(linkage_1d(rts, None), rt_tol)
            assign = np.unique(assign * 2 + a_rt * 3, return_inverse=True)[1].astype(np.int32)
    n_clusters = int(assign.max()) + 1
    if n_clusters == 1:
        labels[:] = start_label
    elif n_clusters == m:
        labels[:] = -1
        n_clusters = 0
    else:
        counts = {}
        for a in assign:
            counts[int(a)] = counts.get(int(a), 0) + 1
        n_clusters = 0
        new = {}
        for a, c in counts.items():            # dict order = first occurrence
            if c < min_samples:
                new[a] = -1
            else:
                new[a] = start_label + n_clusters
                n_clusters += 1
        labels[:] = [new[int(a)] for a in assign]
    return n_clusters


def cluster_group_idx(labels_sorted: np.ndarray):
    """cluster.py:334-359."""
    n = len(labels_sorted)
    i = 0
    while i < n and labels_sorted[i] == -1:
        yield i, i + 1
        i += 1
    j = i
    while j < n:
        i, lab = j, labels_sorted[j]
        while j < n and labels_sorted[j] == lab:
            j += 1
        yield i, j


# -------------------------------------------------------------------------- a11
def medoids_dense(idx_interval, labels_sorted, pdist, order_map) -> np.ndarray:
    """cluster.py:512-553 on a condensed full matrix (pinned by golden)."""
    m = len(idx_interval)
    med = []
    for a, b in cluster_group_idx(labels_sorted):
        if b - a > 1:
            rs = np.zeros(b - a, f32)
            for r in range(b - a):
                for c in range(r + 1, b - a):
                    i, j = order_map[a + r], order_map[a + c]
                    if i > j:
                        i, j = j, i
                    d = pdist[m * i + j - ((i + 2) * (i + 1)) // 2]
                    rs[r] = f32(rs[r] + f32(d))
                    rs[c] = f32(rs[c] + f32(d))
            med.append(idx_interval[a + int(np.argmin(rs))])
        else:
            med.append(idx_interval[a])
    return np.asarray(med, np.int32)


def medoid_scores_sparse(labels: np.ndarray, nb_idx: np.ndarray, nb_dist: np.ndarray) -> np.ndarray:
    """Sparse-graph restatement of the row sum of cluster.py:536-550: for member i of
    a cluster of size m,  s_i = sum (float32, stored order) of dist(i,j) over the
    stored neighbours j of i that carry the same label, plus 1.0 for every other
    member that row i does not store (a missing pair has cosine 0 -> distance 1,
    cluster.py:621-626)."""
    n, k = nb_idx.shape
    labels = np.asarray(labels)
    valid = labels >= 0                                  # noise rows (-1) score +inf
    j = np.where(nb_idx < 0, 0, nb_idx)
    same = ((nb_idx >= 0) & valid[:, None] & (labels[j] == labels[:, None])
            & (nb_idx != np.arange(n)[:, None]))
    s = np.zeros(n, f32)
    for c in range(k):
        s = np.where(same[:, c], (s + nb_dist[:, c]).astype(f32), s)
    n_lab = int(labels.max()) + 1 if valid.any() else 0
    size = np.bincount(labels[valid], minlength=max(n_lab, 1))[np.where(valid, labels, 0)]
    missing = (size - 1 - same.sum(1)).astype(f32)
    return np.where(valid, (s + missing).astype(f32), np.inf).astype(f32)


# -------------------------------------------------------------------------- a12
def assign_global_cluster_labels(labels: np.ndarray, idx: np.ndarray, splits, current_label: int) -> int:
    """cluster.py:556-590, in place."""
    max_label = current_label
    for i in range(len(splits) - 1):
        for j in idx[splits[i]:splits[i + 1]]:
            if labels[j] != -1:
                labels[j] += current_label
                if labels[j] > max_label:
                    max_label = labels[j]
        current_label = max_label + 1
    return max_label


# ------------------------------------------------------------------- the pipeline
def refine_and_number(db_labels: np.ndarray, core_first: np.ndarray, mz_sorted: np.ndarray,
                      rt_sorted: Optional[np.ndarray], tol: float, mode: str,
                      rt_tol: Optional[float]) -> np.ndarray:
    """a10 + numbering.  `db_labels` are DBSCAN labels over the precursor-sorted rows
    (clusters numbered by lowest core index, as both DBSCAN variants do).  Every
    DBSCAN cluster is split by `postprocess_cluster` (members in ascending row
    order); final clusters are numbered in (DBSCAN label, first occurrence) order,
    which is what _cluster_interval + _assign_global_cluster_labels produce block
    after block (cluster.py:293-313, 556-590).  Returns labels with -1 for noise."""
    n = len(db_labels)
    out = np.full(n, -1, np.int32)
    order = np.argsort(db_labels, kind="stable")
    ls = db_labels[order]
    cur = 0
    for a, b in cluster_group_idx(ls):
        if ls[a] == -1:
            continue
        rows = order[a:b]
        lab = np.zeros(b - a, np.int32)
        cur += postprocess_cluster(lab, mz_sorted[rows], None if rt_sorted is None else rt_sorted[rows],
                                   tol, mode, rt_tol, 2, cur)
        out[rows] = lab
    return out


def generate_clusters(mz, intensity, indptr, precursor_mz, rt, *, eps=0.1, precursor_tol=(20.0, "ppm"),
                      rt_tol=None, batch_size=2 ** 15, low_dim=400, n_probe=16, n_neighbors=64,
                      n_neighbors_ann=128, min_mz=101.0, max_mz=1500.0, fragment_tol=0.05,
                      mz_interval=1.0, kmeans_iters=10, hash_seed=0, dbscan="components",
                      dtype=np.float32, return_intermediates=False, n_jobs=1, rescore=False, min_matches=0,
                      clustering="dbscan", linkage="complete"):
    """Whole hot path for ONE charge partition -> (labels int32[N] by dataset row,
    no -1 left; medoids int32[n_labels]: medoids[c] = dataset row representing
    cluster c).  Mirrors cluster.generate_clusters (cluster.py:24-156) with the
    distance+linkage core replaced by ANN + DBSCAN."""
    tol, mode = float(precursor_tol[0]), precursor_tol[1]
    N = len(precursor_mz)
    pmz = np.asarray(precursor_mz, f32)
    order = np.argsort(pmz, kind="stable")                      # cluster.py:73-85
    mzs = pmz[order]
    rts = None if rt is None else np.asarray(rt, f32)[order]
    n_bins, start, _ = get_dim(min_mz, max_mz, fragment_tol)
    X = vectorize(mz, intensity, indptr, start, fragment_tol, n_bins, low_dim, hash_seed,
                  True, order, dtype, width=row_width(low_dim))
    splits = bucket_splits(mzs, tol, mode, batch_size, mz_interval)
    sim = np.full((N, n_neighbors_ann), -np.inf, f32)
    idx = np.full((N, n_neighbors_ann), -1, np.int32)
    def one_bucket(ab):
        a, b = int(ab[0]), int(ab[1])
        Xb = X[a:b].astype(f32)
        nl = n_list_for(b - a)
        C, asg, perm, off = ivf_build(Xb, nl, kmeans_iters)
        sim[a:b], idx[a:b] = ivf_search(Xb, C, asg, perm, off, n_probe, n_neighbors_ann, base=a)

    buckets = list(zip(splits[:-1], splits[1:]))
    if n_jobs > 1 and _klib() is not None:
        # buckets on a thread pool, like the reference's joblib threading backend over its blocks (cluster.py:115-136);
        # the C helper then runs one thread per call (ctypes releases the GIL)
        from concurrent.futures import ThreadPoolExecutor
        _klib().fo_set_threads(1)
        try:
            with ThreadPoolExecutor(max_workers=n_jobs) as pool:
                list(pool.map(one_bucket, buckets))
        finally:
            _klib().fo_set_threads(0)
    else:
        for ab in buckets:
            one_bucket(ab)
    nb_idx, nb_dist = filter_neighbors(sim, idx, mzs, rts, tol, mode, rt_tol, n_neighbors)
    if rescore or clustering == "hierarchical":                 # f4: exact matched-peak distances (cluster.py:593-639)
        nb_dist = rescore_neighbors(nb_idx, nb_dist, mz, intensity, indptr, order, fragment_tol, min_matches)
    if clustering == "hierarchical":                            # f4: the snapshot's linkage + cut (cluster.py:283-290)
        db = linkage_clusters(nb_idx, nb_dist, eps, linkage)
    elif dbscan == "sklearn":
        db = dbscan_sklearn_order(nb_idx, nb_dist, eps)
    else:
        db = dbscan_components(nb_idx, nb_dist, eps)
    lab_sorted = refine_and_number(db, None, mzs, rts, tol, mode, rt_tol)
    # medoids of real clusters
    n_clusters = int(lab_sorted.max()) + 1 if N else 0
    med_sorted = np.zeros(n_clusters, np.int64)
    if n_clusters:
        member = lab_sorted >= 0
        safe = np.where(member, lab_sorted, 0)
        score = medoid_scores_sparse(lab_sorted, nb_idx, nb_dist)
        o = np.lexsort((np.arange(N), score, safe))
        o = o[member[o]]
        first = np.concatenate([[True], safe[o][1:] != safe[o][:-1]])
        med_sorted[safe[o][first]] = o[first]
    labels = np.empty(N, np.int32)
    labels[order] = lab_sorted
    noise = labels == -1                                         # cluster.py:144-155
    n_noise = int(noise.sum())
    labels[noise] = np.arange(n_clusters, n_clusters + n_noise)
    medoids = np.concatenate([order[med_sorted], np.flatnonzero(noise)]).astype(np.int32)
    if return_intermediates:
        return labels, medoids, dict(order=order, X=X, splits=splits, sim=sim, idx=idx,
                                     nb_idx=nb_idx, nb_dist=nb_dist, db=db, lab_sorted=lab_sorted)
    return labels, medoids


# --------------------------------------------------------------------------- f1
PROTON = 1.0072766


def process_spectra(mz: np.ndarray, intensity: np.ndarray, indptr: np.ndarray, precursor_mz: np.ndarray,
                    precursor_charge: np.ndarray, min_peaks: int, min_mz_range: float, mz_min: Optional[float] = None,
                    mz_max: Optional[float] = None, remove_precursor_tolerance: Optional[float] = None,
                    min_intensity: Optional[float] = None, max_peaks_used: Optional[int] = None,
                    scaling: Optional[str] = None):
    """Batch restatement of `process_spectrum` (reference spectrum.py:73-169) over a CSR of raw peaks
    (mz float64, sorted per spectrum; intensity float32; charge 0 = unknown).

    -> valid bool[n], out_indptr i64[n+1] (invalid spectra hold 0 peaks), out_mz f32, out_intensity f32.

    Conventions where the reference leaves arithmetic open (PARITY UNPINNED): m/z comparisons in float64;
    the base-peak threshold `min_intensity * max` in float32; top-`max_peaks_used` ties keep the lower m/z;
    sqrt / log2 computed in float64 and rounded to float32; the L2 norm = (float32)sqrt(l2_norm_sq_tree)
    and a float32 divide.
    """
    n = len(indptr) - 1
    valid = np.zeros(n, bool)
    out_mz, out_it, counts = [], [], np.zeros(n, np.int64)

    def ok(m):                                                     # spectrum.py:27-52
        return len(m) >= max(min_peaks, 1) and (m.max() - m.min()) >= min_mz_range

    for i in range(n):
        m = np.asarray(mz[indptr[i]:indptr[i + 1]], f64)
        it = np.asarray(intensity[indptr[i]:indptr[i + 1]], f32)
        keep = np.ones(len(m), bool)                               # set_mz_range (spectrum.py:135)
        if mz_min is not None:
            keep &= m >= mz_min
        if mz_max is not None:
            keep &= m <= mz_max
        m, it = m[keep], it[keep]
        if not ok(m):
            continue
        if remove_precursor_tolerance is not None:                 # spectrum.py:139-149
            z = abs(int(precursor_charge[i])) if precursor_charge[i] != 0 else 1
            neutral = (f64(precursor_mz[i]) - f64(PROTON)) * f64(z)
            keep = np.ones(len(m), bool)
            for c in range(z, 0, -1):
                keep &= np.abs(m - (neutral / f64(c) + f64(PROTON))) > remove_precursor_tolerance
            m, it = m[keep], it[keep]
            if not ok(m):
                continue
        if min_intensity is not None or max_peaks_used is not None:   # spectrum.py:151-155
            mi = f32(0.0 if min_intensity is None else min_intensity)
            keep = it >= f32(mi * it.max())
            if max_peaks_used is not None and keep.sum() > max_peaks_used:
                idx = np.flatnonzero(keep)
                top = idx[np.argsort(-it[idx], kind="stable")[:max_peaks_used]]
                keep = np.zeros(len(m), bool)
                keep[top] = True
            m, it = m[keep], it[keep]
            if not ok(m):
                continue
        k = len(it)
        if scaling == "root":                                      # spectrum.py:157
            it = np.sqrt(it.astype(f64)).astype(f32)
        elif scaling == "log":
            it = np.log2((f32(1.0) + it).astype(f64)).astype(f32)
        elif scaling == "rank":
            max_rank = max_peaks_used if max_peaks_used is not None else k
            ranks = np.empty(k, np.int64)
            ranks[np.argsort(it, kind="stable")] = np.arange(1, k + 1)
            it = (max_rank - (k - ranks)).astype(f32)
        nrm = f32(np.sqrt(l2_norm_sq_tree(it[None, :])[0]))        # spectrum.py:55-70, 158
        with np.errstate(divide="ignore", invalid="ignore"):
            it = (it / nrm).astype(f32)
        valid[i] = True
        counts[i] = k
        out_mz.append(m.astype(f32))
        out_it.append(it)
    out_indptr = np.concatenate([[0], np.cumsum(counts)]).astype(np.int64)
    cat = lambda xs: np.concatenate(xs) if xs else np.zeros(0, f32)
    return valid, out_indptr, cat(out_mz), cat(out_it)


# --------------------------------------------------------------------------- f4
def cosine_fast(mz_a: np.ndarray, it_a: np.ndarray, mz_b: np.ndarray, it_b: np.ndarray, tol: float) -> Tuple[float, int]:
    """Matched-peak cosine of two spectra, reference similarity.py:17-80: every pair of peaks within the
    fragment tolerance gets cost it_a * it_b (float32), the optimal assignment is taken
    (`scipy.optimize.linear_sum_assignment`, maximize) and the positive pair scores are summed in row
    order (float64).  -> (score clipped to [0, 1], number of matched peaks).  Arithmetic as numba types it:
    `peak_mz - tol` in float64, `abs(peak_mz - other_mz)` in float32 compared against the float64 tolerance."""
    import scipy.optimize
    mz_a = np.asarray(mz_a, f32); mz_b = np.asarray(mz_b, f32)
    it_a = np.asarray(it_a, f32); it_b = np.asarray(it_b, f32)
    if len(mz_a) == 0 or len(mz_b) == 0:
        return 0.0, 0
    tol = f64(tol)
    cost = np.zeros((len(mz_a), len(mz_b)), f32)
    o = 0
    for p in range(len(mz_a)):                                     # similarity.py:45-63
        while o < len(mz_b) - 1 and f64(mz_a[p]) - tol > f64(mz_b[o]):
            o += 1
        q = o
        while q < len(mz_b) and f64(abs(f32(mz_a[p] - mz_b[q]))) <= tol:
            cost[p, q] = it_a[p] * it_b[q]
            q += 1
    rows, cols = scipy.optimize.linear_sum_assignment(cost, maximize=True)     # similarity.py:65-68
    score, n_match = 0.0, 0
    for r, c in zip(rows, cols):                                   # similarity.py:70-78
        if cost[r, c] > 0.0:
            score += float(cost[r, c])
            n_match += 1
    return max(0.0, min(score, 1.0)), n_match


def rescore_neighbors(nb_idx: np.ndarray, nb_dist: np.ndarray, mz: np.ndarray, intensity: np.ndarray,
                      indptr: np.ndarray, order: np.ndarray, tol: float, min_matches: int) -> np.ndarray:
    """Exact re-scoring of the ANN neighbour lists: every stored neighbour's distance becomes
    1 - cosine_fast(query, neighbour), or 1 when fewer than `min_matches` peaks match (reference
    cluster.py:621-630).  Rows / ids are sorted positions; `order` maps them to dataset rows."""
    out = nb_dist.copy()
    n, k = nb_idx.shape
    for i in range(n):
        a = order[i]
        for s in range(k):
            j = nb_idx[i, s]
            if j < 0:
                continue
            b = order[j]
            sim, nm = cosine_fast(mz[indptr[a]:indptr[a + 1]], intensity[indptr[a]:indptr[a + 1]],
                                  mz[indptr[b]:indptr[b + 1]], intensity[indptr[b]:indptr[b + 1]], tol)
            if nm < min_matches:
                sim = 0.0
            out[i, s] = f32(1.0 - sim)
    return out
