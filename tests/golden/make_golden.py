#!/usr/bin/env python3
"""Generate golden vectors from the reference's OWN function bodies.

TEST INFRASTRUCTURE.  Runs only in the build container, where the read-only
reference checkout lives at /root/reference.  Nothing from the reference is
copied: the reference modules are imported from where they lie, with a
throw-away identity shim (built in a temp dir, never committed) standing in for
the import-time dependencies that are absent here (numba, faiss, lance,
fastcluster, spectrum_utils).  The shim contains no arithmetic of the path
except `spectrum_utils.utils.mass_diff` (two lines).

Outputs (data only: inputs + expected outputs) land next to this script as
.npz files and are committed; the GPU box only ever sees those.

    python tests/golden/make_golden.py
"""
import os
import sys
import tempfile
import textwrap

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
REF = "/root/reference"

SHIM = {
    "numba/__init__.py": """
        import contextlib
        def njit(*a, **k):
            if len(a) == 1 and callable(a[0]) and not k:
                return a[0]
            return lambda f: f
        jit = njit
        @contextlib.contextmanager
        def objmode(**k):
            yield
        int64 = 'int64'; float32 = 'float32'
        class _L(list):
            pass
        class _D:
            @staticmethod
            def empty(key_type=None, value_type=None):
                return {}
        class typed:
            List = _L
            Dict = _D
        """,
    "faiss.py": "",
    "lance.py": "class LanceDataset: pass\n",
    "fastcluster.py": "",
    "spectrum_utils/__init__.py": "",
    "spectrum_utils/spectrum.py": "class MsmsSpectrum: pass\n",
    "spectrum_utils/utils.py": """
        def mass_diff(mz1, mz2, mode_is_da):
            return mz1 - mz2 if mode_is_da else (mz1 - mz2) / mz2 * 10**6
        """,
}


def _install_shim():
    d = tempfile.mkdtemp(prefix="falcon_shim_")
    for rel, src in SHIM.items():
        p = os.path.join(d, rel)
        os.makedirs(os.path.dirname(p), exist_ok=True)
        with open(p, "w") as f:
            f.write(textwrap.dedent(src))
    sys.dont_write_bytecode = True
    sys.path.insert(0, d)
    sys.path.insert(0, REF)
    return d


def main():
    if not os.path.isdir(REF):
        sys.exit("reference checkout not present; goldens can only be made in the build container")
    _install_shim()
    import importlib.util

    def load(name, path):
        spec = importlib.util.spec_from_file_location(name, path)
        m = importlib.util.module_from_spec(spec)
        sys.modules[name] = m
        spec.loader.exec_module(m)
        return m

    # Load leaf modules directly (falcon/__init__ and falcon.py need more deps).
    import types
    pkg = types.ModuleType("falcon"); pkg.__path__ = [REF + "/falcon"]; pkg.__version__ = "0"
    sys.modules["falcon"] = pkg
    cpk = types.ModuleType("falcon.cluster"); cpk.__path__ = [REF + "/falcon/cluster"]
    sys.modules["falcon.cluster"] = cpk
    similarity = load("falcon.cluster.similarity", REF + "/falcon/cluster/similarity.py")
    spectrum = load("falcon.cluster.spectrum", REF + "/falcon/cluster/spectrum.py")
    cluster = load("falcon.cluster.cluster", REF + "/falcon/cluster/cluster.py")

    rng = np.random.default_rng(20240917)
    f32 = np.float32
    out = {}

    # ---- a1 get_dim (float32 semantics: pass np.float32 scalars) -------------
    cases = [(101.0, 1500.0, 0.05), (101.0, 1500.0, 0.02), (50.0, 2000.0, 1.0005),
             (0.0, 500.0, 0.1), (101.0, 500.0, 0.05), (200.5, 1800.25, 0.005)]
    gd_in, gd_out = [], []
    for lo, hi, b in cases:
        dim, start, end = spectrum.get_dim(f32(lo), f32(hi), f32(b))
        gd_in.append((lo, hi, b))
        gd_out.append((int(dim), float(f32(start)), float(f32(end))))
    out["get_dim_in"] = np.array(gd_in, np.float64)
    out["get_dim_dim"] = np.array([o[0] for o in gd_out], np.int64)
    out["get_dim_start_end"] = np.array([(o[1], o[2]) for o in gd_out], np.float32)

    # ---- a2 _to_vector (bin indices; f32 mz promoted to f64 arithmetic) ------
    _, start, _ = spectrum.get_dim(f32(101.0), f32(1500.0), f32(0.05))
    min_mz = float(start)          # falcon.py:124 hands the f32 back as a Python float
    bin_size = 0.05
    n_spec = 64
    mzs, ints = [], []
    for s in range(n_spec):
        p = int(rng.integers(0, 70)) if s % 7 else 0       # includes empty spectra
        mz = np.sort(rng.uniform(101.0, 1500.0, p)).astype(f32)
        if p and s % 5 == 0:   # values sitting on/near bin edges
            k = rng.integers(0, 27000, p)
            mz = np.sort((min_mz + k * bin_size + rng.choice([-1e-4, 0.0, 1e-4], p))).astype(f32)
            mz = np.clip(mz, f32(101.0), f32(1500.0))
        it = rng.lognormal(0, 1, p).astype(f32)
        mzs.append(mz); ints.append(it)
    # numba types the Python-float arguments min_mz / bin_size as float64, so the
    # f32 m/z values are promoted.  Under the identity shim
    # NumPy-2 would keep a *Python* float "weak" (f32 arithmetic), so the scalars
    # are handed over as np.float64 to reproduce numba's typing.
    data, indices, indptr = spectrum._to_vector(mzs, ints, np.float64(min_mz), np.float64(bin_size))
    out["tv_min_mz"] = np.float64(min_mz)
    out["tv_bin_size"] = np.float64(bin_size)
    out["tv_mz"] = np.concatenate(mzs).astype(f32)
    out["tv_intensity"] = np.concatenate(ints).astype(f32)
    out["tv_indptr"] = indptr.astype(np.int64)
    out["tv_indices"] = indices.astype(np.int32)
    out["tv_data"] = data.astype(f32)

    # ---- a3 to_vector with a feature-hash projection, norm=False -------------
    # The reference realises "bin -> hash bin, scatter-add" as vectors @ CSR
    # (spectrum.py:240-243).  The projection matrix is built with sklearn's
    # MurmurHash3 (independent implementation), low_dim = 400.
    import scipy.sparse as ss
    from sklearn.utils import murmurhash3_32
    dim = int(spectrum.get_dim(f32(101.0), f32(1500.0), f32(0.05))[0])
    low_dim = 400
    h = np.array([murmurhash3_32(int(i), 0, True) % low_dim for i in range(dim)], np.int64)
    T = ss.csr_matrix((np.ones(dim, f32), (np.arange(dim), h)), (dim, low_dim), f32)
    spectra = [{"mz": m, "intensity": i} for m, i in zip(mzs, ints)]
    vec = spectrum.to_vector(spectra, T, np.float64(min_mz), np.float64(bin_size), dim, False)
    out["tv_hash_lookup_400"] = h.astype(np.uint32)
    out["tv_vectors_unnorm_400"] = np.asarray(vec, f32)
    # MurmurHash3_x86_32 known answers (int32 LE key), seeds 0 and 42.
    keys = np.array([0, 1, 2, 3, 4, 27981, 2**31 - 1, 123456789], np.int64)
    out["mmh3_keys"] = keys
    out["mmh3_seed0"] = np.array([murmurhash3_32(int(k), 0, True) for k in keys], np.uint32)
    out["mmh3_seed42"] = np.array([murmurhash3_32(int(k), 42, True) for k in keys], np.uint32)

    # ---- _norm_intensity (spectrum.py:55-70) ---------------------------------
    x = rng.lognormal(0, 1, 37).astype(f32)
    out["norm_in"] = x
    out["norm_out"] = spectrum._norm_intensity(x).astype(f32)

    # ---- a5 _get_precursor_mz_splits -----------------------------------------
    sp_cases = []
    def splits_case(mz, tol, mode, batch):
        mz = np.sort(np.asarray(mz, f32))
        s = list(cluster._get_precursor_mz_splits(mz, tol, mode, batch))
        sp_cases.append((mz, float(tol), mode, int(batch), np.asarray(s, np.int64)))
    # sparse: gaps split; dense: chunking; last block quirk; singletons
    splits_case(np.concatenate([400 + rng.normal(0, 0.001, 40), 500 + rng.normal(0, 0.001, 7),
                                [650.0], 700 + rng.normal(0, 0.0005, 90)]), 20.0, "ppm", 32)
    splits_case(rng.uniform(400, 401, 500), 20.0, "ppm", 64)
    splits_case(np.concatenate([rng.uniform(400, 400.5, 300), rng.uniform(900, 900.3, 211)]), 0.05, "Da", 50)
    splits_case(np.concatenate([rng.uniform(400, 400.01, 100), [800.0]]), 20.0, "ppm", 33)
    splits_case([500.0], 20.0, "ppm", 8)
    splits_case([500.0, 500.001, 600.0], 20.0, "ppm", 8)
    splits_case(np.concatenate([rng.uniform(300, 300.002, 64), rng.uniform(310, 310.002, 65),
                                rng.uniform(320, 320.002, 63), rng.uniform(330, 330.002, 10)]), 10.0, "ppm", 64)
    for i, (mz, tol, mode, batch, s) in enumerate(sp_cases):
        out[f"splits{i}_mz"] = mz
        out[f"splits{i}_par"] = np.array([tol, 1.0 if mode == "Da" else 0.0, batch], np.float64)
        out[f"splits{i}_out"] = s
    out["splits_n"] = np.int64(len(sp_cases))

    # ---- a10 _linkage / _postprocess_cluster ---------------------------------
    import scipy.cluster.hierarchy as sch
    pp = []
    def pp_case(mzv, rtv, tol, mode, rt_tol, min_samples, start_label):
        mzv = np.asarray(mzv, f32); rtv = np.asarray(rtv, f32)
        labels = np.zeros(len(mzv), np.int32)
        n = cluster._postprocess_cluster(labels, mzv, rtv, tol, mode, rt_tol, min_samples, start_label)
        pp.append((mzv, rtv, tol, mode, rt_tol, min_samples, start_label, labels.copy(), int(n)))
    pp_case([500, 500.001, 500.5, 500.5004], [10, 11, 12, 13], 20.0, "ppm", None, 2, 0)
    pp_case([500.0], [1.0], 20.0, "ppm", None, 2, 3)
    pp_case([500.0, 500.002], [1.0, 2.0], 20.0, "ppm", None, 2, 5)
    pp_case([500.0, 500.2], [1.0, 2.0], 20.0, "ppm", None, 2, 5)
    pp_case(np.concatenate([600 + rng.normal(0, 0.002, 9), 600.3 + rng.normal(0, 0.002, 5), [601.0]]),
            rng.uniform(0, 100, 15), 20.0, "ppm", None, 2, 7)
    pp_case(rng.uniform(700, 700.2, 40), rng.uniform(0, 100, 40), 0.02, "Da", None, 2, 0)
    pp_case(800 + rng.normal(0, 0.004, 30), rng.uniform(0, 600, 30), 20.0, "ppm", 60.0, 2, 11)
    pp_case(rng.uniform(450, 450.05, 25), rng.uniform(0, 50, 25), 20.0, "ppm", 5.0, 2, 0)
    pp_case(np.repeat(f32(512.25), 6), np.arange(6), 20.0, "ppm", None, 2, 2)
    for i, (mzv, rtv, tol, mode, rt_tol, ms, sl, lab, n) in enumerate(pp):
        out[f"pp{i}_mz"] = mzv; out[f"pp{i}_rt"] = rtv
        out[f"pp{i}_par"] = np.array([tol, 1.0 if mode == "Da" else 0.0,
                                      -1.0 if rt_tol is None else rt_tol, ms, sl], np.float64)
        out[f"pp{i}_labels"] = lab; out[f"pp{i}_n"] = np.int64(n)
    out["pp_n"] = np.int64(len(pp))
    # linkage matrices + flat cuts for a few 1-D arrays
    lk = [(np.array([500, 500.001, 500.5, 500.5004], f32), "ppm", 20.0),
          (rng.uniform(700, 700.2, 17).astype(f32), "Da", 0.02),
          (rng.uniform(0, 100, 12).astype(f32), None, 7.5)]
    for i, (v, mode, t) in enumerate(lk):
        Z = cluster._linkage(v, mode)
        out[f"lk{i}_v"] = v
        out[f"lk{i}_par"] = np.array([{"ppm": 0.0, "Da": 1.0, None: 2.0}[mode], t], np.float64)
        out[f"lk{i}_Z"] = np.asarray(Z, np.float64)
        out[f"lk{i}_flat"] = (sch.fcluster(Z, t, "distance") - 1).astype(np.int32)
    out["lk_n"] = np.int64(len(lk))

    # ---- _get_cluster_group_idx ----------------------------------------------
    g = np.array([-1, -1, 0, 0, 0, 1, 1, 2, 5, 5], np.int64)
    out["grp_in"] = g
    out["grp_out"] = np.array(list(cluster._get_cluster_group_idx(g)), np.int64)

    # ---- a12 _assign_global_cluster_labels ------------------------------------
    n = 40
    idx = rng.permutation(n).astype(np.int64)
    splits = [0, 7, 8, 20, 33, 40]
    lab = np.full(n, -1, np.int32)
    for a, b in zip(splits[:-1], splits[1:]):
        k = max(1, (b - a) // 3)
        local = rng.integers(-1, k, b - a)
        lab[idx[a:b]] = local
    out["gl_idx"] = idx; out["gl_splits"] = np.array(splits, np.int64); out["gl_in"] = lab.copy()
    mx = cluster._assign_global_cluster_labels(lab, idx, splits, 0)
    out["gl_out"] = lab.copy(); out["gl_max"] = np.int64(mx)

    # ---- a11 _get_cluster_medoids (condensed full matrix version) -------------
    m = 14
    pts = rng.normal(size=(m, 3))
    D = np.sqrt(((pts[:, None] - pts[None]) ** 2).sum(-1))
    iu = np.triu_indices(m, 1)
    pdist = D[iu]                                   # float64 condensed, like np.zeros default
    labels = np.array([0, 0, 0, 1, 1, 2, 2, 2, 2, 3, 4, 4, 4, 4], np.int32)
    perm = rng.permutation(m)
    labels_p = labels[perm]                         # labels by pdist position
    order_ = np.argsort(labels_p, kind="stable")
    idx_interval = (100 + np.arange(m))[order_].astype(np.int64)
    med = cluster._get_cluster_medoids(idx_interval, labels_p[order_], pdist, order_)
    out["med_pdist"] = pdist; out["med_labels_sorted"] = labels_p[order_]
    out["med_idx_interval"] = idx_interval; out["med_order_map"] = order_.astype(np.int64)
    out["med_out"] = np.asarray(med, np.int32)
    out["condensed_index_1_3_5"] = np.int64(cluster.condensed_index(1, 3, 5))

    # ---- f4 cosine_fast (in-tree exact cosine; informational / next) ----------
    def mk_spec(p):
        mz = np.sort(rng.uniform(101, 1500, p)).astype(f32)
        it = rng.lognormal(0, 1, p).astype(f32); it /= np.linalg.norm(it)
        return mz, it.astype(f32)
    cf = []
    a = mk_spec(30)
    b = (a[0] + f32(0.01), a[1])
    c = mk_spec(25)
    dmz = np.sort(np.concatenate([a[0][:15] + f32(0.03), rng.uniform(101, 1500, 12).astype(f32)])).astype(f32)
    dit = rng.lognormal(0, 1, 27).astype(f32); dit = (dit / np.linalg.norm(dit)).astype(f32)
    for x, y in [(a, b), (a, c), (a, (dmz, dit)), (c, c)]:
        s1 = similarity.SpectrumTuple(f32(500), 2, x[0], x[1])
        s2 = similarity.SpectrumTuple(f32(500), 2, y[0], y[1])
        sim, nm = similarity.cosine_fast(s1, s2, 0.05)
        cf.append((x, y, float(sim), int(nm)))
    for i, (x, y, sim, nm) in enumerate(cf):
        out[f"cf{i}_amz"], out[f"cf{i}_ait"] = x
        out[f"cf{i}_bmz"], out[f"cf{i}_bit"] = y
        out[f"cf{i}_out"] = np.array([sim, nm], np.float64)
    out["cf_n"] = np.int64(len(cf))

    np.savez_compressed(os.path.join(HERE, "reference_functions.npz"), **out)
    print("wrote reference_functions.npz with", len(out), "arrays")

    # ---- a9 DBSCAN: sklearn on sparse precomputed kNN graphs (upstream dep
    # scikit-learn, setup.cfg:34) -----------------------------------------------
    from sklearn.cluster import DBSCAN
    db = {}
    for ci, (n, k, eps) in enumerate([(60, 5, 0.1), (300, 8, 0.1), (300, 8, 0.25), (40, 3, 0.05)]):
        # clustered unit vectors -> asymmetric kNN cosine graph
        cents = rng.normal(size=(max(3, n // 12), 16))
        X = cents[rng.integers(0, len(cents), n)] + 0.08 * rng.normal(size=(n, 16))
        X[: n // 6] = rng.normal(size=(n // 6, 16))          # noise points
        X /= np.linalg.norm(X, axis=1, keepdims=True)
        S = (X @ X.T).astype(f32)
        np.fill_diagonal(S, -2)
        nb_idx = np.argsort(-S, axis=1, kind="stable")[:, :k]
        nb_dist = np.clip(1 - np.take_along_axis(S, nb_idx, 1), 0, 1).astype(f32)
        # drop a few entries to make rows ragged (-1 = empty slot)
        drop = rng.random(nb_idx.shape) < 0.15
        nb_idx = np.where(drop, -1, nb_idx).astype(np.int32)
        rows = np.repeat(np.arange(n), k)[~drop.ravel()]
        cols = nb_idx.ravel()[~drop.ravel()]
        vals = nb_dist.ravel()[~drop.ravel()]
        import scipy.sparse as ss2
        M = ss2.csr_matrix((vals, (rows, cols)), (n, n))
        import warnings
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            lab = DBSCAN(eps=eps, min_samples=2, metric="precomputed").fit_predict(M)
        db[f"db{ci}_idx"] = nb_idx; db[f"db{ci}_dist"] = nb_dist
        db[f"db{ci}_eps"] = np.float64(eps); db[f"db{ci}_labels"] = lab.astype(np.int32)
    db["db_n"] = np.int64(4)
    np.savez_compressed(os.path.join(HERE, "sklearn_dbscan.npz"), **db)
    print("wrote sklearn_dbscan.npz")

    # ---- f4 cosine_fast (similarity.py:17-80): matched-peak cosine, optimal assignment ---------------
    # pairs of spectra: jittered copies (inside / outside the tolerance), dense peak groups (several
    # candidates inside one tolerance window -> assignment components larger than 1x1), empty spectra.
    r2 = np.random.default_rng(4242)
    cf = {k: [] for k in ("a_mz", "a_it", "b_mz", "b_it")}
    a_ptr, b_ptr, tols, scores, nmatch = [0], [0], [], [], []
    for case in range(320):
        tol = float(r2.choice([0.05, 0.02, 0.5]))
        p = int(r2.integers(0, 60)) if case % 9 else int(r2.integers(0, 3))
        mz_a = np.sort(r2.uniform(101.0, 1500.0, p))
        if p > 8 and case % 3 == 0:                          # groups of peaks closer than the tolerance
            g = r2.integers(0, p - 4)
            mz_a[g:g + 4] = mz_a[g] + np.sort(r2.uniform(0, 1.5 * tol, 4))
            mz_a = np.sort(mz_a)
        keep = r2.random(p) < 0.8
        mz_b = mz_a[keep] + r2.choice([0.0, 0.3, 0.9, 1.0, 1.1, 3.0], keep.sum()) * tol * r2.choice([-1, 1], keep.sum())
        mz_b = np.sort(np.concatenate([mz_b, r2.uniform(101.0, 1500.0, int(r2.integers(0, 10)))]))
        it_a = r2.lognormal(0, 1, len(mz_a)); it_b = r2.lognormal(0, 1, len(mz_b))
        if case % 11 == 0 and len(it_a) > 3:
            it_a[:] = 1.0                                     # equal intensities: ties between assignments
        it_a = (it_a / max(np.linalg.norm(it_a), 1e-30)).astype(f32)
        it_b = (it_b / max(np.linalg.norm(it_b), 1e-30)).astype(f32)
        mz_a = mz_a.astype(f32); mz_b = mz_b.astype(f32)
        sa = similarity.SpectrumTuple(f32(500.0), 2, mz_a, it_a)
        sb = similarity.SpectrumTuple(f32(500.0), 2, mz_b, it_b)
        if len(mz_a) == 0 or len(mz_b) == 0:
            sc, nm = 0.0, 0                                   # (the reference never scores an empty spectrum)
        else:
            # (np.float64 tolerance: numba types the Python float as float64; a bare Python float would be
            #  a weak scalar under NumPy 2 and the comparisons would run in float32)
            sc, nm = similarity.cosine_fast(sa, sb, np.float64(tol))
        cf["a_mz"].append(mz_a); cf["a_it"].append(it_a); cf["b_mz"].append(mz_b); cf["b_it"].append(it_b)
        a_ptr.append(a_ptr[-1] + len(mz_a)); b_ptr.append(b_ptr[-1] + len(mz_b))
        tols.append(tol); scores.append(float(sc)); nmatch.append(int(nm))
    np.savez_compressed(os.path.join(HERE, "cosine_fast.npz"),
                        a_mz=np.concatenate(cf["a_mz"]), a_it=np.concatenate(cf["a_it"]), a_ptr=np.array(a_ptr, np.int64),
                        b_mz=np.concatenate(cf["b_mz"]), b_it=np.concatenate(cf["b_it"]), b_ptr=np.array(b_ptr, np.int64),
                        tol=np.array(tols, np.float64), score=np.array(scores, np.float64), n_match=np.array(nmatch, np.int64))
    print("wrote cosine_fast.npz")


if __name__ == "__main__":
    main()
