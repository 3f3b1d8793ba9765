"""Pin the CPU oracle against golden vectors produced by the reference's own
function bodies (tests/golden/make_golden.py) and by scikit-learn."""
import os

import numpy as np
import pytest

from oracle import falcon_oracle as fo

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def test_get_dim(ref_golden):
    g = ref_golden
    for (lo, hi, b), dim, (s, e) in zip(g["get_dim_in"], g["get_dim_dim"], g["get_dim_start_end"]):
        d, start, end = fo.get_dim(lo, hi, b)
        assert d == dim
        assert np.float32(start) == s and np.float32(end) == e
    # SURVEY 8(a) a1 probe
    assert fo.get_dim(101, 1500, 0.05)[0] == 27982


def test_bin_indices(ref_golden):
    g = ref_golden
    idx = fo.bin_indices(g["tv_mz"], float(g["tv_min_mz"]), float(g["tv_bin_size"]))
    assert np.array_equal(idx, g["tv_indices"])


def test_murmurhash(ref_golden):
    g = ref_golden
    assert np.array_equal(fo.murmurhash3_32(g["mmh3_keys"], 0), g["mmh3_seed0"])
    assert np.array_equal(fo.murmurhash3_32(g["mmh3_keys"], 42), g["mmh3_seed42"])
    # SURVEY 8(c) known answers
    assert list(fo.murmurhash3_32(np.arange(5), 0)) == [593689054, 4226891818, 1085422463, 847579505, 1889779975]
    assert list(fo.hash_lookup(10, 400)) == [254, 218, 63, 305, 375, 94, 302, 321, 321, 131]
    assert np.array_equal(fo.hash_lookup(27982, 400), g["tv_hash_lookup_400"])


def test_vectorize_matches_reference_projection(ref_golden):
    """reference to_vector(vectors @ hash-projection), norm=False (spectrum.py:240-243)."""
    g = ref_golden
    V = fo.vectorize(g["tv_mz"], g["tv_intensity"], g["tv_indptr"], float(g["tv_min_mz"]),
                     float(g["tv_bin_size"]), 27982, 400, norm=False)
    ref = g["tv_vectors_unnorm_400"]
    # scipy's CSR product may add colliding peaks in another order: 1 ulp slack
    np.testing.assert_allclose(V, ref, rtol=3e-7, atol=0)
    assert np.array_equal(V != 0, ref != 0)


def test_vectorize_norm_and_edge_cases(ref_golden):
    g = ref_golden
    V = fo.vectorize(g["tv_mz"], g["tv_intensity"], g["tv_indptr"], float(g["tv_min_mz"]),
                     float(g["tv_bin_size"]), 27982, 400)
    n = np.linalg.norm(V.astype(np.float64), axis=1)
    empty = np.diff(g["tv_indptr"]) == 0
    assert empty.any()
    assert np.all(V[empty] == 0)
    np.testing.assert_allclose(n[~empty], 1.0, atol=2e-7)
    # out-of-range peaks are ignored
    V2 = fo.vectorize(np.array([50.0, 200.0, 5000.0], np.float32), np.ones(3, np.float32),
                      np.array([0, 3]), float(g["tv_min_mz"]), 0.05, 27982, 400)
    assert (V2 != 0).sum() == 1 and V2.max() == 1.0


def test_norm_intensity(ref_golden):
    g = ref_golden
    x = g["norm_in"]
    np.testing.assert_allclose(fo.l2_normalize_rows(x[None, :])[0], g["norm_out"], rtol=2e-7)


def test_precursor_splits(ref_golden):
    g = ref_golden
    for i in range(int(g["splits_n"])):
        tol, is_da, batch = g[f"splits{i}_par"]
        s = fo.get_precursor_mz_splits(g[f"splits{i}_mz"], tol, "Da" if is_da else "ppm", int(batch))
        assert np.array_equal(s, g[f"splits{i}_out"]), i


def test_linkage_and_flat(ref_golden):
    g = ref_golden
    for i in range(int(g["lk_n"])):
        mode = {0.0: "ppm", 1.0: "Da", 2.0: None}[float(g[f"lk{i}_par"][0])]
        t = float(g[f"lk{i}_par"][1])
        Z = fo.linkage_1d(g[f"lk{i}_v"], mode)
        ref = g[f"lk{i}_Z"]
        assert np.array_equal(Z[:, [0, 1, 3]], ref[:, [0, 1, 3]])
        np.testing.assert_allclose(Z[:, 2], ref[:, 2], rtol=1e-6)     # f32 vs f64 ppm product
        flat = g[f"lk{i}_flat"]
        assert np.array_equal(fo.fcluster_numbering(Z, t), flat)
        # partition equality of the closed form
        mine = fo.flat_1d(g[f"lk{i}_v"], t, mode)
        assert _same_partition(mine, flat)


def _same_partition(a, b):
    m = {}
    for x, y in zip(a, b):
        if m.setdefault(int(x), int(y)) != int(y):
            return False
    return len(set(m.values())) == len(m)


def test_postprocess_cluster(ref_golden):
    g = ref_golden
    for i in range(int(g["pp_n"])):
        tol, is_da, rt_tol, ms, sl = g[f"pp{i}_par"]
        lab = np.zeros(len(g[f"pp{i}_mz"]), np.int32)
        n = fo.postprocess_cluster(lab, g[f"pp{i}_mz"], g[f"pp{i}_rt"], tol, "Da" if is_da else "ppm",
                                   None if rt_tol < 0 else rt_tol, int(ms), int(sl))
        assert n == int(g[f"pp{i}_n"]), i
        assert np.array_equal(lab, g[f"pp{i}_labels"]), i


def test_group_idx_and_global_labels(ref_golden):
    g = ref_golden
    assert np.array_equal(np.array(list(fo.cluster_group_idx(g["grp_in"]))), g["grp_out"])
    lab = g["gl_in"].copy()
    mx = fo.assign_global_cluster_labels(lab, g["gl_idx"], list(g["gl_splits"]), 0)
    assert mx == int(g["gl_max"]) and np.array_equal(lab, g["gl_out"])


def test_medoids_dense(ref_golden):
    g = ref_golden
    med = fo.medoids_dense(g["med_idx_interval"], g["med_labels_sorted"], g["med_pdist"], g["med_order_map"])
    assert np.array_equal(med, g["med_out"])


def test_dbscan_matches_sklearn(dbscan_golden):
    g = dbscan_golden
    from sklearn.metrics import adjusted_rand_score
    for i in range(int(g["db_n"])):
        idx, dist, eps, ref = g[f"db{i}_idx"], g[f"db{i}_dist"], float(g[f"db{i}_eps"]), g[f"db{i}_labels"]
        lab = fo.dbscan_sklearn_order(idx, dist, eps)
        assert np.array_equal(lab, ref), i
        comp = fo.dbscan_components(idx, dist, eps)
        # The order-independent variant the HIP path implements: identical noise set,
        # and every sklearn cluster lies inside ONE component cluster (sklearn's
        # visiting order can only fragment a density-connected component, never
        # join two).  With k >= 8 neighbours the two agree exactly on these graphs.
        assert np.array_equal(ref == -1, comp == -1), i
        n = len(ref)
        rows = np.arange(n)[:, None]
        core = ((idx >= 0) & (dist <= np.float32(eps)) & (idx != rows)).any(1)
        for c in np.unique(ref[ref >= 0]):
            assert len(np.unique(comp[(ref == c) & core])) == 1, (i, c)
        if idx.shape[1] >= 8:
            assert adjusted_rand_score(ref, comp) >= 0.99, i


def test_cosine_fast_matches_reference_golden():
    """f4: the oracle's matched-peak cosine == the reference's `cosine_fast` (similarity.py:17-80) on 320 pairs
    (jittered copies, peak groups denser than the tolerance, equal intensities, empty spectra)."""
    import os
    g = np.load(os.path.join(os.path.dirname(__file__), "golden", "cosine_fast.npz"))
    n = len(g["tol"])
    assert n == 320 and (g["n_match"] > 0).sum() > 250
    for c in range(n):
        a0, a1, b0, b1 = g["a_ptr"][c], g["a_ptr"][c + 1], g["b_ptr"][c], g["b_ptr"][c + 1]
        sc, nm = fo.cosine_fast(g["a_mz"][a0:a1], g["a_it"][a0:a1], g["b_mz"][b0:b1], g["b_it"][b0:b1], float(g["tol"][c]))
        assert nm == int(g["n_match"][c]), c
        # same matched pairs; the golden was produced without numba, where NumPy 2 keeps the running sum in
        # float32 (`0.0 += np.float32`), while numba -- and the oracle -- accumulate in float64: last-bit slack
        assert abs(sc - float(g["score"][c])) <= 4e-7, c


def test_linkage_clusters_match_the_reference_composition():
    """f4: the per-component restatement equals fcluster(linkage(dense pdist, method), t, "distance") on the whole graph
    (tests/golden/make_linkage_golden.py: reference cluster.py:283-290 with scipy's linkage for fastcluster)."""
    g = np.load(os.path.join(GOLDEN, "linkage.npz"))
    for c in range(int(g["n_cases"])):
        for method in ("single", "complete", "average"):
            lab = fo.linkage_clusters(g[f"c{c}_idx"], g[f"c{c}_dist"], float(g[f"c{c}_t"]), method)
            assert np.array_equal(lab, g[f"c{c}_{method}"]), (c, method)
