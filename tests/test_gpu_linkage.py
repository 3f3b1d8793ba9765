"""f4 on the GPU: hierarchical clustering of the neighbour graph (`fal_linkage_cluster`, `--clustering hierarchical`),
the clustering the reference snapshot ships (cluster.py:283-290), against the golden of that composition and the oracle."""
import os

import numpy as np
import pytest

from oracle import falcon_oracle as fo

pytestmark = pytest.mark.gpu
GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


@pytest.fixture(scope="module")
def ctx():
    from falcon_amd.device import Context
    c = Context(0)
    yield c
    c.close()


@pytest.mark.parametrize("method", ["single", "complete", "average"])
def test_linkage_matches_reference_golden(ctx, method):
    import torch
    g = np.load(os.path.join(GOLDEN, "linkage.npz"))
    for c in range(int(g["n_cases"])):
        idx, dist, t = g[f"c{c}_idx"], g[f"c{c}_dist"], float(g[f"c{c}_t"])
        lab, n_cl = ctx.linkage_cluster(torch.from_numpy(idx).to(ctx.tdev), torch.from_numpy(dist).to(ctx.tdev), t, method)
        lab = lab.cpu().numpy()
        exp = g[f"c{c}_{method}"]
        assert np.array_equal(lab, exp), (c, method, int((lab != exp).sum()))
        assert n_cl == exp.max() + 1


def test_linkage_has_no_group_size_cap(ctx):
    """round 3 refused connected groups of more than 2,048 rows (`FAL_EUNSUPPORTED`) where the reference runs
    fastcluster on whole blocks (cluster.py:277-290).  Groups beyond the one-wave form now take `lk_agglomerate_big_kernel`
    (a 16-wave workgroup, cached nearest partners: the same merges in the same order).  A chain of 3,000 rows with EQUAL
    distances inside the threshold: the tie order (lowest (a, b) first) pairs the rows up -- (0, 1), (2, 3), ... -- and a pair
    cannot take a third row (complete: one missing pair = height 1; average: (1 + 0.05) / 2)."""
    import torch
    n, k = 3000, 4
    idx = np.full((n, k), -1, np.int32)
    dist = np.full((n, k), np.inf, np.float32)
    idx[:-1, 0] = np.arange(1, n)                               # a chain 0 - 1 - 2 - ... inside the threshold
    dist[:-1, 0] = 0.05
    ti, td = torch.from_numpy(idx).to(ctx.tdev), torch.from_numpy(dist).to(ctx.tdev)
    for method in ("complete", "average"):
        lab, n_cl = ctx.linkage_cluster(ti, td, 0.1, method)
        assert n_cl == n // 2
        assert np.array_equal(lab.cpu().numpy(), np.arange(n) // 2), method
    lab, n_cl = ctx.linkage_cluster(ti, td, 0.1, "single")
    assert n_cl == 1 and bool((lab == 0).all())


@pytest.mark.parametrize("m,k", [(2500, 24), (6000, 16)])
def test_linkage_of_a_group_of_thousands_of_rows_equals_scipy(ctx, m, k):
    """one connected group of m rows next to small ones (both agglomeration kernels in one call), distinct heights (random
    geometry), so scipy's dendrogram cut at t is THE answer: labels must be identical, not just ARI-close"""
    import torch
    rng = np.random.default_rng(m)
    # a long noisy curve: neighbours along the curve are within the threshold, the group is one component of m rows
    s = np.arange(m) * 0.006
    pos = np.stack([s, 0.008 * rng.normal(size=m)], 1)
    small = np.concatenate([rng.normal(size=(30, 2)) * 0.01 + np.array([0.0, 50.0 + 5 * g]) for g in range(20)])
    pos = np.concatenate([pos, small])
    n = len(pos)
    nb_idx = np.full((n, k), -1, np.int32)
    nb_dist = np.full((n, k), np.inf, np.float32)
    for i0 in range(0, n, 1000):
        blk = np.sqrt(((pos[i0:i0 + 1000, None] - pos[None]) ** 2).sum(-1)).astype(np.float32)
        for r in range(len(blk)):
            i = i0 + r
            o = np.argpartition(blk[r], k + 1)[:k + 1]
            o = o[np.argsort(blk[r][o], kind="stable")]
            o = o[o != i][:k]
            nb_idx[i, :len(o)] = o
            nb_dist[i, :len(o)] = np.clip(blk[r][o], 0, 0.99)
    ti, td = torch.from_numpy(nb_idx).to(ctx.tdev), torch.from_numpy(nb_dist).to(ctx.tdev)
    for method, t in (("complete", 0.05), ("average", 0.04)):
        ref = fo.linkage_clusters(nb_idx, nb_dist, t, method)
        lab, n_cl = ctx.linkage_cluster(ti, td, t, method)
        lab = lab.cpu().numpy()
        assert np.array_equal(lab, ref), (method, int((lab != ref).sum()))
        assert n_cl == ref.max() + 1
        assert np.bincount(ref[:m][ref[:m] >= 0]).max() < 200          # (the big group is cut into many small clusters)


def test_linkage_large_component_and_ties(ctx):
    """a component of several hundred rows (the global-memory matrix) and exact zero distances (duplicates)"""
    import torch
    rng = np.random.default_rng(7)
    n, k = 900, 24
    pos = np.concatenate([rng.normal(size=(600, 2)) * 0.05, rng.normal(size=(300, 2)) * 0.05 + 5.0])
    pos[100:140] = pos[100]                                            # 40 identical points: zero distances
    d = np.sqrt(((pos[:, None] - pos[None]) ** 2).sum(-1)).astype(np.float32)
    d = np.clip(d, 0, 0.99)
    nb_idx = np.full((n, k), -1, np.int32)
    nb_dist = np.full((n, k), np.inf, np.float32)
    for i in range(n):
        o = np.argsort(d[i], kind="stable")
        o = o[o != i][:k]
        nb_idx[i, :len(o)] = o
        nb_dist[i, :len(o)] = d[i, o]
    for method, t in (("single", 0.03), ("complete", 0.06), ("average", 0.05)):
        lab, n_cl = ctx.linkage_cluster(torch.from_numpy(nb_idx).to(ctx.tdev), torch.from_numpy(nb_dist).to(ctx.tdev), t, method)
        lab = lab.cpu().numpy()
        ref = fo.linkage_clusters(nb_idx, nb_dist, t, method)
        from sklearn.metrics import adjusted_rand_score
        # identical partitions; where equal heights leave the merge order open (the duplicates) ARI guards the claim
        assert adjusted_rand_score(ref, lab) >= 0.999, method
        assert np.array_equal(lab == -1, ref == -1)
        if method == "single":                                          # (complete / average: the duplicates that no list
            assert len(np.unique(lab[100:140])) == 1 and lab[100] >= 0     #  connects are a missing pair = distance 1 apart)


def test_pipeline_hierarchical_clustering_matches_oracle(ctx):
    """`generate_clusters(..., linkage, ann=AnnParams(clustering="hierarchical"))`: re-scored graph -> linkage -> the same
    refinement / medoids / labels as the DBSCAN path, against the oracle's run of the same stages."""
    from falcon_amd import synth
    from falcon_amd.cluster.cluster import AnnParams, SpectrumDataset, generate_clusters, ClusterPipeline
    d = synth.select_charge(synth.generate(5000, seed=11, mz_lo=500.0, mz_hi=520.0), 2)
    ds = SpectrumDataset(d["precursor_mz"], d["retention_time"], d["mz"], d["intensity"], d["indptr"])
    pipe = ClusterPipeline(ctx)
    for method in ("complete", "average", "single"):
        ann = AnnParams(eps=0.35, clustering="hierarchical")
        labels, medoids = generate_clusters(ds, method, 0.35, 4, 20.0, "ppm", None, 0.05, 2 ** 15, ann=ann, pipeline=pipe)
        ref, rmed = fo.generate_clusters(d["mz"], d["intensity"], d["indptr"], d["precursor_mz"], d["retention_time"], eps=0.35,
                                         clustering="hierarchical", linkage=method, min_matches=4)
        assert np.array_equal(labels, ref), method
        assert np.array_equal(medoids, rmed), method
        assert (np.bincount(labels) > 1).sum() > 50
