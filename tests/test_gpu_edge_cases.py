"""Edge cases of the hot path on the GPU (ragged / degenerate inputs the reference's data can produce)."""
import numpy as np
import pytest

from oracle import falcon_oracle as fo
from tests.util import assert_topk_close, assert_topk_exact

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ctx():
    from falcon_amd.device import Context
    c = Context(0)
    yield c
    c.close()


def _unit(n, d, seed):
    rng = np.random.default_rng(seed)
    X = np.abs(rng.normal(size=(n, d))) * (rng.random((n, d)) < 0.1)
    X[:, 0] += 1e-3
    return (X / np.linalg.norm(X, axis=1, keepdims=True)).astype(np.float32)


def test_k_larger_than_bucket_and_probe_larger_than_lists(ctx):
    """k_ann > bucket size pads with (-inf, -1); n_probe > n_list probes every list."""
    import torch
    sizes, nlists = [3, 40, 600, 150], [1, 1, 4, 2]
    off = np.concatenate([[0], np.cumsum(sizes)])
    X = _unit(off[-1], 400, 1)
    idxr = ctx.ivf_build(torch.from_numpy(X).to(ctx.tdev), off, np.array(nlists, np.int32), kmeans_iters=2)
    sim, idx = idxr.search(64, 200)            # n_probe 64 > every n_list, k 200 > some buckets
    sim, idx = sim.cpu().numpy(), idx.cpu().numpy()
    for a, b in zip(off[:-1], off[1:]):
        rs, ri = fo.exhaustive_topk(X[a:b], 200, base=a)
        assert_topk_exact(sim[a:b], idx[a:b], rs, ri)
    assert (idx[:3] >= 0).sum() == 9 and np.all(np.isneginf(sim[:3, 3:]))


def test_zero_vectors_and_duplicates_in_ivf_bucket(ctx):
    """all-zero rows (empty spectra) and exact duplicates inside an IVF bucket: search still equals
    the oracle on the same index; duplicates tie-break by ascending id."""
    import torch
    n = 3000
    X = _unit(n, 400, 2)
    X[100:140] = 0
    X[500:560] = X[500]
    off = np.array([0, n])
    idxr = ctx.ivf_build(torch.from_numpy(X).to(ctx.tdev), off, np.array([32], np.int32), kmeans_iters=3)
    cent, asg, perm, loff = [t.cpu().numpy() for t in idxr.export()]
    sim, idx = idxr.search(8, 64)
    sim, idx = sim.cpu().numpy(), idx.cpu().numpy()
    rs, ri = fo.ivf_search(X, cent, asg, perm, loff, 8, 64)
    assert_topk_exact(sim, idx, rs, ri)
    assert np.array_equal(idx[500, :60], np.arange(500, 560))          # exact ties -> ascending id
    assert np.all(np.isfinite(sim[100:140][idx[100:140] >= 0]))


def test_giant_tie_cluster_through_the_tail(ctx):
    """thousands of identical spectra: one DBSCAN cluster far larger than n_neighbors; the refine
    kernel's O(m^2/64) paths, the medoid tie-break and the label contract must hold."""
    from falcon_amd.cluster.cluster import AnnParams, ClusterPipeline, SpectrumDataset
    rng = np.random.default_rng(3)
    m = 2500
    mz = np.sort(rng.uniform(150, 1400, 40)).astype(np.float32)
    it = rng.random(40).astype(np.float32)
    it /= np.linalg.norm(it)
    other_n = 300
    omz = np.sort(rng.uniform(150, 1400, (other_n, 40)), axis=1).astype(np.float32)
    oit = rng.random((other_n, 40)).astype(np.float32)
    oit /= np.linalg.norm(oit, axis=1, keepdims=True)
    all_mz = np.concatenate([np.tile(mz, m), omz.ravel()])
    all_it = np.concatenate([np.tile(it, m), oit.ravel()])
    indptr = np.arange(0, (m + other_n) * 40 + 1, 40, dtype=np.int64)
    pmz = np.concatenate([np.full(m, 500.25, np.float32) + rng.normal(0, 5e-4, m).astype(np.float32),
                          rng.uniform(400, 600, other_n).astype(np.float32)])
    rt = rng.uniform(0, 100, m + other_n).astype(np.float32)
    ds = SpectrumDataset(pmz, rt, all_mz, all_it, indptr)
    labels, medoids = ClusterPipeline(ctx).run(ds, 20.0, "ppm", None, 0.05, 2 ** 15, AnnParams())
    labels, medoids = labels.cpu().numpy(), medoids.cpu().numpy()
    big = labels[:m]
    # 20 ppm complete linkage splits the jittered precursors into a few groups; each is one cluster
    assert len(np.unique(big)) <= 12 and np.bincount(big).max() >= m // 12
    assert np.array_equal(labels[medoids], np.arange(len(medoids)))
    assert labels.min() == 0 and labels.max() == len(medoids) - 1
    ref, _ = fo.generate_clusters(all_mz, all_it, indptr, pmz, rt)
    from sklearn.metrics import adjusted_rand_score
    assert adjusted_rand_score(ref, labels) >= 0.99


def test_rt_and_da_tolerance_pipeline(ctx):
    from falcon_amd import synth
    from falcon_amd.cluster.cluster import AnnParams, ClusterPipeline, SpectrumDataset
    d = synth.select_charge(synth.generate(6000, seed=31, mz_lo=600, mz_hi=640), 2)
    ds = SpectrumDataset(d["precursor_mz"], d["retention_time"], d["mz"], d["intensity"], d["indptr"])
    p = AnnParams(eps=0.35, n_neighbors=32, n_neighbors_ann=64)
    labels, medoids = ClusterPipeline(ctx).run(ds, 0.01, "Da", 1200.0, 0.05, 2 ** 15, p)
    ref, _ = fo.generate_clusters(d["mz"], d["intensity"], d["indptr"], d["precursor_mz"], d["retention_time"],
                                  eps=0.35, precursor_tol=(0.01, "Da"), rt_tol=1200.0, n_neighbors=32, n_neighbors_ann=64)
    assert np.array_equal(labels.cpu().numpy(), ref)


def test_bad_arguments_are_reported_not_fatal(ctx):
    import torch
    from falcon_amd._lib import FalconHipError
    X = torch.zeros((10, 400), device=ctx.tdev)
    with pytest.raises(FalconHipError):
        ctx.ivf_build(X, np.array([0, 5]), np.array([1], np.int32))          # bucket_off does not end at n
    with pytest.raises(FalconHipError):
        ctx.ivf_build(X, np.array([0, 10]), np.array([64], np.int32))        # more lists than rows
    idxr = ctx.ivf_build(X, np.array([0, 10]), np.array([1], np.int32))
    with pytest.raises(FalconHipError):
        idxr.search(16, 1000)                                                # k_ann beyond FAL_MAX_K_ANN
    with pytest.raises(FalconHipError):
        ctx.vectorize(np.zeros(4, np.float32), np.zeros(4, np.float32), np.array([0, 4]), None, 100.0, 0.05, 100, 402)


def test_dbscan_ignores_neighbour_ids_outside_the_dataset(ctx):
    """neighbour ids >= n (an uninitialised or corrupted neighbour array) are treated as "no neighbour" by the graph stage:
    they once sent the union-find through foreign memory (a GPU box hung for its whole time limit)"""
    import torch
    n, k = 5000, 8
    rng = np.random.default_rng(3)
    idx = np.full((n, k), -1, np.int32)
    dist = np.full((n, k), np.inf, np.float32)
    idx[:-1, 0] = np.arange(1, n)
    dist[:-1, 0] = 0.05
    idx[::7, 1] = rng.integers(n, 2 ** 31 - 1, size=len(idx[::7]))      # garbage ids with small distances
    dist[::7, 1] = 0.01
    ref_idx = idx.copy()
    ref_idx[::7, 1] = -1
    ti, td = torch.from_numpy(idx).to(ctx.tdev), torch.from_numpy(dist).to(ctx.tdev)
    tr = torch.from_numpy(ref_idx).to(ctx.tdev)
    lab, n_cl = ctx.dbscan(ti, td, 0.1)
    lab_ref, n_ref = ctx.dbscan(tr, td, 0.1)
    assert n_cl == n_ref and torch.equal(lab, lab_ref)
    lab, n_cl = ctx.linkage_cluster(ti, td, 0.1, "single")
    lab_ref, n_ref = ctx.linkage_cluster(tr, td, 0.1, "single")
    assert n_cl == n_ref and torch.equal(lab, lab_ref)


def test_plan_and_trim_bound_the_scratch_and_keep_the_results(ctx):
    """`fal_ctx_plan` (code objects + shape-dependent scratch before the first pass) and `fal_ctx_trim` (cached device memory back
    to the driver): the results do not change, trim returns what the passes had grown, a pass after trim runs from empty pools."""
    import torch
    from falcon_amd import synth
    from falcon_amd.cluster.cluster import AnnParams, ClusterPipeline, PartitionRunner, SpectrumDataset
    from oracle import falcon_oracle as fo
    d = synth.select_charge(synth.generate(30000, seed=17, mz_lo=600.0, mz_hi=612.0), 2)         # flat and indexed buckets
    ds = SpectrumDataset(d["precursor_mz"], d["retention_time"], d["mz"], d["intensity"], d["indptr"])
    ref, rmed = fo.generate_clusters(d["mz"], d["intensity"], d["indptr"], d["precursor_mz"], d["retention_time"])
    pipe = ClusterPipeline(device=ctx.device)
    pipe.ctx.plan(0)                                             # (code objects only: what falcon.main() does at start)
    pipe.plan(len(ds), 2 ** 15, AnnParams())
    lab, med = pipe.run(ds, 20.0, "ppm", None, 0.05, 2 ** 15, AnnParams())
    assert np.array_equal(lab.cpu().numpy(), ref) and np.array_equal(med.cpu().numpy(), rmed)
    del lab, med
    pipe.last = {}
    torch.cuda.synchronize()
    free_before = torch.cuda.mem_get_info()[0]
    pipe.trim()
    free_after = torch.cuda.mem_get_info()[0]
    assert free_after - free_before > 50e6, (free_before, free_after)        # the flat scan's hand-off alone is larger
    lab, med = pipe.run(ds, 20.0, "ppm", None, 0.05, 2 ** 15, AnnParams())
    assert np.array_equal(lab.cpu().numpy(), ref) and np.array_equal(med.cpu().numpy(), rmed)
    runner = PartitionRunner(ctx.device, 2)
    try:
        args = (20.0, "ppm", None, 0.05, 2 ** 15, AnnParams())
        runner.plan([ds], *args)
        out = runner.run([ds], *args)
        assert np.array_equal(out[0][0].cpu().numpy(), ref)
        runner.trim()
        out = runner.run([ds], *args)
        assert np.array_equal(out[0][0].cpu().numpy(), ref)
    finally:
        runner.close()
