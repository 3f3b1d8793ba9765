"""GPU parity of the whole hot path, stage by stage and end to end.

Each stage is compared with the oracle run on the SAME inputs (the GPU's own upstream
output), so a discrepancy is pinned to one kernel: integer / index work must be
bit-exact, float entries within north_star's 1e-5.  The end-to-end check is the
north_star gate: ARI >= 0.99 against the oracle's labels."""
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


def _dataset(n, seed=42, charge=2, **kw):
    from falcon_amd import synth
    from falcon_amd.cluster.cluster import SpectrumDataset
    d = synth.select_charge(synth.generate(n, seed=seed, **kw), charge)
    return d, SpectrumDataset(d["precursor_mz"], d["retention_time"], d["mz"], d["intensity"], d["indptr"])


def _check_stages(ctx, d, ds, tol, mode, rt_tol, batch_size, p, pipe=None):
    from falcon_amd.cluster.cluster import ClusterPipeline
    pipe = pipe or ClusterPipeline(ctx)
    labels, medoids = pipe.run(ds, tol, mode, rt_tol, 0.05, batch_size, p, keep_intermediates=True)
    L = {k: (v.cpu().numpy() if hasattr(v, "cpu") else v) for k, v in pipe.last.items() if k != "index"}
    labels, medoids = labels.cpu().numpy(), medoids.cpu().numpy()
    N = len(ds)
    # sort (cluster.py:73-85) -- stable
    order = np.argsort(d["precursor_mz"], kind="stable")
    assert np.array_equal(L["order"], order)
    mzs = d["precursor_mz"][order]
    assert np.array_equal(L["mz_sorted"], mzs)
    rts = d["retention_time"][order] if rt_tol is not None else None
    # a5 buckets
    assert np.array_equal(L["splits"], fo.bucket_splits(mzs, tol, mode, batch_size, p.mz_interval))
    # a1-a3 vectors, bit-exact
    nb, start, _ = fo.get_dim(p.min_mz, p.max_mz, 0.05)
    X = fo.vectorize(d["mz"], d["intensity"], d["indptr"], start, 0.05, nb, p.low_dim, p.hash_seed, True, order,
                     width=fo.row_width(p.low_dim))           # (rows are stored row_width(low_dim) columns wide, zero behind low_dim)
    assert np.array_equal(L["X"], X)
    # a6/a7 on the GPU's own index
    cent, asg, perm, loff = [t.cpu().numpy() for t in pipe.last["index"].export()]
    lb = np.concatenate([[0], np.cumsum(L["n_list"])])
    for b, (a, e) in enumerate(zip(L["splits"][:-1], L["splits"][1:])):
        if e - a == 0:
            continue
        rs, ri = fo.ivf_search(X[a:e], cent[lb[b]:lb[b + 1]], asg[a:e], perm[a:e] - a,
                               loff[lb[b]:lb[b + 1] + 1] - a, p.n_probe, p.n_neighbors_ann, base=a)
        assert_topk_exact(L["sim"][a:e], L["idx"][a:e], rs, ri, what=f"bucket {b}")
    # a8 filter: same inputs -> bit-exact
    ni, nd = fo.filter_neighbors(L["sim"], L["idx"], mzs, rts, tol, mode, rt_tol, p.n_neighbors)
    assert np.array_equal(L["nb_idx"], ni)
    assert np.array_equal(L["nb_dist"], nd)
    # a9 DBSCAN: bit-exact vs the order-independent restatement, ARI vs sklearn's order
    db = fo.dbscan_components(ni, nd, p.eps)
    assert np.array_equal(L["db"], db)
    assert L["n_db"] == db.max() + 1
    from sklearn.metrics import adjusted_rand_score
    import warnings
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        assert adjusted_rand_score(fo.dbscan_sklearn_order(ni, nd, p.eps), db) >= 0.99
    # a10 refinement + numbering
    lab = fo.refine_and_number(db, None, mzs, rts, tol, mode, rt_tol)
    assert np.array_equal(L["lab_sorted"], lab)
    n_cl = int(lab.max()) + 1
    assert L["n_clusters"] == n_cl
    # a11 medoids + a12 labels
    member = lab >= 0
    safe = np.where(member, lab, 0)
    score = fo.medoid_scores_sparse(lab, ni, nd)
    o = np.lexsort((np.arange(N), score, safe))
    o = o[member[o]]
    first = np.concatenate([[True], safe[o][1:] != safe[o][:-1]]) if len(o) else np.zeros(0, bool)
    ref_labels = np.empty(N, np.int32)
    ref_labels[order] = lab
    noise = ref_labels == -1
    ref_labels[noise] = np.arange(n_cl, n_cl + noise.sum())
    ref_med = np.concatenate([order[o[first]], np.flatnonzero(noise)]).astype(np.int32)
    assert np.array_equal(labels, ref_labels)
    assert np.array_equal(medoids, ref_med)
    # contract of the seam (cluster.py:152-156): no -1 left, labels dense, medoid belongs to its cluster
    assert labels.min() == 0 and labels.max() == len(medoids) - 1
    assert np.array_equal(labels[medoids], np.arange(len(medoids)))
    return labels, medoids


def test_stages_flat_buckets(ctx):
    """default options: 1 m/z windows -> small flat buckets (the C2 regime of SURVEY 8d)."""
    from falcon_amd.cluster.cluster import AnnParams
    d, ds = _dataset(12000)
    labels, _ = _check_stages(ctx, d, ds, 20.0, "ppm", None, 2 ** 15, AnnParams())
    # end to end vs the oracle's own run
    ref, _ = fo.generate_clusters(d["mz"], d["intensity"], d["indptr"], d["precursor_mz"], d["retention_time"])
    from sklearn.metrics import adjusted_rand_score
    import warnings
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        assert adjusted_rand_score(ref, labels) >= 0.99
    assert np.array_equal(ref, labels)


def test_stages_ivf_buckets_and_rt(ctx):
    """no windows + small batch_size -> real IVF buckets (n_list > n_probe); RT tolerance on;
    clean synthetic spectra so that clusters form and get split by m/z and RT."""
    from falcon_amd.cluster.cluster import AnnParams
    d, ds = _dataset(16000, seed=5, mz_lo=500.0, mz_hi=520.0)
    p = AnnParams(n_probe=4, n_neighbors=16, n_neighbors_ann=48, mz_interval=0.0, kmeans_iters=3, eps=0.3)
    labels, _ = _check_stages(ctx, d, ds, 20.0, "ppm", 900.0, 4096, p)
    ref, _ = fo.generate_clusters(d["mz"], d["intensity"], d["indptr"], d["precursor_mz"], d["retention_time"],
                                  eps=0.3, rt_tol=900.0, batch_size=4096, n_probe=4, n_neighbors=16,
                                  n_neighbors_ann=48, mz_interval=0.0, kmeans_iters=3)
    assert np.array_equal(ref, labels)      # same index, same sums: the oracle's own run gives the same labels


def test_stages_da_tolerance_low_dim(ctx):
    from falcon_amd.cluster.cluster import AnnParams
    d, ds = _dataset(5000, seed=9, charge=3)
    _check_stages(ctx, d, ds, 0.05, "Da", None, 512, AnnParams(low_dim=64, n_neighbors=8, n_neighbors_ann=20, eps=0.4))


def test_generate_clusters_seam(ctx):
    """the drop-in call itself (reference signature) incl. empty and single-spectrum datasets."""
    from falcon_amd.cluster.cluster import generate_clusters, SpectrumDataset, ClusterPipeline
    pipe = ClusterPipeline(ctx)
    d, ds = _dataset(3000, seed=3)
    labels, medoids = generate_clusters(ds, "complete", 0.1, 0, 20.0, "ppm", None, 0.05, 2 ** 15, pipeline=pipe)
    assert labels.dtype == np.int32 and labels.shape == (len(ds),) and medoids.dtype == np.int32
    ref, rmed, im = fo.generate_clusters(d["mz"], d["intensity"], d["indptr"], d["precursor_mz"],
                                         d["retention_time"], return_intermediates=True)
    assert np.array_equal(labels, ref)
    # medoids: identical -- the oracle's inner products are the kernels' (same k order, symmetric bit for bit)
    assert np.array_equal(labels[medoids], np.arange(len(medoids)))
    assert np.array_equal(medoids, rmed)
    one = SpectrumDataset(d["precursor_mz"][:1], d["retention_time"][:1], d["mz"][:d["indptr"][1]],
                          d["intensity"][:d["indptr"][1]], d["indptr"][:2])
    l1, m1 = generate_clusters(one, "complete", 0.1, 0, 20.0, "ppm", None, 0.05, 2 ** 15, pipeline=pipe)
    assert list(l1) == [0] and list(m1) == [0]
    empty = SpectrumDataset(np.zeros(0, np.float32), np.zeros(0, np.float32), np.zeros(0, np.float32),
                            np.zeros(0, np.float32), np.zeros(1, np.int64))
    l0, m0 = generate_clusters(empty, "complete", 0.1, 0, 20.0, "ppm", None, 0.05, 2 ** 15, pipeline=pipe)
    assert len(l0) == 0 and len(m0) == 0


def test_precursor_splits_golden(ctx, ref_golden):
    """a5 against the reference's own `_get_precursor_mz_splits` (both extra rules off)."""
    g = ref_golden
    for i in range(int(g["splits_n"])):
        tol, is_da, batch = g[f"splits{i}_par"]
        s = ctx.precursor_splits(g[f"splits{i}_mz"], tol, "Da" if is_da else "ppm", int(batch), 0.0, False)
        assert np.array_equal(s, g[f"splits{i}_out"]), i


def test_refine_golden(ctx, ref_golden):
    """a10 against the reference's own `_postprocess_cluster` (one DBSCAN cluster each)."""
    import torch
    g = ref_golden
    for i in range(int(g["pp_n"])):
        tol, is_da, rt_tol, ms, sl = g[f"pp{i}_par"]
        mz, rt = g[f"pp{i}_mz"], g[f"pp{i}_rt"]
        m = len(mz)
        lab = torch.zeros(m, dtype=torch.int32, device=ctx.tdev)
        out, n = ctx.refine_clusters(lab, 1, ctx.to_dev(mz, torch.float32), ctx.to_dev(rt, torch.float32), tol,
                                     "Da" if is_da else "ppm", None if rt_tol < 0 else rt_tol)
        exp = g[f"pp{i}_labels"].astype(np.int64)
        exp = np.where(exp >= 0, exp - int(sl), -1)
        assert n == int(g[f"pp{i}_n"]), i
        assert np.array_equal(out.cpu().numpy(), exp), i


def test_dbscan_golden(ctx, dbscan_golden):
    """a9 against scikit-learn's DBSCAN on sparse precomputed graphs: identical noise set,
    every sklearn cluster inside one GPU cluster, exact agreement for k >= 8."""
    g = dbscan_golden
    import torch
    for i in range(int(g["db_n"])):
        idx, dist, eps, ref = g[f"db{i}_idx"], g[f"db{i}_dist"], float(g[f"db{i}_eps"]), g[f"db{i}_labels"]
        lab, nc = ctx.dbscan(ctx.to_dev(idx, torch.int32), ctx.to_dev(dist, torch.float32), eps)
        lab = lab.cpu().numpy()
        assert np.array_equal(lab, fo.dbscan_components(idx, dist, eps))
        assert np.array_equal(lab == -1, ref == -1)
        if idx.shape[1] >= 8:
            m = {}
            assert all(m.setdefault(int(a), int(b)) == int(b) for a, b in zip(ref, lab))


@pytest.mark.parametrize("mode", ["f16x3", "f16"])
def test_pipeline_f16_scan_modes(ctx, mode):
    """whole path with the f16-MFMA flat scan: "f16x3" (float32 vectors, hi/lo split) must reproduce
    the float32 oracle's clustering (ARI >= 0.99, north_star); "f16" (config 5: low_dim = 800 float16
    vectors) is compared with the oracle run on float16 vectors."""
    from falcon_amd.cluster.cluster import AnnParams, ClusterPipeline
    from sklearn.metrics import adjusted_rand_score
    import warnings
    d, ds = _dataset(8000, seed=13)
    if mode == "f16x3":
        p, kw = AnnParams(scan="f16x3"), {}
    else:
        p, kw = AnnParams(dtype="f16", low_dim=800), dict(low_dim=800, dtype=np.float16)
    labels, medoids = ClusterPipeline(ctx).run(ds, 20.0, "ppm", None, 0.05, 2 ** 15, p)
    labels, medoids = labels.cpu().numpy(), medoids.cpu().numpy()
    ref, _ = fo.generate_clusters(d["mz"], d["intensity"], d["indptr"], d["precursor_mz"], d["retention_time"], **kw)
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        assert adjusted_rand_score(ref, labels) >= 0.99
    assert np.array_equal(labels[medoids], np.arange(len(medoids)))


def test_run_many_equals_run_per_partition(ctx):
    """the software-pipelined multi-partition entry gives exactly the per-partition results (labels, medoids,
    neighbour lists), including an empty partition in the middle."""
    import torch
    from falcon_amd import synth
    from falcon_amd.cluster.cluster import AnnParams, ClusterPipeline, SpectrumDataset
    data = synth.generate(6000, seed=13)
    parts = []
    for ch in (2, 3):
        d = synth.select_charge(data, ch)
        parts.append(SpectrumDataset(d["precursor_mz"], d["retention_time"], d["mz"], d["intensity"], d["indptr"]))
    empty = SpectrumDataset(np.zeros(0, np.float32), np.zeros(0, np.float32), np.zeros(0, np.float32),
                            np.zeros(0, np.float32), np.zeros(1, np.int64))
    parts = [parts[0], empty, parts[1]]
    pipe = ClusterPipeline(ctx)
    p = AnnParams(eps=0.3)
    args = (20.0, "ppm", None, 0.05, 2 ** 15, p)
    ref = []
    for ds in parts:
        lab, med = pipe.run(ds, *args)
        ref.append((lab.cpu().numpy(), med.cpu().numpy(), pipe.last["nb_idx"].cpu().numpy() if len(ds) else None))
    for _ in range(2):                                       # twice: the second call reuses the front context
        outs = pipe.run_many(parts, *args)
        for (lab, med), last, (rl, rm, rnb) in zip(outs, pipe.lasts, ref):
            assert np.array_equal(lab.cpu().numpy(), rl) and np.array_equal(med.cpu().numpy(), rm)
            if rnb is not None:
                assert np.array_equal(last["nb_idx"].cpu().numpy(), rnb)


def test_run_sharded_equals_single_gpu_partition(ctx):
    """one dataset dealt to 3 'ranks' bucket by bucket (run one after the other here, the all-gatherv replaced by
    a list) gives exactly the single-GPU partition; world_size 1 goes through the real entry."""
    from sklearn.metrics import adjusted_rand_score
    from falcon_amd import distributed as fd, synth
    from falcon_amd.cluster.cluster import AnnParams, ClusterPipeline, SpectrumDataset
    d = synth.select_charge(synth.generate(6000, seed=23), 2)
    ds = SpectrumDataset(d["precursor_mz"], d["retention_time"], d["mz"], d["intensity"], d["indptr"])
    pipe = ClusterPipeline(ctx)
    p = AnnParams(eps=0.3)
    args = (20.0, "ppm", None, 0.05, 2 ** 15, p)
    lab1, med1 = pipe.run(ds, *args)
    lab1, med1 = lab1.cpu().numpy(), med1.cpu().numpy()
    parts = [fd.run_sharded(pipe, ds, *args, rank=r, world_size=3, local_only=True) for r in range(3)]
    assert sum(len(x[0]) for x in parts) == len(ds) and min(len(x[0]) for x in parts) > 0
    labels, medoids = fd.merge_shards(len(ds), parts)
    assert sorted(medoids.tolist()) == sorted(med1.tolist())
    assert adjusted_rand_score(lab1, labels) == 1.0
    labels_w1, medoids_w1 = fd.run_sharded(pipe, ds, *args)                 # not under torch.distributed: world of 1
    # same partition and representatives; noise singletons are numbered in sorted order here, in dataset order by `run`
    assert adjusted_rand_score(lab1, labels_w1) == 1.0 and sorted(medoids_w1.tolist()) == sorted(med1.tolist())
    assert np.array_equal(np.unique(labels_w1), np.arange(len(medoids_w1)))
    assert np.array_equal(labels_w1[medoids_w1], np.arange(len(medoids_w1)))          # medoids[c] represents cluster c


@pytest.mark.parametrize("rt_tol", [None, 5.0])
def test_refine_small_and_large_clusters_match_oracle(ctx, rt_tol):
    """a10 on DBSCAN clusters of 2 ... 700 members: the LDS path (<= 64 members) and the global-scratch path of the
    refine kernel against the oracle's `postprocess_cluster` (itself pinned to the reference's by the goldens)."""
    import torch
    rng = np.random.default_rng(12)
    sizes = [2, 3, 17, 63, 64, 65, 130, 700, 1, 40]
    n = sum(sizes)
    lab = np.concatenate([np.full(s, c, np.int32) for c, s in enumerate(sizes)])
    # precursor-sorted rows; inside a cluster a few groups 30 ppm apart (split by the 20 ppm rule) plus jitter
    mz = np.sort(500.0 + 0.015 * rng.integers(0, 4, n) + rng.normal(0, 0.002, n)).astype(np.float32)
    rt = (rng.integers(0, 3, n) * 8.0 + rng.normal(0, 1.0, n)).astype(np.float32)
    perm = rng.permutation(n)                       # clusters interleaved over the sorted rows
    lab = lab[perm]
    exp = lab.copy()
    total = 0
    for c in range(len(sizes)):                     # reference order: cluster by cluster, labels offset by the running total
        idx = np.flatnonzero(lab == c)
        sub = np.zeros(len(idx), np.int32)
        k = fo.postprocess_cluster(sub, mz[idx], rt[idx], 20.0, "ppm", rt_tol, 2, total)
        exp[idx] = sub
        total += k
    out, n_out = ctx.refine_clusters(torch.from_numpy(lab).to(ctx.tdev), len(sizes), ctx.to_dev(mz, torch.float32),
                                     ctx.to_dev(rt, torch.float32), 20.0, "ppm", rt_tol)
    assert n_out == total and total > 10
    assert np.array_equal(out.cpu().numpy(), exp)


def test_dbscan_long_chains_get_one_label_each(ctx):
    """components that are long CHAINS (row i's only neighbour is row i + 1): the union-find trees are deep when the final pass
    maps every row to its root.  That pass once compressed paths while it stored the roots -- another thread's halving store
    could land on top of a row's root and the row got the label slot of a non-root (seen on a 3,000-row chain; rare on
    shallow trees, not impossible).  Repeated: the race was timing dependent."""
    import torch
    n, k, seg = 60000, 4, 3000
    idx = np.full((n, k), -1, np.int32)
    dist = np.full((n, k), np.inf, np.float32)
    i = np.arange(n)
    link = (i % seg) != seg - 1                                # chains of 3,000 rows
    idx[link, 0] = i[link] + 1
    dist[link, 0] = 0.05
    ti, td = torch.from_numpy(idx).to(ctx.tdev), torch.from_numpy(dist).to(ctx.tdev)
    exp = (i // seg).astype(np.int32)
    for rep in range(20):
        lab, n_cl = ctx.dbscan(ti, td, 0.1)
        assert n_cl == n // seg
        assert np.array_equal(lab.cpu().numpy(), exp), rep


def test_star_graphs_more_clusters_than_half_the_rows(ctx):
    """DBSCAN clusters may hold ONE member: cores 0..3 of a star each keep only the non-core border 4 as their
    eps-neighbour, so 5 rows form 4 clusters (the refine arrays were sized for n/2 + 1 clusters).
    Staged ABI a9 -> a10 -> a11/a12 against the oracle."""
    import torch
    m, k = 3000, 4
    n = 5 * m
    nb_idx = np.full((n, k), -1, np.int32)
    nb_dist = np.full((n, k), np.inf, np.float32)
    for s in range(m):
        nb_idx[5 * s:5 * s + 4, 0] = 5 * s + 4          # four cores -> their shared border; the border stores nothing
        nb_dist[5 * s:5 * s + 4, 0] = 0.05
    mz = np.sort(np.repeat(500.0 + 0.5 * np.arange(m), 5).astype(np.float32))
    db, n_db = ctx.dbscan(ctx.to_dev(nb_idx, torch.int32), ctx.to_dev(nb_dist, torch.float32), 0.1)
    ref_db = fo.dbscan_components(nb_idx, nb_dist, 0.1)
    assert np.array_equal(db.cpu().numpy(), ref_db)
    assert n_db == 4 * m and n_db > n // 2 + 1
    lab, n_cl = ctx.refine_clusters(db.clone(), n_db, ctx.to_dev(mz, torch.float32), None, 20.0, "ppm", None)   # (in place)
    ref_lab = fo.refine_and_number(ref_db, None, mz, None, 20.0, "ppm", None)
    assert np.array_equal(lab.cpu().numpy(), ref_lab)
    assert n_cl == m                                    # {core 0, border} survives, the three lone cores become noise
    order = torch.arange(n, dtype=torch.int64, device=ctx.tdev)
    # a11/a12 with single-member clusters in the labels (the unrefined DBSCAN labels): sized by n as well
    labels, medoids = ctx.finalize(db, n_db, order, ctx.to_dev(nb_idx, torch.int32), ctx.to_dev(nb_dist, torch.float32))
    labels, medoids = labels.cpu().numpy(), medoids.cpu().numpy()
    assert len(medoids) == n_db and np.array_equal(labels, ref_db)
    assert np.array_equal(labels[medoids], np.arange(n_db))


def test_pipeline_f16_large_bucket_is_searched_exhaustively(ctx):
    """dtype="f16" (config 5) with a bucket far beyond the flat limit (5,000 spectra in one precursor window, where the
    float32 path would train a 64-list index): the float16 path scans it exhaustively -- same clustering as the
    float32 path told to probe every list."""
    import warnings
    from sklearn.metrics import adjusted_rand_score
    from falcon_amd import synth
    from falcon_amd.cluster.cluster import AnnParams, ClusterPipeline, SpectrumDataset, n_list_rule
    d = synth.select_charge(synth.generate(12000, seed=31), 2)
    pm = (600.0 + (d["precursor_mz"] - d["precursor_mz"].min()) / np.ptp(d["precursor_mz"]) * 0.9).astype(np.float32)
    ds = SpectrumDataset(pm, d["retention_time"], d["mz"], d["intensity"], d["indptr"])
    assert n_list_rule(np.array([len(ds)]), 16)[0] > 16                       # the float32 rule wants an IVF index here
    pipe = ClusterPipeline(ctx)
    args = (2000.0, "ppm", None, 0.05, 2 ** 15)                               # wide tolerance: one bucket, everything comparable
    lab16, med16 = pipe.run(ds, *args, AnnParams(eps=0.3, dtype="f16", low_dim=400))
    lab32, _ = pipe.run(ds, *args, AnnParams(eps=0.3, low_dim=400, n_probe=128))   # n_list <= n_probe: exhaustive fp32
    lab16, lab32, med16 = lab16.cpu().numpy(), lab32.cpu().numpy(), med16.cpu().numpy()
    assert np.array_equal(np.unique(lab16), np.arange(len(med16)))
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        assert adjusted_rand_score(lab32, lab16) >= 0.99          # same vectors up to float16 rounding


def test_partition_runner_concurrent_streams_equal_run(ctx):
    """the charge partitions on concurrent host threads / HIP streams / contexts (PartitionRunner, what bench.py times at 1 M
    spectra) give exactly the results of `run` one partition after the other -- flat and IVF buckets, several passes (the
    runner's contexts and scratch pools are reused), an empty partition in between."""
    import torch
    from falcon_amd import synth
    from falcon_amd.cluster.cluster import AnnParams, ClusterPipeline, PartitionRunner, SpectrumDataset
    data = synth.generate(40000, seed=17, mz_lo=600.0, mz_hi=606.0)          # dense enough for buckets with an index
    parts = []
    for ch in (2, 3):
        d = synth.select_charge(data, ch)
        parts.append(SpectrumDataset(d["precursor_mz"], d["retention_time"], d["mz"], d["intensity"], d["indptr"]))
    empty = SpectrumDataset(np.zeros(0, np.float32), np.zeros(0, np.float32), np.zeros(0, np.float32),
                            np.zeros(0, np.float32), np.zeros(1, np.int64))
    parts = [parts[0], empty, parts[1]]
    args = (20.0, "ppm", None, 0.05, 2 ** 15, AnnParams())
    pipe = ClusterPipeline(ctx)
    ref = []
    for ds in parts:
        lab, med = pipe.run(ds, *args)
        ref.append((lab.cpu().numpy(), med.cpu().numpy()))
    assert int(pipe.last["n_list"].max()) > 1
    # the same partitions as the runner may be handed them: numpy columns (the reference's hand-over), pinned host tensors
    # (uploaded partition by partition on the runner's copy stream, each partition's kernels behind its own bytes) and
    # tensors already on the device
    as_t = lambda ds, f: SpectrumDataset(*[f(torch.from_numpy(np.ascontiguousarray(t))) for t in ds.columns()])
    forms = {"numpy": parts, "pinned": [as_t(ds, lambda t: t.pin_memory()) for ds in parts],
             "device": [as_t(ds, lambda t: t.cuda()) for ds in parts]}
    assert parts[0].on_host() and forms["pinned"][0].on_host() and not forms["device"][0].on_host()
    runner = PartitionRunner(0, 2)
    try:
        for rep in range(3):
            for form, dss in forms.items():
                outs = runner.run(dss, *args)
                for (lab, med), (rl, rm) in zip(outs, ref):
                    assert np.array_equal(lab.cpu().numpy(), rl) and np.array_equal(med.cpu().numpy(), rm), (form, rep)
        # a STREAM of jobs: three jobs queued on the two slots before the first is collected (`submit` / `collect`: the slots take
        # the next job's partitions as they finish the current one's) -- every job's results are those of `run`
        handles = [runner.submit(forms["device"], *args, inputs_ready=True) for _ in range(3)]
        for h in handles:
            for (lab, med), (rl, rm) in zip(runner.collect(h), ref):
                assert np.array_equal(lab.cpu().numpy(), rl) and np.array_equal(med.cpu().numpy(), rm)
        # host partitions WITHOUT retention times (rt_tol None: the column is never read) and with float64 precursors
        no_rt = [SpectrumDataset(ds.precursor_mz.astype(np.float64), None, ds.mz, ds.intensity, ds.indptr) for ds in parts]
        outs = runner.run(no_rt, *args)
        for (lab, med), (rl, rm) in zip(outs, ref):
            assert np.array_equal(lab.cpu().numpy(), rl) and np.array_equal(med.cpu().numpy(), rm)
    finally:
        runner.close()
