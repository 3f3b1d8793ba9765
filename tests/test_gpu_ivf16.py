"""The IVF fine scan with the float16 prefilter (ivf16.hip: f16-MFMA list scan to 16-bit keys, k-th key per query from the keys,
exact float32 refinement of the precursor window, exact fallback) must give BIT-IDENTICAL neighbour lists to the staged path
(`fal_ivf_search_topk` -> `fal_filter_neighbors`: fp32-MFMA fine scan + wavefront select), which test_gpu_search.py /
test_gpu_regimes.py pin to the oracle."""
import numpy as np
import pytest

from tests.test_gpu_search import sparse_unit_vectors, unit_vectors

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ctx():
    from falcon_amd.device import Context
    c = Context(0)
    yield c
    c.close()


def _staged_and_prefiltered(ctx, X, off, nl, mz, rt, n_probe, k_ann, keep, tol, mode, rt_tol):
    import torch
    Xd = torch.from_numpy(X).to(ctx.tdev)
    mz_d = torch.from_numpy(mz).to(ctx.tdev)
    rt_d = None if rt is None else torch.from_numpy(rt).to(ctx.tdev)
    plain = ctx.ivf_build(Xd, off, nl)
    sim, idx = plain.search(n_probe, k_ann)
    e_idx, e_dist = ctx.filter_neighbors(sim, idx, mz_d, rt_d, tol, mode, rt_tol, keep)
    pre = ctx.ivf_build(Xd, off, nl, Xpre=Xd.to(torch.float16).contiguous(), prefilter_which=2)
    g_idx, g_dist = pre.search_neighbors(n_probe, k_ann, mz_d, rt_d, tol, mode, rt_tol, keep)
    cnt = pre.nb_count.cpu().numpy()
    ctx.sync()
    n_fallback = ctx.counter(5)
    e_idx, e_dist, g_idx, g_dist = (t.cpu().numpy() for t in (e_idx, e_dist, g_idx, g_dist))
    bad = np.flatnonzero((g_idx != e_idx).any(1) | (g_dist.view(np.uint32) != e_dist.view(np.uint32)).any(1))
    assert len(bad) == 0, (len(bad), bad[:10], g_idx[bad[0]][:12], e_idx[bad[0]][:12], g_dist[bad[0]][:6], e_dist[bad[0]][:6])
    assert np.array_equal(cnt, (e_idx >= 0).sum(1))
    return e_idx, n_fallback


def _buckets(sizes, d, seed, mz0=500.0, width=1.0):
    off = np.concatenate([[0], np.cumsum(sizes)])
    n = int(off[-1])
    X = unit_vectors(n, d, seed, noise=0.35)
    rng = np.random.default_rng(seed + 1)
    mz = np.concatenate([np.sort(mz0 + b + width * rng.random(s)) for b, s in enumerate(sizes)]).astype(np.float32)
    rt = (rng.random(n) * 100).astype(np.float32)
    return off, X, mz, rt


@pytest.mark.parametrize("d,n_probe,k_ann,keep,tol,mode,rt_tol", [
    (400, 16, 128, 64, 20.0, "ppm", None),        # the production setting (BASELINE configs[2] regime)
    (400, 16, 128, 64, 20.0, "ppm", 30.0),        # + retention time tolerance
    (400, 8, 32, 8, 60.0, "ppm", None),           # small k: the threshold cuts deep into the window
    (400, 16, 128, 64, 0.05, "Da", None),         # Da: wide windows
    (400, 32, 200, 100, 20.0, "ppm", None),       # k_ann > 128, more probes
    (64, 16, 64, 16, 20.0, "ppm", None), (128, 16, 128, 64, 20.0, "ppm", None), (256, 12, 100, 50, 20.0, "ppm", None),
])
def test_ivf_prefilter_neighbours_bit_identical_to_the_staged_path(ctx, d, n_probe, k_ann, keep, tol, mode, rt_tol):
    sizes = [6000, 300, 2500, 9000, 40, 1300]
    nl = np.array([64, 1, 32, 128, 1, 16], np.int32)          # flat buckets in between take the staged flat path
    off, X, mz, rt = _buckets(sizes, d, 31)
    X[off[3] + 10:off[3] + 16] = X[off[3] + 10]               # exact duplicates -> exact ties
    e_idx, n_fb = _staged_and_prefiltered(ctx, X, off, nl, mz, rt if rt_tol is not None else None, n_probe, k_ann, keep,
                                          tol, mode, rt_tol)
    assert (e_idx >= 0).sum() > off[-1] // 8
    assert n_fb < 0.05 * off[-1], n_fb                        # the prefiltered path did the work, not the fallback


@pytest.mark.parametrize("d,n_probe,k_ann,keep", [(400, 16, 128, 64), (256, 12, 100, 50), (128, 16, 128, 64), (64, 16, 64, 16)])
def test_list16r_form_of_the_fine_scan_gives_the_same_lists(ctx, monkeypatch, d, n_probe, k_ann, keep):
    """FALCON_LIST16=r: list16r_kernel (list16r.hip: the list's rows resident in LDS, every wave walks its own 32-query chunks,
    no workgroup barrier in the stream) writes the keys of list16_kernel bit for bit, so the neighbour lists stay those of the
    staged path -- lists of 1 to 4 slices, lists of more than 128 rows (several tiles), probe streams shorter than four
    chunks (waves without work), exact ties."""
    monkeypatch.setenv("FALCON_LIST16", "r")
    sizes = [6000, 300, 2500, 9000, 40, 1300, 5000]
    nl = np.array([64, 1, 32, 128, 1, 16, 32], np.int32)      # ~94 / 78 / 70 / 81 / 156 rows per list
    off, X, mz, rt = _buckets(sizes, d, 37)
    X[off[3] + 10:off[3] + 16] = X[off[3] + 10]
    e_idx, n_fb = _staged_and_prefiltered(ctx, X, off, nl, mz, None, n_probe, k_ann, keep, 20.0, "ppm", None)
    assert (e_idx >= 0).sum() > off[-1] // 8
    assert n_fb < 0.05 * off[-1], n_fb


@pytest.mark.parametrize("d,n_probe,k_ann,keep,tol,mode,rt_tol", [
    (400, 16, 128, 64, 20.0, "ppm", None), (400, 8, 32, 8, 0.05, "Da", 30.0), (128, 16, 64, 16, 20.0, "ppm", None),
    (64, 4, 32, 16, 60.0, "ppm", None),
])
def test_ivf_prefilter_on_sparse_rows_bit_identical_to_the_staged_path(ctx, d, n_probe, k_ann, keep, tol, mode, rt_tol):
    """rows shaped like vectorised spectra (a few dozen non-zero components): the exact chains of the kept pairs run over the
    rows' sparse form (pairs16s_kernel) -- plus rows with exactly 64 / more than 64 non-zeros (dense chain) and all-zero rows; the staged path computes every similarity densely on the fp32 matrix cores"""
    sizes = [6000, 300, 2500, 9000, 40, 1300]
    nl = np.array([64, 1, 32, 128, 1, 16], np.int32)
    off, _, mz, rt = _buckets(sizes, d, 33)
    X = sparse_unit_vectors(int(off[-1]), d, 35, nnz_hi=min(50, d // 2))
    X[off[3] + 10:off[3] + 16] = X[off[3] + 10]
    e_idx, n_fb = _staged_and_prefiltered(ctx, X, off, nl, mz, rt if rt_tol is not None else None, n_probe, k_ann, keep,
                                          tol, mode, rt_tol)
    assert (e_idx >= 0).sum() > off[-1] // 8


def test_mixed_sign_rows_are_detected_and_searched_exactly(ctx):
    """the prefilters' error bound holds for non-negative rows only.  With genuinely mixed-sign rows
    (40 % negative components: the error of a float16 product sum scales with sum |x_i y_i|, not with the similarity) the
    library must notice (fal_ctx_counter 6) and build / search with the exact kernels: index and neighbour lists equal
    the ones computed without any float16 copy, bit for bit -- and equal the oracle's index."""
    import torch
    from oracle import falcon_oracle as fo
    sizes = [6000, 300, 2500]
    nl = np.array([64, 1, 32], np.int32)
    off, _, mz, rt = _buckets(sizes, 400, 37)
    X = sparse_unit_vectors(int(off[-1]), 400, 39, signed=True)
    assert (X < 0).mean() > 0.02
    Xd = torch.from_numpy(X).to(ctx.tdev)
    X16 = Xd.to(torch.float16).contiguous()
    mz_d = torch.from_numpy(mz).to(ctx.tdev)
    plain = ctx.ivf_build(Xd, off, nl, kmeans_iters=4)
    assert ctx.counter(6) == 0                                   # nothing to check without float16 copies
    e = plain.search_neighbors(16, 128, mz_d, None, 20.0, "ppm", None, 64)
    pre = ctx.ivf_build(Xd, off, nl, kmeans_iters=4, Xkm=X16, Xpre=X16, prefilter_which=3)
    assert ctx.counter(6) == 3                                   # indexed rows AND the flat prefilter's rows rejected
    g = pre.search_neighbors(16, 128, mz_d, None, 20.0, "ppm", None, 64)
    for a, b in zip(plain.export(), pre.export()):
        assert torch.equal(a, b)
    assert torch.equal(e[0], g[0]) and torch.equal(e[1].view(torch.int32), g[1].view(torch.int32))
    ctx.sync()
    assert ctx.counter(5) == 0                                   # no prefiltered search ran, so no fallback either
    cent, asg, perm, loff = [t.cpu().numpy() for t in pre.export()]
    C, ra, rperm, roff = fo.ivf_build(X[:6000], 64, 4)
    assert np.array_equal(asg[:6000], ra) and np.array_equal(cent[:64], C)
    # non-negative rows next: the same calls use the prefilters again (the flag belongs to the index, not the context)
    Xp = torch.from_numpy(np.abs(X)).to(ctx.tdev)
    ok = ctx.ivf_build(Xp, off, nl, kmeans_iters=4, Xkm=Xp.to(torch.float16).contiguous(),
                       Xpre=Xp.to(torch.float16).contiguous(), prefilter_which=3)
    assert ctx.counter(6) == 0
    ok.close(), pre.close(), plain.close()


def test_ivf_prefilter_handles_ties_zero_rows_and_few_candidates(ctx):
    """hundreds of identical spectra (more members around the k-th key than the hand-off holds), all-zero rows (every key 0),
    near-duplicates, and a bucket whose lists hold fewer than k_ann candidates per query: still bit-identical"""
    sizes = [5000, 700, 3000]
    nl = np.array([64, 32, 32], np.int32)                     # 700 rows in 32 lists, 16 probed: ~350 candidates; k_ann 400 below
    off, X, mz, _ = _buckets(sizes, 400, 41, mz0=600.0, width=0.02)
    X[100:420] = X[100]                                       # 320 identical rows
    X[500:560] = 0                                            # empty spectra
    base = X[off[2]].copy()
    for i in range(200):                                      # near-duplicates: dense similarities just below 1
        v = base.copy()
        v[(7 * i) % 400] += 1e-3 * (i + 1)
        X[off[2] + 1 + i] = v / np.linalg.norm(v)
    _, n_fb = _staged_and_prefiltered(ctx, X, off, nl, mz, None, 16, 128, 64, 20.0, "ppm", None)
    assert n_fb > 100
    _staged_and_prefiltered(ctx, X, off, nl, mz, None, 16, 250, 64, 20.0, "ppm", None)


def test_ivf_prefilter_long_lists_take_the_second_pass_or_the_staged_scan(ctx):
    """coarse indexes: ~300 rows per list.  8 probes = ~2,400 keys per query (select16's second pass with 64 keys per lane);
    16 probes = ~4,800 keys, more than select16 holds in registers: the search keeps the exact staged scan for such an index."""
    sizes = [9600, 2000]
    nl = np.array([32, 16], np.int32)
    off, X, mz, _ = _buckets(sizes, 128, 53)
    _, n_fb = _staged_and_prefiltered(ctx, X, off, nl, mz, None, 8, 64, 32, 20.0, "ppm", None)
    assert n_fb < 500
    _, n_fb = _staged_and_prefiltered(ctx, X, off, nl, mz, None, 16, 64, 32, 20.0, "ppm", None)
    assert n_fb == 0


def test_coarse_quantiser_from_the_build_keys_with_up_to_512_lists(ctx):
    """buckets with 129..512 lists: the float16 assignment runs in groups of 128 centroids, the keys have up to 512 columns
    and the quantiser reads 32 keys per lane -- the search equals the one through an exactly built index bit for bit"""
    import torch
    sizes = [21000, 3000, 11000]
    nl = np.array([512, 64, 200], np.int32)
    off, X, mz, rt = _buckets(sizes, 128, 67)
    X[off[0]:off[0] + 21000:41] = X[off[0]]                   # identical k-means seeds -> identical centroids (ties)
    Xd = torch.from_numpy(X).to(ctx.tdev)
    plain = ctx.ivf_build(Xd, off, nl, kmeans_iters=3)
    keyed = ctx.ivf_build(Xd, off, nl, kmeans_iters=3, Xkm=Xd.to(torch.float16).contiguous())
    for n_probe in (32, 5):
        s0, i0 = plain.search(n_probe, 64)
        s1, i1 = keyed.search(n_probe, 64)
        assert torch.equal(i0, i1) and torch.equal(s0.view(torch.int32), s1.view(torch.int32))


def test_float16_assignment_and_keyed_quantiser_with_1024_and_2048_lists(ctx):
    """buckets with 513..2,048 lists (`--batch_size 65536` on windows of 40 k+ spectra: SURVEY 8d's C4 row is n_list 1,024): the
    float16 assignment merges up to 16 groups of 128 centroids, the keys have up to 2,048 columns, the quantiser holds 16 / 32
    keys per lane (coarse16x2 / x4) and hands queries with many equal keys to the workgroup-level kernel -- index and search equal
    the exactly built index's bit for bit.  (Round 4 sent such buckets to the exact fp32 kernels: 7.5 x the build time.)"""
    import torch
    sizes = [24000, 3000, 43000]
    nl = np.array([1024, 64, 2048], np.int32)
    off, X, mz, rt = _buckets(sizes, 128, 69)
    for b in (0, 2):                                            # 40 identical k-means seeds -> 40 identical centroids: exact ties,
        seeds = off[b] + (np.arange(nl[b], dtype=np.int64) * sizes[b]) // nl[b]      # more members than a wave's lists hold
        X[seeds[5:45]] = X[seeds[5]]
    Xd = torch.from_numpy(X).to(ctx.tdev)
    plain = ctx.ivf_build(Xd, off, nl, kmeans_iters=3)
    keyed = ctx.ivf_build(Xd, off, nl, kmeans_iters=3, Xkm=Xd.to(torch.float16).contiguous())
    for a, b in zip(plain.export(), keyed.export()):
        assert torch.equal(a, b)                                # centroids, assignment, lists: the same index
    for n_probe in (32, 5):
        s0, i0 = plain.search(n_probe, 64)
        s1, i1 = keyed.search(n_probe, 64)
        assert torch.equal(i0, i1) and torch.equal(s0.view(torch.int32), s1.view(torch.int32))
    mz_d = torch.from_numpy(mz).to(ctx.tdev)
    n0 = plain.search_neighbors(32, 64, mz_d, None, 20.0, "ppm", None, 32)
    pre = ctx.ivf_build(Xd, off, nl, kmeans_iters=3, Xkm=Xd.to(torch.float16).contiguous(), Xpre=Xd.to(torch.float16).contiguous(),
                        prefilter_which=2)
    n1 = pre.search_neighbors(32, 64, mz_d, None, 20.0, "ppm", None, 32)
    assert torch.equal(n0[0], n1[0]) and torch.equal(n0[1].view(torch.int32), n1[1].view(torch.int32))


def test_float16_assignment_with_list_counts_that_are_not_powers_of_two(ctx):
    """the C ABI takes any list count per bucket: 600 lists (group jobs of 4 x 128 + 88 centroids, keys of 640 columns) and
    1,500 (11 x 128 + 92; 32 keys per lane in the quantiser, the last 548 of them padding), low_dim 64 and 400 --
    keyed build and search == the exact build's, bit for bit"""
    import torch
    for d, iters in ((64, 3), (400, 2)):
        sizes = [14000, 31000]
        nl = np.array([600, 1500], np.int32)
        off, X, mz, rt = _buckets(sizes, d, 71)
        Xd = torch.from_numpy(X).to(ctx.tdev)
        plain = ctx.ivf_build(Xd, off, nl, kmeans_iters=iters)
        keyed = ctx.ivf_build(Xd, off, nl, kmeans_iters=iters, Xkm=Xd.to(torch.float16).contiguous())
        for a, b in zip(plain.export(), keyed.export()):
            assert torch.equal(a, b)
        s0, i0 = plain.search(32, 64)
        s1, i1 = keyed.search(32, 64)
        assert torch.equal(i0, i1) and torch.equal(s0.view(torch.int32), s1.view(torch.int32))


def test_a_bucket_beyond_2048_lists_next_to_prefiltered_ones(ctx):
    """more than 2,048 lists in one bucket (a window of > 80 k spectra kept whole by `--batch_size` >= 2^17): that bucket takes
    the exact fp32 assignment while its neighbours keep the float16 one, no keys are left for the quantiser (the staged coarse
    scan serves every bucket) -- index and search still equal the all-exact build's"""
    import torch
    sizes = [90000, 9000, 2500]
    nl = np.array([4096, 128, 32], np.int32)
    off, X, mz, rt = _buckets(sizes, 64, 73)
    Xd = torch.from_numpy(X).to(ctx.tdev)
    plain = ctx.ivf_build(Xd, off, nl, kmeans_iters=2)
    keyed = ctx.ivf_build(Xd, off, nl, kmeans_iters=2, Xkm=Xd.to(torch.float16).contiguous())
    for a, b in zip(plain.export(), keyed.export()):
        assert torch.equal(a, b)
    s0, i0 = plain.search(16, 32)
    s1, i1 = keyed.search(16, 32)
    assert torch.equal(i0, i1) and torch.equal(s0.view(torch.int32), s1.view(torch.int32))
    mz_d = torch.from_numpy(mz).to(ctx.tdev)
    n0 = plain.search_neighbors(16, 32, mz_d, None, 20.0, "ppm", None, 16)
    pre = ctx.ivf_build(Xd, off, nl, kmeans_iters=2, Xkm=Xd.to(torch.float16).contiguous(), Xpre=Xd.to(torch.float16).contiguous(),
                        prefilter_which=2)
    n1 = pre.search_neighbors(16, 32, mz_d, None, 20.0, "ppm", None, 16)
    assert torch.equal(n0[0], n1[0]) and torch.equal(n0[1].view(torch.int32), n1[1].view(torch.int32))


@pytest.mark.parametrize("d,n_probe", [(400, 16), (400, 3), (128, 8), (64, 1)])
def test_coarse_quantiser_from_the_build_keys_gives_the_same_search(ctx, d, n_probe):
    """an index built with the float16 k-means prefilter keeps the final pass's (row, centroid) similarities as 16-bit keys and
    its search takes the n_probe lists from them (coarse16.hip) instead of scanning again: same top-k, bit for bit -- with
    identical centroids (exact ties between lists) and all-zero rows in the data"""
    import torch
    sizes = [6000, 300, 2500, 9000, 1300]
    nl = np.array([64, 1, 32, 128, 16], np.int32)
    off, X, mz, rt = _buckets(sizes, d, 61)
    X[off[3]:off[3] + 9000:70] = X[off[3]]                    # the k-means seeds of bucket 3 (every 70th row): identical centroids
    X[off[0] + 5:off[0] + 40] = 0
    Xd = torch.from_numpy(X).to(ctx.tdev)
    plain = ctx.ivf_build(Xd, off, nl, kmeans_iters=3)
    keyed = ctx.ivf_build(Xd, off, nl, kmeans_iters=3, Xkm=Xd.to(torch.float16).contiguous())
    s0, i0 = plain.search(n_probe, 64)
    s1, i1 = keyed.search(n_probe, 64)
    assert torch.equal(i0, i1) and torch.equal(s0.view(torch.int32), s1.view(torch.int32))
    mz_d = torch.from_numpy(mz).to(ctx.tdev)
    n0 = plain.search_neighbors(n_probe, 64, mz_d, None, 20.0, "ppm", None, 32)
    pre = ctx.ivf_build(Xd, off, nl, kmeans_iters=3, Xkm=Xd.to(torch.float16).contiguous(), Xpre=Xd.to(torch.float16).contiguous(),
                        prefilter_which=2)
    n1 = pre.search_neighbors(n_probe, 64, mz_d, None, 20.0, "ppm", None, 32)
    assert torch.equal(n0[0], n1[0]) and torch.equal(n0[1].view(torch.int32), n1[1].view(torch.int32))


def test_pipeline_with_and_without_ivf_prefilter_is_identical(ctx):
    """whole path on synthetic spectra dense enough for IVF buckets: the prefilter changes nothing but the time"""
    from falcon_amd import synth
    from falcon_amd.cluster.cluster import AnnParams, ClusterPipeline, SpectrumDataset
    d = synth.select_charge(synth.generate(60000, seed=11, mz_lo=600.0, mz_hi=603.0), 2)
    ds = SpectrumDataset(d["precursor_mz"], d["retention_time"], d["mz"], d["intensity"], d["indptr"])
    pipe = ClusterPipeline(ctx)
    outs = []
    for pre in (True, False):
        p = AnnParams(ivf_prefilter=pre)
        lab, med = pipe.run(ds, 20.0, "ppm", None, 0.05, 2 ** 15, p)
        assert int(pipe.last["n_list"].max()) > 1
        outs.append((lab.cpu().numpy(), med.cpu().numpy(), pipe.last["nb_idx"].cpu().numpy(),
                     pipe.last["nb_dist"].cpu().numpy().view(np.uint32), ctx.counter(5)))
    assert np.array_equal(outs[0][0], outs[1][0]) and np.array_equal(outs[0][1], outs[1][1])
    assert np.array_equal(outs[0][2], outs[1][2]) and np.array_equal(outs[0][3], outs[1][3])
    assert outs[0][4] < 0.02 * len(ds)                        # few queries needed the exact fallback


def _rows_of_at_most_64_nonzeros(n, d, seed, nnz_lo=5, nnz_hi=50):
    """rows shaped like vectorised spectra, NONE with more than 64 non-zeros (the precondition of list16s_kernel: an index with a
    wider row keeps the dense gather); near-duplicates in groups, rows with exactly 64 entries, all-zero rows"""
    rng = np.random.default_rng(seed)
    n_groups = max(1, n // 12)
    proto = np.zeros((n_groups, d), np.float32)
    for g in range(n_groups):
        c = rng.choice(d, rng.integers(nnz_lo, min(nnz_hi, d // 2) + 1), replace=False)
        proto[g, c] = rng.random(len(c)).astype(np.float32) + 0.05
    X = proto[rng.integers(0, n_groups, n)].copy()
    X *= np.abs(1.0 + 0.3 * rng.standard_normal(X.shape).astype(np.float32)) * (X != 0)
    if d >= 64:
        for r in rng.choice(n, max(4, n // 100), replace=False):       # rows that fill the record: exactly 64 entries
            X[r] = 0
            X[r, rng.choice(d, 64, replace=False)] = rng.random(64).astype(np.float32) + 0.01
    nrm = np.sqrt((X.astype(np.float64) ** 2).sum(1))
    X = (X / np.maximum(nrm, 1e-30)[:, None]).astype(np.float32)
    X[rng.choice(n, 5, replace=False)] = 0
    assert (X != 0).sum(1).max() <= 64
    return X


@pytest.mark.parametrize("d,n_probe,k_ann,keep,tol,mode,rt_tol", [
    (400, 16, 128, 64, 20.0, "ppm", None), (400, 8, 32, 8, 0.05, "Da", 30.0), (800, 16, 128, 64, 20.0, "ppm", None),
    (256, 12, 100, 50, 20.0, "ppm", None), (128, 16, 64, 16, 20.0, "ppm", None), (64, 4, 32, 16, 60.0, "ppm", None),
])
def test_list16s_sparse_query_records_give_the_staged_lists(ctx, monkeypatch, d, n_probe, k_ann, keep, tol, mode, rt_tol):
    """list16s_kernel (list16s.hip): the probing queries gathered as their 256-byte sparse records and expanded in LDS -- the form
    the production path takes whenever the build had float16 rows (the records are made beside the sparse rows) and no row holds
    more than 64 non-zeros.  Neighbour lists == the staged path's, bit for bit, and == the dense-gather form's (FALCON_LIST16=d);
    lists of 1 to 4 slices, lists of more than 128 rows, probe streams of one and two chunks, exact duplicates, all-zero rows."""
    import torch
    sizes = [6000, 300, 2500, 9000, 40, 1300, 5000, 150]
    nl = np.array([64, 1, 32, 128, 1, 16, 32, 2], np.int32)
    off, _, mz, rt = _buckets(sizes, d, 41)
    X = _rows_of_at_most_64_nonzeros(int(off[-1]), d, 43)
    X[off[3] + 10:off[3] + 16] = X[off[3] + 10]
    rt_u = rt if rt_tol is not None else None
    Xd = torch.from_numpy(X).to(ctx.tdev)
    x16 = Xd.to(torch.float16).contiguous()
    mz_d = torch.from_numpy(mz).to(ctx.tdev)
    rt_d = None if rt_u is None else torch.from_numpy(rt_u).to(ctx.tdev)
    # (the exact float32 assignment kernels end at 512 columns: at 800 the reference index is built with the float16 assignment
    #  too -- the same index by construction, tests/test_gpu_search.py -- and searched with the staged float32 kernels)
    plain = ctx.ivf_build(Xd, off, nl, Xkm=x16 if d > 512 else None)
    sim, idx = plain.search(n_probe, k_ann)
    e_idx, e_dist = ctx.filter_neighbors(sim, idx, mz_d, rt_d, tol, mode, rt_tol, keep)
    out = {}
    for form in ("s", "d", "fused"):
        if form == "d":
            monkeypatch.setenv("FALCON_LIST16", "d")
        if form == "fused":                    # kept16 as the tail of the selection kernel (select16k_kernel: round 6 A/B switch)
            monkeypatch.delenv("FALCON_LIST16")
            monkeypatch.setenv("FALCON_KEPT16", "fused")
        pre = ctx.ivf_build(Xd, off, nl, Xpre=x16, Xkm=x16, prefilter_which=2)       # (Xkm: the build sees float16 rows -> records)
        g_idx, g_dist = pre.search_neighbors(n_probe, k_ann, mz_d, rt_d, tol, mode, rt_tol, keep)
        ctx.sync()
        assert ctx.counter(6) == 0
        out[form] = (g_idx.cpu().numpy(), g_dist.cpu().numpy(), ctx.counter(5))
    e_idx, e_dist = e_idx.cpu().numpy(), e_dist.cpu().numpy()
    for form, (gi, gd, n_fb) in out.items():
        bad = np.flatnonzero((gi != e_idx).any(1) | (gd.view(np.uint32) != e_dist.view(np.uint32)).any(1))
        assert len(bad) == 0, (form, len(bad), bad[:10])
        assert n_fb < 0.05 * off[-1], (form, n_fb)
    assert out["s"][2] == out["d"][2]                 # the same keys -> the same queries took the exact fallback
    assert (e_idx >= 0).sum() > off[-1] // 8


def test_an_index_with_a_row_of_more_than_64_nonzeros_keeps_the_dense_gather(ctx):
    """the sparse records hold 64 entries: one wider row in an indexed bucket and the searches of that index gather dense rows
    (list16_kernel) -- same lists as the staged path"""
    sizes = [6000, 2500]
    nl = np.array([64, 32], np.int32)
    off, _, mz, rt = _buckets(sizes, 400, 45)
    X = sparse_unit_vectors(int(off[-1]), 400, 47)          # (holds rows of 65..160 non-zeros)
    assert (X != 0).sum(1).max() > 64
    import torch
    Xd = torch.from_numpy(X).to(ctx.tdev)
    x16 = Xd.to(torch.float16).contiguous()
    mz_d = torch.from_numpy(mz).to(ctx.tdev)
    plain = ctx.ivf_build(Xd, off, nl)
    sim, idx = plain.search(16, 128)
    e_idx, e_dist = ctx.filter_neighbors(sim, idx, mz_d, None, 20.0, "ppm", None, 64)
    pre = ctx.ivf_build(Xd, off, nl, Xpre=x16, Xkm=x16, prefilter_which=2)
    g_idx, g_dist = pre.search_neighbors(16, 128, mz_d, None, 20.0, "ppm", None, 64)
    assert torch.equal(g_idx, e_idx) and torch.equal(g_dist.view(torch.int32), e_dist.view(torch.int32))
