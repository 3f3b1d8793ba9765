"""GPU parity: a6 (IVF build) and a7 (n_probe search + top-k) vs the oracle."""
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


def unit_vectors(n, d, seed, n_centers=None, noise=0.25):
    rng = np.random.default_rng(seed)
    k = n_centers or max(2, n // 8)
    cent = np.abs(rng.normal(size=(k, d))) * (rng.random((k, d)) < 0.12)
    X = cent[rng.integers(0, k, n)] + noise * np.abs(rng.normal(size=(n, d))) * (rng.random((n, d)) < 0.05)
    X[rng.random(n) < 0.02] = 0          # a few all-zero rows (empty spectra)
    nrm = np.linalg.norm(X, axis=1, keepdims=True)
    return (X / np.where(nrm > 0, nrm, 1)).astype(np.float32)


@pytest.mark.parametrize("d,k", [(400, 128), (400, 64), (64, 16), (512, 200), (200, 7), (128, 32), (256, 100)])
def test_flat_exhaustive_topk(ctx, d, k):
    """flat buckets == brute-force cosine top-k inside each bucket (SURVEY 8c ground truth).  Sizes cover the three forms of
    the fp32 scan: one block per bucket (<= 32 rows: dense_tiny4_kernel), groups of four tiles sharing a candidate stream
    (dense4_kernel: partial groups, partial last tiles, many groups), and dense_kernel (low_dim 512)."""
    import torch
    sizes = [1, 2, 31, 32, 33, 100, 128, 129, 257, 700, 1500, 7, 19, 30]
    off = np.concatenate([[0], np.cumsum(sizes)])
    X = unit_vectors(off[-1], d, 3)
    X[off[4]:off[4] + 5] = X[off[4]]                 # exact duplicates -> exact ties
    idxr = ctx.ivf_build(torch.from_numpy(X).to(ctx.tdev), off, np.ones(len(sizes), np.int32))
    sim, idx = idxr.search(16, k)
    sim, idx = sim.cpu().numpy(), idx.cpu().numpy()
    for a, b in zip(off[:-1], off[1:]):
        rs, ri = fo.exhaustive_topk(X[a:b], k, base=a)
        assert_topk_exact(sim[a:b], idx[a:b], rs, ri, what=f"bucket {a}:{b}")
        # exact ties are broken by ascending id, bit for bit
    a = off[4]
    assert np.array_equal(idx[a, :5], np.arange(a, a + 5))


def test_dense4ab_two_waves_per_simd_form_gives_the_same_similarities(ctx, monkeypatch):
    """`FALCON_DENSE4=ab` (dense4ab.hip, round 6): the flat scan with the query tile split along K between two waves per tile --
    wave A runs the first half of every block's chain, hands the accumulators over through LDS, wave B finishes and stores.
    Same chain, same stores: similarities, ids and tie order of the oracle, bit for bit (low_dim 400; groups with idle tiles,
    partial last tiles, one-chunk and many-chunk buckets)."""
    import torch
    monkeypatch.setenv("FALCON_DENSE4", "ab")
    sizes = [33, 100, 128, 129, 257, 700, 1500, 64, 4000, 31]
    off = np.concatenate([[0], np.cumsum(sizes)])
    X = unit_vectors(off[-1], 400, 5)
    X[off[4]:off[4] + 5] = X[off[4]]
    idxr = ctx.ivf_build(torch.from_numpy(X).to(ctx.tdev), off, np.ones(len(sizes), np.int32))
    sim, idx = idxr.search(16, 128)
    sim, idx = sim.cpu().numpy(), idx.cpu().numpy()
    for a, b in zip(off[:-1], off[1:]):
        rs, ri = fo.exhaustive_topk(X[a:b], 128, base=a)
        assert_topk_exact(sim[a:b], idx[a:b], rs, ri, what=f"bucket {a}:{b}")


def test_small_flat_buckets_beyond_low_dim_512_are_exact(ctx):
    """float16 vectors at low_dim 800 (BASELINE configs[4]): flat buckets of fewer than 64 rows are scanned by exact fmaf chains on
    the vector ALU (flat_exact_small_kernel) -- similarities bit-identical to the oracle's chain over the float32 images of the
    float16 rows, tie order included; larger flat buckets stay on the f16 matrix cores (within 2e-6)."""
    import torch
    sizes = [3, 40, 63, 1, 33, 200, 17]
    off = np.concatenate([[0], np.cumsum(sizes)])
    X = unit_vectors(off[-1], 800, 29).astype(np.float16)
    X[off[1] + 2:off[1] + 6] = X[off[1] + 2]                  # exact ties
    Ximg = X.astype(np.float32)                               # the float32 image of the float16 rows (fal_vectorize_f16_image)
    idxr = ctx.ivf_build(torch.from_numpy(Ximg).to(ctx.tdev), off, np.ones(len(sizes), np.int32),
                         X16=torch.from_numpy(X).to(ctx.tdev))
    k = 32
    sim, idx = idxr.search(16, k)
    sim, idx = sim.cpu().numpy(), idx.cpu().numpy()
    for b, (a, e) in enumerate(zip(off[:-1], off[1:])):
        rs, ri = fo.exhaustive_topk(Ximg[a:e], k, base=a)
        if e - a < 64:
            assert_topk_exact(sim[a:e], idx[a:e], rs, ri, what=f"bucket {b} ({e - a} rows)")
        else:                                                 # the f16-MFMA scan: float32 accumulation in the pipe's own order
            fin = np.isfinite(rs)
            assert np.array_equal(np.isfinite(sim[a:e]), fin) and np.abs(sim[a:e][fin] - rs[fin]).max() <= 2e-6


def test_flat_many_small_batches(ctx, monkeypatch):
    """forces several scan/select batches through a tiny sims buffer."""
    import torch
    sizes = [300] * 12
    off = np.concatenate([[0], np.cumsum(sizes)])
    X = unit_vectors(off[-1], 400, 5)
    idxr = ctx.ivf_build(torch.from_numpy(X).to(ctx.tdev), off, np.ones(len(sizes), np.int32))
    sim, idx = idxr.search(16, 32)
    sim, idx = sim.cpu().numpy(), idx.cpu().numpy()
    for a, b in zip(off[:-1], off[1:]):
        rs, ri = fo.exhaustive_topk(X[a:b], 32, base=a)
        assert_topk_exact(sim[a:b], idx[a:b], rs, ri)


@pytest.mark.parametrize("sizes,nlists,d", [
    ([3000], [64], 400), ([900, 40, 2500, 130], [16, 1, 32, 2], 400), ([1200], [16], 64),
    # the shared-stream assignment kernel: two centroid groups (128 + 72, last tile 8 centroids), a 33-list bucket
    # (second tile holds one centroid), three 2,048-row segments
    ([5000, 2100], [200, 33], 400),
    # more than 512 lists: the row-resident assignment kernel
    ([24000], [600], 64),
])
def test_ivf_build_matches_oracle(ctx, sizes, nlists, d):
    import torch
    off = np.concatenate([[0], np.cumsum(sizes)])
    X = unit_vectors(off[-1], d, 11)
    idxr = ctx.ivf_build(torch.from_numpy(X).to(ctx.tdev), off, np.array(nlists, np.int32), kmeans_iters=4)
    cent, asg, perm, loff = [t.cpu().numpy() for t in idxr.export()]
    lb = np.concatenate([[0], np.cumsum(nlists)])
    assert loff[0] == 0 and loff[-1] == off[-1]
    assert np.array_equal(np.sort(perm), np.arange(off[-1]))
    for b, (a, e) in enumerate(zip(off[:-1], off[1:])):
        C, ra, rperm, roff = fo.ivf_build(X[a:e], nlists[b], 4)
        # the oracle sums in the kernels' k order (oracle/kordered.c): every assignment, list and centroid is identical
        assert np.array_equal(asg[a:e], ra), (b, (asg[a:e] != ra).sum())
        assert np.array_equal(loff[lb[b]:lb[b + 1] + 1] - a, roff)
        assert np.array_equal(perm[a:e] - a, rperm)
        if nlists[b] > 1:
            # ordered float32 sums + fixed-order norm: centroids are bit-identical
            assert np.array_equal(cent[lb[b]:lb[b + 1]], C)


def sparse_unit_vectors(n, d, seed, nnz_lo=20, nnz_hi=50, signed=False):
    """rows shaped like vectorised spectra: a few dozen non-zero components, near-duplicates in groups; plus rows with exactly
    64 and with more than 64 non-zeros (the k-means update reads those dense) and all-zero rows.  Non-negative like hashed
    spectra unless `signed` (then ~40 % of the components are negative: outside the float16 prefilters' precondition)"""
    rng = np.random.default_rng(seed)
    n_groups = max(1, n // 12)
    proto = np.zeros((n_groups, d), np.float32)
    for g in range(n_groups):
        c = rng.choice(d, rng.integers(nnz_lo, nnz_hi + 1), replace=False)
        proto[g, c] = rng.random(len(c)).astype(np.float32) + 0.05
    X = proto[rng.integers(0, n_groups, n)].copy()
    X *= np.abs(1.0 + 0.3 * rng.standard_normal(X.shape).astype(np.float32)) * (X != 0)
    if signed:
        X *= np.where(rng.random(X.shape) < 0.4, -1.0, 1.0).astype(np.float32)
    for r in rng.choice(n, max(4, n // 50), replace=False):        # wide rows
        k = 64 if (r % 3 == 0 or d <= 65) else int(rng.integers(65, min(d, 160) + 1))
        X[r] = 0
        X[r, rng.choice(d, k, replace=False)] = rng.random(k).astype(np.float32) + 0.01
    nrm = np.sqrt((X.astype(np.float64) ** 2).sum(1))
    X = (X / np.maximum(nrm, 1e-30)[:, None]).astype(np.float32)
    X[rng.choice(n, 5, replace=False)] = 0
    return X


@pytest.mark.parametrize("sizes,nlists,d,half", [
    ([3000, 500, 2200], [64, 1, 32], 400, False), ([5000, 2100], [200, 33], 400, True), ([1500], [16], 64, False),
    ([2600, 40], [48, 1], 128, True),
])
def test_ivf_build_on_sparse_rows_matches_oracle(ctx, sizes, nlists, d, half):
    """k-means on rows with few non-zero components: the centroid update sums the members' non-zero entries only (rows sorted
    by list, centroid_update_kernel) -- centroids, assignments and lists still equal the oracle's dense sums bit for bit"""
    import torch
    off = np.concatenate([[0], np.cumsum(sizes)])
    X = sparse_unit_vectors(off[-1], d, 17, nnz_hi=min(50, d // 2))
    Xd = torch.from_numpy(X).to(ctx.tdev)
    idxr = ctx.ivf_build(Xd, off, np.array(nlists, np.int32), kmeans_iters=5,
                         Xkm=Xd.to(torch.float16).contiguous() if half else None)
    cent, asg, perm, loff = [t.cpu().numpy() for t in idxr.export()]
    lb = np.concatenate([[0], np.cumsum(nlists)])
    assert loff[0] == 0 and loff[-1] == off[-1]
    for b, (a, e) in enumerate(zip(off[:-1], off[1:])):
        C, ra, rperm, roff = fo.ivf_build(X[a:e], nlists[b], 5)
        assert np.array_equal(asg[a:e], ra), (b, (asg[a:e] != ra).sum())
        assert np.array_equal(loff[lb[b]:lb[b + 1] + 1] - a, roff)
        assert np.array_equal(perm[a:e] - a, rperm)
        if nlists[b] > 1:
            assert np.array_equal(cent[lb[b]:lb[b + 1]].view(np.uint32), C.view(np.uint32))


@pytest.mark.parametrize("n_probe,k", [(4, 32), (16, 128), (64, 64)])
def test_ivf_search_matches_oracle_on_same_index(ctx, n_probe, k):
    import torch
    sizes, nlists = [2600, 700, 50, 4000], [32, 8, 1, 64]
    off = np.concatenate([[0], np.cumsum(sizes)])
    X = unit_vectors(off[-1], 400, 21)
    idxr = ctx.ivf_build(torch.from_numpy(X).to(ctx.tdev), off, np.array(nlists, np.int32), kmeans_iters=3)
    cent, asg, perm, loff = [t.cpu().numpy() for t in idxr.export()]
    sim, idx = idxr.search(n_probe, k)
    sim, idx = sim.cpu().numpy(), idx.cpu().numpy()
    lb = np.concatenate([[0], np.cumsum(nlists)])
    for b, (a, e) in enumerate(zip(off[:-1], off[1:])):
        lo = loff[lb[b]:lb[b + 1] + 1] - a
        rs, ri = fo.ivf_search(X[a:e], cent[lb[b]:lb[b + 1]], asg[a:e], perm[a:e] - a, lo, n_probe, k, base=a)
        assert_topk_exact(sim[a:e], idx[a:e], rs, ri, what=f"bucket {b}")


def test_ivf_exhaustive_probe_equals_bruteforce(ctx):
    """n_probe >= n_list: the IVF path must return the brute-force result."""
    import torch
    sizes, nlists = [1500, 800], [16, 8]
    off = np.concatenate([[0], np.cumsum(sizes)])
    X = unit_vectors(off[-1], 400, 33)
    idxr = ctx.ivf_build(torch.from_numpy(X).to(ctx.tdev), off, np.array(nlists, np.int32), kmeans_iters=2)
    sim, idx = idxr.search(16, 64)
    sim, idx = sim.cpu().numpy(), idx.cpu().numpy()
    for a, b in zip(off[:-1], off[1:]):
        rs, ri = fo.exhaustive_topk(X[a:b], 64, base=a)
        assert_topk_exact(sim[a:b], idx[a:b], rs, ri)


def _f16_planes(X):
    hi = X.astype(np.float16)
    lo = ((X - hi.astype(np.float32)) * np.float32(2048)).astype(np.float16)
    return hi, lo


@pytest.mark.parametrize("planes,d,k", [(2, 400, 128), (2, 400, 16), (1, 800, 64), (1, 400, 128), (2, 64, 8)])
def test_f16_mfma_flat_scan(ctx, planes, d, k):
    """f16 matrix-core scan of flat buckets: planes=2 (hi/lo split of float32 rows) must agree with
    the float32 brute force to north_star's 1e-5 (observed ~3e-7); planes=1 (config 5: float16
    vectors) must equal the brute force over the SAME float16 rows."""
    import torch
    sizes = [1, 31, 64, 65, 127, 128, 129, 300, 700, 1500]
    off = np.concatenate([[0], np.cumsum(sizes)])
    X = unit_vectors(off[-1], d, 17)
    hi, lo = _f16_planes(X)
    if planes == 2:
        X16 = np.ascontiguousarray(np.stack([hi, lo], 1))
        Xref = X
        Xdev = torch.from_numpy(X).to(ctx.tdev)
    else:
        X16 = hi
        Xref = hi.astype(np.float32)
        Xdev = None
    idxr = ctx.ivf_build(Xdev, off, np.ones(len(sizes), np.int32), X16=torch.from_numpy(X16).to(ctx.tdev))
    sim, idx = idxr.search(16, k)
    sim, idx = sim.cpu().numpy(), idx.cpu().numpy()
    worst = 0.0
    for a, b in zip(off[:-1], off[1:]):
        rs, ri = fo.exhaustive_topk(Xref[a:b], k, base=a)
        assert_topk_close(sim[a:b], idx[a:b], rs, ri, Xref[a:b], base=a, what=f"bucket {a}:{b}")
        ok = ri >= 0
        worst = max(worst, float(np.abs(sim[a:b][ok] - rs[ok]).max()))
    assert worst < 2e-6, worst


def test_vectorize_split16(ctx):
    """FAL_DTYPE_SPLIT16 = bit-exact hi/lo planes of the float32 vectors."""
    from falcon_amd import synth
    d = synth.generate(2000, seed=4)
    nb, start, _ = fo.get_dim(101, 1500, 0.05)
    V = ctx.vectorize(d["mz"], d["intensity"], d["indptr"], None, start, 0.05, nb, 400, dtype="split16").cpu().numpy()
    ref = fo.vectorize(d["mz"], d["intensity"], d["indptr"], start, 0.05, nb, 400)
    hi, lo = _f16_planes(ref)
    assert V.shape == (2000, 2, 400)
    assert np.array_equal(V[:, 0], hi) and np.array_equal(V[:, 1], lo)
    rec = V[:, 0].astype(np.float64) + V[:, 1].astype(np.float64) / 2048
    assert np.abs(rec - ref).max() < 3e-7


@pytest.mark.parametrize("k_ann,keep,tol,mode,rt_tol", [
    (128, 64, 20.0, "ppm", None),        # the bench configuration: a handful of survivors per row
    (128, 64, 5.0, "Da", None),          # everything passes: > 64 survivors -> truncation to n_neighbors
    (256, 200, 5.0, "Da", None),         # 4 keys per lane
    (128, 5, 300.0, "ppm", 30.0),        # RT filter, tiny n_neighbors
    (16, 64, 50.0, "ppm", None),         # n_neighbors > k_ann: padded rows
])
def test_search_neighbors_equals_search_then_filter(ctx, k_ann, keep, tol, mode, rt_tol):
    """a7+a8 fused (`fal_ivf_search_neighbors`) == `fal_ivf_search_topk` -> `fal_filter_neighbors`, bit for bit,
    on flat and IVF buckets (exact duplicates included: ties in similarity)."""
    import torch
    sizes, nlists = [1, 40, 700, 1500, 2600, 90], [1, 1, 1, 1, 8, 1]
    off = np.concatenate([[0], np.cumsum(sizes)])
    n = int(off[-1])
    X = unit_vectors(n, 400, 11)
    X[off[2]:off[2] + 6] = X[off[2]]
    rng = np.random.default_rng(5)
    mz = np.sort(500.0 + rng.random(n).astype(np.float32) * 0.05).astype(np.float32)    # ~100 ppm wide
    rt = (rng.random(n) * 100).astype(np.float32)
    idxr = ctx.ivf_build(torch.from_numpy(X).to(ctx.tdev), off, np.array(nlists, np.int32), kmeans_iters=3)
    mz_d, rt_d = torch.from_numpy(mz).to(ctx.tdev), torch.from_numpy(rt).to(ctx.tdev)
    sim, idx = idxr.search(4, k_ann)
    e_idx, e_dist = ctx.filter_neighbors(sim, idx, mz_d, rt_d, tol, mode, rt_tol, keep)
    g_idx, g_dist = idxr.search_neighbors(4, k_ann, mz_d, rt_d, tol, mode, rt_tol, keep)
    e_idx, e_dist, g_idx, g_dist = (t.cpu().numpy() for t in (e_idx, e_dist, g_idx, g_dist))
    assert (e_idx >= 0).sum() > 0
    assert np.array_equal(g_idx, e_idx)
    assert np.array_equal(g_dist.view(np.uint32), e_dist.view(np.uint32))


@pytest.mark.parametrize("sizes,nlists,d", [([3000], [64], 400), ([900, 40, 2500, 9000], [16, 1, 32, 128], 400),
                                           ([5000, 2100], [100, 33], 128), ([2600], [32], 64),
                                           # more than 128 lists: groups of 128 centroids + a merge over the groups (BASELINE configs[3])
                                           ([21000, 3000, 11000], [512, 64, 200], 400), ([9000, 5200], [129, 300], 128)])
def test_ivf_build_with_f16_prefilter_is_identical(ctx, sizes, nlists, d):
    """a6 with the float16 prefilter (assign16.hip: arg-max on the f16 matrix cores, close calls re-evaluated exactly):
    assignments, centroids and lists are bit-identical to the exact build -- and therefore to the oracle's."""
    import torch
    off = np.concatenate([[0], np.cumsum(sizes)])
    X = unit_vectors(off[-1], d, 13)
    X[50:60] = X[50]                                     # duplicates: exact ties between rows, equal sims to every centroid
    Xd = torch.from_numpy(X).to(ctx.tdev)
    exact = ctx.ivf_build(Xd, off, np.array(nlists, np.int32), kmeans_iters=5)
    pre = ctx.ivf_build(Xd, off, np.array(nlists, np.int32), kmeans_iters=5, Xkm=Xd.to(torch.float16).contiguous())
    for a, b in zip(exact.export(), pre.export()):
        assert torch.equal(a, b)
    C, ra, rperm, roff = fo.ivf_build(X[off[-2]:off[-1]], nlists[-1], 5)
    cent, asg, perm, loff = [t.cpu().numpy() for t in pre.export()]
    assert np.array_equal(asg[off[-2]:off[-1]], ra)
    lb = np.concatenate([[0], np.cumsum(nlists)])
    assert np.array_equal(cent[lb[-2]:lb[-1]], C)


def test_flat_scan_random_bucket_mixes(ctx):
    """the three forms of the fp32 flat scan on random mixes of bucket sizes (1 .. 1,400 rows: one-block buckets, partial and
    many 4-tile groups side by side in one launch, persistent workgroups pulling them in size order) and low_dims: top-k
    similarities and ids bit-identical to the brute-force oracle in every bucket"""
    import torch
    rng = np.random.default_rng(2026)
    for case in range(8):
        d = int(rng.choice([64, 128, 200, 256, 400, 400]))
        k = int(rng.choice([16, 64, 128]))
        nb = int(rng.integers(3, 30))
        sizes = [int(x) for x in np.concatenate([rng.integers(1, 40, nb // 2), rng.integers(33, 1400, nb - nb // 2)])]
        rng.shuffle(sizes)
        off = np.concatenate([[0], np.cumsum(sizes)])
        X = unit_vectors(off[-1], d, 100 + case)
        idxr = ctx.ivf_build(torch.from_numpy(X).to(ctx.tdev), off, np.ones(len(sizes), np.int32))
        sim, idx = idxr.search(16, k)
        sim, idx = sim.cpu().numpy(), idx.cpu().numpy()
        for a, b in zip(off[:-1], off[1:]):
            rs, ri = fo.exhaustive_topk(X[a:b], k, base=a)
            assert_topk_exact(sim[a:b], idx[a:b], rs, ri, what=f"case {case} d {d} bucket {a}:{b}")
