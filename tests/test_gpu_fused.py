"""The flat-bucket cosine scan with the top-k kept on chip (fused.hip: f16-MFMA prefilter + exact float32 refinement of
the precursor window + exact fallback) must give BIT-IDENTICAL neighbour lists to the staged path
(`fal_ivf_search_topk` -> `fal_filter_neighbors`), which the other tests pin to the oracle."""
import numpy as np
import pytest

from tests.test_gpu_search import unit_vectors

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ctx():
    from falcon_amd.device import Context
    c = Context(0)
    yield c
    c.close()


def _staged_and_fused(ctx, X, off, mz, rt, k_ann, keep, tol, mode, rt_tol, n_probe=16):
    import torch
    nl = np.ones(len(off) - 1, np.int32)
    Xd = torch.from_numpy(X).to(ctx.tdev)
    mz_d = torch.from_numpy(mz).to(ctx.tdev)
    rt_d = None if rt is None else torch.from_numpy(rt).to(ctx.tdev)
    plain = ctx.ivf_build(Xd, off, nl)
    sim, idx = plain.search(n_probe, k_ann)
    e_idx, e_dist = ctx.filter_neighbors(sim, idx, mz_d, rt_d, tol, mode, rt_tol, keep)
    pre = ctx.ivf_build(Xd, off, nl, Xpre=Xd.to(torch.float16).contiguous())
    g_idx, g_dist = pre.search_neighbors(n_probe, k_ann, mz_d, rt_d, tol, mode, rt_tol, keep)
    cnt = pre.nb_count.cpu().numpy()
    ctx.sync()
    n_fallback = ctx.counter(5)
    e_idx, e_dist, g_idx, g_dist = (t.cpu().numpy() for t in (e_idx, e_dist, g_idx, g_dist))
    bad = np.flatnonzero((g_idx != e_idx).any(1) | (g_dist.view(np.uint32) != e_dist.view(np.uint32)).any(1))
    assert len(bad) == 0, (len(bad), bad[:10], g_idx[bad[0]][:12], e_idx[bad[0]][:12], g_dist[bad[0]][:6], e_dist[bad[0]][:6])
    assert np.array_equal(cnt, (e_idx >= 0).sum(1))
    return e_idx, n_fallback


@pytest.mark.parametrize("d,k_ann,keep,tol,mode,rt_tol", [
    (400, 128, 64, 20.0, "ppm", None),          # the production setting
    (400, 128, 64, 20.0, "ppm", 30.0),          # + retention time tolerance
    (400, 32, 8, 40.0, "ppm", None),            # small k: the threshold cuts deep into the window
    (400, 128, 64, 0.05, "Da", None),           # Da: wide windows (whole small buckets)
    (400, 200, 100, 10.0, "ppm", None),         # k_ann > 128
    (64, 64, 16, 20.0, "ppm", None), (128, 128, 64, 20.0, "ppm", None), (256, 100, 50, 20.0, "ppm", None),
])
def test_fused_neighbours_bit_identical_to_the_staged_path(ctx, d, k_ann, keep, tol, mode, rt_tol):
    sizes = [1, 2, 31, 33, 100, 129, 130, 257, 700, 1248, 1500, 2400, 5, 640]
    off = np.concatenate([[0], np.cumsum(sizes)])
    n = int(off[-1])
    X = unit_vectors(n, d, 17, noise=0.35)
    X[off[8]:off[8] + 6] = X[off[8]]                                 # exact duplicates -> exact ties
    rng = np.random.default_rng(3)
    # precursor-sorted rows, ~1 m/z per bucket like the production windows
    mz = np.concatenate([np.sort(500.0 + b + rng.random(s)) for b, s in enumerate(sizes)]).astype(np.float32)
    rt = (rng.random(n) * 100).astype(np.float32)
    e_idx, n_fb = _staged_and_fused(ctx, X, off, mz, rt if rt_tol is not None else None, k_ann, keep, tol, mode, rt_tol)
    assert (e_idx >= 0).sum() > n // 4
    if mode == "ppm":
        assert n_fb < 0.2 * n, n_fb                                   # the on-chip path did the work, not the fallback


def test_fused_handles_ties_zero_rows_and_crowded_bins(ctx):
    """hundreds of identical spectra (every similarity 1.0 -> one histogram bin, more members than the chip keeps),
    all-zero rows (every similarity 0.0) and near-duplicates: everything funnels through the exact fallback and
    still equals the staged path bit for bit."""
    sizes = [900, 400, 1300]
    off = np.concatenate([[0], np.cumsum(sizes)])
    n = int(off[-1])
    X = unit_vectors(n, 400, 23)
    X[100:420] = X[100]                                               # 320 identical rows
    X[500:560] = 0                                                    # empty spectra
    X[off[1]:off[2]] = X[off[1]]                                      # a bucket of identical rows
    base = X[off[2]].copy()
    for i in range(200):                                              # near-duplicates: dense similarities just below 1
        v = base.copy()
        v[(7 * i) % 400] += 1e-3 * (i + 1)
        X[off[2] + 1 + i] = v / np.linalg.norm(v)
    rng = np.random.default_rng(4)
    mz = np.concatenate([np.sort(600.0 + b + 0.02 * rng.random(s)) for b, s in enumerate(sizes)]).astype(np.float32)
    _, n_fb = _staged_and_fused(ctx, X, off, mz, None, 128, 64, 20.0, "ppm", None)
    assert n_fb > 500


def test_pipeline_with_and_without_prefilter_is_identical(ctx):
    """whole path on synthetic spectra: the prefilter changes nothing but the time"""
    import dataclasses
    from falcon_amd import synth
    from falcon_amd.cluster.cluster import AnnParams, ClusterPipeline, SpectrumDataset
    d = synth.select_charge(synth.generate(60000, seed=7, mz_lo=500.0, mz_hi=560.0), 2)
    ds = SpectrumDataset(d["precursor_mz"], d["retention_time"], d["mz"], d["intensity"], d["indptr"])
    pipe = ClusterPipeline(ctx)
    outs = []
    for pre in (True, False):
        p = AnnParams(prefilter=pre)
        lab, med = pipe.run(ds, 20.0, "ppm", None, 0.05, 2 ** 15, p)
        outs.append((lab.cpu().numpy(), med.cpu().numpy(), pipe.last["nb_idx"].cpu().numpy(),
                     pipe.last["nb_dist"].cpu().numpy().view(np.uint32), ctx.counter(5)))
    assert np.array_equal(outs[0][0], outs[1][0]) and np.array_equal(outs[0][1], outs[1][1])
    assert np.array_equal(outs[0][2], outs[1][2]) and np.array_equal(outs[0][3], outs[1][3])
    assert outs[0][4] < 0.05 * len(ds)                                # few queries needed the exact fallback


def test_prefilter_error_bound_has_margin(ctx):
    """the bound the kernel relies on, |f16-MFMA similarity - exact| <= 1.3e-3 * approx + 2e-6, measured on hashed
    spectra with the f16 scan kernel itself (float16 rows, float32 accumulation) against float64."""
    import torch
    from falcon_amd import device as _device, synth
    d = synth.select_charge(synth.generate(6000, seed=9, mz_lo=700.0, mz_hi=701.0), 2)
    n_bins, start, _ = _device.get_dim(101.0, 1500.0, 0.05)
    X = ctx.vectorize(d["mz"], d["intensity"], d["indptr"], None, start, 0.05, n_bins, 400, 0, True, "f32")
    X16 = ctx.vectorize(d["mz"], d["intensity"], d["indptr"], None, start, 0.05, n_bins, 400, 0, True, "f16")
    n = X.shape[0]
    idx = ctx.ivf_build(None, np.array([0, n]), np.array([1], np.int32), X16=X16)
    sim, ids = idx.search(1, 128)
    exact = (X.double() @ X.double().T)
    ref = torch.gather(exact, 1, ids.long())
    err = (sim.double() - ref).abs()
    bound = 1.3e-3 * sim.double().abs() + 2e-6
    ratio = float((err / bound).max())
    # the float16 rounding of the two factors alone reaches 2 * 2^-11 = 9.8e-4 when few peaks overlap: the bound is
    # tight by construction; what is left covers the float32 accumulation
    assert ratio < 0.9, ratio
