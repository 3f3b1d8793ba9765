"""`--low_dim` is a free integer in the reference (README.md:114-117); the cosine kernels are instantiated for rows of 64, 128,
256, 400 and 800 columns.  The path hashes modulo low_dim and stores the rows `row_width(low_dim)` columns wide with zeros behind
(`fal_vectorize_rows`): a zero column adds fma(0 * 0, acc) = acc to every k-ordered chain, so every kernel downstream runs at the
padded width and the oracle -- which pads the same way (`fo.row_width`) -- is matched bit for bit: vectors, k-means index, n_probe
search, neighbour lists, labels and medoids, on the staged and on the production path.  float32 rows of more than 400 columns
(low_dim 401..800) take the two-K-half forms of the fp32 matrix kernels (scan.hip dense4_kernel MODE 1 / 2, ivf_fine.hip)."""
import numpy as np
import pytest

from oracle import falcon_oracle as fo
from tests.test_gpu_pipeline import _check_stages
from tests.test_gpu_regimes import _check_index_equals_oracle, _check_production_equals_oracle, _dense_dataset

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ctx():
    from falcon_amd.device import Context
    c = Context(0)
    yield c
    c.close()


def test_row_width_rule(ctx):
    from falcon_amd import device
    from falcon_amd._lib import FalconHipError
    for low_dim in (1, 7, 64, 65, 128, 200, 256, 257, 400, 401, 504, 799, 800):
        assert device.row_width(low_dim) == fo.row_width(low_dim)
    assert [device.row_width(x) for x in (64, 128, 256, 400, 800)] == [64, 128, 256, 400, 800]
    for bad in (0, 801, 1024):
        with pytest.raises(FalconHipError):
            device.row_width(bad)


@pytest.mark.parametrize("low_dim", [1, 7, 10, 100, 200, 257, 399, 401, 504, 799])
def test_padded_rows_equal_the_oracle_bit_for_bit(ctx, low_dim):
    """hash modulo low_dim, rows `row_width(low_dim)` wide: float32 rows, their float16 rounding, the float16 vectors' image --
    all the oracle's bits; the columns behind low_dim are zero and the first low_dim columns are the unpadded vectors"""
    from falcon_amd import device, synth
    d = synth.generate(3000, seed=12)
    nb, start, _ = fo.get_dim(101.0, 1500.0, 0.05)
    W = device.row_width(low_dim)
    order = np.argsort(d["precursor_mz"], kind="stable")
    ref = fo.vectorize(d["mz"], d["intensity"], d["indptr"], start, 0.05, nb, low_dim, 0, True, order, width=W)
    assert ref.shape == (3000, W) and not ref[:, low_dim:].any() and ref[:, :low_dim].any(0).all()
    assert np.array_equal(ref[:, :low_dim], fo.vectorize(d["mz"], d["intensity"], d["indptr"], start, 0.05, nb, low_dim, 0, True, order))
    X = ctx.vectorize(d["mz"], d["intensity"], d["indptr"], order, start, 0.05, nb, low_dim, 0, True, "f32", width=W)
    assert np.array_equal(X.cpu().numpy(), ref)
    X, X16 = ctx.vectorize(d["mz"], d["intensity"], d["indptr"], order, start, 0.05, nb, low_dim, 0, True, "f32+f16", width=W)
    assert np.array_equal(X.cpu().numpy(), ref) and np.array_equal(X16.cpu().numpy(), ref.astype(np.float16))
    Xi, X16 = ctx.vectorize(d["mz"], d["intensity"], d["indptr"], order, start, 0.05, nb, low_dim, 0, True, "f16+image", width=W)
    assert np.array_equal(X16.cpu().numpy(), ref.astype(np.float16))
    assert np.array_equal(Xi.cpu().numpy(), ref.astype(np.float16).astype(np.float32))
    # the hash bins themselves: MurmurHash3 mod low_dim (bit-exact vs sklearn's, tests/test_oracle_golden.py)
    assert np.array_equal(device.hash_lookup(2000, low_dim), fo.hash_lookup(2000, low_dim))


@pytest.mark.parametrize("low_dim", [200, 504, 800])
def test_flat_buckets_staged_and_production(ctx, low_dim):
    """flat buckets of a few hundred rows (several 128-row groups of the shared-stream kernel; the 20 ppm gap rule cuts the
    windows further) + tiny ones: every stage against the oracle, then the production path.  504 / 800: float32 rows of 800 columns -- the
    two-K-half passes of dense4_kernel; buckets of up to 32 rows by exact chains."""
    from falcon_amd.cluster.cluster import AnnParams, ClusterPipeline
    from falcon_amd import synth
    from falcon_amd.cluster.cluster import SpectrumDataset
    a = synth.select_charge(synth.generate(5000, seed=81, mz_lo=600.0, mz_hi=610.0), 2)            # ~350 rows per window
    b = synth.select_charge(synth.generate(7000, seed=82, mz_lo=620.0, mz_hi=623.0), 2)            # ~1,600 rows per window
    c = synth.select_charge(synth.generate(300, seed=83, mz_lo=700.0, mz_hi=720.0), 2)             # ~10 rows per window
    parts = [a, b, c]
    d = {k: np.concatenate([p[k] for p in parts]) for k in ("precursor_mz", "retention_time", "mz", "intensity")}
    cnt = np.concatenate([np.diff(p["indptr"]) for p in parts])
    d["indptr"] = np.concatenate([[0], np.cumsum(cnt)]).astype(np.int64)
    ds = SpectrumDataset(d["precursor_mz"], d["retention_time"], d["mz"], d["intensity"], d["indptr"])
    p = AnnParams(low_dim=low_dim, n_probe=32)                # (n_probe 32: buckets of up to 2,495 rows stay flat)
    pipe = ClusterPipeline(ctx)
    labels, medoids = _check_stages(ctx, d, ds, 20.0, "ppm", None, 2 ** 15, p, pipe=pipe)
    L = pipe.last
    sizes = np.diff(L["splits"])
    assert (np.asarray(L["n_list"]) == 1).all() and sizes.max() > 500 and (sizes <= 32).sum() >= 5 and (sizes > 128).sum() >= 3, sorted(sizes)[-8:]
    assert L["X"].shape[1] == fo.row_width(low_dim)
    ref, rmed = fo.generate_clusters(d["mz"], d["intensity"], d["indptr"], d["precursor_mz"], d["retention_time"], low_dim=low_dim,
                                     n_probe=32)
    assert np.array_equal(labels, ref) and np.array_equal(medoids, rmed)
    prod = ClusterPipeline(ctx)
    pl, pm = prod.run(ds, 20.0, "ppm", None, 0.05, 2 ** 15, p)
    assert np.array_equal(pl.cpu().numpy(), ref) and np.array_equal(pm.cpu().numpy(), rmed)
    assert np.array_equal(prod.last["nb_idx"].cpu().numpy(), L["nb_idx"].cpu().numpy())
    assert np.array_equal(prod.last["nb_dist"].cpu().numpy(), L["nb_dist"].cpu().numpy())


@pytest.mark.parametrize("low_dim", [200, 504])
def test_128_list_regime_index_and_production_path(ctx, low_dim):
    """index + production path at low_dim 200 and 504 in the 128-list regime (three 1 m/z windows of ~9 k
    spectra, n_probe 16, k_ann 128): staged stages, the k-means index and the production path (float16 prefilters ON: rows of
    256 / 800 columns on assign16 / list16) against the oracle, bit for bit."""
    from falcon_amd.cluster.cluster import AnnParams, ClusterPipeline
    d, ds = _dense_dataset(39000, 600.0, 603.0, seed=71)
    p = AnnParams(low_dim=low_dim)
    pipe = ClusterPipeline(ctx)
    labels, medoids = _check_stages(ctx, d, ds, 20.0, "ppm", None, 2 ** 15, p, pipe=pipe)
    L = pipe.last
    assert (np.asarray(L["n_list"]) == 128).sum() >= 3 and L["X"].shape[1] == fo.row_width(low_dim)
    X = L["X"].cpu().numpy()
    assert _check_index_equals_oracle(L, X, p.kmeans_iters) >= 3
    ref, rmed = fo.generate_clusters(d["mz"], d["intensity"], d["indptr"], d["precursor_mz"], d["retention_time"], low_dim=low_dim)
    assert np.array_equal(labels, ref) and np.array_equal(medoids, rmed)
    _check_production_equals_oracle(ctx, ds, p, L, ref, rmed)


@pytest.mark.parametrize("low_dim", [200, 504])
def test_float16_vectors_at_an_off_grid_low_dim(ctx, low_dim):
    """`--dtype f16` at a low_dim between the instantiated widths: the production path == the oracle's float16 run (round, then
    the float32 path on 256 / 800-column rows)"""
    from falcon_amd.cluster.cluster import AnnParams, ClusterPipeline
    d, ds = _dense_dataset(26000, 600.0, 602.0, seed=77)
    p = AnnParams(dtype="f16", low_dim=low_dim)
    pipe = ClusterPipeline(ctx)
    labels, medoids = pipe.run(ds, 20.0, "ppm", None, 0.05, 2 ** 15, p)
    assert (np.asarray(pipe.last["n_list"]) == 128).sum() >= 2
    ref, rmed, im = fo.generate_clusters(d["mz"], d["intensity"], d["indptr"], d["precursor_mz"], d["retention_time"],
                                         low_dim=low_dim, dtype=np.float16, return_intermediates=True)
    assert np.array_equal(pipe.last["nb_idx"].cpu().numpy(), im["nb_idx"])
    splits, n_list = np.asarray(pipe.last["splits"]), np.asarray(pipe.last["n_list"])
    indexed = np.repeat(n_list > 1, np.diff(splits))
    gd = pipe.last["nb_dist"].cpu().numpy()
    assert np.array_equal(gd[indexed], im["nb_dist"][indexed])
    np.testing.assert_allclose(gd[~indexed], im["nb_dist"][~indexed], atol=2e-6, rtol=0)      # (flat: f16 matrix cores >= 64 rows)
    assert np.array_equal(labels.cpu().numpy(), ref) and np.array_equal(medoids.cpu().numpy(), rmed)
