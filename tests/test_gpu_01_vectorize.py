"""GPU parity: a1-a3 (bin geometry, bin index, feature hashing, L2) vs the oracle and
vs the reference goldens.  Bit-exact for the integer work AND for the float32 vectors
(ordered adds + fixed-order norm make the kernel reproduce the oracle exactly)."""
import numpy as np
import pytest

from oracle import falcon_oracle as fo

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ctx():
    from falcon_amd.device import Context
    c = Context(0)
    yield c
    c.close()


def test_bin_indices_golden(ctx, ref_golden):
    g = ref_golden
    out = ctx.to_vector_indices(g["tv_mz"], float(g["tv_min_mz"]), float(g["tv_bin_size"])).cpu().numpy()
    assert np.array_equal(out, g["tv_indices"])


def test_bin_indices_boundaries(ctx):
    """every bin edge +- 1 ulp (needs true float64 division)."""
    _, start, _ = fo.get_dim(101, 1500, 0.05)
    edges = (start + np.arange(0, 27982) * 0.05).astype(np.float32)
    mz = np.concatenate([edges, np.nextafter(edges, np.float32(0)), np.nextafter(edges, np.float32(1e9))])
    out = ctx.to_vector_indices(mz, start, 0.05).cpu().numpy()
    assert np.array_equal(out, fo.bin_indices(mz, start, 0.05))


def test_vectorize_golden_unnormalised(ctx, ref_golden):
    g = ref_golden
    V = ctx.vectorize(g["tv_mz"], g["tv_intensity"], g["tv_indptr"], None, float(g["tv_min_mz"]),
                      float(g["tv_bin_size"]), 27982, 400, normalize=False).cpu().numpy()
    ref = g["tv_vectors_unnorm_400"]
    assert np.array_equal(V != 0, ref != 0)          # hash bins bit-exact
    np.testing.assert_allclose(V, ref, rtol=3e-7, atol=0)
    assert np.array_equal(V, fo.vectorize(g["tv_mz"], g["tv_intensity"], g["tv_indptr"], float(g["tv_min_mz"]),
                                          float(g["tv_bin_size"]), 27982, 400, norm=False))


@pytest.mark.parametrize("low_dim,dtype", [(400, "f32"), (800, "f32"), (64, "f32"), (1024, "f32"),
                                            (400, "f16"), (800, "f16")])
def test_vectorize_vs_oracle(ctx, low_dim, dtype):
    from falcon_amd import synth
    d = synth.generate(3000, seed=7)
    # ragged extras: empty spectrum, >64 peaks (two chunks), many collisions, out-of-range peaks
    rng = np.random.default_rng(1)
    extra_mz = [np.zeros(0, np.float32),
                np.sort(rng.uniform(101, 1500, 150)).astype(np.float32),
                np.full(40, 500.02, np.float32),
                np.array([20.0, 100.0, 300.0, 1500.0, 1500.2, 9000.0], np.float32)]
    mz = np.concatenate([d["mz"]] + extra_mz)
    it = np.concatenate([d["intensity"]] + [rng.lognormal(0, 1, len(x)).astype(np.float32) for x in extra_mz])
    indptr = np.concatenate([d["indptr"], d["indptr"][-1] + np.cumsum([len(x) for x in extra_mz])])
    n = len(indptr) - 1
    order = rng.permutation(n)
    nb, start, _ = fo.get_dim(101, 1500, 0.05)
    V = ctx.vectorize(mz, it, indptr, order, start, 0.05, nb, low_dim, dtype=dtype).cpu().numpy()
    ref = fo.vectorize(mz, it, indptr, start, 0.05, nb, low_dim, row_order=order,
                       dtype=np.float16 if dtype == "f16" else np.float32)
    assert V.dtype == ref.dtype and V.shape == (n, low_dim)
    assert np.array_equal(V, ref)
    # row-wise norms
    nrm = np.linalg.norm(V.astype(np.float64), axis=1)
    nz = np.diff(indptr)[order] > 0
    tol = 2e-3 if dtype == "f16" else 3e-7
    assert np.all(np.abs(nrm[nz & (nrm > 0)] - 1) < tol)


@pytest.mark.parametrize("low_dim", [64, 400, 800])
def test_vectorize_pair_equals_the_two_single_outputs(ctx, low_dim):
    """`fal_vectorize_pair`: float32 rows and their float16 rounding from ONE pass over the peaks == the two separate calls,
    bit for bit, with a row order (gather) and empty spectra in the input"""
    import torch
    from falcon_amd import synth
    d = synth.generate(3000, seed=5)
    _, start, _ = fo.get_dim(101, 1500, 0.05)
    order = torch.from_numpy(np.random.default_rng(1).permutation(3000)[:2500].astype(np.int64))
    args = (d["mz"], d["intensity"], d["indptr"], order, start, 0.05, 27982, low_dim, 0, True)
    a32, a16 = ctx.vectorize(*args, "f32+f16")
    b32, b16 = ctx.vectorize(*args, "f32"), ctx.vectorize(*args, "f16")
    assert torch.equal(a32.view(torch.int32), b32.view(torch.int32)) and torch.equal(a16.view(torch.int16), b16.view(torch.int16))


def test_host_helpers(ref_golden):
    from falcon_amd.device import get_dim, hash_lookup
    g = ref_golden
    for (lo, hi, b), dim, (s, e) in zip(g["get_dim_in"], g["get_dim_dim"], g["get_dim_start_end"]):
        d, start, end = get_dim(lo, hi, b)
        assert d == dim and np.float32(start) == s and np.float32(end) == e
    assert np.array_equal(hash_lookup(27982, 400), g["tv_hash_lookup_400"])
