"""f1 on the GPU: `fal_process_spectra` against the oracle's `process_spectra` (reference spectrum.py:73-169)."""
import numpy as np
import pytest

from oracle import falcon_oracle as fo
from tests.prep_cases import OPTION_SETS, raw_spectra

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ctx():
    from falcon_amd.device import Context
    c = Context(0)
    yield c
    c.close()


def _same_f32(a, b):
    return np.array_equal(a.view(np.uint32), b.view(np.uint32))


@pytest.mark.parametrize("opts", OPTION_SETS)
def test_process_spectra_matches_oracle(ctx, opts):
    mz, it, indptr, pmz, ch = raw_spectra(1200, 17, max_peaks=900)
    e_valid, e_ip, e_mz, e_it = fo.process_spectra(mz, it, indptr, pmz, ch, **opts)
    valid, ip, omz, oit = ctx.process_spectra(mz, it, indptr, pmz, ch, **opts)
    valid, ip, omz, oit = valid.cpu().numpy(), ip.cpu().numpy(), omz.cpu().numpy(), oit.cpu().numpy()
    assert 100 < e_valid.sum() < len(e_valid)
    assert np.array_equal(valid, e_valid)                     # the same spectra survive
    assert np.array_equal(ip, e_ip)                           # with the same number of peaks
    assert _same_f32(omz, e_mz)                               # the same peaks, bit for bit
    if opts["scaling"] == "log":                              # log2 comes from two libms: last-bit differences allowed
        np.testing.assert_allclose(oit, e_it, rtol=3e-7, atol=0)
    else:
        assert np.array_equal(np.isnan(oit), np.isnan(e_it))
        ok = ~np.isnan(e_it)
        assert _same_f32(oit[ok], e_it[ok])                   # intensities bit-identical


def test_process_spectra_long_and_degenerate(ctx):
    """a spectrum of 20,000 peaks (hundreds of 64-lane chunks), empty spectra, an empty batch."""
    rng = np.random.default_rng(2)
    big = np.sort(rng.uniform(100, 1500, 20000))
    mz = np.concatenate([big, [150.0, 700.0]])
    it = np.concatenate([rng.gamma(0.7, 1000.0, 20000), [5.0, 6.0]]).astype(np.float32)
    indptr = np.array([0, 0, 20000, 20000, 20002], np.int64)
    pmz, ch = np.array([500.0, 640.0, 500.0, 500.0]), np.array([2, 3, 0, 1], np.int32)
    opts = dict(min_peaks=2, min_mz_range=100.0, mz_min=101.0, mz_max=1500.0, remove_precursor_tolerance=1.5,
                min_intensity=0.001, max_peaks_used=300, scaling="rank")
    e = fo.process_spectra(mz, it, indptr, pmz, ch, **opts)
    g = [t.cpu().numpy() for t in ctx.process_spectra(mz, it, indptr, pmz, ch, **opts)]
    assert e[0].tolist() == [False, True, False, True] and np.array_equal(g[0], e[0]) and np.array_equal(g[1], e[1])
    assert _same_f32(g[2], e[2]) and _same_f32(g[3], e[3])
    v, ip, m, i = ctx.process_spectra(np.zeros(0), np.zeros(0, np.float32), np.zeros(1, np.int64), np.zeros(0),
                                      np.zeros(0, np.int32), 5, 250.0)
    assert v.numel() == 0 and ip.cpu().tolist() == [0] and m.numel() == 0
