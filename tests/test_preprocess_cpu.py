"""f1 on the CPU: the oracle's batch `process_spectra` against the host `process_spectrum` (both restate
reference spectrum.py:73-169; spectrum_utils is absent -> PARITY UNPINNED, the two restatements pin each other)."""
import numpy as np
import pytest

from oracle import falcon_oracle as fo
from tests.prep_cases import OPTION_SETS, raw_spectra


@pytest.mark.parametrize("opts", OPTION_SETS)
def test_oracle_process_spectra_matches_host_process_spectrum(opts):
    from falcon_amd.cluster.spectrum import process_spectrum
    mz, it, indptr, pmz, ch = raw_spectra(150, 3, max_peaks=400)
    valid, oip, omz, oit = fo.process_spectra(mz, it, indptr, pmz, ch, **opts)
    host_opts = dict(opts)
    if host_opts["scaling"] == "off":
        host_opts["scaling"] = None
    n_valid = 0
    for i in range(len(pmz)):
        spec = dict(identifier=str(i), precursor_mz=float(pmz[i]), precursor_charge=int(ch[i]) or None,
                    retention_time=0.0, mz=mz[indptr[i]:indptr[i + 1]], intensity=it[indptr[i]:indptr[i + 1]])
        out = process_spectrum(spec, **host_opts)
        assert (out is not None) == bool(valid[i]), i
        if out is None:
            assert oip[i + 1] == oip[i]
            continue
        n_valid += 1
        a, b = oip[i], oip[i + 1]
        assert np.array_equal(out["mz"], omz[a:b])                       # same peaks survive, bit for bit
        np.testing.assert_allclose(out["intensity"], oit[a:b], rtol=3e-6, atol=0)
        if np.any(it[indptr[i]:indptr[i + 1]] > 0):                      # (an all-zero spectrum normalises to NaN, as numpy does)
            assert abs(float(np.linalg.norm(oit[a:b].astype(np.float64))) - 1.0) < 1e-6
    assert 20 < n_valid < len(pmz)


def test_oracle_process_spectra_edge_cases():
    # empty batch, empty spectrum, single peak, unknown charge treated as 1+
    v, ip, m, i = fo.process_spectra(np.zeros(0), np.zeros(0, np.float32), np.zeros(1, np.int64), np.zeros(0), np.zeros(0, np.int32),
                                     5, 250.0)
    assert len(v) == 0 and ip.tolist() == [0]
    mz = np.array([100.0, 200.0, 500.0 - 1.0072766 + 1.0072766, 900.0], np.float64)
    it = np.array([1, 2, 3, 4], np.float32)
    v, ip, m, i = fo.process_spectra(mz, it, np.array([0, 0, 4], np.int64), np.array([500.0, 500.0]), np.array([0, 0], np.int32),
                                     2, 100.0, remove_precursor_tolerance=0.5)
    assert v.tolist() == [False, True] and ip.tolist() == [0, 0, 3]
    assert m.tolist() == [100.0, 200.0, 900.0]                           # the peak on the 1+ precursor is gone
