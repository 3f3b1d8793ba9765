"""Raw-spectrum generator shared by the f1 (preprocessing) tests."""
import numpy as np


def raw_spectra(n, seed, max_peaks=900):
    """-> mz f64 (sorted per spectrum), intensity f32, indptr, precursor_mz f64, charge i32 (0 = unknown).
    Covers: empty / tiny spectra, peaks outside the m/z window, peaks at every precursor charge state,
    intensity ties (quantised intensities), spectra longer than a wave's 64 lanes x several chunks."""
    rng = np.random.default_rng(seed)
    sizes = rng.integers(0, max_peaks, n)
    small = rng.random(n) < 0.1
    sizes[small] = rng.integers(0, 8, int(small.sum()))
    mz, it, pmz, ch = [], [], [], []
    for p in sizes:
        z = int(rng.integers(0, 5))
        precursor = float(rng.uniform(300, 1200))
        m = rng.uniform(50, 1900, p)
        zz = max(z, 1)
        neutral = (precursor - 1.0072766) * zz
        for c in range(1, zz + 1):                              # plant peaks on / next to the precursor ions
            if p > 4 * c:
                m[2 * c] = neutral / c + 1.0072766 + rng.uniform(-0.4, 0.4)
                m[2 * c + 1] = neutral / c + 1.0072766 + rng.choice([-1.5, 1.5, 1.4999, 1.5001])
        if p > 10:
            m[-1] = 101.0                                       # exactly on the window edge
            m[-2] = 1500.0
        m = np.sort(np.round(m, 4))
        i = np.round(rng.gamma(0.7, 1000.0, p), 0 if rng.random() < 0.5 else 2).astype(np.float32)   # ties
        if p > 3 and rng.random() < 0.2:
            i[:] = i[0]                                         # a flat spectrum: everything ties
        mz.append(m); it.append(i); pmz.append(precursor); ch.append(z)
    indptr = np.concatenate([[0], np.cumsum(sizes)]).astype(np.int64)
    cat = lambda xs, dt: np.concatenate(xs).astype(dt) if len(xs) else np.zeros(0, dt)
    return cat(mz, np.float64), cat(it, np.float32), indptr, np.array(pmz, np.float64), np.array(ch, np.int32)


OPTION_SETS = [
    dict(min_peaks=5, min_mz_range=250.0, mz_min=101.0, mz_max=1500.0, remove_precursor_tolerance=1.5,
         min_intensity=0.01, max_peaks_used=50, scaling="off"),       # falcon's defaults (config.py)
    dict(min_peaks=5, min_mz_range=250.0, mz_min=101.0, mz_max=1500.0, remove_precursor_tolerance=1.5,
         min_intensity=0.01, max_peaks_used=50, scaling="root"),
    dict(min_peaks=5, min_mz_range=250.0, mz_min=101.0, mz_max=1500.0, remove_precursor_tolerance=1.5,
         min_intensity=0.01, max_peaks_used=50, scaling="log"),
    dict(min_peaks=5, min_mz_range=250.0, mz_min=101.0, mz_max=1500.0, remove_precursor_tolerance=1.5,
         min_intensity=0.01, max_peaks_used=50, scaling="rank"),
    dict(min_peaks=1, min_mz_range=0.0, mz_min=None, mz_max=None, remove_precursor_tolerance=None,
         min_intensity=None, max_peaks_used=None, scaling="rank"),    # nothing filtered, rank over every peak
    dict(min_peaks=10, min_mz_range=100.0, mz_min=200.0, mz_max=None, remove_precursor_tolerance=0.5,
         min_intensity=None, max_peaks_used=150, scaling="root"),     # top-150 only
    dict(min_peaks=3, min_mz_range=10.0, mz_min=None, mz_max=900.0, remove_precursor_tolerance=None,
         min_intensity=0.05, max_peaks_used=None, scaling=None),      # base-peak filter only
]
