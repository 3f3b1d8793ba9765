"""Host-side spectrum preprocessing (reference falcon/cluster/spectrum.py:27-169).

`process_spectrum` is the step immediately before the hot path.  The
spectrum_utils 0.3.5 calls it makes are restated from SURVEY Appendix B [recollection];
PARITY UNPINNED (spectrum_utils is absent).  Bin geometry (`get_dim`) is served by the
library's host entry point (bit-compatible with spectrum.py:172-199).
"""
from __future__ import annotations

from typing import Dict, Optional

import numpy as np

from ..device import get_dim  # noqa: F401  (re-export: falcon.py:124 calls spectrum.get_dim)

PROTON = 1.0072766


def _valid(mz: np.ndarray, min_peaks: int, min_mz_range: float) -> bool:
    """spectrum.py:27-52"""
    return len(mz) >= min_peaks and mz[-1] - mz[0] >= min_mz_range


def process_spectrum(spec: Dict, min_peaks: int, min_mz_range: float, mz_min: Optional[float] = None,
                     mz_max: Optional[float] = None, remove_precursor_tolerance: Optional[float] = None,
                     min_intensity: Optional[float] = None, max_peaks_used: Optional[int] = None,
                     scaling: Optional[str] = None) -> Optional[Dict]:
    """spectrum.py:73-169 on a plain dict (identifier, precursor_mz, precursor_charge,
    retention_time, mz, intensity, filename)."""
    mz = np.asarray(spec["mz"], np.float64)
    it = np.asarray(spec["intensity"], np.float32)
    order = np.argsort(mz, kind="stable")
    mz, it = mz[order], it[order]
    keep = np.ones(len(mz), bool)                                            # set_mz_range (135)
    if mz_min is not None:
        keep &= mz >= mz_min
    if mz_max is not None:
        keep &= mz <= mz_max
    mz, it = mz[keep], it[keep]
    if not _valid(mz, min_peaks, min_mz_range):
        return None
    charge = spec.get("precursor_charge")
    if remove_precursor_tolerance is not None:                               # 139-149
        z = abs(int(charge)) if charge else 1
        neutral = (spec["precursor_mz"] - PROTON) * z
        keep = np.ones(len(mz), bool)
        for c in range(z, 0, -1):
            keep &= np.abs(mz - (neutral / c + PROTON)) > remove_precursor_tolerance
        mz, it = mz[keep], it[keep]
        if not _valid(mz, min_peaks, min_mz_range):
            return None
    if min_intensity is not None or max_peaks_used is not None:              # 151-155
        mi = 0.0 if min_intensity is None else min_intensity
        keep = it >= mi * it.max()
        if max_peaks_used is not None and keep.sum() > max_peaks_used:
            idx = np.flatnonzero(keep)
            top = idx[np.argsort(-it[idx], kind="stable")[:max_peaks_used]]
            keep = np.zeros(len(mz), bool)
            keep[top] = True
        mz, it = mz[keep], it[keep]
        if not _valid(mz, min_peaks, min_mz_range):
            return None
    if scaling == "root":                                                    # 157
        it = np.sqrt(it)
    elif scaling == "log":
        it = np.log2(1.0 + it)
    elif scaling == "rank":
        max_rank = max_peaks_used if max_peaks_used is not None else len(it)
        ranks = np.empty(len(it), np.float32)
        ranks[np.argsort(it, kind="stable")] = np.arange(len(it), dtype=np.float32) + 1
        it = (max_rank - (len(it) - ranks)).astype(np.float32)
    it = (it / np.linalg.norm(it)).astype(np.float32)                        # 158, 55-70
    return {
        "identifier": spec["identifier"], "precursor_mz": np.float32(spec["precursor_mz"]),
        "precursor_charge": charge, "mz": mz.astype(np.float32), "intensity": it,
        "retention_time": np.float32(spec.get("retention_time", -1)), "filename": spec.get("filename", ""),
    }
