import sys, numpy as np
sys.path.insert(0, ".")
from oracle import falcon_oracle as fo
from tests.prep_cases import OPTION_SETS, raw_spectra
from falcon_amd.device import Context
ctx = Context(0)
mz, it, indptr, pmz, ch = raw_spectra(1200, 17, max_peaks=900)
for oi, opts in enumerate(OPTION_SETS[:1]):
    e = fo.process_spectra(mz, it, indptr, pmz, ch, **opts)
    g = [t.cpu().numpy() for t in ctx.process_spectra(mz, it, indptr, pmz, ch, **opts)]
    bad = np.flatnonzero(g[0] != e[0])
    print("opts", oi, "valid mismatches", len(bad), bad[:10])
    cg, ce = np.diff(g[1]), np.diff(e[1])
    badc = np.flatnonzero(cg != ce)
    print("count mismatches", len(badc), badc[:10], cg[badc[:10]], ce[badc[:10]])
    for i in list(bad[:3]) + list(badc[:3]):
        a, b = indptr[i], indptr[i + 1]
        print("spec", i, "raw peaks", b - a, "charge", ch[i], "pmz", pmz[i], "gpu", g[0][i], cg[i], "oracle", e[0][i], ce[i])
        # step-by-step oracle counts
        m = mz[a:b]; k = (m >= 101.0) & (m <= 1500.0); print("  range keep", k.sum(), "span", m[k].max() - m[k].min() if k.any() else None)
