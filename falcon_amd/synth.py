"""Deterministic synthetic MS/MS peak lists.

Spectra arrive the way the hot path receives them from `process_spectrum`
(reference spectrum.py:134-169): m/z inside [min_mz, max_mz], at most
`max_peaks_used` (50, config.py:169-171) peaks sorted by m/z, intensities
L2-normalised (spectrum.py:158).

Block `b` of up to 1,000,000 spectra is drawn from `default_rng([seed, b])`
(seed 42 = reference seed.py:6), so shards generate independently.
"""
from __future__ import annotations

import numpy as np

BLOCK = 1_000_000
N_TEMPLATE_PEAKS = 50
N_NOISE_PEAKS = 5
MAX_PEAKS = 50
SKEW_SIGMA = 1.0          # `skew=True`: log-normal occupancy of the 1 m/z precursor windows (sigma of the log)
SKEW_MIN_PEAKS = 5        # ... and 5..50 peaks per template instead of 50


def skew_window_weights(seed: int, mz_lo: float, mz_hi: float) -> np.ndarray:
    """`skew=True`: the probability of every 1 m/z precursor window of [mz_lo, mz_hi) -- LogNormal(0, SKEW_SIGMA) weights drawn
    from the seed alone (every block of a dataset uses the same windows), normalised.  With 800 windows the fullest holds
    ~70x the median's spectra: a 32 k-row window and a few of 8-14 k rows among 500-row ones at 1 M spectra (seed 42) (real precursor
    distributions are not uniform; SURVEY 8d's recipe is)."""
    n_win = max(1, int(np.ceil(mz_hi - mz_lo)))
    w = np.random.default_rng([seed, 0x5CE3]).lognormal(0.0, SKEW_SIGMA, n_win)
    return w / w.sum()


def _block(n: int, block: int, seed: int, mz_lo: float, mz_hi: float, skew: bool = False):
    rng = np.random.default_rng([seed, block])
    f32 = np.float32
    n_single = int(round(0.2 * n))
    sizes = 1 + rng.poisson(7, size=max(16, int((n - n_single) / 7.5) + 64))
    cs = np.cumsum(sizes)
    n_cl = int(np.searchsorted(cs, n - n_single, side="left")) + 1
    sizes = sizes[:n_cl].copy()
    sizes[-1] -= cs[n_cl - 1] - (n - n_single)
    if sizes[-1] <= 0:
        sizes = sizes[:-1]
    sizes = np.concatenate([sizes, np.ones(n - int(sizes.sum()), np.int64)])
    n_t = len(sizes)
    tmpl = np.repeat(np.arange(n_t), sizes)                      # template of every spectrum
    assert len(tmpl) == n
    # templates
    if skew:
        pw = skew_window_weights(seed, mz_lo, mz_hi)
        win = rng.choice(len(pw), size=n_t, p=pw)
        t_pmz = np.minimum(mz_lo + win + rng.random(n_t), np.nextafter(f32(mz_hi), f32(0))).astype(f32)
    else:
        t_pmz = rng.uniform(mz_lo, mz_hi, n_t).astype(f32)
    t_charge = np.where(rng.random(n_t) < 0.7, 2, 3).astype(np.int8)
    t_mz = np.sort(rng.uniform(101.0, 1500.0, (n_t, N_TEMPLATE_PEAKS)), axis=1).astype(f32)
    t_int = rng.lognormal(0.0, 1.0, (n_t, N_TEMPLATE_PEAKS)).astype(f32)
    t_valid = None
    if skew:                                                     # SKEW_MIN_PEAKS .. 50 peaks per template, a random subset
        k_t = rng.integers(SKEW_MIN_PEAKS, N_TEMPLATE_PEAKS + 1, n_t)
        ranks = np.argsort(np.argsort(rng.random((n_t, N_TEMPLATE_PEAKS)), axis=1), axis=1)
        t_valid = ranks < k_t[:, None]
    # members
    P = N_TEMPLATE_PEAKS + N_NOISE_PEAKS
    mz = np.empty((n, P), f32)
    it = np.empty((n, P), f32)
    mz[:, :N_TEMPLATE_PEAKS] = t_mz[tmpl] + rng.normal(0.0, 0.005, (n, N_TEMPLATE_PEAKS)).astype(f32)
    it[:, :N_TEMPLATE_PEAKS] = t_int[tmpl] * rng.lognormal(0.0, 0.2, (n, N_TEMPLATE_PEAKS)).astype(f32)
    keep = rng.random((n, P)) >= 0.10                            # 10 % peak dropout
    keep[:, N_TEMPLATE_PEAKS:] = True
    if t_valid is not None:
        keep[:, :N_TEMPLATE_PEAKS] &= t_valid[tmpl]
        keep[:, N_TEMPLATE_PEAKS:] = np.arange(N_NOISE_PEAKS)[None, :] < ((k_t[tmpl] + 9) // 10)[:, None]      # 1 noise peak per 10
    base = np.max(np.where(keep[:, :N_TEMPLATE_PEAKS], it[:, :N_TEMPLATE_PEAKS], 0), axis=1, keepdims=True)
    base = np.maximum(base, f32(1e-6))
    mz[:, N_TEMPLATE_PEAKS:] = rng.uniform(101.0, 1500.0, (n, N_NOISE_PEAKS)).astype(f32)
    it[:, N_TEMPLATE_PEAKS:] = (rng.uniform(0.0, 0.05, (n, N_NOISE_PEAKS)) * base).astype(f32)
    keep &= (mz >= f32(101.0)) & (mz <= f32(1500.0))             # set_mz_range (spectrum.py:135)
    # keep the MAX_PEAKS most intense (filter_intensity, spectrum.py:153)
    it_k = np.where(keep, it, -1.0)
    kth = np.partition(it_k, P - MAX_PEAKS, axis=1)[:, P - MAX_PEAKS][:, None]
    keep &= it_k >= kth
    # sort by m/z, invalid last
    key = np.where(keep, mz, np.inf)
    o = np.argsort(key, axis=1, kind="stable")
    mz = np.take_along_axis(mz, o, 1)
    it = np.take_along_axis(it, o, 1)
    keep = np.take_along_axis(keep, o, 1)
    it = np.where(keep, it, 0)
    nrm = np.sqrt((it.astype(np.float64) ** 2).sum(1, keepdims=True))
    it = (it / np.maximum(nrm, 1e-30)).astype(f32)               # _norm_intensity (spectrum.py:55-70)
    counts = keep.sum(1)
    pmz = (t_pmz[tmpl].astype(np.float64) * (1.0 + rng.normal(0.0, 3e-6, n))).astype(f32)
    rt = rng.uniform(0.0, 7200.0, n).astype(f32)
    charge = t_charge[tmpl]
    # shuffle so that nothing arrives pre-sorted
    perm = rng.permutation(n)
    keep, mz, it, counts = keep[perm], mz[perm], it[perm], counts[perm]
    return dict(mz=mz[keep], intensity=it[keep], counts=counts.astype(np.int64),
                precursor_mz=pmz[perm], retention_time=rt[perm], precursor_charge=charge[perm],
                truth=tmpl[perm].astype(np.int64))


def generate(n: int, seed: int = 42, first_block: int = 0, mz_lo: float = 400.0, mz_hi: float = 1200.0, skew: bool = False):
    """-> dict(mz f32[nnz], intensity f32[nnz], indptr i64[n+1], precursor_mz f32[n],
    retention_time f32[n], precursor_charge i8[n], truth i64[n]).
    `skew`: log-normal occupancy of the 1 m/z precursor windows and 5..50 peaks per spectrum (`skew_window_weights`); the
    default (False) is SURVEY 8d's recipe and its random stream, unchanged."""
    parts, done, b, t_off = [], 0, first_block, 0
    while done < n:
        m = min(BLOCK, n - done)
        p = _block(m, b, seed, mz_lo, mz_hi, skew)
        p["truth"] = p["truth"] + t_off
        t_off = int(p["truth"].max()) + 1
        parts.append(p)
        done += m
        b += 1
    cat = lambda k: np.concatenate([p[k] for p in parts])
    counts = cat("counts")
    indptr = np.zeros(n + 1, np.int64)
    np.cumsum(counts, out=indptr[1:])
    return dict(mz=cat("mz"), intensity=cat("intensity"), indptr=indptr,
                precursor_mz=cat("precursor_mz"), retention_time=cat("retention_time"),
                precursor_charge=cat("precursor_charge"), truth=cat("truth"))


def select_charge(data: dict, charge: int) -> dict:
    """Sub-dataset of one precursor charge (the per-charge partition of falcon.py:151-160)."""
    sel = np.flatnonzero(data["precursor_charge"] == charge)
    counts = np.diff(data["indptr"])[sel]
    indptr = np.zeros(len(sel) + 1, np.int64)
    np.cumsum(counts, out=indptr[1:])
    src = np.repeat(data["indptr"][:-1][sel] - indptr[:-1], counts) + np.arange(int(counts.sum()))
    out = {k: data[k][sel] for k in ("precursor_mz", "retention_time", "precursor_charge", "truth")}
    out.update(mz=data["mz"][src], intensity=data["intensity"][src], indptr=indptr, rows=sel)
    return out


# ---------------------------------------------------------------------------------------------
# The same recipe on the GPU (torch, plumbing only): bench.py generates its workloads here because the
# numpy generator needs ~20 s of host time per million spectra.  Same distributions, parameters and
# layout as `_block` / `generate` / `select_charge`; a different random stream (torch's Philox), so the
# spectra are statistically -- not bitwise -- the numpy ones.  Parity tests keep the numpy generator
# (the oracle runs on its output); tests/test_host_logic.py::test_device_generator_matches_the_numpy_recipe_statistically compares the statistics of the two.
# ---------------------------------------------------------------------------------------------
def _block_device(n: int, block: int, seed: int, mz_lo: float, mz_hi: float, dev, skew: bool = False):
    import torch
    g = torch.Generator(device=dev)
    g.manual_seed(int(seed) * 1_000_003 + int(block))
    f32 = torch.float32

    def uni(lo, hi, shape):
        return torch.rand(shape, generator=g, device=dev, dtype=torch.float64) * (hi - lo) + lo

    def randn(shape, dtype=f32):
        return torch.randn(shape, generator=g, device=dev, dtype=dtype)

    n_single = int(round(0.2 * n))
    m = max(16, int((n - n_single) / 7.5) + 64)
    sizes = 1 + torch.poisson(torch.full((m,), 7.0, device=dev), generator=g).long()
    cs = torch.cumsum(sizes, 0)
    n_cl = int(torch.searchsorted(cs, torch.tensor([n - n_single], device=dev)).item()) + 1
    sizes = sizes[:n_cl].clone()
    sizes[-1] -= cs[n_cl - 1] - (n - n_single)
    if int(sizes[-1].item()) <= 0:
        sizes = sizes[:-1]
    sizes = torch.cat([sizes, torch.ones(n - int(sizes.sum().item()), dtype=torch.long, device=dev)])
    n_t = sizes.numel()
    tmpl = torch.repeat_interleave(torch.arange(n_t, device=dev), sizes)
    if skew:                                                                           # (the numpy generator's windows)
        pw = torch.from_numpy(skew_window_weights(seed, mz_lo, mz_hi)).to(dev)
        win = torch.multinomial(pw, n_t, replacement=True, generator=g)
        t_pmz = (mz_lo + win.double() + torch.rand(n_t, generator=g, device=dev, dtype=torch.float64)).to(f32)
        t_pmz = torch.minimum(t_pmz, torch.nextafter(torch.tensor(mz_hi, dtype=f32, device=dev), torch.zeros((), dtype=f32, device=dev)))
    else:
        t_pmz = uni(mz_lo, mz_hi, (n_t,)).to(f32)
    t_charge = torch.where(torch.rand(n_t, generator=g, device=dev) < 0.7, 2, 3).to(torch.int8)
    t_mz = torch.sort(uni(101.0, 1500.0, (n_t, N_TEMPLATE_PEAKS)), dim=1).values.to(f32)
    t_int = torch.exp(randn((n_t, N_TEMPLATE_PEAKS)))                                  # LogNormal(0, 1)
    t_valid = None
    if skew:
        k_t = torch.randint(SKEW_MIN_PEAKS, N_TEMPLATE_PEAKS + 1, (n_t,), generator=g, device=dev)
        ranks = torch.argsort(torch.argsort(torch.rand((n_t, N_TEMPLATE_PEAKS), generator=g, device=dev), dim=1), dim=1)
        t_valid = ranks < k_t[:, None]
    P, T = N_TEMPLATE_PEAKS + N_NOISE_PEAKS, N_TEMPLATE_PEAKS
    mz = torch.empty((n, P), dtype=f32, device=dev)
    it = torch.empty((n, P), dtype=f32, device=dev)
    mz[:, :T] = t_mz[tmpl] + randn((n, T)) * 0.005
    it[:, :T] = t_int[tmpl] * torch.exp(randn((n, T)) * 0.2)
    keep = torch.rand((n, P), generator=g, device=dev) >= 0.10                         # 10 % peak dropout
    keep[:, T:] = True
    if t_valid is not None:
        keep[:, :T] &= t_valid[tmpl]
        keep[:, T:] = torch.arange(N_NOISE_PEAKS, device=dev)[None, :] < ((k_t[tmpl] + 9) // 10)[:, None]
    base = torch.where(keep[:, :T], it[:, :T], torch.zeros((), device=dev)).amax(1, keepdim=True).clamp_min(1e-6)
    mz[:, T:] = uni(101.0, 1500.0, (n, N_NOISE_PEAKS)).to(f32)
    it[:, T:] = (uni(0.0, 0.05, (n, N_NOISE_PEAKS)) * base.double()).to(f32)
    keep &= (mz >= 101.0) & (mz <= 1500.0)
    it_k = torch.where(keep, it, torch.full((), -1.0, device=dev))
    kth = torch.kthvalue(it_k, P - MAX_PEAKS + 1, dim=1, keepdim=True).values         # keep the MAX_PEAKS most intense
    keep &= it_k >= kth
    key = torch.where(keep, mz, torch.full((), float("inf"), device=dev))
    o = torch.argsort(key, dim=1, stable=True)
    mz, it, keep = torch.gather(mz, 1, o), torch.gather(it, 1, o), torch.gather(keep, 1, o)
    it = torch.where(keep, it, torch.zeros((), device=dev))
    nrm = torch.sqrt((it.double() ** 2).sum(1, keepdim=True)).clamp_min(1e-30)
    it = (it.double() / nrm).to(f32)
    counts = keep.sum(1)
    pmz = (t_pmz[tmpl].double() * (1.0 + randn((n,), torch.float64) * 3e-6)).to(f32)
    rt = uni(0.0, 7200.0, (n,)).to(f32)
    charge = t_charge[tmpl]
    perm = torch.randperm(n, generator=g, device=dev)
    keep, mz, it, counts = keep[perm], mz[perm], it[perm], counts[perm]
    return dict(mz=mz[keep], intensity=it[keep], counts=counts.long(), precursor_mz=pmz[perm], retention_time=rt[perm],
                precursor_charge=charge[perm], truth=tmpl[perm].long())


def generate_device(n: int, device, seed: int = 42, first_block: int = 0, mz_lo: float = 400.0, mz_hi: float = 1200.0,
                    skew: bool = False):
    """`generate` on a torch device -> the same dict with device tensors (indptr i64[n+1])."""
    import torch
    parts, done, b, t_off = [], 0, first_block, 0
    while done < n:
        m = min(BLOCK, n - done)
        p = _block_device(m, b, seed, mz_lo, mz_hi, device, skew)
        p["truth"] = p["truth"] + t_off
        t_off = int(p["truth"].max().item()) + 1
        parts.append(p)
        done += m
        b += 1
    cat = lambda k: torch.cat([p[k] for p in parts])
    indptr = torch.zeros(n + 1, dtype=torch.int64, device=device)
    torch.cumsum(cat("counts"), 0, out=indptr[1:])
    return dict(mz=cat("mz"), intensity=cat("intensity"), indptr=indptr, precursor_mz=cat("precursor_mz"),
                retention_time=cat("retention_time"), precursor_charge=cat("precursor_charge"), truth=cat("truth"))


def select_charge_device(data: dict, charge: int) -> dict:
    """`select_charge` for the device form."""
    import torch
    sel = torch.nonzero(data["precursor_charge"] == charge).flatten()
    ip = data["indptr"]
    counts = (ip[1:] - ip[:-1])[sel]
    indptr = torch.zeros(sel.numel() + 1, dtype=torch.int64, device=ip.device)
    torch.cumsum(counts, 0, out=indptr[1:])
    src = torch.repeat_interleave(ip[:-1][sel] - indptr[:-1], counts) + torch.arange(int(indptr[-1].item()), device=ip.device)
    out = {k: data[k][sel] for k in ("precursor_mz", "retention_time", "precursor_charge", "truth")}
    out.update(mz=data["mz"][src], intensity=data["intensity"][src], indptr=indptr, rows=sel)
    return out
