"""BASELINE configs at full size through size-independent properties: the oracle cannot follow at these sizes, the
invariants of the path can.  configs[1]: 1,000,000 spectra (numpy generator, the data of the parity tests);
configs[2]'s dataset on ONE GPU: 10,000,000 spectra float32 (the north star's target size; IVF regime, n_list 128);
configs[4]: 10,000,000 spectra, low_dim 800, float16.  The 10 M datasets come from the device generator
(falcon_amd.synth.generate_device: the same recipe on the GPU)."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _parts(ctx, data, select):
    import torch
    from falcon_amd.cluster.cluster import SpectrumDataset
    parts = []
    for ch in (2, 3):
        c = select(data, ch)
        parts.append(SpectrumDataset(*[ctx.to_dev(c[k], torch.float32) for k in ("precursor_mz", "retention_time", "mz", "intensity")],
                                     ctx.to_dev(c["indptr"], torch.int64)))
    return parts


def _check_partition(ctx, ds, lab, med, lab2, med2, last, p, dist_tol, check_values=2000, shard=False):
    """`shard`: (lab, med, last) are one share of a bucket-sharded run -- labels / medoids / neighbour ids refer to the share's
    rows in sorted order, last["rows"] maps them to dataset rows."""
    import torch
    n = int(lab.numel())
    if shard:
        last = dict(last, order=last["rows"])
        sorted_pos = torch.arange(n, device=lab.device)
    # determinism: the same inputs give the same bits (no atomics-order dependence anywhere on the path)
    assert torch.equal(lab, lab2) and torch.equal(med, med2)
    # label contract (cluster.py:144-155): dense ids, no -1, medoids[c] represents cluster c
    assert int(lab.min()) == 0 and int(lab.max()) == med.numel() - 1
    assert torch.equal(lab[med.long()], torch.arange(med.numel(), device=lab.device, dtype=lab.dtype))
    counts = torch.bincount(lab.long(), minlength=med.numel())
    assert int(counts.min()) >= 1
    nb_idx, nb_dist, cnt, order = last["nb_idx"], last["nb_dist"], last["nb_count"], last["order"]
    # neighbour lists: front-packed, sorted by (distance, id), inside the precursor tolerance, never the row itself
    k = nb_idx.shape[1]
    col = torch.arange(k, device=nb_idx.device)[None, :]
    valid = nb_idx >= 0
    assert torch.equal(valid, col < cnt[:, None])
    rows = torch.arange(n, device=nb_idx.device)[:, None].expand(-1, k)
    assert not bool((nb_idx == rows)[valid].any())
    d0, d1 = nb_dist[:, :-1], nb_dist[:, 1:]
    both = valid[:, 1:]
    assert bool((d0[both] <= d1[both]).all())
    assert bool(((nb_dist[valid] >= 0) & (nb_dist[valid] <= 1)).all())
    mz_sorted = ds.precursor_mz[order]
    assert bool((mz_sorted[1:] >= mz_sorted[:-1]).all())
    i_idx = rows[valid]
    j_idx = nb_idx[valid].long()
    dist_ij = nb_dist[valid]
    ppm = (mz_sorted[i_idx] - mz_sorted[j_idx]).abs() / mz_sorted[j_idx] * 1e6
    assert float(ppm.max()) <= 20.0 + 1e-3
    # symmetry of the scan: where i stores j AND j stores i, the two distances are the same bits
    key = i_idx * n + j_idx
    rkey = j_idx * n + i_idx
    skey, perm = torch.sort(key)
    pos = torch.searchsorted(skey, rkey).clamp(max=skey.numel() - 1)
    mutual = skey[pos] == rkey
    assert int(mutual.sum()) > 1000
    assert torch.equal(dist_ij[mutual], dist_ij[perm][pos][mutual])
    del key, rkey, skey, perm, pos
    # the stored distances are the cosine distances of the hashed vectors: re-vectorise the endpoints of a sample of
    # edges and compare with a float64 inner product (north_star: 1e-5 on the sparse entries in float32)
    from falcon_amd import device as _device
    g = torch.Generator(device="cpu").manual_seed(5)
    pick = torch.randint(0, i_idx.numel(), (check_values,), generator=g).to(i_idx.device)
    n_bins, start, _ = _device.get_dim(p.min_mz, p.max_mz, 0.05)
    vec = lambda r: ctx.vectorize(ds.mz, ds.intensity, ds.indptr, order[r], start, 0.05, n_bins, p.low_dim, p.hash_seed, True,
                                  "f16" if p.dtype == "f16" else "f32").double()
    cos = (vec(i_idx[pick]) * vec(j_idx[pick])).sum(1)
    exp = (1.0 - cos).clamp(0.0, 1.0)
    assert float((exp - dist_ij[pick].double()).abs().max()) <= dist_tol
    # clusters never cross the precursor tolerance chain: members of one cluster span < 1 m/z (one window)
    lab_sorted = lab[sorted_pos] if shard else lab[order]
    lo = torch.full((med.numel(),), float("inf"), device=lab.device).scatter_reduce(0, lab_sorted.long(), mz_sorted, "amin")
    hi = torch.full((med.numel(),), float("-inf"), device=lab.device).scatter_reduce(0, lab_sorted.long(), mz_sorted, "amax")
    assert float((hi - lo).max()) < 1.0
    return n


def _run_and_check(n_total, p, generator, dist_tol, skew=False):
    import torch
    from falcon_amd import synth
    from falcon_amd.cluster.cluster import ClusterPipeline
    from falcon_amd.device import Context
    ctx = Context(0)
    pipe = ClusterPipeline(ctx)
    if generator == "device":
        data = synth.generate_device(n_total, ctx.tdev, seed=42, skew=skew)
        parts = _parts(ctx, data, synth.select_charge_device)
    else:
        parts = _parts(ctx, synth.generate(n_total, seed=42, skew=skew), synth.select_charge)
    args = (20.0, "ppm", None, 0.05, 2 ** 15, p)
    outs = pipe.run_many(parts, *args)
    lasts = [dict(x) for x in pipe.lasts]
    if skew:                                                      # flat and indexed buckets side by side, tiny to 2^15 rows
        nl = np.concatenate([np.asarray(x["n_list"]) for x in lasts])
        assert nl.min() == 1 and nl.max() >= 256 and (nl == 1).sum() > 100
    again = pipe.run_many(parts, *args)
    total, sizes = 0, []
    for ds, (lab, med), (lab2, med2), last in zip(parts, outs, again, lasts):
        total += _check_partition(ctx, ds, lab, med, lab2, med2, last, p, dist_tol)
        sizes.append(int(med.numel()))
    assert total == n_total
    ctx.close()
    return sizes


def test_fifty_million_spectra_config4_invariants():
    """BASELINE configs[3] at its own size on ONE GPU: 50,000,000 spectra, n_probe 32, n_neighbors_ann 128 -- 1 m/z windows of
    ~44,000 (charge 2) / ~19,000 (charge 3) spectra, cut further by the reference's batch_size rule (blocks of at most 2^15
    rows, cluster.py:197-207): buckets of up to 32,768 rows, n_list 512 / 256.  The working set of one pass (~260 GB) does not fit next to the
    dataset, so the buckets run in 4 shares (`ClusterPipeline.run_chunked`, the multi-GPU partition executed in turn); every
    share goes through the invariants, and the shares must cover every spectrum exactly once."""
    import torch
    from falcon_amd import synth
    from falcon_amd.cluster.cluster import AnnParams, ClusterPipeline
    from falcon_amd.device import Context
    n_total, n_chunks = 50_000_000, 4
    ctx = Context(0)
    pipe = ClusterPipeline(ctx)
    data = synth.generate_device(n_total, ctx.tdev, seed=42)
    parts = _parts(ctx, data, synth.select_charge_device)
    del data
    torch.cuda.empty_cache()
    p = AnnParams(n_probe=32, n_neighbors_ann=128)
    args = (20.0, "ppm", None, 0.05, 2 ** 15, p)
    seen = [torch.zeros(len(ds), dtype=torch.int32, device=ctx.tdev) for ds in parts]
    n_list_max, first = [0], {}

    def check(ch, outs, lasts):
        for j, ((lab, med), last) in enumerate(zip(outs, lasts)):
            if last["rows"].numel() == 0:
                continue
            seen[j][last["rows"]] += 1
            n_list_max[0] = max(n_list_max[0], int(np.max(last["n_list"])))
            if ch == 0:
                first[j] = (lab.clone(), med.clone())
            _check_partition(ctx, parts[j], lab, med, lab, med, last, p, 1e-5, shard=True)

    res = pipe.run_chunked(parts, *args, n_chunks=n_chunks, on_chunk=check)
    assert n_list_max[0] == 512
    assert all(bool((s == 1).all()) for s in seen)
    for ds, (lab, med) in zip(parts, res):
        assert int(lab.min()) == 0 and int(lab.max()) == med.numel() - 1
        assert torch.equal(lab[med.long()], torch.arange(med.numel(), device=lab.device, dtype=lab.dtype))
    # determinism: share 0 again gives the same bits
    again = pipe.run_many(parts, *args, shard=(0, n_chunks))
    for j, (lab, med) in enumerate(again):
        assert torch.equal(lab, first[j][0]) and torch.equal(med, first[j][1])
    assert ctx.counter(6) == 0
    ctx.close()


def test_one_million_spectra_invariants():
    from falcon_amd.cluster.cluster import AnnParams
    _run_and_check(1_000_000, AnnParams(), "numpy", 1e-5)


def test_ten_million_spectra_f32_invariants():
    """the north star's target size on ONE GPU: 1 m/z windows of ~8,750 (charge 2) / ~3,750 (charge 3) spectra ->
    IVF buckets with n_list 128 / 64, n_probe 16"""
    from falcon_amd.cluster.cluster import AnnParams
    _run_and_check(10_000_000, AnnParams(), "device", 1e-5)


def test_ten_million_spectra_f16_low_dim_800_invariants():
    """BASELINE configs[4]: 10 M spectra, low_dim 800, float16 vectors (f16 MFMA, float32 accumulation)"""
    from falcon_amd.cluster.cluster import AnnParams
    _run_and_check(10_000_000, AnnParams(dtype="f16", low_dim=800), "device", 2e-5)


def test_skewed_two_million_spectra_invariants():
    """a workload that is NOT uniform: log-normal occupancy of the 1 m/z precursor windows (the fullest
    ~70x the median: a 65 k-row window and a few of 15-30 k rows -- cut at batch_size, n_list up to 512 -- among 300-row flat ones) and 5..50
    peaks per spectrum (`synth.generate_device(skew=True)`): the same invariants"""
    from falcon_amd.cluster.cluster import AnnParams
    _run_and_check(2_000_000, AnnParams(), "device", 1e-5, skew=True)
