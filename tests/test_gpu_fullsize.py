"""BASELINE configs[1] at full size (1,000,000 synthetic spectra) through size-independent properties: the oracle
cannot follow at this size, the invariants of the path can."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def test_one_million_spectra_invariants():
    import torch
    from falcon_amd import synth
    from falcon_amd.cluster.cluster import AnnParams, ClusterPipeline, SpectrumDataset
    from falcon_amd.device import Context
    ctx = Context(0)
    pipe = ClusterPipeline(ctx)
    data = synth.generate(1_000_000, seed=42)
    p = AnnParams()
    args = (20.0, "ppm", None, 0.05, 2 ** 15, p)
    parts = []
    for ch in (2, 3):
        c = synth.select_charge(data, ch)
        parts.append(SpectrumDataset(*[ctx.to_dev(c[k], torch.float32) for k in ("precursor_mz", "retention_time", "mz", "intensity")],
                                     ctx.to_dev(c["indptr"], torch.int64)))
    outs = pipe.run_many(parts, *args)
    lasts = [dict(x) for x in pipe.lasts]
    again = pipe.run_many(parts, *args)
    total = 0
    for ds, (lab, med), (lab2, med2), last in zip(parts, outs, again, lasts):
        n = len(ds)
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
        j = nb_idx.clamp(min=0).long()
        ppm = (mz_sorted[:, None] - mz_sorted[j]).abs() / mz_sorted[j] * 1e6
        assert float(ppm[valid].max()) <= 20.0 + 1e-3
        # symmetry of the scan: where i stores j AND j stores i, the two distances are the same bits
        i_idx = rows[valid]
        j_idx = nb_idx[valid].long()
        dist_ij = nb_dist[valid]
        key = i_idx * n + j_idx
        rkey = j_idx * n + i_idx
        skey, perm = torch.sort(key)
        pos = torch.searchsorted(skey, rkey).clamp(max=skey.numel() - 1)
        mutual = skey[pos] == rkey
        assert int(mutual.sum()) > 1000
        assert torch.equal(dist_ij[mutual], dist_ij[perm][pos][mutual])
        total += n
    assert total == 1_000_000
    ctx.close()
