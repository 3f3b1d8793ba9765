"""f4 on the GPU: `fal_rescore_neighbors` (matched-peak cosine, reference similarity.py:17-80) against the
reference's own golden pairs and against the oracle on pipeline neighbour lists."""
import os

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


def test_rescore_matches_reference_cosine_fast_golden(ctx):
    """the 320 golden pairs as a dataset: spectrum 2c = a_c, 2c + 1 = b_c, row 2c stores neighbour 2c + 1."""
    import torch
    g = np.load(os.path.join(os.path.dirname(__file__), "golden", "cosine_fast.npz"))
    n = len(g["tol"])
    mz, it, sizes = [], [], []
    for c in range(n):
        for key, ptr in (("a", g["a_ptr"]), ("b", g["b_ptr"])):
            mz.append(g[key + "_mz"][ptr[c]:ptr[c + 1]]); it.append(g[key + "_it"][ptr[c]:ptr[c + 1]])
            sizes.append(ptr[c + 1] - ptr[c])
    mz, it = np.concatenate(mz), np.concatenate(it)
    indptr = np.concatenate([[0], np.cumsum(sizes)]).astype(np.int64)
    order = np.arange(2 * n, dtype=np.int64)
    k = 4
    for tol in np.unique(g["tol"]):
        sel = np.flatnonzero(g["tol"] == tol)
        for min_matches in (0, 3, 12):
            nb_idx = np.full((2 * n, k), -1, np.int32)
            nb_idx[2 * sel, 1] = 2 * sel + 1                     # (slot 1: holes before a valid entry are fine)
            nb_dist = np.full((2 * n, k), np.inf, np.float32)
            out = ctx.rescore_neighbors(torch.from_numpy(nb_idx).to(ctx.tdev), torch.from_numpy(nb_dist).to(ctx.tdev),
                                        mz, it, indptr, order, float(tol), min_matches).cpu().numpy()
            sim = np.where(g["n_match"][sel] >= min_matches, g["score"][sel], 0.0)
            np.testing.assert_allclose(out[2 * sel, 1], (1.0 - sim).astype(np.float32), rtol=0, atol=2e-7)
            # n_match is exercised through the cut: pairs at exactly min_matches stay, one below go to 1.0
            assert np.all(out[2 * sel, 1][g["n_match"][sel] < min_matches] == 1.0)
            assert np.all(np.isinf(out[:, 0])) and np.all(np.isinf(out[2 * sel + 1]))      # untouched slots


def test_rescore_pipeline_lists_match_oracle(ctx):
    import torch
    from falcon_amd import synth
    from falcon_amd.cluster.cluster import AnnParams, ClusterPipeline, SpectrumDataset
    d = synth.select_charge(synth.generate(2500, seed=5), 2)
    ds = SpectrumDataset(d["precursor_mz"], d["retention_time"], d["mz"], d["intensity"], d["indptr"])
    pipe = ClusterPipeline(ctx)
    p = AnnParams(eps=0.3)
    pipe.run(ds, 20.0, "ppm", None, 0.05, 2 ** 15, p, keep_intermediates=True)
    nb_idx, nb_dist, order = (pipe.last[k].cpu().numpy() for k in ("nb_idx", "nb_dist", "order"))
    assert (nb_idx >= 0).sum() > 2000
    for min_matches in (0, 6):
        ref = fo.rescore_neighbors(nb_idx, nb_dist, d["mz"], d["intensity"], d["indptr"], order, 0.05, min_matches)
        out = ctx.rescore_neighbors(pipe.last["nb_idx"], pipe.last["nb_dist"].clone(), d["mz"], d["intensity"], d["indptr"],
                                    order, 0.05, min_matches).cpu().numpy()
        valid = nb_idx >= 0
        assert np.array_equal(np.isinf(out), ~valid)
        assert np.array_equal(out[valid].view(np.uint32), ref[valid].view(np.uint32))       # bit-identical
    # and end to end: the option changes the distances DBSCAN sees, labels stay a valid partition
    labels, medoids = pipe.run(ds, 20.0, "ppm", None, 0.05, 2 ** 15, AnnParams(eps=0.3, rescore=True, min_matches=6))
    labels = labels.cpu().numpy()
    assert labels.min() == 0 and np.array_equal(np.unique(labels), np.arange(labels.max() + 1))
