"""Determinism under repetition: the index of BASELINE configs[3]'s regime -- buckets with up to 512
lists, float16 k-means assignment in groups of 128 centroids, 512-column keys feeding the coarse quantiser
(`coarse16w_kernel<32>` + the workgroup-level hand-off for queries with more than 16 tied lists) -- is built and searched
`FALCON_STRESS_REPS` (default 200) times on fixed data.  Every repetition must equal the exact (all-fp32) build + staged
search bit for bit, and repetition 0.  One such mismatch was seen once in round 2 (~1 in 35 runs of the test file) and never
reproduced; this test is the standing guard, `tools/stress_build.py` the long-running form."""
import os

import numpy as np
import pytest

from tests.test_gpu_search import sparse_unit_vectors, unit_vectors

pytestmark = pytest.mark.gpu
REPS = int(os.environ.get("FALCON_STRESS_REPS", "200"))


@pytest.fixture(scope="module")
def ctx():
    from falcon_amd.device import Context
    c = Context(0)
    yield c
    c.close()


def _bits(t):
    import torch
    return t.view(torch.int32) if t.dtype == torch.float32 else t


@pytest.mark.parametrize("sparse", [False, True])
def test_keyed_512_list_index_repeats_bit_for_bit(ctx, sparse):
    import torch
    sizes = [21000, 3000, 11000]
    nl = np.array([512, 64, 200], np.int32)
    off = np.concatenate([[0], np.cumsum(sizes)])
    n = int(off[-1])
    X = sparse_unit_vectors(n, 128, 67) if sparse else unit_vectors(n, 128, 67, noise=0.35)
    X[off[0]:off[0] + 21000:41] = X[off[0]]                    # identical k-means seeds -> identical centroids (ties)
    X[off[2] + 7:off[2] + 300:3] = 0                           # all-zero rows: every key equal, everything is a member
    Xd = torch.from_numpy(X).to(ctx.tdev)
    X16 = Xd.to(torch.float16).contiguous()
    mz = torch.from_numpy(np.concatenate([np.sort(500.0 + b + np.random.default_rng(b).random(s))
                                          for b, s in enumerate(sizes)]).astype(np.float32)).to(ctx.tdev)
    plain = ctx.ivf_build(Xd, off, nl, kmeans_iters=3)
    ref_index = [_bits(t).clone() for t in plain.export()]
    ref = {}
    for n_probe in (32, 5):
        s, i = plain.search(n_probe, 64)
        ref[n_probe] = (_bits(s).clone(), i.clone())
    ref_nb = tuple(_bits(t).clone() for t in plain.search_neighbors(32, 128, mz, None, 20.0, "ppm", None, 64))
    plain.close()
    bad = []
    for rep in range(REPS):
        keyed = ctx.ivf_build(Xd, off, nl, kmeans_iters=3, Xkm=X16, Xpre=X16, prefilter_which=2)
        for name, a, b in zip(("centroids", "assign", "perm", "list_off"), ref_index, keyed.export()):
            if not torch.equal(a, _bits(b)):
                bad.append((rep, "index " + name, int((a != _bits(b)).sum())))
        for n_probe in (32, 5):
            s, i = keyed.search(n_probe, 64)                   # probes from the keys (coarse16.hip), staged exact fine scan
            if not (torch.equal(ref[n_probe][1], i) and torch.equal(ref[n_probe][0], _bits(s))):
                bad.append((rep, f"search n_probe={n_probe}", int((ref[n_probe][1] != i).any(1).sum())))
        nb = keyed.search_neighbors(32, 128, mz, None, 20.0, "ppm", None, 64)      # the production path (ivf16.hip)
        if not (torch.equal(ref_nb[0], nb[0]) and torch.equal(ref_nb[1], _bits(nb[1]))):
            bad.append((rep, "search_neighbors", int((ref_nb[0] != nb[0]).any(1).sum())))
        keyed.close()
        if len(bad) > 5:
            break
    assert not bad, bad
