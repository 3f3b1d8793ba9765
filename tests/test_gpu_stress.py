"""Determinism under repetition, attributed: the index of BASELINE configs[3]'s regime -- buckets
with up to 512 lists at low_dim 128, float16 k-means assignment in groups of 128 centroids, 512-column keys feeding the
coarse quantiser -- is built `FALCON_STRESS_REPS` (default 200) times on fixed data by BOTH build paths:

  plain  = `fal_ivf_build`        (all-fp32: assign_kernel, LDS-DMA double buffer; round 3 shipped it with a bare s_barrier on
                                   the loop back-edge -- rows raced their DMA on a cold box, a handful of rows went to the wrong
                                   list, and every "keyed == plain" comparison turned red)
  keyed  = `fal_ivf_build_x16`    (float16 prefilter + exact re-evaluation; production)

and EVERY repetition of either is compared with the oracle's index (`fo.ivf_build`, computed once): the test says which
side left the oracle, in which repetition, and prints the first differing row with its best / runner-up margin.  All
repetitions run; nothing stops at the first mismatch.  The searches of the keyed index must equal the plain index's staged
search of repetition 0 (itself pinned to the oracle by tests/test_gpu_search.py).  `tools/stress_build.py` is the long-running
form (second-stream HBM traffic, low_dim 64 / 128)."""
import os

import numpy as np
import pytest

from oracle import falcon_oracle as fo
from tests.test_gpu_search import sparse_unit_vectors, unit_vectors

pytestmark = pytest.mark.gpu
REPS = int(os.environ.get("FALCON_STRESS_REPS", "200"))
SIZES = [21000, 3000, 11000]
NLIST = [512, 64, 200]
ITERS = 3


@pytest.fixture(scope="module")
def ctx():
    from falcon_amd.device import Context
    c = Context(0)
    yield c
    c.close()


def _bits(t):
    import torch
    return t.view(torch.int32) if t.dtype == torch.float32 else t


def stress_data(sparse, d=128):
    off = np.concatenate([[0], np.cumsum(SIZES)])
    n = int(off[-1])
    X = sparse_unit_vectors(n, d, 67) if sparse else unit_vectors(n, d, 67, noise=0.35)
    X[off[0]:off[0] + 21000:41] = X[off[0]]                    # identical k-means seeds -> identical centroids (ties)
    X[off[2] + 7:off[2] + 300:3] = 0                           # all-zero rows: every key equal, everything is a member
    return X, off


def oracle_index(X, off):
    """(centroids, assign, perm, list_off) of the whole multi-bucket index as `IvfIndex.export` lays it out"""
    cents, asg, perm, loff = [], [], [], [np.zeros(1, np.int64)]
    for b, (a, e) in enumerate(zip(off[:-1], off[1:])):
        C, ra, rperm, roff = fo.ivf_build(X[a:e], NLIST[b], ITERS)
        cents.append(C)
        asg.append(ra)
        perm.append(rperm + a)
        loff.append(roff[1:] + a)
    return (np.concatenate(cents).astype(np.float32), np.concatenate(asg).astype(np.int32),
            np.concatenate(perm).astype(np.int32), np.concatenate(loff).astype(np.int64))


def describe_first_difference(X, off, ref, got):
    """which row went to another list, and how close the decision was (exact k-ordered similarities against the ORACLE's
    final centroids: best, runner-up and the list the GPU chose)"""
    rows = np.flatnonzero(ref[1] != got[1])
    if len(rows) == 0:
        w = np.flatnonzero((ref[0].view(np.uint32) != got[0].view(np.uint32)).any(1))
        return f"assignments equal; {len(w)} centroid rows differ, first list {w[:3]}"
    i = int(rows[0])
    b = int(np.searchsorted(off, i, side="right") - 1)
    lb = np.concatenate([[0], np.cumsum(NLIST)])
    s = fo.sims_f32(X[i:i + 1], ref[0][lb[b]:lb[b + 1]])[0]
    o = np.argsort(-s.astype(np.float64), kind="stable")
    return (f"{len(rows)} rows assigned differently; first row {i} (bucket {b}): oracle list {ref[1][i]} GPU list {got[1][i]}; "
            f"exact sims vs the oracle's centroids: best {s[o[0]]:.9g} (list {o[0]}), runner-up {s[o[1]]:.9g} (list {o[1]}), "
            f"GPU's choice {s[got[1][i]]:.9g}")


def compare_index(ref, exported):
    got = [t.cpu().numpy() for t in exported]
    names = ("centroids", "assign", "perm", "list_off")
    diff = [nm for nm, a, b in zip(names, ref, got)
            if not np.array_equal(a.view(np.uint32) if a.dtype == np.float32 else a, b.view(np.uint32) if b.dtype == np.float32 else b)]
    return diff, got


@pytest.mark.parametrize("sparse", [False, True])
def test_keyed_and_plain_512_list_index_equal_the_oracle_every_repetition(ctx, sparse):
    import torch
    X, off = stress_data(sparse)
    nl = np.array(NLIST, np.int32)
    ref = oracle_index(X, off)
    Xd = torch.from_numpy(X).to(ctx.tdev)
    X16 = Xd.to(torch.float16).contiguous()
    mz = torch.from_numpy(np.concatenate([np.sort(500.0 + b + np.random.default_rng(b).random(s))
                                          for b, s in enumerate(SIZES)]).astype(np.float32)).to(ctx.tdev)
    bad, search_ref = [], None
    for rep in range(REPS):
        plain = ctx.ivf_build(Xd, off, nl, kmeans_iters=ITERS)                     # rebuilt every repetition
        diff, got = compare_index(ref, plain.export())
        if diff:
            bad.append((rep, "PLAIN build left the oracle", diff, describe_first_difference(X, off, ref, got)))
        if search_ref is None and not diff:
            search_ref = {}
            for n_probe in (32, 5):
                s, i = plain.search(n_probe, 64)
                search_ref[n_probe] = (_bits(s).clone(), i.clone())
            search_ref["nb"] = tuple(_bits(t).clone() for t in plain.search_neighbors(32, 128, mz, None, 20.0, "ppm", None, 64))
        plain.close()
        keyed = ctx.ivf_build(Xd, off, nl, kmeans_iters=ITERS, Xkm=X16, Xpre=X16, prefilter_which=2)
        diff, got = compare_index(ref, keyed.export())
        if diff:
            bad.append((rep, "KEYED build left the oracle", diff, describe_first_difference(X, off, ref, got)))
        elif search_ref is not None:
            for n_probe in (32, 5):
                s, i = keyed.search(n_probe, 64)               # probes from the keys (coarse16.hip), staged exact fine scan
                if not (torch.equal(search_ref[n_probe][1], i) and torch.equal(search_ref[n_probe][0], _bits(s))):
                    bad.append((rep, f"keyed search n_probe={n_probe}", int((search_ref[n_probe][1] != i).any(1).sum())))
            nb = keyed.search_neighbors(32, 128, mz, None, 20.0, "ppm", None, 64)      # the production path (ivf16.hip)
            if not (torch.equal(search_ref["nb"][0], nb[0]) and torch.equal(search_ref["nb"][1], _bits(nb[1]))):
                bad.append((rep, "keyed search_neighbors", int((search_ref["nb"][0] != nb[0]).any(1).sum())))
        keyed.close()
    assert search_ref is not None, "no repetition of the plain build equalled the oracle"
    assert not bad, f"{len(bad)} findings in {REPS} repetitions; first 8: {bad[:8]}"
