"""Shared helpers of the parity tests."""
import numpy as np


def assert_topk_close(sim, idx, ref_sim, ref_idx, X=None, base=0, tol=1e-5, what=""):
    """Top-k lists agree up to the float tolerance of north_star (1e-5 on the sparse
    entries): similarities match rank by rank, and wherever the ids differ the two
    candidates are tied within the tolerance (checked against float64 inner products
    when the vectors X are given)."""
    assert sim.shape == ref_sim.shape and idx.shape == ref_idx.shape, what
    pad = ref_idx < 0
    assert np.array_equal(idx < 0, pad), f"{what}: padding differs"
    assert np.all(np.isneginf(sim[pad])), what
    np.testing.assert_allclose(sim[~pad], ref_sim[~pad], atol=tol, rtol=0, err_msg=what)
    diff = (idx != ref_idx) & ~pad
    if diff.any():
        r, c = np.nonzero(diff)
        # the id the GPU chose must be as good as the reference's at that rank
        if X is not None:
            Xd = X.astype(np.float64)
            s_gpu = np.einsum("ij,ij->i", Xd[r], Xd[idx[r, c] - base])
            assert np.all(np.abs(s_gpu - ref_sim[r, c]) <= tol), f"{what}: id differs without a tie"
        # and the two rows hold the same ids up to tie swaps: every GPU id missing from the
        # reference row sits at the k-th boundary value
        for i in np.unique(r):
            extra = np.setdiff1d(idx[i][idx[i] >= 0], ref_idx[i][ref_idx[i] >= 0])
            if len(extra):
                kth = ref_sim[i][~pad[i]].min()
                js = np.isin(idx[i], extra)
                assert np.all(np.abs(sim[i][js] - kth) <= tol), f"{what}: row {i} differs beyond ties"
    return int(diff.sum())


def assert_topk_exact(sim, idx, ref_sim, ref_idx, what=""):
    """Bit-exact form: the oracle sums in the kernels' own k order (oracle/kordered.c), so similarities,
    ids, tie order and padding must all be identical."""
    from oracle import falcon_oracle as fo
    assert fo.have_kordered(), "oracle/_build/libkordered.so is missing: run __graft_entry__.build()"
    assert np.array_equal(idx, ref_idx), f"{what}: ids differ in {(idx != ref_idx).any(1).sum()} rows"
    assert np.array_equal(sim, ref_sim), f"{what}: sims differ, max {np.nanmax(np.abs(np.where(np.isfinite(ref_sim), sim - ref_sim, 0)))}"
