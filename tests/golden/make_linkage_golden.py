#!/usr/bin/env python3
"""Golden vectors for f4 (`--clustering hierarchical`): sparse neighbour graphs and the flat clusters the reference's own
composition gives on them,

    labels = sch.fcluster(fastcluster.linkage(pdist, linkage), distance_threshold, "distance") - 1      (cluster.py:283-290)

with scipy.cluster.hierarchy.linkage standing in for fastcluster (not installed; same dendrogram up to the order of
equal heights) and the dense `pdist` built from the sparse graph with "missing pair = 1" (cluster.py:621-626).
Run in the build container; writes tests/golden/linkage.npz (data only).

    python tests/golden/make_linkage_golden.py
"""
import os

import numpy as np
import scipy.cluster.hierarchy as sch
from scipy.spatial.distance import squareform

HERE = os.path.dirname(os.path.abspath(__file__))


def graph(n, k, seed, n_groups):
    """kNN-like graph: points in groups (small within-group distances), a few cross edges, directed and truncated"""
    rng = np.random.default_rng(seed)
    g = rng.integers(0, n_groups, n)
    pos = rng.normal(size=(n, 3)) * 0.03 + rng.normal(size=(n_groups, 3))[g]
    d = np.sqrt(((pos[:, None] - pos[None]) ** 2).sum(-1))
    d = np.clip(d / 1.2, 0, 0.999).astype(np.float32)
    nb_idx = np.full((n, k), -1, np.int32)
    nb_dist = np.full((n, k), np.inf, np.float32)
    for i in range(n):
        o = np.argsort(d[i], kind="stable")
        o = o[o != i][:k]
        keep = rng.random(len(o)) > 0.15                     # drop some directions: asymmetric lists
        o = o[keep]
        nb_idx[i, :len(o)] = o
        nb_dist[i, :len(o)] = d[i, o]
    return nb_idx, nb_dist


def reference_labels(nb_idx, nb_dist, t, method):
    n, k = nb_idx.shape
    D = np.ones((n, n), np.float64)
    for i in range(n):
        for s in range(k):
            j = nb_idx[i, s]
            if j >= 0 and j != i:
                D[i, j] = nb_dist[i, s]
    D = np.minimum(D, D.T)
    np.fill_diagonal(D, 0.0)
    pdist = squareform(D, checks=False)
    labels = sch.fcluster(sch.linkage(pdist, method), float(np.float32(t)), "distance") - 1        # cluster.py:283-290
    # canonical numbering: clusters by lowest row, groups of one row = -1 (cluster.py:441-454)
    out = np.full(n, -1, np.int32)
    nxt = 0
    for i in range(n):
        if out[i] == -1:
            mm = np.flatnonzero(labels == labels[i])
            if len(mm) >= 2:
                out[mm] = nxt
                nxt += 1
            else:
                out[i] = -2
    out[out == -2] = -1
    return out


def main():
    out = {}
    cases = [(60, 6, 1, 8, 0.08), (300, 10, 2, 25, 0.05), (800, 16, 3, 60, 0.1), (500, 8, 4, 5, 0.06), (200, 12, 5, 200, 0.3)]
    out["n_cases"] = len(cases)
    for c, (n, k, seed, groups, t) in enumerate(cases):
        nb_idx, nb_dist = graph(n, k, seed, groups)
        out[f"c{c}_idx"], out[f"c{c}_dist"], out[f"c{c}_t"] = nb_idx, nb_dist, np.float32(t)
        for method in ("single", "complete", "average"):
            out[f"c{c}_{method}"] = reference_labels(nb_idx, nb_dist, t, method)
    np.savez_compressed(os.path.join(HERE, "linkage.npz"), **out)
    print("wrote linkage.npz:", {k: (v.shape if hasattr(v, "shape") else v) for k, v in out.items() if k.endswith("complete")})


if __name__ == "__main__":
    main()
