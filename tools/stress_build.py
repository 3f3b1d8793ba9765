"""repeat the build of an index (exact k-means and float16-prefiltered k-means) and its search many times on the same data and
report anything that differs between repetitions: a determinism / race hunt."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from falcon_amd.device import Context
from tests.test_gpu_search import unit_vectors, sparse_unit_vectors

reps = int(sys.argv[1]) if len(sys.argv) > 1 else 50
sparse = len(sys.argv) > 2 and sys.argv[2] == "sparse"
ctx = Context(0)
sizes = [21000, 3000, 11000]
nl = np.array([512, 64, 200], np.int32)
off = np.concatenate([[0], np.cumsum(sizes)])
n = int(off[-1])
X = sparse_unit_vectors(n, 128, 67) if sparse else unit_vectors(n, 128, 67, noise=0.35)
X[off[0]:off[0] + 21000:41] = X[off[0]]
Xd = torch.from_numpy(X).to(ctx.tdev)
X16 = Xd.to(torch.float16).contiguous()
ref = None
bad = 0
for it in range(reps):
    plain = ctx.ivf_build(Xd, off, nl, kmeans_iters=3)
    keyed = ctx.ivf_build(Xd, off, nl, kmeans_iters=3, Xkm=X16)
    ep = [t.cpu().numpy() for t in plain.export()]
    ek = [t.cpu().numpy() for t in keyed.export()]
    res = []
    for n_probe in (32, 5):
        s0, i0 = plain.search(n_probe, 64)
        s1, i1 = keyed.search(n_probe, 64)
        res.append((s0.cpu().numpy().view(np.uint32), i0.cpu().numpy(), s1.cpu().numpy().view(np.uint32), i1.cpu().numpy()))
    names = ["cent", "asg", "perm", "loff"]
    msgs = []
    for nm, a, b in zip(names, ep, ek):
        if not np.array_equal(a.view(np.uint32) if a.dtype == np.float32 else a, b.view(np.uint32) if b.dtype == np.float32 else b):
            msgs.append(f"plain/keyed {nm} differ at {np.flatnonzero((a != b).reshape(len(a), -1).any(1))[:5]}")
    for j, (s0, i0, s1, i1) in enumerate(res):
        if not (np.array_equal(i0, i1) and np.array_equal(s0, s1)):
            rows = np.flatnonzero((i0 != i1).any(1) | (s0 != s1).any(1))
            msgs.append(f"search {j}: plain/keyed differ in {len(rows)} rows, first {rows[:5]}")
    cur = (ep, ek, res)
    if ref is None:
        ref = cur
    else:
        for nm, a, b in zip(names, ref[0], ep):
            if not np.array_equal(a, b, equal_nan=True):
                msgs.append(f"plain {nm} differs from repetition 0")
        for nm, a, b in zip(names, ref[1], ek):
            if not np.array_equal(a, b, equal_nan=True):
                msgs.append(f"keyed {nm} differs from repetition 0")
        for j in range(2):
            for t, nm in enumerate(["plain sims", "plain idx", "keyed sims", "keyed idx"]):
                if not np.array_equal(ref[2][j][t], res[j][t]):
                    rows = np.flatnonzero((ref[2][j][t] != res[j][t]).any(1))
                    msgs.append(f"search {j} {nm} differs from repetition 0 in {len(rows)} rows, first {rows[:5]}")
    if msgs:
        bad += 1
        print(f"rep {it}:", "; ".join(msgs), flush=True)
    plain.close(); keyed.close()
print(f"{reps} repetitions, {bad} with differences")
