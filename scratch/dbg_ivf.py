import sys, numpy as np, torch
sys.path.insert(0, '.')
from falcon_amd.device import Context
from oracle import falcon_oracle as fo
from tests.test_gpu_search import unit_vectors
ctx = Context(0)
X = unit_vectors(3000, 400, 11)
for iters in (3, 4, 4):
    idx = ctx.ivf_build(torch.from_numpy(X).to(ctx.tdev), np.array([0, 3000]), np.array([64], np.int32), iters)
    cent, asg, perm, loff = [t.cpu().numpy() for t in idx.export()]
    C, ra, rperm, roff = fo.ivf_build(X, 64, iters)
    print(iters, 'agree', (asg == ra).mean(), 'cent equal', np.array_equal(cent, C), np.abs(cent - C).max())
    if iters == 0:
        S = X @ C.T
        bad = np.flatnonzero(asg != ra)[:5]
        for b in bad:
            print(' row', b, 'gpu', asg[b], S[b, asg[b]], 'ref', ra[b], S[b, ra[b]], 'rownorm', np.linalg.norm(X[b]))
