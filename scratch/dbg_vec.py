import sys, numpy as np
sys.path.insert(0, '.')
from falcon_amd import synth
from falcon_amd.device import Context
from oracle import falcon_oracle as fo
ctx = Context(0)
d = synth.generate(3000, seed=7)
nb, start, _ = fo.get_dim(101, 1500, 0.05)
V = ctx.vectorize(d['mz'], d['intensity'], d['indptr'], None, start, 0.05, nb, 400).cpu().numpy()
U = ctx.vectorize(d['mz'], d['intensity'], d['indptr'], None, start, 0.05, nb, 400, normalize=False).cpu().numpy()
ref = fo.vectorize(d['mz'], d['intensity'], d['indptr'], start, 0.05, nb, 400)
refu = fo.vectorize(d['mz'], d['intensity'], d['indptr'], start, 0.05, nb, 400, norm=False)
print('unnorm equal', np.array_equal(U, refu))
bad = np.argwhere(V != ref)
print('mismatch elems', len(bad), 'rows', len(np.unique(bad[:, 0])))
r = bad[0, 0]
nr = fo.l2_norm_sq_tree(refu[r:r+1])[0]
print('row', r, 'nr', repr(nr), 'sqrtf', repr(np.sqrt(np.float32(nr))), 'inv ref', repr((np.float64(1)/np.float64(np.sqrt(np.float32(nr)))).astype(np.float32)))
nzc = np.flatnonzero(refu[r])
print('gpu inv candidates', np.unique(V[r, nzc] / refu[r, nzc])[:5], 'ref', np.unique(ref[r, nzc] / refu[r, nzc])[:5])
ulp = np.abs(V.view(np.int32).astype(np.int64) - ref.view(np.int32).astype(np.int64))
print('max ulp', ulp.max())
