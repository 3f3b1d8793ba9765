"""Debug: the failing d=128 case of tests/test_gpu_fused.py, row by row."""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from falcon_amd.device import Context
from tests.test_gpu_search import unit_vectors
from oracle import falcon_oracle as fo

d, k_ann, keep = int(sys.argv[1]) if len(sys.argv) > 1 else 128, 128, 64
ctx = Context(0)
sizes = [1, 2, 31, 33, 100, 129, 130, 257, 700, 1248, 1500, 2400, 5, 640]
off = np.concatenate([[0], np.cumsum(sizes)]); n = int(off[-1])
X = unit_vectors(n, d, 17, noise=0.35); X[off[8]:off[8] + 6] = X[off[8]]
rng = np.random.default_rng(3)
mz = np.concatenate([np.sort(500.0 + b + rng.random(s)) for b, s in enumerate(sizes)]).astype(np.float32)
nl = np.ones(len(sizes), np.int32)
Xd = torch.from_numpy(X).to(ctx.tdev); mz_d = torch.from_numpy(mz).to(ctx.tdev)
plain = ctx.ivf_build(Xd, off, nl)
sim, idx = plain.search(16, k_ann)
e_idx, e_dist = ctx.filter_neighbors(sim, idx, mz_d, None, 20.0, "ppm", None, keep)
pre = ctx.ivf_build(Xd, off, nl, Xpre=Xd.to(torch.float16).contiguous())
g_idx, g_dist = pre.search_neighbors(16, k_ann, mz_d, None, 20.0, "ppm", None, keep)
ctx.sync()
ptr, nrows = ctx.counter(6), ctx.counter(7)
hip = ctypes.CDLL("libamdhip64.so")
thr = np.zeros((nrows, 8), np.float32)
hip.hipMemcpy(ctypes.c_void_p(thr.ctypes.data), ctypes.c_void_p(ptr), ctypes.c_size_t(thr.nbytes), 2)
thr_i = thr.view(np.int32)
memv = np.zeros((nrows, 64), np.float32); memi = np.zeros((nrows, 64), np.uint32)
hip.hipMemcpy(ctypes.c_void_p(memv.ctypes.data), ctypes.c_void_p(ptr + thr.nbytes), ctypes.c_size_t(memv.nbytes), 2)
hip.hipMemcpy(ctypes.c_void_p(memi.ctypes.data), ctypes.c_void_p(ptr + thr.nbytes + memv.nbytes), ctypes.c_size_t(memi.nbytes), 2)
sim, idx, e_idx, g_idx, e_dist, g_dist = (t.cpu().numpy() for t in (sim, idx, e_idx, g_idx, e_dist, g_dist))
bad = np.flatnonzero((g_idx != e_idx).any(1))
print("bad rows", len(bad), "fallback", ctx.counter(5))
X16 = X.astype(np.float16).astype(np.float64)
for row in bad[:6]:
    b = np.searchsorted(off, row, side="right") - 1
    a0, a1 = off[b], off[b + 1]
    T_exact = sim[row, k_ann - 1]; id_T = idx[row, k_ann - 1]
    approx = X16[a0:a1] @ X16[row]
    T_approx = np.sort(approx)[-k_ann]
    L, U, T, eps = thr[row, :4]; bstar, nabove, mc, flags = thr_i[row, 4:8]
    print(f"row {row} bucket {b} nc {a1-a0}: exact T128 {T_exact:.7f} (id {id_T}) approx-T(f64 of f16 rows) {T_approx:.7f} | kernel L {L:.7f} U {U:.7f} T~ {T:.7f} eps {eps:.2e} b* {bstar} nabove {nabove} m0 {mc & 0xffff} m1 {mc >> 16} flags {flags}")
    print("   expected", e_idx[row][:6], e_dist[row][:6], " got", g_idx[row][:6], g_dist[row][:6])
    ex = set(e_idx[row][e_idx[row] >= 0]); gt = set(g_idx[row][g_idx[row] >= 0])
    for c in sorted(ex ^ gt):
        s_ex = float(fo.sims_f32(X[row:row+1], X[c:c+1])[0, 0])
        print(f"      cand {c}: exact sim {s_ex:.7f} approx {float(X16[c] @ X16[row]):.7f} in_expected {c in ex} in_got {c in gt}")
    m0, m1 = mc & 0xffff, mc >> 16
    mv = np.concatenate([memv[row, :m0], memv[row, 32:32 + m1]])
    print("   members:", len(mv), "range", mv.min() if len(mv) else None, mv.max() if len(mv) else None, " count approx >= T~:", int((approx >= T).sum()), " count approx > binhi:", None)
