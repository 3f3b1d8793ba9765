import sys, time, numpy as np, torch
sys.path.insert(0, '.')
from falcon_amd import synth, device as dv
from falcon_amd.cluster.cluster import AnnParams, ClusterPipeline, SpectrumDataset, n_list_rule
data = synth.generate(1000000)
ctx = dv.Context(0)
c = synth.select_charge(data, int(sys.argv[1]) if len(sys.argv) > 1 else 2)
ds = SpectrumDataset(*[ctx.to_dev(c[k], torch.float32) for k in ("precursor_mz", "retention_time", "mz", "intensity")], ctx.to_dev(c["indptr"], torch.int64))
p = AnnParams()
pipe = ClusterPipeline(ctx)
for _ in range(2): pipe.run(ds, 20.0, "ppm", None, 0.05, 2**15, p)
torch.cuda.synchronize()
T = {}
def tick(name, t0):
    T[name] = T.get(name, 0) + (time.perf_counter() - t0) * 1e3
for rep in range(5):
    t_all = time.perf_counter()
    t = time.perf_counter(); n_bins, start, _ = dv.get_dim(p.min_mz, p.max_mz, 0.05); tick("get_dim", t)
    t = time.perf_counter(); order, mzs = ctx.sort_by_precursor(ds.precursor_mz); tick("sort", t)
    t = time.perf_counter(); splits = ctx.precursor_splits(mzs, 20.0, "ppm", 2**15, 1.0); tick("splits", t)
    t = time.perf_counter(); X = ctx.vectorize(ds.mz, ds.intensity, ds.indptr, order, start, 0.05, n_bins, 400); tick("vectorize", t)
    t = time.perf_counter(); nl = n_list_rule(np.diff(splits), 16); tick("n_list_rule", t)
    t = time.perf_counter(); index = ctx.ivf_build(X, splits, nl, 10); tick("build", t)
    t = time.perf_counter(); sim, idx = index.search(16, 128); tick("search", t)
    t = time.perf_counter(); nbi, nbd = ctx.filter_neighbors(sim, idx, mzs, None, 20.0, "ppm", None, 64); tick("filter", t)
    t = time.perf_counter(); labels, medoids, _, _ = ctx.cluster_graph(nbi, nbd, 0.1, mzs, None, 20.0, "ppm", None, order); tick("cluster_graph(sync)", t)
    t = time.perf_counter(); index.close(); tick("close", t)
    t = time.perf_counter(); torch.cuda.synchronize(); tick("final sync", t)
    tick("TOTAL", t_all)
for k, v in T.items(): print(f"{k:22s} {v/5:8.3f} ms")
