import sys, time, torch, numpy as np
sys.path.insert(0, ".")
from falcon_amd import synth
from falcon_amd.cluster.cluster import AnnParams, ClusterPipeline, SpectrumDataset
from falcon_amd.device import Context
scan = sys.argv[1] if len(sys.argv) > 1 else "f32"
ctx = Context(0); pipe = ClusterPipeline(ctx)
data = synth.generate(1000000, seed=42)
parts = []
for ch in (2, 3):
    c = synth.select_charge(data, ch)
    parts.append(SpectrumDataset(*[ctx.to_dev(c[k], torch.float32) for k in ("precursor_mz", "retention_time", "mz", "intensity")], ctx.to_dev(c["indptr"], torch.int64)))
p = AnnParams(scan=scan)
args = (20.0, "ppm", None, 0.05, 32768, p)
def step_many():
    outs = pipe.run_many(parts, *args)
    return torch.cat([o[0] for o in outs]).cpu()
def step_serial():
    outs = [pipe.run(ds, *args) for ds in parts]
    return torch.cat([o[0] for o in outs]).cpu()
for name, fn in (("serial", step_serial), ("many", step_many), ("serial", step_serial), ("many", step_many)):
    ts = []
    for i in range(8):
        torch.cuda.synchronize(); t = time.perf_counter(); fn(); torch.cuda.synchronize(); ts.append((time.perf_counter() - t) * 1e3)
    print(scan, name, " ".join(f"{x:.2f}" for x in ts))
# phases of run_many, host-side timing
import falcon_amd.cluster.cluster as cl
orig_front, orig_search, orig_graph = pipe._front, pipe._search, pipe._graph
T = []
def wrap(name, f):
    def g(*a, **k):
        t = time.perf_counter(); r = f(*a, **k); T.append((name, (time.perf_counter() - t) * 1e3)); return r
    return g
pipe._front, pipe._search, pipe._graph = wrap("front", orig_front), wrap("search", orig_search), wrap("graph", orig_graph)
for i in range(3):
    T.clear(); torch.cuda.synchronize(); t = time.perf_counter(); step_many(); torch.cuda.synchronize()
    print("many phases", " ".join(f"{n}={v:.2f}" for n, v in T), f"total={(time.perf_counter()-t)*1e3:.2f}")
