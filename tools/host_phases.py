"""host time of the three phases of ClusterPipeline.run for one charge partition of the 1 M workload: `_front` (ends with a
synchronisation), `_search` (enqueue only) and `_graph` (ends with one), next to the GPU time of the whole pass."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from falcon_amd import synth
from falcon_amd.cluster.cluster import AnnParams, ClusterPipeline, SpectrumDataset
dev = torch.device("cuda:0")
data = synth.generate_device(1_000_000, dev)
c = synth.select_charge_device(data, int(sys.argv[1]) if len(sys.argv) > 1 else 2)
ds = SpectrumDataset(c["precursor_mz"], c["retention_time"], c["mz"], c["intensity"], c["indptr"])
pipe = ClusterPipeline(device=0)
p = AnnParams()
a = (20.0, "ppm", None)
for _ in range(3): pipe.run(ds, *a, 0.05, 2 ** 15, p)
torch.cuda.synchronize()
T = {"front": 0.0, "search (enqueue)": 0.0, "graph": 0.0, "total": 0.0}
reps = 20
for _ in range(reps):
    t0 = time.perf_counter()
    st = pipe._front(pipe.ctx, ds, *a, 2 ** 15, p)
    t1 = time.perf_counter()
    pipe._search(ds, st, *a, 0.05, p, False)
    t2 = time.perf_counter()
    lab, med, last = pipe._graph(st, *a, p, False)
    t3 = time.perf_counter()
    torch.cuda.synchronize()
    t4 = time.perf_counter()
    T["front"] += t1 - t0; T["search (enqueue)"] += t2 - t1; T["graph"] += t3 - t2; T["total"] += t4 - t0
print(len(ds), "rows;", {k: round(v / reps * 1e3, 3) for k, v in T.items()}, "ms")
