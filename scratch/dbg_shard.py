import sys, numpy as np, torch, traceback
sys.path.insert(0, ".")
from falcon_amd import distributed as fd, synth
from falcon_amd.cluster.cluster import AnnParams, ClusterPipeline, SpectrumDataset
from falcon_amd.device import Context
ctx = Context(0)
d = synth.select_charge(synth.generate(6000, seed=23), 2)
ds = SpectrumDataset(d["precursor_mz"], d["retention_time"], d["mz"], d["intensity"], d["indptr"])
pipe = ClusterPipeline(ctx)
p = AnnParams(eps=0.3)
args = (20.0, "ppm", None, 0.05, 2 ** 15, p)
try:
    for r in range(3):
        out = fd.run_sharded(pipe, ds, *args, rank=r, world_size=3, local_only=True)
        print("rank", r, len(out[0]), len(out[2]), flush=True)
except Exception:
    traceback.print_exc()
