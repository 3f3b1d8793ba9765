"""10 M spectra (float32, the IVF regime) once through the whole path: for rocprofv3 --kernel-trace --stats."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from falcon_amd import synth
from falcon_amd.cluster.cluster import AnnParams, ClusterPipeline, SpectrumDataset
from falcon_amd.device import Context

n = int(sys.argv[1]) if len(sys.argv) > 1 else 10_000_000
dtype = sys.argv[2] if len(sys.argv) > 2 else "f32"
ctx = Context(0)
pipe = ClusterPipeline(ctx)
data = synth.generate_device(n, ctx.tdev)
parts = []
for ch in (2, 3):
    c = synth.select_charge_device(data, ch)
    parts.append(SpectrumDataset(c["precursor_mz"], c["retention_time"], c["mz"], c["intensity"], c["indptr"]))
del data
p = AnnParams(dtype=dtype, low_dim=800 if dtype == "f16" else 400)
args = (20.0, "ppm", None, 0.05, 2 ** 15, p)
for _ in range(3):
    pipe.run_many(parts, *args)
torch.cuda.synchronize()
print("done")
