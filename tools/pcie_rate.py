"""Whole-job rate when the seam hands HOST buffers (pageable numpy / pinned tensors) instead of device-resident ones."""
import sys, time, numpy as np, torch
sys.path.insert(0, ".")
from falcon_amd import synth
from falcon_amd.cluster.cluster import AnnParams, ClusterPipeline, SpectrumDataset
from falcon_amd.device import Context
ctx = Context(0); pipe = ClusterPipeline(ctx)
data = synth.generate(1_000_000, seed=42)
p = AnnParams(); args = (20.0, "ppm", None, 0.05, 2 ** 15, p)
keys = ("precursor_mz", "retention_time", "mz", "intensity")
host, pinned, dev = [], [], []
for ch in (2, 3):
    c = synth.select_charge(data, ch)
    host.append(SpectrumDataset(*[np.ascontiguousarray(c[k], np.float32) for k in keys], c["indptr"].astype(np.int64)))
    pinned.append(SpectrumDataset(*[torch.from_numpy(np.ascontiguousarray(c[k], np.float32)).pin_memory() for k in keys],
                                  torch.from_numpy(c["indptr"].astype(np.int64)).pin_memory()))
    dev.append(SpectrumDataset(*[ctx.to_dev(c[k], torch.float32) for k in keys], ctx.to_dev(c["indptr"], torch.int64)))
mb = sum(sum(np.asarray(getattr(d, k)).nbytes for k in ("precursor_mz", "retention_time", "mz", "intensity", "indptr")) for d in host) / 1e6
for name, parts in (("device-resident", dev), ("pinned host", pinned), ("pageable host", host)):
    for _ in range(3): pipe.run_many(parts, *args)
    torch.cuda.synchronize(); t = time.perf_counter()
    for _ in range(10):
        outs = pipe.run_many(parts, *args); lab = torch.cat([o[0] for o in outs]).cpu()
    torch.cuda.synchronize(); dt = (time.perf_counter() - t) / 10
    print(f"{name:16s} {dt*1e3:7.2f} ms/step  {1e6/dt/1e6:7.1f} M spectra/s   (input {mb:.0f} MB)")
