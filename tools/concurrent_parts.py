"""charge partitions one after the other (run_many) vs concurrently (PartitionRunner: a host thread + stream + context each)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from falcon_amd import synth
from falcon_amd.cluster.cluster import AnnParams, ClusterPipeline, PartitionRunner, SpectrumDataset
from falcon_amd.device import Context

n = int(sys.argv[1]) if len(sys.argv) > 1 else 10_000_000
ctx = Context(0)
data = synth.generate_device(n, ctx.tdev)
parts = []
for ch in (2, 3):
    c = synth.select_charge_device(data, ch)
    parts.append(SpectrumDataset(c["precursor_mz"], c["retention_time"], c["mz"], c["intensity"], c["indptr"]))
del data
args = (20.0, "ppm", None, 0.05, 2 ** 15, AnnParams())
pipe = ClusterPipeline(ctx)
runner = PartitionRunner(0, 2)
def t(fn, reps):
    for _ in range(2): fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(reps): outs = fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps * 1e3, outs
reps = 5 if n > 2_000_000 else 30
def full(fn):
    def g():
        outs = fn()
        cur, lab = 0, []
        for l, m in outs:
            lab.append(l + cur); cur += int(m.numel())
        return torch.cat(lab).cpu()
    return g
for rnd in range(3):
    a, oa = t(full(lambda: pipe.run_many(parts, *args)), reps)
    b, ob = t(full(lambda: runner.run(parts, *args)), reps)
    print(f"n={n} round {rnd}: run_many {a:.2f} ms   PartitionRunner {b:.2f} ms   identical {bool(torch.equal(oa, ob))}")
runner.close()
