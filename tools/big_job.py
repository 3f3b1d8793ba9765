import sys, time, traceback, torch
sys.path.insert(0, '.')
from falcon_amd import synth
from falcon_amd.cluster.cluster import AnnParams, PartitionRunner, SpectrumDataset
from falcon_amd.device import Context
ctx = Context(0)
n = int(sys.argv[1]); chunks = int(sys.argv[2])
data = synth.generate_device(n, ctx.tdev)
parts = []
for ch in (2, 3):
    c = synth.select_charge_device(data, ch)
    parts.append(SpectrumDataset(c["precursor_mz"], c["retention_time"], c["mz"], c["intensity"], c["indptr"]))
del data, c
torch.cuda.empty_cache()
free, total = torch.cuda.mem_get_info(); print("after dataset: used GB", (total - free) / 1e9, flush=True)
runner = PartitionRunner(0, 2)
args = (20.0, "ppm", None, 0.05, 2 ** 15, AnnParams(n_probe=32))
try:
    for i in range(2):
        t0 = time.perf_counter()
        outs = runner.run_chunked(parts, *args, n_chunks=chunks)
        torch.cuda.synchronize()
        free, total = torch.cuda.mem_get_info()
        print("pass", i, (time.perf_counter() - t0) * 1e3, "ms; used GB", (total - free) / 1e9, "torch reserved", torch.cuda.memory_reserved() / 1e9, flush=True)
except Exception:
    traceback.print_exc()
    free, total = torch.cuda.mem_get_info(); print("at failure: used GB", (total - free) / 1e9, "torch reserved", torch.cuda.memory_reserved() / 1e9, "allocated", torch.cuda.memory_allocated() / 1e9)
runner.close()
