"""The per-GPU share of the N-GPU job of bench.py --gpus N, measured on ONE GPU: a dataset of N x 1 M spectra, every simulated
rank runs the path on its own precursor buckets (run_many(shard=(rank, N)): same LPT deal as the real run).  No exchange, no
concurrency between ranks: a PROJECTION of the compute side of the multi-GPU run, not a measurement of it."""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from falcon_amd import synth
from falcon_amd.cluster.cluster import AnnParams, ClusterPipeline, SpectrumDataset
from falcon_amd.device import Context

world = int(sys.argv[1]) if len(sys.argv) > 1 else 8
per_gpu = int(sys.argv[2]) if len(sys.argv) > 2 else 1_000_000
ctx = Context(0)
pipe = ClusterPipeline(ctx)
data = synth.generate_device(world * per_gpu, ctx.tdev)
parts = []
for ch in (2, 3):
    c = synth.select_charge_device(data, ch)
    parts.append(SpectrumDataset(c["precursor_mz"], c["retention_time"], c["mz"], c["intensity"], c["indptr"]))
del data
args = (20.0, "ppm", None, 0.05, 2 ** 15, AnnParams())
out = {"world": world, "spectra": world * per_gpu, "ranks": []}
only = [int(x) for x in sys.argv[3].split(',')] if len(sys.argv) > 3 else list(range(world))
for rank in only:
    for _ in range(2):
        pipe.run_many(parts, *args, shard=(rank, world))
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    reps = 3
    for _ in range(reps):
        outs = pipe.run_many(parts, *args, shard=(rank, world))
    torch.cuda.synchronize()
    ms = (time.perf_counter() - t0) / reps * 1e3
    rows = sum(int(o[0].numel()) for o in outs)
    # the same share with the two charge partitions on concurrent streams (PartitionRunner)
    if "runner" not in globals():
        from falcon_amd.cluster.cluster import PartitionRunner
        runner = PartitionRunner(0, 2)
    for _ in range(2):
        runner.run(parts, *args, shard=(rank, world))
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        runner.run(parts, *args, shard=(rank, world))
    torch.cuda.synchronize()
    ms2 = (time.perf_counter() - t0) / reps * 1e3
    out["ranks"].append({"rank": rank, "rows": rows, "ms": round(ms, 2), "ms_concurrent_partitions": round(ms2, 2)})
    print(f"rank {rank}: {rows} rows in {ms:.1f} ms (pipelined partitions), {ms2:.1f} ms (concurrent partitions)", flush=True)
worst = max(r["ms"] for r in out["ranks"])
out["slowest_rank_ms"] = worst
out["projected_spectra_per_s"] = world * per_gpu / (worst * 1e-3)
out["note"] = ("one GPU running every rank's share in turn: the compute side of the window-sharded job, every phase of the rank "
               "included (window histogram, deal, its own sort, the path); the all-gatherv of the neighbour lists is not included")
print(json.dumps(out))
