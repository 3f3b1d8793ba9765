"""Where a rank's time goes in the bucket-sharded job (strong scaling: ONE fixed dataset, default 10 M spectra, on `world`
GPUs), measured on one GPU for rank 0: host-synchronised phases of `ClusterPipeline` (front = window histogram, deal, the rank's
own sort and bucket boundaries; search; graph), next to the single-GPU pass of the same dataset.
python tools/shard_phases.py [world] [spectra]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from falcon_amd import synth
from falcon_amd.cluster.cluster import AnnParams, ClusterPipeline, SpectrumDataset
from falcon_amd.device import Context

world = int(sys.argv[1]) if len(sys.argv) > 1 else 8
n = int(sys.argv[2]) if len(sys.argv) > 2 else 10_000_000
ctx = Context(0)
pipe = ClusterPipeline(ctx)
data = synth.generate_device(n, ctx.tdev)
parts = []
for ch in (2, 3):
    c = synth.select_charge_device(data, ch)
    parts.append(SpectrumDataset(c["precursor_mz"], c["retention_time"], c["mz"], c["intensity"], c["indptr"]))
del data
p = AnnParams()
A = (20.0, "ppm", None)


def timed(fn):
    torch.cuda.synchronize(); t = time.perf_counter(); r = fn(); torch.cuda.synchronize()
    return r, (time.perf_counter() - t) * 1e3


def one_pass(shard):
    T = {"plan": 0.0, "front": 0.0, "search": 0.0, "graph": 0.0}
    if shard is not None:
        owners, T["plan"] = timed(lambda: pipe.plan_shards(ctx, parts, 2 ** 15, p, shard[1]))
    for j, ds in enumerate(parts):
        if shard is not None:
            def front():
                s2 = pipe._front_windows(ctx, ds, *A, 2 ** 15, p, shard, owner=owners[j])
                s2["order_local"] = torch.arange(s2["rows"].numel(), dtype=torch.int64, device=ctx.tdev)
                return s2
            sub, t = timed(front); T["front"] += t
            run = dict(sub, order=sub["rows"])
        else:
            st, t = timed(lambda: pipe._front(ctx, ds, *A, 2 ** 15, p)); T["front"] += t
            run = st
        _, t = timed(lambda: pipe._search(ds, run, *A, 0.05, p, False)); T["search"] += t
        if shard is not None:
            run["order"] = sub["order_local"]
        _, t = timed(lambda: pipe._graph(run, *A, p, False)); T["graph"] += t
    return T


for name, shard in (("single GPU", None), (f"rank 0 of {world}", (0, world))):
    for _ in range(2):
        one_pass(shard)
    T = one_pass(shard)
    tot = sum(T.values())
    print(f"{name}: " + "  ".join(f"{k} {v:.2f}" for k, v in T.items()) + f"  | sum {tot:.2f} ms", flush=True)
    for _ in range(2):
        pipe.run_many(parts, *A, 0.05, 2 ** 15, p, shard=shard)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(3):
        pipe.run_many(parts, *A, 0.05, 2 ** 15, p, shard=shard)
    torch.cuda.synchronize()
    print(f"{name}: run_many {(time.perf_counter() - t0) / 3 * 1e3:.2f} ms per pass", flush=True)
