"""Where a rank's time goes in the bucket-sharded job (strong scaling: ONE fixed dataset, default 10 M spectra, on `world`
GPUs), measured on one GPU for rank 0: host-synchronised phases of `ClusterPipeline` (front = sort + bucket boundaries of the
WHOLE dataset, replicated on every rank; restrict = the LPT deal + the rank's row subset; search; graph), next to the
single-GPU pass of the same dataset.   python tools/shard_phases.py [world] [spectra]"""
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
    T = {"front": 0.0, "restrict": 0.0, "search": 0.0, "graph": 0.0}
    for ds in parts:
        st, t = timed(lambda: pipe._front(ctx, ds, *A, 2 ** 15, p)); T["front"] += t
        if shard is not None:
            def restrict():
                s2 = pipe._restrict(ctx, st, p, shard)
                s2["order_local"] = torch.arange(s2["rows"].numel(), dtype=torch.int64, device=ctx.tdev)
                return s2
            sub, t = timed(restrict); T["restrict"] += t
            run = dict(sub, order=sub["rows"])
        else:
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
