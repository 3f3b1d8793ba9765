import sys, time, numpy as np, torch
sys.path.insert(0, '.')
from falcon_amd import synth, device as dv
from falcon_amd.cluster.cluster import AnnParams, ClusterPipeline, SpectrumDataset, n_list_rule
N = int(sys.argv[1]); scan = sys.argv[2] if len(sys.argv) > 2 else "f32"
n_probe = int(sys.argv[3]) if len(sys.argv) > 3 else 16
dtype = sys.argv[4] if len(sys.argv) > 4 else "f32"
low_dim = int(sys.argv[5]) if len(sys.argv) > 5 else 400
mz_lo = float(sys.argv[6]) if len(sys.argv) > 6 else 400.0      # a narrower precursor range = denser buckets (BASELINE configs[3] regime)
mz_hi = float(sys.argv[7]) if len(sys.argv) > 7 else 1200.0
batch_size = int(sys.argv[8]) if len(sys.argv) > 8 else 2 ** 15   # (2**16 with ~44 k-row windows: n_list 1,024 -- SURVEY 8d's C4 row)
ctx = dv.Context(0); pipe = ClusterPipeline(ctx)
t = time.time(); data = synth.generate_device(N, ctx.tdev, mz_lo=mz_lo, mz_hi=mz_hi); print(f"generated {N} in {time.time()-t:.1f}s", flush=True)
p = AnnParams(scan=scan, n_probe=n_probe, dtype=dtype, low_dim=low_dim)
parts = []
for ch in (2, 3):
    c = synth.select_charge_device(data, ch)
    parts.append(SpectrumDataset(*[ctx.to_dev(c[k], torch.float32) for k in ("precursor_mz", "retention_time", "mz", "intensity")], ctx.to_dev(c["indptr"], torch.int64)))
del data
REPS = 4
for rep in range(REPS):
    torch.cuda.synchronize(); t = time.time()
    if rep == REPS - 1: ctx.enable_timing(True)
    tot = 0; st = {}
    for ds in parts:
        labels, medoids = pipe.run(ds, 20.0, "ppm", None, 0.05, batch_size, p)
        tot += int(medoids.numel())
        if rep == REPS - 1:
            for k in ("vectorize", "build", "coarse", "scan", "select", "filter", "dbscan", "tail"):
                st[k] = st.get(k, 0) + ctx.stage_ms(k)[0]
            st["pairs"] = st.get("pairs", 0) + ctx.counter(0); st["coarse_pairs"] = st.get("coarse_pairs", 0) + ctx.counter(1)
    torch.cuda.synchronize(); dt = time.time() - t
    print(f"rep {rep}: {N} spectra in {dt*1e3:.1f} ms -> {N/dt/1e6:.1f} M spectra/s, {tot} clusters, mem {torch.cuda.max_memory_allocated()/2**30:.1f} GiB torch", flush=True)
print({k: (round(v, 2) if isinstance(v, float) else v) for k, v in st.items()}, "n_list max", int(np.max(pipe.last["n_list"])))
