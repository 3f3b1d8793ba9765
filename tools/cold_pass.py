"""What `falcon.main()` actually pays: ONE pass per charge partition in a fresh process (reference falcon.py:153-193 calls
generate_clusters once per charge; nothing is warm).  Times, in a fresh process: library load, context creation, then the first,
second and third pass of the job on the production runner (PartitionRunner, two slots) -- the first pass grows every scratch slot,
pool block and torch allocation from nothing and loads the kernels' code objects.

    python tools/cold_pass.py [spectra] [--plan] [--mz-hi 1200] [--n_probe 16] [--low_dim 400]
prints one JSON line.  `--plan`: size the scratch before the first pass (`fal_ctx_plan` through PartitionRunner.plan)."""
import argparse, json, os, sys, time
t_proc = time.perf_counter()
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
ap = argparse.ArgumentParser()
ap.add_argument("spectra", type=int, nargs="?", default=1_000_000)
ap.add_argument("--plan", action="store_true")
ap.add_argument("--mz-hi", type=float, default=1200.0)
ap.add_argument("--n_probe", type=int, default=16)
ap.add_argument("--low_dim", type=int, default=400)
ap.add_argument("--passes", type=int, default=4)
args = ap.parse_args()
import torch
t0 = time.perf_counter()
from falcon_amd import _lib, synth
from falcon_amd.cluster.cluster import AnnParams, PartitionRunner, SpectrumDataset
_lib.load()
t_lib = time.perf_counter() - t0
dev = torch.device("cuda", 0)
data = synth.generate_device(args.spectra, dev, mz_hi=args.mz_hi)          # (torch kernels: HIP is initialised, our code objects are not)
parts = []
for ch in (2, 3):
    c = synth.select_charge_device(data, ch)
    parts.append(SpectrumDataset(c["precursor_mz"], c["retention_time"], c["mz"], c["intensity"], c["indptr"]))
del data, c
torch.cuda.synchronize()
torch.cuda.empty_cache()
p = AnnParams(n_probe=args.n_probe, low_dim=args.low_dim)
run_args = (20.0, "ppm", None, 0.05, 2 ** 15, p)
t0 = time.perf_counter()
runner = PartitionRunner(0, 2)
t_plan = None
if args.plan:
    t1 = time.perf_counter()
    runner.plan(parts, *run_args)
    torch.cuda.synchronize()
    t_plan = (time.perf_counter() - t1) * 1e3
ms = []
t_first = None
for i in range(args.passes):
    torch.cuda.synchronize()
    t1 = time.perf_counter()
    outs = runner.run(parts, *run_args)
    labels = torch.cat([o[0] for o in outs]).cpu()
    ms.append((time.perf_counter() - t1) * 1e3)
    if t_first is None:
        t_first = (time.perf_counter() - t0) * 1e3, (time.perf_counter() - t_proc) * 1e3
free, total = torch.cuda.mem_get_info()
print(json.dumps({"spectra": args.spectra, "plan": args.plan, "lib_load_ms": t_lib * 1e3, "plan_ms": t_plan,
                  "pass_ms": [round(x, 2) for x in ms], "cold_over_steady": ms[0] / min(ms[1:]),
                  "first_pass_from_runner_creation_ms": round(t_first[0], 2),
                  "process_ms_to_first_labels_incl_torch_import_and_data_generation": round(t_first[1], 1),
                  "device_memory_used_GB": (total - free) / 1e9, "clusters": int(labels.max()) + 1}))
runner.close()
