"""A/B of the forms of the indexed fine scan (FALCON_LIST16 unset: the default -- list16s_kernel, sparse query records, since round 6 | "d":
list16_kernel, dense query rows | "r": list16r_kernel; FALCON_AB_FORMS=lockstep,d picks the forms, "lockstep" = unset) on one GPU: the same
dataset, serial staged passes in turn (HIP events around every stage), labels compared.

    python tools/list16_ab.py [spectra] [mz_lo] [mz_hi] [n_probe] [rounds]
defaults: 2,500,000 spectra in 400-600 m/z = the bucket density of the 10 M job (8,750 rows per charge-2 window, 128 lists)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from falcon_amd import synth
from falcon_amd.cluster.cluster import AnnParams, ClusterPipeline, SpectrumDataset
from falcon_amd.device import Context

n = int(sys.argv[1]) if len(sys.argv) > 1 else 2_500_000
mz_lo = float(sys.argv[2]) if len(sys.argv) > 2 else 400.0
mz_hi = float(sys.argv[3]) if len(sys.argv) > 3 else 600.0
n_probe = int(sys.argv[4]) if len(sys.argv) > 4 else 16
rounds = int(sys.argv[5]) if len(sys.argv) > 5 else 3
forms = os.environ.get("FALCON_AB_FORMS", "lockstep,r").split(",")
ctx = Context(0)
data = synth.generate_device(n, ctx.tdev, mz_lo=mz_lo, mz_hi=mz_hi)
parts = []
for ch in (2, 3):
    c = synth.select_charge_device(data, ch)
    parts.append(SpectrumDataset(c["precursor_mz"], c["retention_time"], c["mz"], c["intensity"], c["indptr"]))
del data
args = (20.0, "ppm", None, 0.05, 2 ** 15, AnnParams(n_probe=n_probe, low_dim=int(os.environ.get("FALCON_AB_LOW_DIM", "400")),
                                                    dtype=os.environ.get("FALCON_AB_DTYPE", "f32")))
pipe = ClusterPipeline(ctx)
STAGES = ("vectorize", "build", "coarse", "scan", "select", "filter", "dbscan", "tail", "kernel")


def one_pass(form):
    os.environ.pop("FALCON_L16_KNOCK", None)
    if form == "lockstep":
        os.environ.pop("FALCON_LIST16", None)
    elif form[0] == "k":             # k1 / k2 / k3: the default form with FALCON_L16_KNOCK (timing experiments, wrong results)
        os.environ.pop("FALCON_LIST16", None)
        os.environ["FALCON_L16_KNOCK"] = form[1:]
        os.environ["FALCON_TIMING_EXPERIMENTS"] = "1"
    else:
        os.environ["FALCON_LIST16"] = form
    tot = {k: 0.0 for k in STAGES}
    labs = []
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for ds in parts:
        lab, med = pipe.run(ds, *args)
        labs.append(lab)
        for k in STAGES:
            tot[k] += ctx.stage_ms(k)[0]
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) * 1e3, tot, torch.cat(labs).cpu()


for f in forms:                      # untimed: scratch pools to their steady-state size
    one_pass(f)
ctx.enable_timing(True)
ref = None
for rnd in range(rounds):
    for f in forms:
        ms, tot, lab = one_pass(f)
        if ref is None:
            ref = lab
        same = bool(torch.equal(ref, lab))
        print(f"n={n} [{mz_lo:.0f},{mz_hi:.0f}) n_probe={n_probe} round {rnd} form={f:9s} pass {ms:8.2f} ms  list16 {tot['kernel']:7.2f}  "
              f"scan {tot['scan']:7.2f}  select {tot['select']:7.2f}  build {tot['build']:7.2f}  coarse {tot['coarse']:6.2f}  "
              f"labels identical {same}", flush=True)
print("n_list max", int(pipe.last["n_list"].max()))
