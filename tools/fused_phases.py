"""Timing experiment: the fused flat scan with phases switched off (FALCON_FUSED_DBG bits; results invalid)."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from falcon_amd import device as _device, synth
from falcon_amd.cluster.cluster import n_list_rule
from falcon_amd.device import Context

n = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
modes = [int(x) for x in sys.argv[2].split(",")] if len(sys.argv) > 2 else \
    [0, 64, 1 + 64, 2 + 64, 3 + 64, 4 + 64, 8 + 64, 16 + 64, 32 + 4 + 64, 32 + 4 + 8 + 64, 32 + 4 + 8 + 16 + 64, 3 + 4 + 8 + 16 + 64]
ctx = Context(0)
data = synth.generate_device(n, ctx.tdev)
c = synth.select_charge_device(data, 2)
order, mzs = ctx.sort_by_precursor(c["precursor_mz"])
splits = ctx.precursor_splits(mzs, 20.0, "ppm", 2 ** 15, 1.0)
n_list = n_list_rule(np.diff(splits), 16)
n_bins, start, _ = _device.get_dim(101.0, 1500.0, 0.05)
vec = lambda dt: ctx.vectorize(c["mz"], c["intensity"], c["indptr"], order, start, 0.05, n_bins, 400, 0, True, dt)
X, X16 = vec("f32"), vec("f16")
print("rows", X.shape[0], "buckets", len(n_list), "flat", int((n_list == 1).sum()), "max", int(np.diff(splits).max()), flush=True)
index = ctx.ivf_build(X, splits, n_list, 10, Xpre=X16)
plain = ctx.ivf_build(X, splits, n_list, 10)
ctx.enable_timing(True)
for _ in range(2):
    plain.search_neighbors(16, 128, mzs, None, 20.0, "ppm", None, 64)
print("staged: scan %.3f select %.3f" % (ctx.stage_ms("scan")[0], ctx.stage_ms("select")[0]), flush=True)
for dbg in modes:
    os.environ["FALCON_FUSED_DBG"] = str(dbg)
    for _ in range(2):
        index.search_neighbors(16, 128, mzs, None, 20.0, "ppm", None, 64)
    s, f = ctx.stage_ms("scan")[0], ctx.stage_ms("select")[0]
    print(f"dbg={dbg:3d} fused {s:.3f} ms  fallback kernel {f:.3f} ms  fallback rows {ctx.counter(5)}", flush=True)

# ---- in-kernel time stamps (dbg bit 128): cycles per phase, per workgroup -------------------------------------------
import ctypes
os.environ["FALCON_FUSED_DBG"] = "128"
index.search_neighbors(16, 128, mzs, None, 20.0, "ppm", None, 64)
ctx.sync()
ptr, nwg = ctx.counter(6), ctx.counter(7)
hip = ctypes.CDLL("libamdhip64.so")
buf = np.zeros((nwg, 10), np.uint64)
hip.hipMemcpy(ctypes.c_void_p(buf.ctypes.data), ctypes.c_void_p(ptr), ctypes.c_size_t(buf.nbytes), 2)
live = buf[:, 0] > 0
st = buf[live].astype(np.int64)
nc = st[:, 8]
names = ["pass1", "binsearch", "pass2", "Ttilde", "band", "exact-resolve", "rank+store"]
print("workgroups", live.sum(), "of grid", nwg, " with passes (nc>128):", int((nc > 128).sum()))
t0, t1 = st[:, 0].min(), st[:, 7].max()
print("kernel span cycles (s_memtime ticks)", t1 - t0)
big = nc > 128
for name, sel in (("nc>128", big), ("nc<=128", ~big)):
    if sel.sum() == 0:
        continue
    s = st[sel]
    print(name, "count", sel.sum(), "mean nc", s[:, 8].mean(), "mean band chunks (wave 0)", s[:, 9].mean())
    prev = s[:, 0]
    for i, nme in enumerate(names):
        cur = np.where(s[:, i + 1] > 0, s[:, i + 1], prev)
        print(f"   {nme:14s} mean {np.mean(cur - prev):10.0f} ticks   total/CU {np.sum(cur - prev) / 256:12.0f}")
        prev = cur
    chunks = np.ceil(s[:, 8] / 32)
    if name == "nc>128":
        p1 = np.where(s[:, 1] > 0, s[:, 1] - s[:, 0], 0)
        print("   pass1 ticks per chunk:", np.sum(p1) / np.sum(chunks))
