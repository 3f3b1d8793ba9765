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
os.environ['FALCON_FUSED_SPLIT_TIMERS'] = '1'
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
    t = {k: ctx.stage_ms(k)[0] for k in ("build", "scan", "select", "filter")}
    print(f"dbg={dbg:4d} approx {t['build']:.3f}  band {t['scan']:.3f}  resolve {t['select']:.3f}  fallback {t['filter']:.3f} ms  fallback rows {ctx.counter(5)}", flush=True)


import ctypes
hip = ctypes.CDLL("libamdhip64.so")
why = np.zeros(4, np.int32)
hip.hipMemcpy(ctypes.c_void_p(why.ctypes.data), ctypes.c_void_p(ctx.counter(7)), ctypes.c_size_t(16), 2)
print("fallback rows", why[0], "reasons: approx-stage (members / T~)", why[1], " kept overflow", why[2], " need out of range", why[3])
thr = np.zeros((X.shape[0], 8), np.float32)
hip.hipMemcpy(ctypes.c_void_p(thr.ctypes.data), ctypes.c_void_p(ctx.counter(6)), ctypes.c_size_t(thr.nbytes), 2)
ti = thr.view(np.int32)
big = np.diff(splits)
rows_big = np.concatenate([np.arange(a, b) for a, b in zip(splits[:-1], splits[1:]) if b - a > 128])
mc = ti[rows_big, 6]
m0, m1 = mc & 0xffff, mc >> 16
print("members per half: mean", (m0 + m1).mean() / 2, "max", max(m0.max(), m1.max()), " rows with a full half:", int(((m0 >= 20) | (m1 >= 20)).sum()), "flags set", int((ti[rows_big, 7] & 2 > 0).sum()))
