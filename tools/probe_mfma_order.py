"""Which summation order does the fp32 cosine kernel implement?  Compares the GPU's flat-bucket sims
(fal_ivf_search_topk with k = bucket size) with the candidate orders of oracle/kordered.c::fo_sims_mode."""
import ctypes as C
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from falcon_amd.device import Context
from oracle import falcon_oracle as fo

lib = fo._klib()
lib.fo_sims_mode.argtypes = [C.c_void_p, C.c_int64, C.c_void_p, C.c_int64, C.c_int, C.c_int, C.c_void_p]
ctx = Context(0)
rng = np.random.default_rng(1)
for d in (400, 64):
    n = 200
    X = np.abs(rng.normal(size=(n, d))).astype(np.float32) * (rng.random((n, d)) < 0.3)
    X /= np.maximum(np.linalg.norm(X, axis=1, keepdims=True), 1e-9)
    X = np.ascontiguousarray(X, np.float32)
    idx = ctx.ivf_build(torch.from_numpy(X).to(ctx.tdev), np.array([0, n]), np.array([1], np.int32))
    sim, ids = idx.search(1, 128)
    sim, ids = sim.cpu().numpy(), ids.cpu().numpy()
    for mode in range(5):
        S = np.empty((n, n), np.float32)
        lib.fo_sims_mode(X.ctypes.data, n, X.ctypes.data, n, d, mode, S.ctypes.data)
        ref = np.take_along_axis(S, ids.astype(np.int64), 1)
        print(f"d={d} mode={mode}: bit-equal entries {(ref == sim).mean():.6f}  max|diff| {np.abs(ref - sim).max():.3e}")
