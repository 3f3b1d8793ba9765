import sys, os, numpy as np, torch
sys.path.insert(0, '.')
from falcon_amd.device import Context
ctx = Context(0)
for nb, bs in ((64, 4096), (512, 1024), (4096, 256)):
    n = nb * bs
    X = torch.rand((n, 400), device=ctx.tdev); X = X / X.norm(dim=1, keepdim=True)
    off = np.arange(nb + 1, dtype=np.int64) * bs
    idx = ctx.ivf_build(X, off, np.ones(nb, np.int32))
    ctx.enable_timing(True)
    for _ in range(2): sim, ids = idx.search(16, 128)
    ms, k = ctx.stage_ms("scan")
    ctx.enable_timing(False)
    print(f"{nb:5d} x {bs:5d}: scan {ms:8.3f} ms ({k} launches) full-matrix rate {2*400*nb*bs*bs/ms/1e9:7.1f} TFLOP/s")
    idx.close()
