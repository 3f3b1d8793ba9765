import sys, numpy as np, torch
sys.path.insert(0, '.')
from falcon_amd.device import Context
ctx = Context(0)
rng = np.random.default_rng(0)
for nb, bs in ((8, 4096), (32, 2048), (64, 1024), (256, 512), (1024, 256), (2048, 128)):
    n = nb * bs
    X = torch.rand((n, 400), device=ctx.tdev)
    X = X / X.norm(dim=1, keepdim=True)
    off = np.arange(nb + 1, dtype=np.int64) * bs
    import os
    X16 = None
    if os.environ.get('F16'):
        hi = X.half(); lo = ((X - hi.float()) * 2048).half(); X16 = torch.stack([hi, lo], 1).contiguous()
    idx = ctx.ivf_build(X, off, np.ones(nb, np.int32), X16=X16)
    ctx.enable_timing(True)
    for _ in range(2):
        sim, ids = idx.search(16, 128)
    ms, k = ctx.stage_ms("scan")
    sel, _ = ctx.stage_ms("select")
    ctx.enable_timing(False)
    pairs = nb * bs * bs
    print(f"{nb:5d} buckets x {bs:5d}: scan {ms:8.3f} ms ({k} launches) {2*400*pairs/ms/1e9:7.1f} TFLOP/s   select {sel:7.3f} ms ({sel/n*1e3:.3f} us/query)")
    idx.close()
