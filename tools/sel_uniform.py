import sys, numpy as np, torch
sys.path.insert(0, '.')
from falcon_amd.device import Context
ctx = Context(0)
for nb, bs in ((256, 512), (128, 1024)):
    n = nb * bs
    X = torch.rand((n, 400), device=ctx.tdev); X = X / X.norm(dim=1, keepdim=True)
    off = np.arange(nb + 1, dtype=np.int64) * bs
    idx = ctx.ivf_build(X, off, np.ones(nb, np.int32))
    ctx.enable_timing(True)
    for _ in range(2):
        sim, ids = idx.search(16, 128)
    sel, k = ctx.stage_ms("select")
    ctx.enable_timing(False)
    print(f"{nb} x {bs}: select {sel:7.3f} ms  {sel/n*1e6:.1f} ns/query")
    idx.close()
