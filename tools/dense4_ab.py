import os, sys, numpy as np, torch
sys.path.insert(0, '.')
from falcon_amd.device import Context
ctx = Context(0)
nb, rows, d = 64, 8192, 400
X = torch.rand(nb * rows, d, device=ctx.tdev)
X = (X * (torch.rand_like(X) < 0.12)).contiguous()
X = X / X.norm(dim=1, keepdim=True).clamp_min(1e-9)
off = np.arange(nb + 1, dtype=np.int64) * rows
nl = np.ones(nb, np.int32)
ideal_ms = nb * (rows // 32) * (rows // 32 + 1) / 2 * 200 * 64 / 1024 / 2.4e9 * 1e3
res = {}
modes = [("4", 0), ("ab", 0), ("4", 0), ("ab", 0)] + ([("ab", k) for k in (1, 2, 4, 3, 7, 8, 15)] if os.environ.get("FALCON_AB_EXPERIMENT") else [])
for mode, knock in modes:
    os.environ["FALCON_DENSE4"] = mode
    os.environ["FALCON_AB_KNOCK"] = str(knock)
    os.environ["FALCON_TIMING_EXPERIMENTS"] = "1"
    idx = ctx.ivf_build(X, off, nl)
    idx.search(1, 128); ctx.sync()
    ctx.enable_timing(True)
    s, i = idx.search(1, 128); ctx.sync()
    ms, k = ctx.stage_ms("kernel")
    ctx.enable_timing(False)
    print(mode, "knock", knock, "kernel ms", round(ms, 3), "launches", k, "ideal", round(ideal_ms, 3), "pipe busy", round(ideal_ms / ms, 3), flush=True)
    if knock == 0:
        res.setdefault(mode, (s, i))
        assert torch.equal(res[mode][0], s) and torch.equal(res["4"][1], i) and torch.equal(res["4"][0].view(torch.int32), s.view(torch.int32))
    idx.close()
