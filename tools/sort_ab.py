"""precursor sort A/B: rocprim default (merge sort below 1 M keys) vs Onesweep radix passes (FALCON_SORT_ONESWEEP=1)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from falcon_amd.device import Context

ctx = Context(0)
for n in (50_000, 700_000, 1_000_000, 4_400_000, 10_000_000):
    g = torch.Generator(device=ctx.tdev).manual_seed(1)
    x = (300 + 1200 * torch.rand(n, device=ctx.tdev, generator=g)).float()
    x[::7] = x[3]                                    # ties: stability matters
    for _ in range(3):
        o, m = ctx.sort_by_precursor(x)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(20):
        o, m = ctx.sort_by_precursor(x)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / 20
    ref = torch.sort(x, stable=True)
    ok = bool(torch.equal(ref.indices, o)) and bool(torch.equal(ref.values, m))
    print(f"n={n}: {dt * 1e3:.3f} ms  identical to torch stable sort: {ok}")
