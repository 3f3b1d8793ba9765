"""Race hunt for the index build (VERDICT r3 next #1d): repeat the exact (all-fp32, `assign_kernel`) and the float16-prefiltered
k-means build of a 512 / 64 / 200-list index many times on fixed data and compare EVERY repetition with the oracle's index
(computed once on the CPU).  Options make a late LDS-DMA likely: low_dim 64 / 128 (a chunk's compute phase is only 32 / 64 MFMAs
long) and `--hammer` (a second stream copies a few GB back and forth all the time: the row DMAs queue behind it in HBM).

    python tools/stress_build.py REPS [--d 128] [--sparse] [--hammer] [--lib PATH]

`--lib` loads another build of the same library (A/B against a kept round-3 .so: it must FAIL there and pass here)."""
import argparse
import os
import sys
import threading

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
ap = argparse.ArgumentParser()
ap.add_argument("reps", type=int, nargs="?", default=50)
ap.add_argument("--d", type=int, default=128)
ap.add_argument("--sparse", action="store_true")
ap.add_argument("--hammer", action="store_true")
ap.add_argument("--lib", default=None)
ap.add_argument("--keyed", action="store_true", help="also repeat the float16-prefiltered build")
args = ap.parse_args()

from falcon_amd import _lib
if args.lib:
    _lib.LIB_PATH = os.path.abspath(args.lib)
    _lib._SIGNATURES = {k: v for k, v in _lib._SIGNATURES.items()}      # (an older build may lack newer exports)
import numpy as np
import torch
from falcon_amd.device import Context
from tests.test_gpu_stress import NLIST, ITERS, compare_index, describe_first_difference, oracle_index, stress_data

if args.lib:
    import ctypes
    probe = ctypes.CDLL(_lib.LIB_PATH)
    _lib._SIGNATURES = {k: v for k, v in _lib._SIGNATURES.items() if hasattr(probe, k)}
ctx = Context(0)
X, off = stress_data(args.sparse, args.d)
ref = oracle_index(X, off)
nl = np.array(NLIST, np.int32)
Xd = torch.from_numpy(X).to(ctx.tdev)
X16 = Xd.to(torch.float16).contiguous()

stop = False


def hammer():
    s2 = torch.cuda.Stream(device=ctx.tdev)
    a = torch.empty(1 << 30, dtype=torch.uint8, device=ctx.tdev)
    b = torch.empty(1 << 30, dtype=torch.uint8, device=ctx.tdev)
    with torch.cuda.stream(s2):
        while not stop:
            for _ in range(8):
                b.copy_(a)
                a.copy_(b)
            s2.synchronize()


th = None
if args.hammer:
    th = threading.Thread(target=hammer, daemon=True)
    th.start()

bad = {"plain": 0, "keyed": 0}
for it in range(args.reps):
    for which in (("plain", "keyed") if args.keyed else ("plain",)):
        index = ctx.ivf_build(Xd, off, nl, kmeans_iters=ITERS, Xkm=X16 if which == "keyed" else None)
        diff, got = compare_index(ref, index.export())
        if diff:
            bad[which] += 1
            if bad[which] <= 10:
                print(f"rep {it}: {which} build left the oracle in {diff}: {describe_first_difference(X, off, ref, got)}", flush=True)
        index.close()
stop = True
if th:
    th.join()
print(f"lib={os.path.basename(_lib.LIB_PATH)} d={args.d} sparse={args.sparse} hammer={args.hammer}: {args.reps} repetitions, "
      f"{bad['plain']} plain and {bad['keyed']} keyed builds differ from the oracle")
sys.exit(1 if (bad["plain"] or bad["keyed"]) else 0)
