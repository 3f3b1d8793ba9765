#!/usr/bin/env python3
"""Headline benchmark: spectra clustered / second through the whole hot path
(vectorise -> buckets -> IVF/flat cosine scan -> top-k -> filter -> DBSCAN -> refine ->
medoids/labels) on synthetic peak lists, plus the cosine kernel's roofline fraction.

    python bench.py --gpus 1 --steps 3 --warmup 1
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W

A "step" = one pass of the hot path over this rank's shard (default 1,000,000 synthetic
spectra = BASELINE.json configs[1]; both charge partitions, like falcon.py:151-193, software-pipelined
on the one GPU by `ClusterPipeline.run_many`; `--serial` runs them strictly one after the other).
Inputs are resident in HBM before the timed region; the step ends with the labels on the
host.  Before the W warmup steps three untimed passes bring the scratch pool / caching
allocator to their steady-state sizes (setup, like the data generation).
Weak scaling: every rank owns an independent shard (its own 1M-spectrum block of
the generator = its own (charge, bucket) units); the only collective is the result
all-gatherv (CSR neighbour lists + labels by default, asynchronous: it travels while the next step
computes; `--exchange labels|none` for the cheaper exchanges).

Prints ONE JSON line on rank 0.
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

PEAK_MFMA_F32_TFLOPS = 157.3     # /opt/skills/guides/MI355X_MICROARCH.md "Peak FP32 (matrix)"
PEAK_HBM_GBS = 8000.0            # ... "HBM3E peak BW 8.0 TB/s spec"
PEAK_MFMA_F16_TFLOPS = 2500.0    # ... "Peak BF16/FP16 MFMA ~2.5 PF dense"


def cpu_baseline(data, params, seconds_target=20.0):
    """The oracle (numpy restatement, kind "port") timed on this box's host cores on a bounded
    sample of the SAME workload: the charge-2 spectra of a precursor-m/z slice, which keeps
    the bucket density (and so the work per spectrum) of the full run."""
    from oracle import falcon_oracle as fo
    from falcon_amd import synth
    try:
        from threadpoolctl import threadpool_limits
    except Exception:                                    # pragma: no cover
        threadpool_limits = None
    c2 = synth.select_charge(data, 2)
    pm = c2["precursor_mz"]
    lo = 600.0
    width = 200.0          # ~175k spectra: 10-30 s of single-thread numpy
    sel = np.flatnonzero((pm >= lo) & (pm < lo + width))
    if len(sel) < 256:
        sel = np.arange(min(len(pm), 20000))
    counts = np.diff(c2["indptr"])[sel]
    indptr = np.zeros(len(sel) + 1, np.int64)
    np.cumsum(counts, out=indptr[1:])
    src = np.repeat(c2["indptr"][:-1][sel] - indptr[:-1], counts) + np.arange(int(counts.sum()))
    args = (c2["mz"][src], c2["intensity"][src], indptr, pm[sel], c2["retention_time"][sel])
    kw = dict(eps=params.eps, low_dim=params.low_dim, n_probe=params.n_probe, n_neighbors=params.n_neighbors,
              n_neighbors_ann=params.n_neighbors_ann, mz_interval=params.mz_interval,
              kmeans_iters=params.kmeans_iters)
    ctxm = threadpool_limits(limits=1) if threadpool_limits else None
    t0 = time.perf_counter()
    if ctxm:
        with ctxm:
            fo.generate_clusters(*args, **kw)
    else:
        fo.generate_clusters(*args, **kw)
    dt = time.perf_counter() - t0
    return {"value": len(sel) / dt, "unit": "spectra/s", "cores": 1, "kind": "port",
            "sample": f"{len(sel)} charge-2 spectra with precursor m/z in [{lo:.0f},{lo + width:.0f}) of the "
                      f"rank-0 shard (same bucket density), oracle/falcon_oracle.py, numpy 1 thread, "
                      f"{dt:.1f} s"}


def pmc_traffic(args):
    """HBM bytes per scan launch from the rocprofv3 PMC passes committed under profiles/ (FETCH_SIZE x 2
    + WRITE_SIZE, collected in separate passes as MI355X_MICROARCH.md prescribes).  Counters cannot be
    read from inside the run, so the figure is only reported for the workload it was measured on."""
    fn = os.path.join(ROOT, "profiles", "r1_pmc_hbm_traffic_per_step.json")
    default = (args.spectra == 1_000_000 and args.low_dim == 400 and args.n_neighbors_ann == 128
               and args.mz_interval == 1.0 and args.batch_size == 2 ** 15)
    if not default or not os.path.isfile(fn):
        return None
    try:
        with open(fn) as f:
            return float(json.load(f)["scan"]["hbm_bytes_per_launch"])
    except Exception:
        return None


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--spectra", type=int, default=1_000_000, help="spectra per GPU")
    ap.add_argument("--low_dim", type=int, default=400)
    ap.add_argument("--n_probe", type=int, default=16)
    ap.add_argument("--n_neighbors", type=int, default=64)
    ap.add_argument("--n_neighbors_ann", type=int, default=128)
    ap.add_argument("--eps", type=float, default=0.10)
    ap.add_argument("--batch_size", type=int, default=2 ** 15)
    ap.add_argument("--mz_interval", type=float, default=1.0)
    ap.add_argument("--exchange", choices=["neighbors", "labels", "none"], default="neighbors")
    ap.add_argument("--scan", choices=["f32", "f16x3"], default="f32",
                    help="flat-scan arithmetic: exact fp32 MFMA (default) or hi/lo float16 split on the f16 MFMA")
    ap.add_argument("--dtype", choices=["f32", "f16"], default="f32", help="vector dtype (f16 = BASELINE config 5)")
    ap.add_argument("--rescore", action="store_true",
                    help="re-score the neighbours with the matched-peak cosine before DBSCAN (SURVEY 8f-4; not the headline)")
    ap.add_argument("--force-exchange", action="store_true",
                    help="run the exchange step's device work (CSR packing, payload) at 1 GPU too (no collective)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--serial", action="store_true",
                    help="run the charge partitions strictly one after the other (default: software-pipelined, "
                         "ClusterPipeline.run_many)")
    ap.add_argument("--overlap", action="store_true",
                    help="run the charge partitions on two host threads / two streams (PartitionRunner)")
    args = ap.parse_args()

    import torch
    import torch.distributed as dist
    from falcon_amd import synth
    from falcon_amd import distributed as fdist
    from falcon_amd.cluster.cluster import AnnParams, ClusterPipeline, PartitionRunner, SpectrumDataset
    from falcon_amd.device import Context

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if not torch.cuda.is_available():
        sys.exit("bench.py needs a HIP device (falcon_amd has no CPU fallback)")
    # test hooks (1-GPU box): FALCON_BENCH_DEVICE pins every rank to one device, FALCON_BENCH_BACKEND=gloo
    # swaps RCCL for gloo so the N-rank control flow can be exercised where only one GPU exists
    local_rank = int(os.environ.get("FALCON_BENCH_DEVICE", local_rank))
    backend = os.environ.get("FALCON_BENCH_BACKEND", "nccl")
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
        else:
            dist.init_process_group(backend, rank=rank, world_size=world)

    ctx = Context(local_rank)
    pipe = ClusterPipeline(ctx)
    runner = PartitionRunner(local_rank, 2) if args.overlap else None
    p = AnnParams(eps=args.eps, low_dim=args.low_dim, n_probe=args.n_probe, n_neighbors=args.n_neighbors,
                  n_neighbors_ann=args.n_neighbors_ann, mz_interval=args.mz_interval, scan=args.scan, dtype=args.dtype,
                  rescore=args.rescore, min_matches=6 if args.rescore else 0)

    # ---- this rank's shard: its own generator blocks, resident in HBM ------------------------
    blocks_per_rank = (args.spectra + synth.BLOCK - 1) // synth.BLOCK
    data = synth.generate(args.spectra, seed=42, first_block=rank * blocks_per_rank)
    parts = []
    for charge in (2, 3):                                       # falcon.py:151-160
        c = synth.select_charge(data, charge)
        parts.append(SpectrumDataset(ctx.to_dev(c["precursor_mz"], torch.float32),
                                     ctx.to_dev(c["retention_time"], torch.float32),
                                     ctx.to_dev(c["mz"], torch.float32), ctx.to_dev(c["intensity"], torch.float32),
                                     ctx.to_dev(c["indptr"], torch.int64)))
    n_local = sum(len(x) for x in parts)
    row_offset = rank * args.spectra

    def barrier():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
            torch.cuda.synchronize()

    run_args = (20.0, "ppm", None, 0.05, args.batch_size, p)
    exchanging = args.exchange != "none" and (world > 1 or args.force_exchange)
    keep_nb = args.exchange == "neighbors" and exchanging
    exchange = fdist.SparseGraphExchange(dev)
    pending, csr_buf = [], {}

    def step(collect=None):
        """one pass of the hot path over this rank's shard; `collect` != None: serial, per-stage timing"""
        if runner is not None and collect is None:
            outs = runner.run(parts, *run_args)                                   # partitions on two threads/streams
            lasts = [pp.last for pp in runner.last_pipes] if keep_nb else []
        elif collect is None and not args.serial:
            outs = pipe.run_many(parts, *run_args)                                # partitions software-pipelined
            lasts = [dict(x) for x in pipe.lasts] if keep_nb else []
        else:
            outs, lasts = [], []
            for ds in parts:
                outs.append(pipe.run(ds, *run_args))
                if keep_nb:
                    lasts.append(dict(pipe.last))
                if collect is not None:
                    collect.append({k: ctx.stage_ms(k) for k in ("vectorize", "build", "coarse", "scan", "select",
                                                                  "filter", "dbscan", "tail")}
                                   | {"pairs": ctx.counter(0), "coarse_pairs": ctx.counter(1), "issued": ctx.counter(4),
                                      "scan_launches": ctx.counter(2), "sims_bytes": ctx.counter(3), "n": len(ds)})
        labels_all, current = [], 0
        for labels, medoids in outs:
            labels_all.append(labels + current)                  # falcon.py:189-193
            current += int(medoids.numel())
        labels = torch.cat(labels_all)
        if not exchanging:
            return labels.cpu()
        # ---- the one exchange step: all-gatherv of the sparse neighbour lists (CSR, ids ->
        # global sorted rows of the job) + labels; asynchronous: it travels while the next step computes
        if args.exchange == "neighbors":
            rows = sum(last["nb_idx"].shape[0] for last in lasts)
            if "buf" not in csr_buf:
                cap = rows * args.n_neighbors
                csr_buf["buf"] = (torch.empty(rows + 1, dtype=torch.int64, device=dev),
                                  torch.empty(cap, dtype=torch.int32, device=dev),
                                  torch.empty(cap, dtype=torch.float32, device=dev))
            row0 = 0
            for last in lasts:                                   # charge partitions chain into one CSR on the device
                csr = ctx.neighbors_to_csr(last["nb_idx"], last["nb_dist"], row_offset + row0, out=csr_buf["buf"], row0=row0,
                                           nb_count=last.get("nb_count"))
                row0 += last["nb_idx"].shape[0]
        else:                                                    # labels only: an empty graph
            csr = (torch.zeros(labels.numel() + 1, dtype=torch.int64, device=dev),
                   torch.empty(1, dtype=torch.int32, device=dev), torch.empty(1, dtype=torch.float32, device=dev))
        handle = exchange.start(csr[0], csr[1], csr[2], labels, current)
        done = finish_pending()
        pending.append(handle)
        return done

    def finish_pending():
        if not pending:
            return None
        g = exchange.finish(pending.pop())
        # the gathered graph and the globally unique labels of the whole job stay device-resident on every rank;
        # a rank copies its own shard's labels to the host
        return g["labels"][rank].cpu()

    # setup, like the data generation above: the first passes grow the library's scratch pool and torch's caching
    # allocator to their steady-state sizes (GB-sized hipMallocs, tens of ms each) -- prime them before the contract's
    # W warmup steps so that neither W nor the timed K steps contain one-off allocations
    for _ in range(3):
        step()
    finish_pending()
    for _ in range(args.warmup):
        step()
    finish_pending()
    barrier()
    trace = os.environ.get("FALCON_BENCH_TRACE") is not None      # per-step wall times on stderr (adds a sync per step)
    t0 = time.perf_counter()
    for _ in range(args.steps):
        ts = time.perf_counter()
        step()
        if trace:
            torch.cuda.synchronize()
            print(f"[bench] step {(time.perf_counter() - ts) * 1e3:.2f} ms", file=sys.stderr)
    finish_pending()                                             # the last exchange lands inside the timed region
    barrier()
    dt = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([dt], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())

    # ---- per-kernel timing (HIP events on the kernels' own stream), outside the timed region --
    ctx.enable_timing(True)
    stages = []
    step(stages)
    finish_pending()
    ctx.enable_timing(False)

    if rank == 0:
        d = args.low_dim
        pairs = sum(s["pairs"] for s in stages)
        scan_ms = sum(s["scan"][0] for s in stages)
        scan_launches = sum(s["scan"][1] for s in stages)
        flops = 2.0 * d * pairs
        achieved_tf = flops / (scan_ms * 1e-3) / 1e12 if scan_ms > 0 else 0.0
        n_rows = sum(s["n"] for s in stages)
        algo_bytes = n_rows * (2 * d * 4 + 8 * args.n_neighbors_ann)          # SURVEY 8(d) compulsory bytes
        stage_ms = {k: round(sum(s[k][0] for s in stages), 3) for k in
                    ("vectorize", "build", "coarse", "scan", "select", "filter", "dbscan", "tail")}
        f16_path = args.scan == "f16x3" or args.dtype == "f16"
        elem = 2 if args.dtype == "f16" else 4
        algo_bytes = n_rows * (2 * d * elem + 8 * args.n_neighbors_ann)
        hbm_gbs = algo_bytes / (scan_ms * 1e-3) / 1e9 if scan_ms > 0 else 0.0
        if f16_path:
            # issued matrix work: 3 f16 MFMAs per k-step for the split, 1 for plain float16 rows
            issued_tf = achieved_tf * (3 if args.scan == "f16x3" and args.dtype == "f32" else 1)
            mfma_frac, hbm_frac = issued_tf / PEAK_MFMA_F16_TFLOPS, hbm_gbs / PEAK_HBM_GBS
            if hbm_frac >= mfma_frac:
                roof = {"bound": "hbm", "achieved": hbm_gbs, "peak": PEAK_HBM_GBS, "unit": "GB/s", "frac": hbm_frac}
            else:
                roof = {"bound": "mfma", "achieved": issued_tf, "peak": PEAK_MFMA_F16_TFLOPS, "unit": "TFLOP/s",
                        "frac": mfma_frac}
            roof.update({"kernel": "scan16_kernel (f16 MFMA 32x32x16, LDS-staged) + dense_kernel for buckets < 64",
                         "traffic": None, "mfma_f16_frac": mfma_frac, "hbm_frac": hbm_frac,
                         "algorithmic_tflops": achieved_tf})
        out = {
            "metric": "spectra clustered/sec @1/2/4/8 GPU; cosine-kernel HBM GB/s vs roofline",
            "value": n_local * world * args.steps / dt,
            "unit": "spectra/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": dt / args.steps * 1e3,
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": ("f16" if args.dtype == "f16" else "f16x3 (f32 vectors as hi/lo float16, f32 accumulate)"
                      if args.scan == "f16x3" else "f32"),
            "data": "synthetic",
            "config": {"workload": f"{args.spectra} synthetic spectra per GPU (charges 2+3), low_dim={d}, "
                                   f"n_neighbors={args.n_neighbors}, n_neighbors_ann={args.n_neighbors_ann}, "
                                   f"n_probe={args.n_probe}, eps={args.eps}, precursor_tol=20ppm, "
                                   f"mz_interval={args.mz_interval}, batch_size={args.batch_size}",
                       "exchange": (args.exchange + " (CSR all-gatherv, overlapped with the next step)") if exchanging else "none",
                       "partitions": "2 host threads / 2 streams" if args.overlap else "serial",
                       "parallelism": f"bucket-sharded x{world}"},
            "roofline": {"kernel": "dense_kernel<.,STORE> (flat buckets) / ivf_list4_kernel (IVF buckets): cosine scan, fp32 MFMA 32x32x2",
                         "bound": "mfma", "achieved": achieved_tf, "peak": PEAK_MFMA_F32_TFLOPS,
                         "unit": "TFLOP/s", "frac": achieved_tf / PEAK_MFMA_F32_TFLOPS, "traffic": pmc_traffic(args),
                         "traffic_unit": "HBM bytes per launch (PMC, profiles/r1_pmc_hbm_traffic_per_step.json)",
                         "flops_per_launch": flops / max(scan_launches, 1),
                         "launches": scan_launches, "avg_launch_ms": scan_ms / max(scan_launches, 1),
                         "pairs_per_step": pairs,
                         # the kernel computes each bucket's similarity matrix on/above the diagonal only
                         # (bit-identical by symmetry): machine flops actually issued, tile padding included
                         "issued_tflops": (2.0 * d * sum(s["issued"] for s in stages) / (scan_ms * 1e-3) / 1e12
                                           if scan_ms > 0 else 0.0)},
            "roofline_hbm": {"kernel": "cosine scan", "bound": "hbm",
                             "achieved": algo_bytes / (scan_ms * 1e-3) / 1e9 if scan_ms > 0 else 0.0,
                             "peak": PEAK_HBM_GBS, "unit": "GB/s",
                             "frac": (algo_bytes / (scan_ms * 1e-3) / 1e9 / PEAK_HBM_GBS) if scan_ms > 0 else 0.0,
                             "algorithmic_bytes": algo_bytes},
            "stage_ms": stage_ms,
        }
        if f16_path:
            roof.update({"launches": scan_launches, "avg_launch_ms": scan_ms / max(scan_launches, 1),
                         "pairs_per_step": pairs})
            out["roofline"] = roof
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(data, p)
        line = json.dumps(out)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()
    if rank == 0:
        # RCCL's banner / warnings sit in the C stdio buffer of this process (NCCL_DEBUG output goes to
        # stdout): flush them first so that the JSON line is the LAST line of stdout
        import ctypes
        ctypes.CDLL(None).fflush(None)
        sys.stdout.flush()
        print(line, flush=True)


if __name__ == "__main__":
    main()
