#!/usr/bin/env python3
"""Headline benchmark: spectra clustered / second through the whole hot path
(vectorise -> buckets -> IVF/flat cosine scan -> top-k -> filter -> DBSCAN -> refine ->
medoids/labels) on synthetic peak lists, plus the cosine kernel's roofline fraction.

    python bench.py --gpus 1 --steps 100 --warmup 2
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W

The job is ONE dataset of N x `--spectra` synthetic spectra (N = GPUs; default 1,000,000 per GPU =
BASELINE.json configs[1] at N = 1), both charge partitions (falcon.py:151-193).  A "step" = one pass of the
whole hot path over that dataset:

  * every rank sorts the precursors of the whole dataset and derives the SAME precursor buckets (cheap,
    deterministic), buckets are dealt to ranks by longest-processing-time on their scan cost, a rank runs
    vectorise -> ... -> labels on its own buckets (no data-path collective: no neighbour pair crosses a bucket,
    reference cluster.py:107-141);
  * ONE all-gatherv (RCCL over xGMI) of the CSR neighbour lists + labels + dataset rows gives every rank the
    global sparse graph and the globally unique labels (rank-major offsets like falcon.py:189-193), which the
    rank copies to the host.  At N = 1 there is nothing to exchange and the step is the single-GPU pipeline.

Inputs are resident in HBM before the timed region (`value`); `value_host_to_host` times the same step with the
peak arrays starting in pinned host memory.  Rank 0 prints ONE JSON line; at N = 1 it also carries
`configs`: the 10 M-spectra float32 run (BASELINE configs[2]'s dataset on one GPU = the north star's target size)
and the 10 M / low_dim 800 / float16 run (configs[4]), and `cpu_baseline` (the oracle on all host cores).
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
STAGES = ("vectorize", "build", "coarse", "scan", "select", "filter", "dbscan", "tail")


def cpu_baseline(host, params, seconds_hint=20.0):
    """The oracle (numpy + oracle/kordered.c, kind "port") on ALL host cores: precursor buckets are clustered on a
    thread pool the way the reference runs its blocks (joblib threading backend, cluster.py:115-136).  Bounded
    sample of the SAME workload: the charge-2 spectra of a precursor-m/z slice, which keeps the bucket density (and
    so the work per spectrum) of the full run."""
    from oracle import falcon_oracle as fo
    pm = host["precursor_mz"]
    lo, width = 600.0, 120.0          # >= 100 k spectra at the default density
    sel = np.flatnonzero((pm >= lo) & (pm < lo + width))
    if len(sel) < 256:
        sel = np.arange(min(len(pm), 20000))
    counts = np.diff(host["indptr"])[sel]
    indptr = np.zeros(len(sel) + 1, np.int64)
    np.cumsum(counts, out=indptr[1:])
    src = np.repeat(host["indptr"][:-1][sel] - indptr[:-1], counts) + np.arange(int(counts.sum()))
    args = (host["mz"][src], host["intensity"][src], indptr, pm[sel], host["retention_time"][sel])
    kw = dict(eps=params.eps, low_dim=params.low_dim, n_probe=params.n_probe, n_neighbors=params.n_neighbors,
              n_neighbors_ann=params.n_neighbors_ann, mz_interval=params.mz_interval,
              kmeans_iters=params.kmeans_iters)
    cores = os.cpu_count() or 1
    t0 = time.perf_counter()
    fo.generate_clusters(*args, n_jobs=cores, **kw)
    dt = time.perf_counter() - t0
    return {"value": len(sel) / dt, "unit": "spectra/s", "cores": cores, "kind": "port",
            "sample": f"{len(sel)} charge-2 spectra with precursor m/z in [{lo:.0f},{lo + width:.0f}) of the dataset "
                      f"(same bucket density, same parameters), oracle/falcon_oracle.py + oracle/kordered.c (fmaf, AVX2), "
                      f"buckets on a pool of {cores} threads, {dt:.1f} s"}


def pmc_traffic(args):
    """HBM bytes per step of the cosine kernels from the rocprofv3 PMC passes committed under profiles/ (FETCH_SIZE x 2
    + WRITE_SIZE, separate passes, MI355X_MICROARCH.md).  Counters cannot be read from inside the run, so the
    figure is only reported for the workload it was measured on."""
    default = (args.spectra == 1_000_000 and args.low_dim == 400 and args.n_neighbors_ann == 128
               and args.mz_interval == 1.0 and args.batch_size == 2 ** 15 and args.scan == "f32" and args.dtype == "f32")
    if not default:
        return None, None
    for fn in ("r2_pmc_hbm_traffic_per_step.json", "r1_pmc_hbm_traffic_per_step.json"):
        path = os.path.join(ROOT, "profiles", fn)
        if os.path.isfile(path):
            try:
                with open(path) as f:
                    return float(json.load(f)["scan"]["hbm_bytes_per_launch"]), "profiles/" + fn
            except Exception:
                pass
    return None, None


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=100)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--spectra", type=int, default=1_000_000, help="spectra per GPU (the dataset holds gpus x spectra)")
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
    ap.add_argument("--prefilter", action="store_true",
                    help="flat buckets: top-k kept on chip (f16-MFMA prefilter + exact float32 window, fused.hip; same results)")
    ap.add_argument("--no-ivf-prefilter", action="store_true",
                    help="buckets with an index: exact fp32-MFMA fine scan of every probed pair instead of the f16-MFMA prefilter + "
                         "exact float32 refinement (ivf16.hip; same results)")
    ap.add_argument("--rescore", action="store_true",
                    help="re-score the neighbours with the matched-peak cosine before DBSCAN (SURVEY 8f-4; not the headline)")
    ap.add_argument("--generator", choices=["device", "numpy"], default="device",
                    help="synthetic data: falcon_amd.synth on the GPU (default; same recipe, torch random stream) or the "
                         "numpy generator the parity tests use (~20 s of host time per million spectra)")
    ap.add_argument("--force-exchange", action="store_true",
                    help="run the exchange step's device work (CSR packing, payload) at 1 GPU too (no collective)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-configs", action="store_true", help="skip the 10 M-spectra configurations of the `configs` array")
    ap.add_argument("--configs-spectra", type=int, default=10_000_000)
    ap.add_argument("--partitions", choices=["auto", "concurrent", "pipelined"], default="auto",
                    help="how the two charge partitions of a step are scheduled: concurrent = a host thread + HIP stream + context "
                         "each (PartitionRunner; the reference clusters its blocks on a thread pool, cluster.py:115-136), pipelined = "
                         "one stream, the next partition's sort under the current scan (ClusterPipeline.run_many); auto = concurrent "
                         "on one GPU for the headline workload, pipelined for the 10 M configurations and for N > 1")
    ap.add_argument("--serial", action="store_true",
                    help="run the charge partitions strictly one after the other (default: software-pipelined, "
                         "ClusterPipeline.run_many)")
    args = ap.parse_args()

    import torch
    import torch.distributed as dist
    from falcon_amd import synth
    from falcon_amd import distributed as fdist
    from falcon_amd.cluster.cluster import AnnParams, ClusterPipeline, SpectrumDataset
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

    def params(**kw):
        base = dict(eps=args.eps, low_dim=args.low_dim, n_probe=args.n_probe, n_neighbors=args.n_neighbors,
                    n_neighbors_ann=args.n_neighbors_ann, mz_interval=args.mz_interval, scan=args.scan, dtype=args.dtype,
                    rescore=args.rescore, min_matches=6 if args.rescore else 0, prefilter=args.prefilter,
                    ivf_prefilter=not args.no_ivf_prefilter)
        base.update(kw)
        return AnnParams(**base)

    def make_parts(n_total, first_block=0, mz_lo=400.0, mz_hi=1200.0):
        """the dataset, split by precursor charge (falcon.py:151-160), resident in HBM -> [SpectrumDataset], host view"""
        if args.generator == "device":
            data = synth.generate_device(n_total, dev, seed=42, first_block=first_block, mz_lo=mz_lo, mz_hi=mz_hi)
            sel = lambda c: synth.select_charge_device(data, c)
        else:
            data = synth.generate(n_total, seed=42, first_block=first_block, mz_lo=mz_lo, mz_hi=mz_hi)
            sel = lambda c: synth.select_charge(data, c)
        parts = []
        for charge in (2, 3):
            c = sel(charge)
            parts.append(SpectrumDataset(ctx.to_dev(c["precursor_mz"], torch.float32),
                                         ctx.to_dev(c["retention_time"], torch.float32),
                                         ctx.to_dev(c["mz"], torch.float32), ctx.to_dev(c["intensity"], torch.float32),
                                         ctx.to_dev(c["indptr"], torch.int64)))
        return parts

    def barrier():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
            torch.cuda.synchronize()

    # ---- the dataset: world x spectra, every rank holds it (a rank only touches the peaks of its own buckets) ----
    n_total = world * args.spectra
    parts = make_parts(n_total)
    part_off = np.concatenate([[0], np.cumsum([len(x) for x in parts])])
    p = params()
    run_args = (20.0, "ppm", None, 0.05, args.batch_size, p)
    shard = (rank, world) if world > 1 else None
    exchanging = args.exchange != "none" and (world > 1 or args.force_exchange)
    keep_nb = args.exchange == "neighbors" and exchanging
    exchange = fdist.SparseGraphExchange(dev)
    pending, csr_buf = [], {}

    def collect_stages(n):
        return ({k: ctx.stage_ms(k) for k in STAGES}
                | {"pairs": ctx.counter(0), "coarse_pairs": ctx.counter(1), "issued": ctx.counter(4),
                   "scan_launches": ctx.counter(2), "sims_bytes": ctx.counter(3), "n": n})

    concurrent = {"on": False}
    runner = None

    def step(parts, run_args, collect=None):
        """one pass of the hot path over the dataset; `collect` != None: serial, per-stage timing, no exchange"""
        if collect is None and not args.serial and concurrent["on"]:
            outs = runner.run(parts, *run_args)                                   # partitions on concurrent streams
            lasts = [dict(pp.last) for pp in runner.last_pipes]
        elif collect is None and not args.serial:
            outs = pipe.run_many(parts, *run_args, shard=shard)                   # partitions software-pipelined
            lasts = pipe.lasts
        else:
            outs, lasts = [], []
            for ds in parts:
                if shard is not None:
                    o = pipe.run_many([ds], *run_args, shard=shard)
                    outs.append(o[0])
                    lasts.append(pipe.lasts[0])
                else:
                    outs.append(pipe.run(ds, *run_args))
                    lasts.append(dict(pipe.last))
                if collect is not None:
                    collect.append(collect_stages(int(outs[-1][0].numel())))
        if not exchanging or collect is not None:
            labels_all, current = [], 0
            for labels, medoids in outs:
                labels_all.append(labels + current)                  # falcon.py:189-193
                current += int(medoids.numel())
            return torch.cat(labels_all).cpu()
        # ---- the one exchange step: all-gatherv of the sparse neighbour lists (CSR, ids -> dataset
        # rows of the job) + labels + dataset rows; asynchronous: it travels while the next step computes
        handle, _, _ = fdist.start_graph_exchange(ctx, exchange, outs, lasts, part_off, args.n_neighbors, shard is not None,
                                                  with_neighbors=args.exchange == "neighbors", csr_buf=csr_buf)
        done = finish_pending()
        pending.append(handle)
        return done

    def finish_pending():
        if not pending:
            return None
        g = exchange.finish(pending.pop())
        # the gathered graph stays device-resident on every rank; the globally unique labels of the WHOLE dataset
        # are assembled from the shards and copied to the host
        if g["rows"]:
            return fdist.SparseGraphExchange.assemble_labels(g, n_total).cpu()
        return g["labels"][rank].cpu()

    def timed(parts, run_args, steps, warmup, prime=3):
        # setup, like the data generation: the first passes grow the library's scratch pool and torch's caching
        # allocator to their steady-state sizes (GB-sized hipMallocs, tens of ms each) -- prime them before the
        # contract's W warmup steps so that neither W nor the timed K steps contain one-off allocations
        for _ in range(prime):
            step(parts, run_args)
        finish_pending()
        for _ in range(warmup):
            step(parts, run_args)
        finish_pending()
        barrier()
        t0 = time.perf_counter()
        for _ in range(steps):
            step(parts, run_args)
        finish_pending()                                         # the last exchange lands inside the timed region
        barrier()
        dt = time.perf_counter() - t0
        if world > 1:
            t = torch.tensor([dt], dtype=torch.float64, device=dev)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            dt = float(t.item())
        return dt

    def staged(parts, run_args):
        """per-kernel timing (HIP events on the kernels' own stream), outside any timed region"""
        ctx.enable_timing(True)
        stages = []
        step(parts, run_args, stages)
        ctx.enable_timing(False)
        return stages

    if args.partitions == "concurrent" or (args.partitions == "auto" and world == 1 and not exchanging and not args.serial):
        from falcon_amd.cluster.cluster import PartitionRunner
        runner = PartitionRunner(local_rank, 2)
        concurrent["on"] = True
    dt = timed(parts, run_args, args.steps, args.warmup)
    stages = staged(parts, run_args)

    # ---- host-to-host: the peak arrays start in pinned host memory, labels end on the host -------
    h2h = None
    if world == 1:
        pinned = [[t.cpu().pin_memory() for t in (x.precursor_mz, x.retention_time, x.mz, x.intensity, x.indptr)] for x in parts]

        def upload():
            return [SpectrumDataset(*[t.to(dev, non_blocking=True) for t in ts]) for ts in pinned]

        k2 = max(1, min(args.steps, 20))
        for _ in range(2):
            step(upload(), run_args)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(k2):
            step(upload(), run_args)
        torch.cuda.synchronize()
        h2h = (time.perf_counter() - t0) / k2
        del pinned

    def summarize(stages, d, p, elem):
        pairs = sum(s["pairs"] for s in stages)
        ms = {k: round(sum(s[k][0] for s in stages), 3) for k in STAGES}
        scan_ms = ms["scan"]
        topk_ms = ms["scan"] + ms["select"] + ms["filter"]
        n_rows = sum(s["n"] for s in stages)
        launches = sum(s["scan"][1] for s in stages)
        flops = 2.0 * d * pairs
        algo_bytes = n_rows * (2 * d * elem + 8 * p.n_neighbors_ann)          # SURVEY 8(d) compulsory bytes
        tf = lambda f, t: f / (t * 1e-3) / 1e12 if t > 0 else 0.0
        gbs = lambda b, t: b / (t * 1e-3) / 1e9 if t > 0 else 0.0
        build_flops = 2.0 * d * sum(s["coarse_pairs"] for s in stages) * (p.kmeans_iters + 1)
        return dict(pairs=pairs, stage_ms=ms, scan_ms=scan_ms, topk_ms=topk_ms, n_rows=n_rows, launches=launches,
                    flops=flops, algo_bytes=algo_bytes, scan_tflops=tf(flops, scan_ms), cosine_tflops=tf(flops, topk_ms),
                    issued_tflops=tf(2.0 * d * sum(s["issued"] for s in stages), scan_ms),
                    hbm_gbs_scan=gbs(algo_bytes, scan_ms), hbm_gbs_cosine=gbs(algo_bytes, topk_ms),
                    build_tflops=tf(build_flops, ms["build"]), build_flops=build_flops,
                    coarse_tflops=tf(2.0 * d * sum(s["coarse_pairs"] for s in stages), ms["coarse"]))

    # ---- the 10 M configurations on ONE GPU (north star target size; BASELINE configs[2] dataset / configs[4]) --
    extra = []
    if world == 1 and not args.no_configs and rank == 0:
        del parts
        torch.cuda.empty_cache()
        if args.partitions == "auto":
            concurrent["on"] = False          # 10 M spectra: two concurrent partitions lose (2.4x slower; tools/concurrent_parts.py)
        big = make_parts(args.configs_spectra)
        for name, kw in (("f32", dict(low_dim=400, dtype="f32", scan="f32")),
                         ("f16", dict(low_dim=800, dtype="f16", scan="f32")),
                         # BASELINE configs[3]'s bucket regime on one GPU: the same number of spectra in a quarter of the precursor
                         # range (buckets of 20-35 k rows: n_list 512) with that config's n_probe = 32
                         ("f32-dense", dict(low_dim=400, dtype="f32", scan="f32", n_probe=32))):
            if name == "f32-dense":
                del big
                torch.cuda.empty_cache()
                big = make_parts(args.configs_spectra, mz_lo=400.0, mz_hi=600.0)
            pc = params(**kw)
            ra = (20.0, "ppm", None, 0.05, args.batch_size, pc)
            try:
                dtc = timed(big, ra, 3, 1, prime=1)
                sc = summarize(staged(big, ra), pc.low_dim, pc, 2 if name == "f16" else 4)
            except Exception as e:                                # pragma: no cover -- reported, never hidden
                extra.append({"workload": f"{args.configs_spectra} spectra {name}", "error": repr(e)[:300]})
                continue
            n_big = sum(len(x) for x in big)
            peak = PEAK_MFMA_F16_TFLOPS if name == "f16" else PEAK_MFMA_F32_TFLOPS
            extra.append({
                "workload": f"{args.configs_spectra} synthetic spectra on 1 GPU (charges 2+3"
                            f"{', precursor m/z 400-600' if name == 'f32-dense' else ''}), low_dim={pc.low_dim} {name.split('-')[0]}, "
                            f"n_neighbors={pc.n_neighbors}, n_neighbors_ann={pc.n_neighbors_ann}, n_probe={pc.n_probe}, "
                            f"eps={pc.eps}, precursor_tol=20ppm, mz_interval={pc.mz_interval}",
                "baseline_config": {"f32": "configs[2] dataset on one GPU", "f16": "configs[4]",
                                    "f32-dense": "configs[3] regime (n_list 512, n_probe 32, n_neighbors_ann 128) at one GPU's size: "
                                                 "precursors in 400-600 m/z"}[name],
                "steps": 3, "ms_per_step": dtc / 3 * 1e3, "value": n_big * 3 / dtc, "unit": "spectra/s", "dtype": name.split("-")[0],
                "stage_ms": sc["stage_ms"], "pairs_per_step": sc["pairs"],
                "cosine_kernel": {"scan_ms": sc["scan_ms"], "scan_plus_topk_ms": sc["topk_ms"],
                                  "scan_tflops": sc["scan_tflops"], "scan_frac_of_mfma_peak": sc["scan_tflops"] / peak,
                                  "scan_plus_topk_tflops": sc["cosine_tflops"],
                                  "scan_plus_topk_frac_of_mfma_peak": sc["cosine_tflops"] / peak,
                                  "algorithmic_bytes": sc["algo_bytes"],
                                  "scan_plus_topk_frac_of_hbm_roof": sc["hbm_gbs_cosine"] / PEAK_HBM_GBS,
                                  "mfma_peak_tflops": peak},
                "kmeans": {"build_ms": sc["stage_ms"]["build"], "algorithmic_tflops": sc["build_tflops"],
                           "note": "2*d*n_list flop per row and pass, 11 passes; computed on the f16 matrix cores (float16 prefilter) "
                                   "with exact float32 re-evaluation of close calls: not fp32-MFMA work, the index is identical"},
                "coarse": {"ms": sc["stage_ms"]["coarse"], "tflops": sc["coarse_tflops"]},
                "fine_scan": ("float16 rows scanned on the f16 matrix cores (config 5 vectors)" if name == "f16" else
                              ("exact fp32-MFMA scan of every probed pair + wavefront select" if args.no_ivf_prefilter else
                               "f16-MFMA list scan to 16-bit keys + k-th key per query (stage scan/select), exact fp32-MFMA "
                               "similarities of the precursor window + exact resolution of the k-th key: bit-identical to the "
                               "exact scan (tests/test_gpu_ivf16.py); the TFLOP/s above count the algorithmic 2*d flop per probed "
                               "pair, most of which run as float16")),
                "prefilter_fallback_rows": int(ctx.counter(5)),
            })
        del big
        torch.cuda.empty_cache()
        parts = make_parts(n_total)

    if rank == 0:
        d = args.low_dim
        elem = 2 if args.dtype == "f16" else 4
        s = summarize(stages, d, p, elem)
        f16_path = args.scan == "f16x3" or args.dtype == "f16"
        traffic, traffic_src = pmc_traffic(args)
        if f16_path:
            # issued matrix work: 3 f16 MFMAs per k-step for the split, 1 for plain float16 rows
            issued_tf = s["scan_tflops"] * (3 if args.scan == "f16x3" and args.dtype == "f32" else 1)
            mfma_frac, hbm_frac = issued_tf / PEAK_MFMA_F16_TFLOPS, s["hbm_gbs_scan"] / PEAK_HBM_GBS
            if hbm_frac >= mfma_frac:
                roof = {"bound": "hbm", "achieved": s["hbm_gbs_scan"], "peak": PEAK_HBM_GBS, "unit": "GB/s", "frac": hbm_frac}
            else:
                roof = {"bound": "mfma", "achieved": issued_tf, "peak": PEAK_MFMA_F16_TFLOPS, "unit": "TFLOP/s",
                        "frac": mfma_frac}
            roof.update({"kernel": "scan16_kernel (f16 MFMA 32x32x16, LDS-staged) + dense_kernel for buckets < 64",
                         "traffic": None, "mfma_f16_frac": mfma_frac, "hbm_frac": hbm_frac,
                         "algorithmic_tflops": s["scan_tflops"]})
        else:
            roof = {"kernel": "dense_kernel<.,STORE> (flat buckets) / ivf_list4_kernel (IVF buckets): cosine scan, fp32 MFMA 32x32x2",
                    "bound": "mfma", "achieved": s["scan_tflops"], "peak": PEAK_MFMA_F32_TFLOPS, "unit": "TFLOP/s",
                    "frac": s["scan_tflops"] / PEAK_MFMA_F32_TFLOPS, "traffic": traffic,
                    "traffic_unit": f"HBM bytes per launch (PMC, {traffic_src})" if traffic_src else None,
                    "flops_per_launch": s["flops"] / max(s["launches"], 1),
                    # the kernel computes each bucket's similarity matrix on/above the diagonal only (bit-identical by
                    # symmetry): `issued` = machine flops actually run through the matrix pipe, tile padding included
                    "issued_tflops": s["issued_tflops"], "issued_frac": s["issued_tflops"] / PEAK_MFMA_F32_TFLOPS}
        roof.update({"launches": s["launches"], "avg_launch_ms": s["scan_ms"] / max(s["launches"], 1),
                     "pairs_per_step": s["pairs"],
                     # SURVEY 8d defines the cosine kernel as list scan + top-k: the same algorithmic work over the
                     # scan AND the select / filter launches
                     "scan_plus_topk": {"ms": s["topk_ms"], "tflops": s["cosine_tflops"],
                                        "frac_of_f32_mfma_peak": s["cosine_tflops"] / PEAK_MFMA_F32_TFLOPS,
                                        "algorithmic_bytes": s["algo_bytes"], "gbs": s["hbm_gbs_cosine"],
                                        "frac_of_hbm_roof": s["hbm_gbs_cosine"] / PEAK_HBM_GBS}})
        out = {
            "metric": "spectra clustered/sec @1/2/4/8 GPU; cosine-kernel HBM GB/s vs roofline",
            "value": n_total * args.steps / dt,
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
            "config": {"workload": f"one dataset of {n_total} synthetic spectra ({args.spectra} per GPU; charges 2+3), low_dim={d}, "
                                   f"n_neighbors={args.n_neighbors}, n_neighbors_ann={args.n_neighbors_ann}, "
                                   f"n_probe={args.n_probe}, eps={args.eps}, precursor_tol=20ppm, "
                                   f"mz_interval={args.mz_interval}, batch_size={args.batch_size}",
                       "generator": f"falcon_amd.synth ({args.generator}; SURVEY 8d recipe, seed 42)",
                       "exchange": (args.exchange + " (CSR all-gatherv of neighbour lists + labels + rows, overlapped with "
                                    "the next step)") if exchanging else "none",
                       "partitions": ("serial" if args.serial else
                                      "concurrent: a host thread + HIP stream + context per charge partition (PartitionRunner)"
                                      if runner is not None else "software-pipelined (ClusterPipeline.run_many)"),
                       "parallelism": (f"precursor buckets of the one dataset dealt to {world} GPUs (LPT), no data-path "
                                       "collective, one all-gatherv") if world > 1 else "1 GPU",
                       "note": ("weak scaling keeps the spectra per GPU fixed; the dataset's precursor range does not grow, so "
                                "buckets get denser with N and the work per spectrum rises (flat -> IVF regime)") if world > 1 else None},
            "roofline": roof,
            "stage_ms": s["stage_ms"],
        }
        if h2h is not None:
            out["value_host_to_host"] = n_total / h2h
            out["ms_per_step_host_to_host"] = h2h * 1e3
        if extra:
            out["configs"] = extra
        if world == 1 and not args.no_cpu_baseline:
            host = {k: getattr(parts[0], k).cpu().numpy() for k in ("precursor_mz", "retention_time", "mz", "intensity", "indptr")}
            out["cpu_baseline"] = cpu_baseline(host, p)
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
