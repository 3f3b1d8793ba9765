#!/usr/bin/env python3
"""Headline benchmark: spectra clustered / second through the whole hot path
(vectorise -> buckets -> IVF/flat cosine scan -> top-k -> filter -> DBSCAN -> refine ->
medoids/labels) on synthetic peak lists, plus the cosine kernel's roofline fraction.

    python bench.py --gpus 1 --steps 100 --warmup 2
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W

The job is ONE dataset, all its charge partitions (falcon.py:151-193).  `--scaling weak` (default): N x `--spectra`
synthetic spectra (N = GPUs; 1,000,000 per GPU = BASELINE.json configs[1] at N = 1): N statistically identical blocks
of the configs[1] workload (same 400-1200 m/z precursor range, different random templates), block k's charges
relabelled (2, 3) -> (2 + 2k, 3 + 2k) -- a dataset of 2 N charge partitions whose work per spectrum is that of the
1 M workload by construction.  (Widening the precursor range with N instead does NOT keep the work per spectrum: the
20 ppm tolerance grows with m/z, the gap rule cuts less often, buckets and pair counts grow -- measured, profiles/NOTES.md.)
`--scaling strong`: the FIXED `--spectra-total` dataset (default 10,000,000 in 400-1200 m/z = BASELINE configs[2]) on
N GPUs.  A "step" = one pass of the whole hot path over that dataset:

  * every rank histograms the 1-m/z precursor windows of every partition (4 bytes per spectrum) and derives the SAME
    deal (`distributed.deal_job`): whole charge partitions by longest-processing-time where that balances (many like
    partitions), otherwise every (charge, window) unit on its own; it sorts, buckets and runs vectorise -> ... -> labels
    on its own spectra only (no data-path collective: a window's buckets depend on its own spectra only, and no
    neighbour pair crosses a bucket, reference cluster.py:107-141);
  * ONE all-gatherv (RCCL over xGMI) of the CSR neighbour lists + labels + dataset rows gives every rank the
    global sparse graph and the globally unique labels (rank-major offsets like falcon.py:189-193), which the
    rank copies to the host.  At N = 1 there is nothing to exchange and the step is the single-GPU pipeline.

Inputs are resident in HBM before the timed region (`value`); `value_host_to_host` is the throughput of the same steps
with the peak arrays starting in pinned host memory: the upload of step i + 1 travels on a copy stream under
the kernels of step i (`ms_per_step_host_to_host_latency` = one step alone, upload then compute).  Rank 0 prints ONE JSON
line; at N = 1 it also carries `configs`: the 10 M-spectra float32 run (BASELINE configs[2]'s dataset on one GPU = the
north star's target size; repeated as the top-level `north_star` object), the 10 M / low_dim 800 / float16 run
(configs[4]), the configs[3] bucket regime at 10 M, SURVEY 8d's C4 row as written (batch_size 65536: n_list 1,024), a
skewed 2 M workload (log-normal window occupancy, 5..50 peaks per spectrum) and configs[3] itself (50 M spectra, n_probe
32, in 4 bucket shares; its roofline from a staged pass over one share), each with its own `roofline`, and
`cpu_baseline` (the oracle on all host cores).
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


def _slice_by_precursor(host, lo, hi):
    """the spectra with precursor m/z in [lo, hi) of one charge partition (host arrays) -> generate_clusters' arguments"""
    pm = host["precursor_mz"]
    sel = np.flatnonzero((pm >= lo) & (pm < hi))
    counts = np.diff(host["indptr"])[sel]
    indptr = np.zeros(len(sel) + 1, np.int64)
    np.cumsum(counts, out=indptr[1:])
    src = np.repeat(host["indptr"][:-1][sel] - indptr[:-1], counts) + np.arange(int(counts.sum()))
    return (host["mz"][src], host["intensity"][src], indptr, pm[sel], host["retention_time"][sel]), len(sel)


def cpu_baseline(hosts, params, budget_s=75.0):
    """SURVEY 8d's CPU leg, same run, same box: (i) the oracle (numpy + oracle/kordered.c, kind "port") on ALL host cores --
    precursor buckets on a thread pool the way the reference runs its blocks (joblib threading backend, cluster.py:115-136);
    (ii) the library-grade baseline (numpy sgemm top-k + sklearn.DBSCAN, oracle/library_baseline.py).  Both at ~10 k / ~100 k
    / 1 M spectra of the SAME dataset: a size below the full one is the precursor-m/z slice [600, 600 + w) of every charge
    partition (w chosen for the size: same bucket density, so the same work per spectrum as the full run); sizes that do not
    fit the time budget are reported as "not run", never extrapolated.  `hosts`: the charge partitions as host arrays.
    -> (the contract's `cpu_baseline` object = the port at the largest size that ran, the per-size list)"""
    from oracle import falcon_oracle as fo
    from oracle import library_baseline as lb
    cores = os.cpu_count() or 1
    n_all = sum(len(h["precursor_mz"]) for h in hosts)
    kw = dict(eps=params.eps, low_dim=params.low_dim, n_probe=params.n_probe, n_neighbors=params.n_neighbors,
              n_neighbors_ann=params.n_neighbors_ann, mz_interval=params.mz_interval, kmeans_iters=params.kmeans_iters)
    lkw = dict(eps=params.eps, low_dim=params.low_dim, n_neighbors=params.n_neighbors, n_neighbors_ann=params.n_neighbors_ann,
               mz_interval=params.mz_interval)
    lo_all = min(float(h["precursor_mz"].min()) for h in hosts)
    hi_all = max(float(h["precursor_mz"].max()) for h in hosts) + 1.0
    sizes, spent, t_start = [], {"port": 0.0, "library": 0.0}, time.perf_counter()
    rate = {"port": None, "library": None}
    for target in (10_000, 100_000, 1_000_000):
        if target > n_all * 1.05:
            sizes.append({"spectra": target, "port": "not run (larger than the dataset)", "library": "not run (larger than the dataset)"})
            continue
        full = target >= n_all * 0.95
        width = (hi_all - lo_all) if full else (hi_all - lo_all) * target / n_all
        lo = lo_all if full else max(lo_all, min(600.0, hi_all - width))
        work = [_slice_by_precursor(h, lo, lo + width) for h in hosts]
        n_sel = sum(w[1] for w in work)
        entry = {"spectra": n_sel, "sample": ("the whole dataset" if full else
                                               f"precursor m/z in [{lo:.0f}, {lo + width:.1f}) of every charge partition (same bucket density)")}
        for kind in ("port", "library"):
            left = budget_s - (time.perf_counter() - t_start)
            est = n_sel / rate[kind] if rate[kind] else 0.0
            if est > left or left <= 0:
                entry[kind] = f"not run (estimated {est:.0f} s at the previous size's rate, {max(left, 0):.0f} s of the budget left)"
                continue
            t0 = time.perf_counter()
            for a, n_w in work:
                if n_w == 0:
                    continue
                if kind == "port":
                    fo.generate_clusters(*a, n_jobs=cores, **kw)
                else:
                    lb.cluster_partition(*a, **lkw)
            dt = time.perf_counter() - t0
            rate[kind] = n_sel / dt
            entry[kind] = {"value": n_sel / dt, "unit": "spectra/s", "seconds": round(dt, 2)}
        sizes.append(entry)
    ran = [e for e in sizes if isinstance(e.get("port"), dict)]
    if not ran:                                      # a dataset below the smallest size: nothing was timed
        return {"value": None, "unit": "spectra/s", "cores": cores, "kind": "port",
                "sample": f"not run: the dataset ({n_all} spectra) is smaller than the smallest baseline size", "sizes": sizes}
    best = ran[-1]
    main = {"value": best["port"]["value"], "unit": "spectra/s", "cores": cores, "kind": "port",
            "sample": f"{best['spectra']} spectra ({best['sample']}), both charge partitions, same parameters; "
                      f"oracle/falcon_oracle.py + oracle/kordered.c (fmaf, AVX2), buckets on a pool of {cores} threads, "
                      f"{best['port']['seconds']} s",
            "library_baseline": {"what": "numpy sgemm per bucket + argpartition top-k + sklearn.cluster.DBSCAN(metric='precomputed') "
                                         "(oracle/library_baseline.py): exhaustive inside a bucket, BLAS threads = all cores",
                                 "cores": cores},
            "sizes": sizes}
    return main


def pmc_traffic(workload):
    """HBM bytes per launch of a workload's dominant cosine kernel from the rocprofv3 PMC passes committed under profiles/
    (FETCH_SIZE x 2 + WRITE_SIZE, separate passes, MI355X_MICROARCH.md).  Counters cannot be read from inside the run, so
    the figure is only reported for the workloads it was measured on (profiles/r6_pmc_traffic.json -- r5 / r4 / r3 as the fallback --, written by
    tools/pmc_traffic.sh on the GPU box).  -> (bytes per launch or None, source file or None)"""
    for fn in ("r6_pmc_traffic.json", "r5_pmc_traffic.json", "r4_pmc_traffic.json", "r3_pmc_traffic.json"):
        path = os.path.join(ROOT, "profiles", fn)
        if os.path.isfile(path):
            try:
                with open(path) as f:
                    e = json.load(f).get(workload)
                if e:
                    return float(e["hbm_bytes_per_launch"]), f"profiles/{fn}: {e.get('kernel', '')}"
            except Exception:
                pass
    if workload == "headline":
        path = os.path.join(ROOT, "profiles", "r2_pmc_hbm_traffic_per_step.json")
        if os.path.isfile(path):
            with open(path) as f:
                return float(json.load(f)["scan"]["hbm_bytes_per_launch"]), "profiles/r2_pmc_hbm_traffic_per_step.json"
    return None, None


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=100)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--scaling", choices=["weak", "strong"], default="weak",
                    help="weak: gpus x --spectra spectra = gpus blocks of the 1 M workload as 2 x gpus charge partitions; "
                         "strong: the fixed --spectra-total dataset in 400-1200 m/z on every number of GPUs")
    ap.add_argument("--spectra", type=int, default=1_000_000, help="weak scaling: spectra per GPU")
    ap.add_argument("--spectra-total", type=int, default=10_000_000, help="strong scaling: spectra of the dataset")
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
    ap.add_argument("--with-strong", action="store_true",
                    help="N > 1, weak mode: also time the FIXED --spectra-total dataset on the same ranks in the same run "
                         "(`strong_scaling` in the JSON line: the window-by-window deal of one shared dataset).  Off by default: the "
                         "driver's scaling runs stay exactly the weak job")
    ap.add_argument("--exchange-only", action="store_true",
                    help="collective self-check: run ONLY the exchange step (SparseGraphExchange: sizes all-gather + one padded "
                         "payload all-gather over RCCL) with synthetic ragged payloads on the N ranks, verify every rank's received "
                         "bytes, print one JSON line and exit (0 = ok)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-cold", action="store_true",
                    help="skip the cold-pass measurement (tools/cold_pass.py in fresh child processes, before this process touches the GPU)")
    ap.add_argument("--no-configs", action="store_true", help="skip the 10 M / 50 M configurations of the `configs` array")
    ap.add_argument("--configs-spectra", type=int, default=10_000_000)
    ap.add_argument("--big-spectra", type=int, default=50_000_000,
                    help="BASELINE configs[3] at its own size (n_probe 32, n_neighbors_ann 128), run in --big-chunks bucket shares "
                         "(ClusterPipeline.run_chunked); 0 = skip")
    ap.add_argument("--big-chunks", type=int, default=4,
                    help="bucket shares of the 50 M configuration.  Round 6 (tools/big_job.py, alone in a process): 4 shares 0.86 s per "
                         "pass with 191 GB of the device in use; 2 shares ask for 309 GB -- more than the 288 GB of HBM (the driver "
                         "oversubscribes into host memory: 2.7 s per pass), so the default stays 4.  The pools are trimmed between the "
                         "configurations (fal_ctx_trim); a run with fewer shares that runs out of memory falls back to 4 and says so")
    ap.add_argument("--skew-spectra", type=int, default=2_000_000,
                    help="the skewed-workload entry of `configs` (synth skew=True: log-normal window occupancy, 5..50 peaks); 0 = skip")
    ap.add_argument("--partitions", choices=["auto", "concurrent", "pipelined"], default="auto",
                    help="how the two charge partitions of a step are scheduled: concurrent = a host thread + HIP stream + context "
                         "each (PartitionRunner; the reference clusters its blocks on a thread pool, cluster.py:115-136), pipelined = "
                         "one stream, the next partition's sort under the current scan (ClusterPipeline.run_many); auto = concurrent "
                         "(the 50 M configuration runs its bucket shares pipelined: ClusterPipeline.run_chunked)")
    ap.add_argument("--serial", action="store_true",
                    help="run the charge partitions strictly one after the other (default: software-pipelined, "
                         "ClusterPipeline.run_many)")
    args = ap.parse_args()

    # ---- what falcon.main() pays: ONE pass per charge in a fresh process.  Measured in fresh child processes BEFORE this process
    # touches the GPU (tools/cold_pass.py: library load, first pass incl. fal_ctx_plan, the following passes) -----------------
    cold = None
    if int(os.environ.get("WORLD_SIZE", "1")) == 1 and not args.no_cold and not args.exchange_only:
        import subprocess
        cold = {}
        for n_cold in (1_000_000, 10_000_000):
            try:
                r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "cold_pass.py"), str(n_cold), "--passes", "4"],
                                   capture_output=True, text=True, timeout=600)
                line = [l for l in r.stdout.splitlines() if l.startswith("{")]
                cold[str(n_cold)] = json.loads(line[-1]) if line else {"error": (r.stderr or r.stdout)[-300:]}
            except Exception as e:                                # pragma: no cover -- reported, never hidden
                cold[str(n_cold)] = {"error": repr(e)[:300]}

    import torch
    import torch.distributed as dist
    from falcon_amd import synth
    from falcon_amd import distributed as fdist
    from falcon_amd.cluster.cluster import AnnParams, ClusterPipeline, PartitionRunner, SpectrumDataset, n_list_rule
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

    def rccl_info():
        try:
            v = torch.cuda.nccl.version()
            v = ".".join(str(x) for x in v) if isinstance(v, (tuple, list)) else str(v)
        except Exception as e:                                    # pragma: no cover
            v = f"unavailable ({e!r})"
        return {"backend": backend if world > 1 else "none (1 rank)", "version": v, "ranks": world,
                "HSA_ENABLE_IPC_MODE_LEGACY": os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY")}

    if args.exchange_only:
        res = fdist.exchange_self_check(dev, n_neighbors=args.n_neighbors)
        res.update({"mode": "exchange-only", "n_gpus": world, "rccl": rccl_info()})
        if world > 1:
            dist.barrier()
            dist.destroy_process_group()
        if rank == 0:
            import ctypes
            ctypes.CDLL(None).fflush(None)
            print(json.dumps(res), flush=True)
        sys.exit(0 if res["ok"] else 1)

    ctx = Context(local_rank)
    pipe = ClusterPipeline(ctx)

    def params(**kw):
        base = dict(eps=args.eps, low_dim=args.low_dim, n_probe=args.n_probe, n_neighbors=args.n_neighbors,
                    n_neighbors_ann=args.n_neighbors_ann, mz_interval=args.mz_interval, scan=args.scan, dtype=args.dtype,
                    rescore=args.rescore, min_matches=6 if args.rescore else 0, prefilter=args.prefilter,
                    ivf_prefilter=not args.no_ivf_prefilter)
        base.update(kw)
        return AnnParams(**base)

    def make_parts(n_total, first_block=0, mz_lo=400.0, mz_hi=1200.0, replicas=1, skew=False):
        """the dataset, split by precursor charge (falcon.py:151-160), resident in HBM -> [SpectrumDataset].
        `replicas` = R > 1: R statistically identical blocks of n_total / R spectra each (generator blocks first_block,
        first_block + 1, ...), block k's charges relabelled (2, 3) -> (2 + 2k, 3 + 2k): a dataset of 2R charge partitions"""
        parts = []
        per = n_total // replicas
        for k in range(replicas):
            fb = first_block + k * ((per + synth.BLOCK - 1) // synth.BLOCK)
            if args.generator == "device":
                data = synth.generate_device(per, dev, seed=42, first_block=fb, mz_lo=mz_lo, mz_hi=mz_hi, skew=skew)
                sel = lambda c: synth.select_charge_device(data, c)
            else:
                data = synth.generate(per, seed=42, first_block=fb, mz_lo=mz_lo, mz_hi=mz_hi, skew=skew)
                sel = lambda c: synth.select_charge(data, c)
            for charge in (2, 3):
                c = sel(charge)
                parts.append(SpectrumDataset(ctx.to_dev(c["precursor_mz"], torch.float32),
                                             ctx.to_dev(c["retention_time"], torch.float32),
                                             ctx.to_dev(c["mz"], torch.float32), ctx.to_dev(c["intensity"], torch.float32),
                                             ctx.to_dev(c["indptr"], torch.int64)))
            del data
        return parts

    def barrier():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
            torch.cuda.synchronize()

    # ---- the dataset (every rank holds it; a rank only touches the peaks of its own buckets) ------------------------
    strong = args.scaling == "strong"
    n_total = args.spectra_total if strong else world * args.spectra
    mz_lo, mz_hi = 400.0, 1200.0
    replicas = 1 if strong else world
    parts = make_parts(n_total, mz_lo=mz_lo, mz_hi=mz_hi, replicas=replicas)
    job = {"n_total": n_total, "part_off": np.concatenate([[0], np.cumsum([len(x) for x in parts])])}     # what step() works on
    p = params()
    run_args = (20.0, "ppm", None, 0.05, args.batch_size, p)
    shard = (rank, world) if world > 1 else None
    exchanging = args.exchange != "none" and (world > 1 or args.force_exchange)
    exchange = fdist.SparseGraphExchange(dev)
    pending, csr_buf = [], {}
    # spectra per 1 m/z window and charge-2 partition: beyond ~1,600 the windows get an index (n_list > n_probe)
    ivf_regime = 0.7 * (n_total // replicas) / max(mz_hi - mz_lo, 1.0) > 1600

    def collect_stages(n):
        return ({k: ctx.stage_ms(k) for k in STAGES + ("kernel",)}
                | {"pairs": ctx.counter(0), "coarse_pairs": ctx.counter(1), "issued": ctx.counter(4),
                   "pairs_outside": ctx.counter(7), "issued_outside": ctx.counter(8),
                   "scan_launches": ctx.counter(2), "sims_bytes": ctx.counter(3), "n": n})

    concurrent = {"on": False}
    runner = None

    def step(parts, run_args, collect=None, chunks=1):
        """one pass of the hot path over the dataset; `collect` != None: serial, per-stage timing, no exchange"""
        if chunks > 1 and collect is None and not args.serial and concurrent["on"] and os.environ.get("FALCON_BENCH_CHUNKED") != "pipelined":
            outs = runner.run_chunked(parts, *run_args, n_chunks=chunks)          # (configs[3] at its own size, one GPU:
            lasts = []                                                            #  every share's partitions on concurrent slots)
        elif chunks > 1:
            outs = pipe.run_chunked(parts, *run_args, n_chunks=chunks)
            lasts = []
        elif collect is None and not args.serial and concurrent["on"]:
            outs = runner.run(parts, *run_args, shard=shard)                      # partitions on concurrent streams
            lasts = runner.lasts
        elif collect is None and not args.serial:
            outs = pipe.run_many(parts, *run_args, shard=shard)                   # partitions software-pipelined
            lasts = pipe.lasts
        else:
            outs, lasts = [], []
            owners = pipe.plan_shards(ctx, parts, run_args[4], run_args[5], world, tol=run_args[:2]) if shard is not None else None
            for j, ds in enumerate(parts):
                if shard is not None:
                    o = pipe.run_many([ds], *run_args, shard=(rank, world, [owners[j]]))
                    outs.append(o[0])
                    lasts.append(pipe.lasts[0])
                else:
                    outs.append(pipe.run(ds, *run_args))
                    lasts.append(dict(pipe.last))
                if collect is not None:
                    collect.append(collect_stages(int(outs[-1][0].numel())))
        if not exchanging or collect is not None or chunks > 1:
            labels_all, current = [], 0
            for labels, medoids in outs:
                labels_all.append(labels + current)                  # falcon.py:189-193
                current += int(medoids.numel())
            return torch.cat(labels_all).cpu()
        # ---- the one exchange step: all-gatherv of the sparse neighbour lists (CSR, ids -> dataset
        # rows of the job) + labels + dataset rows; asynchronous: it travels while the next step computes
        handle, _, _ = fdist.start_graph_exchange(ctx, exchange, outs, lasts, job["part_off"], args.n_neighbors, shard is not None,
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
            return fdist.SparseGraphExchange.assemble_labels(g, job["n_total"]).cpu()
        return g["labels"][rank].cpu()

    def timed(parts, run_args, steps, warmup, prime=3, chunks=1):
        # setup, like the data generation: the first passes grow the library's scratch pool and torch's caching
        # allocator to their steady-state sizes (GB-sized hipMallocs, tens of ms each) -- prime them before the
        # contract's W warmup steps so that neither W nor the timed K steps contain one-off allocations
        for _ in range(prime):
            step(parts, run_args, chunks=chunks)
        finish_pending()
        for _ in range(warmup):
            step(parts, run_args, chunks=chunks)
        finish_pending()
        barrier()
        t0 = time.perf_counter()
        for _ in range(steps):
            step(parts, run_args, chunks=chunks)
        finish_pending()                                         # the last exchange lands inside the timed region
        barrier()
        dt = time.perf_counter() - t0
        if world > 1:
            t = torch.tensor([dt], dtype=torch.float64, device=dev)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            dt = float(t.item())
        return dt

    def staged(parts, run_args):
        """per-kernel timing (HIP events on the kernels' own stream), outside any timed region.  The serial pass runs on the
        pipeline's own context (the timed steps ran on the partition runner's): one untimed pass first, so that this context's
        scratch has its steady-state size too -- a GB-sized hipMalloc inside a stage shows up as tens of ms of that stage"""
        step(parts, run_args, [])
        ctx.enable_timing(True)
        stages = []
        step(parts, run_args, stages)
        ctx.enable_timing(False)
        return stages

    def staged_share(parts, run_args, chunks):
        """per-kernel timing of ONE bucket share of a job that runs in `chunks` shares (configs[3] at 50 M spectra): share 0 of the
        same deal `run_chunked` executes, partition by partition, serial, HIP events around every stage"""
        owners = pipe.plan_shards(ctx, parts, run_args[4], run_args[5], chunks, tol=run_args[:2])

        def one(collect):
            for j, ds in enumerate(parts):
                o = pipe.run_many([ds], *run_args, shard=(0, chunks, [owners[j]]))
                if collect is not None:
                    collect.append(collect_stages(int(o[0][0].numel())))
        one(None)
        ctx.enable_timing(True)
        stages = []
        one(stages)
        ctx.enable_timing(False)
        return stages

    # two charge partitions on concurrent streams win in both regimes once the pools are warm (tools/concurrent_parts.py: 1 M
    # 6.0 vs 6.7 ms, 10 M 137 vs 145 ms; a rank's share of the 10 M job at 8 GPUs 19.9 vs 23.9 ms, tools/shard_share.py)
    want_concurrent = args.partitions == "concurrent" or (args.partitions == "auto" and not args.serial)
    if want_concurrent:
        runner = PartitionRunner(local_rank, 2)
        concurrent["on"] = True
    dt = timed(parts, run_args, args.steps, args.warmup)
    stages = staged(parts, run_args)

    # ---- N > 1, weak mode: the FIXED dataset (BASELINE configs[2]) on the same ranks as well, so that the line also carries the
    # window-by-window deal of one shared dataset (the weak workload deals whole partitions) -------------------------
    strong_extra = None
    if world > 1 and not strong and args.with_strong:
        try:
            del parts
            torch.cuda.empty_cache()
            sparts = make_parts(args.spectra_total, mz_lo=mz_lo, mz_hi=mz_hi, replicas=1)
            job.update({"n_total": args.spectra_total, "part_off": np.concatenate([[0], np.cumsum([len(x) for x in sparts])])})
            steps_s = 3
            dts = timed(sparts, run_args, steps_s, 1, prime=2)
            strong_extra = {"workload": f"one dataset of {args.spectra_total} synthetic spectra (BASELINE configs[2]) dealt window by "
                                        f"window to {world} GPUs, one all-gatherv per step", "scaling": "strong", "steps": steps_s,
                            "ms_per_step": dts / steps_s * 1e3, "value": args.spectra_total * steps_s / dts, "unit": "spectra/s",
                            "one_gpu_reference": "configs[0] of the N = 1 line (the same dataset on one GPU)"}
            del sparts
        except Exception as e:                                    # pragma: no cover -- reported, never hidden
            strong_extra = {"error": repr(e)[:300]}
        torch.cuda.empty_cache()
        parts = make_parts(n_total, mz_lo=mz_lo, mz_hi=mz_hi, replicas=replicas)
        job.update({"n_total": n_total, "part_off": np.concatenate([[0], np.cumsum([len(x) for x in parts])])})

    # ---- host-to-host: the peak arrays start in pinned host memory, labels end on the host -------
    h2h = h2h_latency = None
    h2h_multi = None
    if world > 1 and concurrent["on"]:
        # N ranks: every rank holds the job's partitions in pinned HOST memory; per step it uploads the precursor columns (the
        # deal needs them: 4 bytes per spectrum) and the peaks of ITS windows only (PartitionRunner.run -> take_rows)
        try:
            host_parts = [SpectrumDataset(*[t.cpu().pin_memory() for t in (x.precursor_mz, x.retention_time, x.mz, x.intensity, x.indptr)])
                          for x in parts]
            ds_bytes = sum(sum(t.numel() * t.element_size() for t in hp.columns()) for hp in host_parts)
            ctxs = lambda: [pl.ctx for pl in runner.pipelines] + [getattr(pl, "_front_ctx", None) for pl in runner.pipelines] + \
                           ([runner._planner.ctx] if hasattr(runner, "_planner") else [])
            for _ in range(2):
                step(host_parts, run_args)
            finish_pending()
            barrier()
            b0 = sum(c.h2d_bytes for c in ctxs() if c is not None)
            k2 = 3
            t0 = time.perf_counter()
            for _ in range(k2):
                step(host_parts, run_args)
            finish_pending()
            barrier()
            dth = time.perf_counter() - t0
            b1 = sum(c.h2d_bytes for c in ctxs() if c is not None)
            if world > 1:
                tt = torch.tensor([dth], dtype=torch.float64, device=dev)
                dist.all_reduce(tt, op=dist.ReduceOp.MAX)
                dth = float(tt.item())
            h2h_multi = {"value_host_to_host": n_total * k2 / dth, "ms_per_step_host_to_host": dth / k2 * 1e3, "steps": k2,
                         "uploaded_bytes_per_step_rank0": (b1 - b0) / k2, "dataset_bytes": ds_bytes,
                         "uploaded_fraction_rank0": (b1 - b0) / k2 / max(ds_bytes, 1),
                         "what": "partitions in pinned host memory on every rank; a rank uploads the precursor columns and the peaks "
                                 "of its own windows (host-side CSR gather), then runs the path and the exchange"}
            del host_parts
        except Exception as e:                                    # pragma: no cover -- reported, never hidden
            h2h_multi = {"error": repr(e)[:300]}
    if world == 1:
        pinned = [[t.cpu().pin_memory() for t in (x.precursor_mz, x.retention_time, x.mz, x.intensity, x.indptr)] for x in parts]
        copy_stream = torch.cuda.Stream(device=dev)

        def upload(on=None):
            """-> the dataset on the device (+ the event that says the bytes have arrived)"""
            if on is None:
                return [SpectrumDataset(*[t.to(dev, non_blocking=True) for t in ts]) for ts in pinned], None
            with torch.cuda.stream(on):
                up = [SpectrumDataset(*[t.to(dev, non_blocking=True) for t in ts]) for ts in pinned]
                ev = torch.cuda.Event()
                ev.record(on)
            return up, ev

        k2 = 30 if args.steps >= 20 else max(1, args.steps)    # (its own step count: a 20-step average of a 12 ms latency is noisy)
        # one step alone (latency).  Two streams: the runner takes the pinned host columns themselves -- it uploads the
        # partitions one after the other on a copy stream and starts a partition's kernels when its own bytes have arrived
        # (PartitionRunner.run); one stream: upload, then compute.
        host_parts = [SpectrumDataset(*ts) for ts in pinned]
        via_runner = concurrent["on"] and shard is None and not os.environ.get("FALCON_BENCH_H2H_UPFRONT")      # (A/B switch)
        one = (lambda: step(host_parts, run_args)) if via_runner else (lambda: step(upload()[0], run_args))
        for _ in range(5):                                     # (the copy stream's allocator pool and the runner's slots reach
            one()                                              #  their steady state within three steps: tools/h2h_timeline.py)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(k2):
            one()
        torch.cuda.synchronize()
        h2h_latency = (time.perf_counter() - t0) / k2
        # a stream of datasets: the upload of step i + 1 (copy stream, PCIe) under the kernels of step i, into two sets of
        # device buffers used in turn (no allocation inside the loop)
        slots = [[[torch.empty_like(t, device=dev) for t in ts] for ts in pinned] for _ in range(2)]
        free_ev = [None, None]                                # compute of the step that last read slot b has been enqueued ...

        def upload_into(b):
            with torch.cuda.stream(copy_stream):
                if free_ev[b] is not None:
                    copy_stream.wait_event(free_ev[b])        # ... and finished, before the slot is overwritten
                for dst, src in zip(slots[b], pinned):
                    for d_t, s_t in zip(dst, src):
                        d_t.copy_(s_t, non_blocking=True)
                ev = torch.cuda.Event()
                ev.record(copy_stream)
            return [SpectrumDataset(*ts) for ts in slots[b]], ev

        nxt = upload_into(0)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for i in range(k2):
            cur, ev = nxt
            torch.cuda.current_stream(dev).wait_event(ev)
            if i + 1 < k2:
                nxt = upload_into((i + 1) & 1)
            step(cur, run_args)                               # (ends with the labels on the host: every kernel of it is done)
            free_ev[i & 1] = torch.cuda.Event()
            free_ev[i & 1].record(torch.cuda.current_stream(dev))
        torch.cuda.synchronize()
        h2h = (time.perf_counter() - t0) / k2
        del slots
        del pinned, nxt, cur

    def fo_steps(low_dim):
        """MFMA steps of the float16 list scan = padded row width / 16"""
        from falcon_amd.device import row_width
        return row_width(low_dim) // 16

    def summarize(stages, d, p, elem):
        pairs = sum(s["pairs"] for s in stages)
        ms = {k: round(sum(s[k][0] for s in stages), 3) for k in STAGES + ("kernel",)}
        topk_ms = ms["scan"] + ms["select"] + ms["filter"]
        n_rows = sum(s["n"] for s in stages)
        launches = sum(s["kernel"][1] for s in stages)
        flops = 2.0 * d * pairs
        # the part of the work that stage 8's launches (the dominant kernel) do: all but the one-block buckets
        kflops = 2.0 * d * (pairs - sum(s.get("pairs_outside", 0) for s in stages))
        kshare = kflops / flops if flops > 0 else 1.0
        algo_bytes = n_rows * (2 * d * elem + 8 * p.n_neighbors_ann)          # SURVEY 8(d) compulsory bytes
        tf = lambda f, t: f / (t * 1e-3) / 1e12 if t > 0 else 0.0
        gbs = lambda b, t: b / (t * 1e-3) / 1e9 if t > 0 else 0.0
        return dict(pairs=pairs, stage_ms={k: v for k, v in ms.items() if k != "kernel"}, kernel_ms=ms["kernel"],
                    scan_ms=ms["scan"], topk_ms=topk_ms, n_rows=n_rows, launches=launches, flops=flops, algo_bytes=algo_bytes,
                    kernel_flops=kflops, kernel_share=kshare,
                    kernel_tflops=tf(kflops, ms["kernel"]), cosine_tflops=tf(flops, topk_ms),
                    issued_tflops=tf(2.0 * d * sum(s["issued"] - s.get("issued_outside", 0) for s in stages), ms["kernel"]),
                    hbm_gbs_kernel=gbs(algo_bytes * kshare, ms["kernel"]), hbm_gbs_cosine=gbs(algo_bytes, topk_ms))

    def roofline_of(s, kernel, mfma_peak, mfma_name, workload_key, issued_factor=1.0, note=None):
        """The dominant cosine kernel against the roof that bounds it.  `achieved` = ALGORITHMIC work per launch (2 d flop per
        (query, candidate) pair the search has to look at; SURVEY 8d's compulsory bytes per spectrum for the HBM view) / the
        kernel's average launch duration (HIP events on its own stream, `fal_ctx_stage_ms("kernel")`).  Both views are
        computed; `bound` is the roof the kernel sits closer to."""
        launches = max(s["launches"], 1)
        mfma_frac = s["kernel_tflops"] * issued_factor / mfma_peak
        hbm_frac = s["hbm_gbs_kernel"] / PEAK_HBM_GBS
        traffic, src = pmc_traffic(workload_key)
        if hbm_frac >= mfma_frac:
            r = {"bound": "hbm", "achieved": s["hbm_gbs_kernel"], "peak": PEAK_HBM_GBS, "unit": "GB/s", "frac": hbm_frac}
        else:
            r = {"bound": "mfma", "achieved": s["kernel_tflops"] * issued_factor, "peak": mfma_peak, "unit": "TFLOP/s",
                 "frac": mfma_frac}
        r.update({"kernel": kernel, "roof": mfma_name if r["bound"] == "mfma" else "HBM3E 8 TB/s",
                  "traffic": traffic, "traffic_unit": f"HBM bytes per launch (PMC, {src})" if src else None,
                  "launches": s["launches"], "avg_launch_ms": s["kernel_ms"] / launches,
                  "pairs_per_launch": s["pairs"] * s["kernel_share"] / launches, "flops_per_launch": s["kernel_flops"] / launches,
                  "algorithmic_bytes_per_launch": s["algo_bytes"] * s["kernel_share"] / launches,
                  "frac_of_mfma_peak": mfma_frac, "frac_of_hbm_roof": hbm_frac,
                  # SURVEY 8d defines the cosine kernel as list scan + top-k: the same algorithmic work over every launch of
                  # the scan / select / filter stages (exact pair chains and the k-th key resolution included)
                  "scan_plus_topk": {"ms": s["topk_ms"], "tflops": s["cosine_tflops"],
                                     "frac_of_mfma_peak": s["cosine_tflops"] * issued_factor / mfma_peak,
                                     "algorithmic_bytes": s["algo_bytes"], "gbs": s["hbm_gbs_cosine"],
                                     "frac_of_hbm_roof": s["hbm_gbs_cosine"] / PEAK_HBM_GBS}})
        if note:
            r["note"] = note
        return r

    # ---- the larger configurations on ONE GPU (north star target size; BASELINE configs[2] dataset / [3] / [4]) --------
    extra = []
    if world == 1 and not args.no_configs and rank == 0:
        del parts
        torch.cuda.empty_cache()
        was_concurrent = concurrent["on"]
        plan = [("f32", args.configs_spectra, dict(low_dim=400, dtype="f32", scan="f32"), 1200.0, 1, args.batch_size),
                # `--low_dim` is a free integer (README.md:114-117): 200 runs on rows padded to 256 columns -- next to 256 itself
                ("f32-d200", args.configs_spectra, dict(low_dim=200, dtype="f32", scan="f32"), 1200.0, 1, args.batch_size),
                ("f32-d256", args.configs_spectra, dict(low_dim=256, dtype="f32", scan="f32"), 1200.0, 1, args.batch_size),
                ("f16", args.configs_spectra, dict(low_dim=800, dtype="f16", scan="f32"), 1200.0, 1, args.batch_size),
                # BASELINE configs[3]'s bucket regime on one GPU: the same number of spectra in a quarter of the precursor
                # range (buckets of 20-35 k rows: n_list 512) with that config's n_probe = 32
                ("f32-dense", args.configs_spectra, dict(low_dim=400, dtype="f32", scan="f32", n_probe=32), 600.0, 1, args.batch_size),
                # SURVEY 8d's C4 row as written: 1 m/z windows of ~44 k charge-2 spectra kept whole by `--batch_size 65536`
                # (config.py:119-124) -> n_list 1,024, n_probe 32: more than 512 lists per bucket (VERDICT r4 missing #4)
                ("f32-b64k", args.configs_spectra, dict(low_dim=400, dtype="f32", scan="f32", n_probe=32), 560.0, 1, 2 ** 16),
                # beyond the float16 gate: 1 m/z windows of ~175 k charge-2 spectra kept whole by
                # `--batch_size 262144` -> n_list 4,096: more than the 2,048 lists the float16 assignment / key quantiser serve,
                # those buckets take the exact float32 kernels (the charge-3 windows, 75 k rows / 1,024 lists, stay on the float16
                # paths): the cost of the cliff as a stated number
                ("f32-b256k", args.configs_spectra, dict(low_dim=400, dtype="f32", scan="f32"), 440.0, 1, 2 ** 18),
                # a workload that is NOT uniform (synth skew=True: log-normal occupancy of the 1 m/z windows, 5..50 peaks per
                # spectrum): 300-row flat buckets next to 2^15-row indexed ones in one job
                ("f32-skew", args.skew_spectra, dict(low_dim=400, dtype="f32", scan="f32"), 1200.0, 1, args.batch_size)]
        plan = [e for e in plan if e[1] > 0]
        if args.big_spectra > 0:
            # ... and configs[3] itself: 50 M spectra; the working set of one pass (~260 GB) does not fit beside the dataset,
            # so the precursor buckets run in 4 shares one after the other (ClusterPipeline.run_chunked; --big-chunks)
            plan.append(("f32-50M", args.big_spectra, dict(low_dim=400, dtype="f32", scan="f32", n_probe=32, n_neighbors_ann=128),
                         1200.0, args.big_chunks, args.batch_size))
        big, big_key = None, None
        for name, n_cfg, kw, hi, chunks, batch in plan:
            skew = name == "f32-skew"
            chunks_note = None
            if big_key != (n_cfg, hi, skew):
                del big
                torch.cuda.empty_cache()
                big = make_parts(n_cfg, mz_hi=hi, skew=skew)
                big_key = (n_cfg, hi, skew)
            pc = params(**kw)
            ra = (20.0, "ppm", None, 0.05, batch, pc)
            # the scratch of the previous configuration goes back to the driver (fal_ctx_trim): every entry starts from the
            # pools a fresh process would have, and the 50 M job has the device to itself
            if runner is not None:
                runner.trim()
            pipe.trim()
            # (five timed steps after three priming passes: with three a single late scratch growth -- a GB-sized hipMalloc of
            #  one of the two partition contexts -- showed up as + 40 ms on the mean of a 120 ms step)
            steps_c = 5 if chunks == 1 else 3
            slow = name == "f32-b256k"                            # (seconds per pass: two timed steps behind one priming pass)
            if slow:
                steps_c = 2
            try:
                try:
                    dtc = timed(big, ra, steps_c, 0 if slow else 1, prime=1 if slow else (3 if chunks == 1 else 2), chunks=chunks)
                except Exception as e:
                    if chunks < 2 or chunks >= 4 or "memory" not in repr(e).lower():
                        raise
                    chunks_note = f"{chunks} shares ran out of device memory ({repr(e)[:120]}); ran in 4"
                    del e
                    import gc
                    gc.collect()
                    runner.trim()
                    pipe.trim()
                    chunks = 4
                    dtc = timed(big, ra, steps_c, 1, prime=2, chunks=chunks)
                if chunks == 1:
                    sc = summarize(staged(big, ra), pc.low_dim, pc, 2 if name == "f16" else 4)
                else:                                             # one bucket share of the chunked job, its own staged pass
                    sc = summarize(staged_share(big, ra, chunks), pc.low_dim, pc, 4)
            except Exception as e:                                # pragma: no cover -- reported, never hidden
                extra.append({"workload": f"{n_cfg} spectra {name}", "error": repr(e)[:300]})
                continue
            n_big = sum(len(x) for x in big)
            entry = {
                "workload": f"{n_cfg} synthetic spectra on 1 GPU (charges 2+3, precursor m/z 400-{hi:.0f}), low_dim={pc.low_dim} "
                            f"{name.split('-')[0]}, n_neighbors={pc.n_neighbors}, n_neighbors_ann={pc.n_neighbors_ann}, "
                            f"n_probe={pc.n_probe}, eps={pc.eps}, precursor_tol=20ppm, mz_interval={pc.mz_interval}, batch_size={batch}"
                            + (f", precursor buckets in {chunks} shares run one after the other" if chunks > 1 else ""),
                "baseline_config": {"f32": "configs[2] dataset on one GPU", "f16": "configs[4]",
                                    "f32-d200": "configs[2] dataset on one GPU at --low_dim 200 (rows padded to 256 columns)",
                                    "f32-d256": "configs[2] dataset on one GPU at --low_dim 256",
                                    "f32-dense": "configs[3] regime (n_list 512, n_probe 32, n_neighbors_ann 128) at one GPU's size: "
                                                 "precursors in 400-600 m/z",
                                    "f32-b64k": "SURVEY 8d's C4 row as written (43,750-row buckets, n_list 1,024, n_probe 32) at one "
                                                "GPU's size: precursors in 400-560 m/z, batch_size 65536",
                                    "f32-b256k": "none: the cliff beyond 2,048 lists per bucket -- precursors in 400-440 m/z, batch_size 262144: "
                                                 "175 k-row buckets, n_list 4,096 (exact float32 k-means / coarse / fine scan)",
                                    "f32-skew": "none (VERDICT r4 next #7): configs[1]'s parameters on a skewed dataset -- log-normal window "
                                                "occupancy (fullest window ~70x the median), 5..50 peaks per spectrum",
                                    "f32-50M": "configs[3] (50 M spectra, n_probe 32, n_neighbors_ann 128) at its own size"}[name],
                "steps": steps_c, "ms_per_step": dtc / steps_c * 1e3, "value": n_big * steps_c / dtc, "unit": "spectra/s",
                "dtype": name.split("-")[0]}
            if chunks_note:
                entry["shares_note"] = chunks_note
            if chunks > 1:
                entry["staged_share"] = {"what": f"share 0 of {chunks} (the deal run_chunked executes), serial staged pass",
                                         "spectra": sc["n_rows"], "stage_ms": sc["stage_ms"], "pairs": sc["pairs"]}
                entry["roofline"] = roofline_of(
                    sc, "list16s_kernel<25> (f16 MFMA 32x32x16, list-major), one bucket share of the 50 M job", PEAK_MFMA_F16_TFLOPS,
                    "f16 MFMA 2.5 PFLOP/s dense", None,
                    note="work and launches of ONE of the job's bucket shares (its own serial staged pass)")
            elif sc is not None:
                entry.update({"stage_ms": sc["stage_ms"], "pairs_per_step": sc["pairs"]})
                if name == "f16" and not pc.f16_index:
                    entry["roofline"] = roofline_of(sc, "scan16_kernel<50,1> (f16 MFMA 32x32x16, LDS-staged candidates, exhaustive)",
                                                    PEAK_MFMA_F16_TFLOPS, "f16 MFMA 2.5 PFLOP/s dense", None)
                elif name == "f16":
                    entry["roofline"] = roofline_of(
                        sc, "list16_kernel<50> (f16 MFMA 32x32x16, list-major; float16 vectors: n_probe lists per query through the "
                            "k-means index, exact float32 chains over the vectors' images)", PEAK_MFMA_F16_TFLOPS,
                        "f16 MFMA 2.5 PFLOP/s dense", "10M_f16")
                elif name == "f32-b256k":
                    entry["roofline"] = roofline_of(
                        sc, "ivf_list4_kernel<50> (fp32 MFMA 32x32x2, list-major: the exact fine scan of the 4,096-list buckets) + "
                            "list16s_kernel<25> (f16 MFMA: the 1,024-list buckets of the other charge)", PEAK_MFMA_F16_TFLOPS,
                        "f16 MFMA 2.5 PFLOP/s dense", None,
                        note="stage 8 sums the launches of BOTH cosine kernels (fp32 and f16 matrix work mixed: the matrix fraction is "
                             "taken against the f16 peak and is a lower bound; the HBM view is the one to read); the job's time is the "
                             "exact float32 k-means assignment of the 4,096-list buckets (stage_ms.build)")
                elif args.no_ivf_prefilter:
                    entry["roofline"] = roofline_of(sc, "ivf_list4_kernel<50> (fp32 MFMA 32x32x2, list-major)",
                                                    PEAK_MFMA_F32_TFLOPS, "fp32 MFMA 157.3 TFLOP/s", None)
                else:
                    entry["roofline"] = roofline_of(
                        sc, f"list16s_kernel<{fo_steps(pc.low_dim)}> (f16 MFMA 32x32x16, list-major: <= 128 rows of a list resident, the queries "
                            "probing it arrive as 256-byte sparse records and are expanded in LDS)", PEAK_MFMA_F16_TFLOPS,
                        "f16 MFMA 2.5 PFLOP/s dense",
                        {"f32": "10M_f32", "f32-dense": "10M_f32_dense", "f32-b64k": "10M_f32_b64k"}.get(name),
                        note="the kernel scans every probed (query, candidate) pair on the f16 matrix cores to 16-bit keys; the exact "
                             "float32 work (pair chains, k-th key resolution: stages scan / select) is in scan_plus_topk")
                entry["prefilter_fallback_rows"] = int(ctx.counter(5))
            if skew:
                # balance of the job as the multi-GPU deal would cut it (window histogram -> cost model -> deal_job), 8 ranks
                counts = ctx.window_counts([ds.precursor_mz for ds in big], pc.mz_interval)
                costs = fdist.window_costs(counts, batch, pc.n_probe, pc.mz_interval, ra[:2], pc.n_neighbors_ann, pc.n_neighbors)
                owners = fdist.deal_job(list(costs), 8)
                loads = sum(np.bincount(o, weights=c, minlength=8) for o, c in zip(owners, costs))
                occ = counts[0][counts[0] > 0]
                entry["skew"] = {"window_occupancy_charge2": {"max": int(occ.max()), "median": float(np.median(occ)), "windows": int(len(occ))},
                                 "n_list_of_the_fullest_bucket": int(n_list_rule(np.array([min(int(occ.max()), batch)]), pc.n_probe)[0]),
                                 "deal_8_ranks_worst_over_mean_modelled": float(loads.max() / loads.mean()),
                                 "scan_plus_topk_ms_per_Gpair": sc["topk_ms"] / max(sc["pairs"], 1) * 1e9}
            extra.append(entry)
        del big
        if runner is not None:                                    # (the 50 M job's pools go back before the 1 M dataset returns)
            runner.trim()
        pipe.trim()
        torch.cuda.empty_cache()
        concurrent["on"] = was_concurrent
        parts = make_parts(n_total, mz_lo=mz_lo, mz_hi=mz_hi, replicas=replicas)

    if rank == 0:
        d = args.low_dim
        elem = 2 if args.dtype == "f16" else 4
        s = summarize(stages, d, p, elem)
        if args.dtype == "f16" or args.scan == "f16x3":
            # issued matrix work: 3 f16 MFMAs per k-step for the split, 1 for plain float16 rows
            roof = roofline_of(s, "scan16_kernel (f16 MFMA 32x32x16, LDS-staged) + dense_kernel for buckets < 64",
                               PEAK_MFMA_F16_TFLOPS, "f16 MFMA 2.5 PFLOP/s dense", None,
                               issued_factor=3.0 if (args.scan == "f16x3" and args.dtype == "f32") else 1.0)
        elif ivf_regime and not args.no_ivf_prefilter:
            roof = roofline_of(s, "list16s_kernel<25> (f16 MFMA 32x32x16, list-major)", PEAK_MFMA_F16_TFLOPS,
                               "f16 MFMA 2.5 PFLOP/s dense", "10M_f32" if n_total == 10_000_000 and world == 1 else None)
        else:
            default = (n_total == 1_000_000 and world == 1 and d == 400 and args.n_neighbors_ann == 128 and args.mz_interval == 1.0
                       and args.batch_size == 2 ** 15 and not args.prefilter)
            roof = roofline_of(s, "dense4_kernel<50> (flat buckets of more than 32 rows, symmetric, four tiles per workgroup share one "
                                  "candidate stream through LDS): cosine scan, fp32 MFMA 32x32x2",
                               PEAK_MFMA_F32_TFLOPS, "fp32 MFMA 157.3 TFLOP/s", "headline" if default else None)
            # the kernel computes each bucket's similarity matrix on/above the diagonal only (bit-identical by symmetry):
            # `issued` = machine flops actually run through the matrix pipe, tile padding included
            roof.update({"issued_tflops": s["issued_tflops"], "issued_frac": s["issued_tflops"] / PEAK_MFMA_F32_TFLOPS})
        roof["profile"] = ("avg_launch_ms comes from a serial per-stage pass (HIP events); it agrees with the kernel's average in "
                           "profiles/r6_bench_1M_pipelined_kernel_stats.csv (rocprofv3 --kernel-trace --stats of `python3 bench.py "
                           "--partitions pipelined --no-configs --no-cpu-baseline --no-cold`: one stream); in the default two-stream run "
                           "(round 5's trace: profiles/r5_bench_1M_kernel_stats.csv) the two partitions' kernels overlap and stretch each other")
        out = {
            "metric": "spectra clustered/sec @1/2/4/8 GPU; cosine-kernel HBM GB/s vs roofline",
            "value": n_total * args.steps / dt,
            "unit": "spectra/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": dt / args.steps * 1e3,
            "higher_is_better": True,
            "scaling": args.scaling,
            "vs_baseline": None,
            "dtype": ("f16" if args.dtype == "f16" else "f16x3 (f32 vectors as hi/lo float16, f32 accumulate)"
                      if args.scan == "f16x3" else "f32"),
            "data": "synthetic",
            "config": {"workload": f"one dataset of {n_total} synthetic spectra "
                                   + (f"(fixed; {n_total // world} per GPU" if strong else f"({args.spectra} per GPU")
                                   + f"; charges 2+3; precursor m/z {mz_lo:.0f}-{mz_hi:.0f}), low_dim={d}, "
                                   f"n_neighbors={args.n_neighbors}, n_neighbors_ann={args.n_neighbors_ann}, "
                                   f"n_probe={args.n_probe}, eps={args.eps}, precursor_tol=20ppm, "
                                   f"mz_interval={args.mz_interval}, batch_size={args.batch_size}",
                       "generator": ("device (falcon_amd.synth.generate_device: torch RNG on the GPU, seed 42; statistically SURVEY 8d's "
                                     "recipe -- same distributions and parameters, not its PCG64 stream bit for bit)"
                                     if args.generator == "device" else
                                     "numpy (falcon_amd.synth.generate: SURVEY 8d's recipe, numpy default_rng([42, block]), PCG64)"),
                       "exchange": (args.exchange + " (CSR all-gatherv of neighbour lists + labels + rows, overlapped with "
                                    "the next step)") if exchanging else "none",
                       "partitions": ("serial" if args.serial else
                                      "concurrent: a host thread + HIP stream + context per charge partition (PartitionRunner)"
                                      if concurrent["on"] else "software-pipelined (ClusterPipeline.run_many)"),
                       "parallelism": (f"(charge, precursor window) units of the one dataset dealt to {world} GPUs "
                                       "(distributed.deal_job: whole charge partitions where that balances, else window by window), "
                                       "no data-path collective, one all-gatherv")
                                      if world > 1 else "1 GPU",
                       "note": (("strong scaling: the dataset is fixed, every GPU takes 1/N of its (charge, window) units" if strong
                                 else f"weak scaling: the dataset is {world} statistically identical blocks of the 1 M BASELINE "
                                      "configs[1] workload (same precursor range), block k's charges relabelled (2, 3) -> (2 + 2k, "
                                      f"3 + 2k): {2 * world} charge partitions, per-GPU work fixed by construction")
                                if world > 1 else None)},
            "roofline": roof,
            "stage_ms": s["stage_ms"],
            "rccl": rccl_info(),
        }
        if h2h is not None:
            # SURVEY 8d's contractual timing: peak arrays in pinned host memory -> labels on the host.  `value_host_to_host` is ONE
            # step alone (with two partition streams the runner uploads partition by partition and starts a partition's kernels
            # behind its own bytes); the stream form (upload of step i + 1 on a copy stream under the kernels of step i) has its
            # own key since round 4 (round 3 reported it under the old one)
            out["value_host_to_host"] = n_total / h2h_latency
            out["ms_per_step_host_to_host"] = h2h_latency * 1e3
            out["value_host_to_host_pipelined"] = n_total / h2h
            out["ms_per_step_host_to_host_pipelined"] = h2h * 1e3
            out["value_note"] = ("`value` = inputs resident in HBM when the timed region starts (the bench contract); "
                                 "`value_host_to_host` = SURVEY 8d's host-to-host figure, one step alone (409 MB over PCIe, a charge "
                                 "partition's kernels start when its own bytes have arrived: PartitionRunner.run on host columns; one "
                                 "stream: upload, then compute); `value_host_to_host_pipelined` = a stream of datasets, the next upload under the current "
                                 "step's kernels")
        if cold is not None:
            # `cold_ms`: the first pass of a fresh process (what falcon.main() runs per charge: reference falcon.py:153-193) next to
            # the steady-state `ms_per_step`; the child process's own later passes are its steady state (no priming beyond them)
            out["cold_pass"] = {"what": "tools/cold_pass.py in a fresh child process per size: pass_ms[0] = the first pass of a fresh "
                                        "process (kernels' code objects, scratch, torch's allocator all cold; it calls fal_ctx_plan "
                                        "itself), pass_ms[1:] = the passes behind it; lib_load_ms = dlopen + symbol binding",
                                **cold}
            c1 = cold.get(str(n_total)) or {}
            if "pass_ms" in c1:
                out["cold_ms"] = c1["pass_ms"][0]
        if h2h_multi is not None:
            out["host_to_host"] = h2h_multi
            if "value_host_to_host" in h2h_multi:
                out["value_host_to_host"] = h2h_multi["value_host_to_host"]
        if strong_extra is not None:
            out["strong_scaling"] = strong_extra
        if extra:
            out["configs"] = extra
            # The path BASELINE.json's north_star names -- 10 M spectra on one GPU through "IVF list build + n_probe query" -- as a
            # first-class object: `value` above is configs[1] (1 M spectra: every bucket flat, build = coarse = 0)
            ns = next((e for e in extra if e.get("baseline_config") == "configs[2] dataset on one GPU" and "roofline" in e), None)
            if ns is not None:
                r = ns["roofline"]
                out["north_star"] = {
                    "workload": ns["workload"], "value": ns["value"], "unit": "spectra/s", "ms_per_step": ns["ms_per_step"],
                    "stage_ms": ns["stage_ms"], "target": ">= 10 M spectra end-to-end on one MI355X, cosine kernel >= 0.50 of the HBM roof",
                    "cosine_kernel_scan_plus_topk": {"ms": r["scan_plus_topk"]["ms"],
                                                     "frac_of_hbm_roof": r["scan_plus_topk"]["frac_of_hbm_roof"],
                                                     "frac_of_f16_mfma_peak": r["scan_plus_topk"]["frac_of_mfma_peak"],
                                                     "algorithmic_bytes": r["scan_plus_topk"]["algorithmic_bytes"]},
                    "list16_kernel": {"avg_launch_ms": r["avg_launch_ms"], "launches": r["launches"],
                                      "frac_of_f16_mfma_peak": r["frac_of_mfma_peak"], "frac_of_hbm_roof": r["frac_of_hbm_roof"],
                                      "traffic": r["traffic"]},
                    "index_build_ms": ns["stage_ms"]["build"], "coarse_ms": ns["stage_ms"]["coarse"]}
        if world == 1 and not args.no_cpu_baseline:
            hosts = [{k: getattr(x, k).cpu().numpy() for k in ("precursor_mz", "retention_time", "mz", "intensity", "indptr")}
                     for x in parts]
            try:
                out["cpu_baseline"] = cpu_baseline(hosts, p)
            except Exception as e:                                # the GPU figures above must not be lost to the baseline leg
                out["cpu_baseline"] = {"value": None, "unit": "spectra/s", "cores": os.cpu_count() or 1, "kind": "port",
                                       "sample": f"failed: {e!r}"}
        line = json.dumps(out)
    if runner is not None:
        runner.close()
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
