"""Worker of tests/test_gpu_00_world2.py: one rank of a world-size-N job on ONE GPU (gloo carries the collectives; RCCL refuses
two ranks on one device).  Launched by `python -m torch.distributed.run`; never imported by pytest.

Every rank builds the same dataset (numpy generator, same seed), runs the HIP pipeline on its own precursor buckets
(`ClusterPipeline.run_many(shard=(rank, world))`), and takes part in the ONE exchange step (`start_graph_exchange` ->
`SparseGraphExchange`: CSR neighbour lists with dataset-row ids + labels + rows).  Rank 0 also runs the single-rank
pipeline on the whole dataset.  Results go to <outdir>/rank<r>.npz for the test to compare."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main(outdir, n_spectra, mz_lo, mz_hi, mode="device"):
    import torch
    import torch.distributed as dist
    from falcon_amd import distributed as fdist, synth
    from falcon_amd.cluster.cluster import AnnParams, ClusterPipeline, PartitionRunner, SpectrumDataset
    from falcon_amd.device import Context

    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    torch.cuda.set_device(0)
    dev = torch.device("cuda", 0)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    ctx = Context(0)
    pipe = ClusterPipeline(ctx)
    data = synth.generate(n_spectra, seed=77, mz_lo=mz_lo, mz_hi=mz_hi)
    parts = []
    for charge in (2, 3):
        c = synth.select_charge(data, charge)
        parts.append(SpectrumDataset(*[ctx.to_dev(c[k], dt) for k, dt in (
            ("precursor_mz", torch.float32), ("retention_time", torch.float32), ("mz", torch.float32),
            ("intensity", torch.float32), ("indptr", torch.int64))]))
    part_off = np.concatenate([[0], np.cumsum([len(x) for x in parts])])
    n_total = int(part_off[-1])
    p = AnnParams()
    args = (20.0, "ppm", None, 0.05, 2 ** 15, p)
    exchange = fdist.SparseGraphExchange(dev)
    out = {}
    runner, host_parts = None, None
    if mode == "host":
        # the partitions stay in host memory (what generate_clusters is handed, cluster.py:73-85): a rank uploads the precursor
        # columns (4 bytes per spectrum) for the deal and then the peaks of ITS windows only (PartitionRunner.run -> run_many)
        runner = PartitionRunner(0, 2)
        host_parts = []
        for charge in (2, 3):
            c = synth.select_charge(data, charge)
            host_parts.append(SpectrumDataset(c["precursor_mz"], c["retention_time"], c["mz"], c["intensity"], c["indptr"]))
        out["dataset_bytes"] = np.int64(sum(sum(np.asarray(t).nbytes for t in ds.columns()) for ds in host_parts))
    for rep in range(2):                                  # twice: the second pass reuses every scratch buffer
        if runner is not None:
            outs = runner.run(host_parts, *args, shard=(rank, world))
            lasts = runner.lasts
            if rep == 0:
                ctxs = [pl.ctx for pl in runner.pipelines] + [getattr(pl, "_front_ctx", None) for pl in runner.pipelines]
                ctxs += [runner._planner.ctx] if hasattr(runner, "_planner") else []
                out["h2d_bytes"] = np.int64(sum(c.h2d_bytes for c in ctxs if c is not None))
        else:
            outs = pipe.run_many(parts, *args, shard=(rank, world))
            lasts = pipe.lasts
        handle, _, n_local = fdist.start_graph_exchange(ctx, exchange, outs, lasts, part_off, p.n_neighbors, True)
        g = exchange.finish(handle)
        labels = fdist.SparseGraphExchange.assemble_labels(g, n_total).cpu().numpy()
        # the gathered sparse graph as per-row lists of the WHOLE dataset: ragged -> (row, neighbour, distance bits) triples
        rows = torch.cat([torch.repeat_interleave(g["rows"][r].long(), g["counts"][r].long()) for r in range(world)])
        idx = torch.cat(g["idx"]).long()
        dbits = torch.cat(g["dist"]).view(torch.int32).long()
        out[f"labels{rep}"] = labels
        out[f"edges{rep}"] = torch.stack([rows, idx, dbits]).cpu().numpy()
        out[f"n_local{rep}"] = np.int64(n_local)
        out[f"rows_local{rep}"] = np.int64(sum(int(o[0].numel()) for o in outs))
        out["n_list_max"] = np.int64(max([int(np.max(l["n_list"])) for l in lasts if "n_list" in l and len(l["n_list"])] + [0]))
    if rank == 0:
        single = pipe.run_many(parts, *args)
        labs, cur, edges = [], 0, []
        for j, ((lab, med), last) in enumerate(zip(single, pipe.lasts)):
            labs.append((lab + cur).cpu().numpy())
            cur += int(med.numel())
            order = last["order"]                                             # sorted position -> dataset row of the partition
            nb_idx, nb_dist = last["nb_idx"], last["nb_dist"]
            keep = nb_idx >= 0
            r = order[:, None].expand_as(nb_idx)[keep] + int(part_off[j])
            c = order[nb_idx.clamp(min=0).long()][keep] + int(part_off[j])
            edges.append(torch.stack([r, c, nb_dist[keep].view(torch.int32).long()]))
        out["single_labels"] = np.concatenate(labs)
        out["single_edges"] = torch.cat(edges, 1).cpu().numpy()
        out["single_n_list_max"] = np.int64(max(int(np.max(l["n_list"])) for l in pipe.lasts))
    np.savez(os.path.join(outdir, f"rank{rank}.npz"), **out)
    if runner is not None:
        runner.close()
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    main(sys.argv[1], int(sys.argv[2]), float(sys.argv[3]), float(sys.argv[4]), sys.argv[5] if len(sys.argv) > 5 else "device")
