"""The multi-GPU exchange on CPU: world_size 2, `gloo` backend (same code path as RCCL)."""
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _worker(rank, world, port, q):
    sys.path.insert(0, ROOT)
    import torch
    import torch.distributed as dist
    from falcon_amd import distributed as fd
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        n = [5, 3][rank]
        k = 4
        rng = np.random.default_rng(rank)
        nb_idx = torch.from_numpy(rng.integers(-1, n, (n, k)).astype(np.int32))
        nb_dist = torch.from_numpy(rng.random((n, k)).astype(np.float32))
        counts = fd.allgather_counts(n, torch.device("cpu"))
        off = sum(counts[:rank])
        gi, gd, c2 = fd.allgatherv_neighbors(nb_idx, nb_dist, off)
        labels = torch.arange(n, dtype=torch.int32) % 2            # 2 local labels per rank
        gl, lc = fd.allgatherv_labels(labels, 2)
        q.put((rank, counts, gi.numpy(), gd.numpy(), gl.numpy(), lc, nb_idx.numpy(), nb_dist.numpy()))
    finally:
        dist.destroy_process_group()


def test_allgatherv_world2_gloo():
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29500 + (os.getpid() % 2000)
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted([q.get(timeout=180) for _ in procs], key=lambda x: x[0])
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    (r0, counts0, gi0, gd0, gl0, lc0, i0, d0), (r1, counts1, gi1, gd1, gl1, lc1, i1, d1) = res
    assert counts0 == counts1 == [5, 3] and lc0 == lc1 == [2, 2]
    # every rank holds the same global graph: rank order, ids shifted to global rows, -1 kept
    exp_idx = np.concatenate([i0, np.where(i1 >= 0, i1 + 5, i1)])
    exp_dist = np.concatenate([d0, d1])
    for gi, gd in ((gi0, gd0), (gi1, gd1)):
        assert np.array_equal(gi, exp_idx) and np.array_equal(gd, exp_dist)
    exp_lab = np.concatenate([np.arange(5) % 2, np.arange(3) % 2 + 2])
    assert np.array_equal(gl0, exp_lab) and np.array_equal(gl1, exp_lab)


def _ell_to_csr(nb_idx, nb_dist, off):
    """numpy statement of fal_neighbors_to_csr (row order kept, ids shifted)."""
    valid = nb_idx >= 0
    indptr = np.concatenate([[0], np.cumsum(valid.sum(1))]).astype(np.int64)
    return indptr, (nb_idx[valid] + off).astype(np.int32), nb_dist[valid].astype(np.float32)


def _csr_worker(rank, world, port, q):
    sys.path.insert(0, ROOT)
    import torch
    import torch.distributed as dist
    from falcon_amd import distributed as fd
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        ex = fd.SparseGraphExchange(torch.device("cpu"))
        handles, sent = [], []
        for step in range(2):                                   # two exchanges in flight, finished in order
            n = [[7, 4], [2, 9]][step][rank]
            k = 5
            rng = np.random.default_rng(10 * step + rank)
            nb_idx = rng.integers(-1, n, (n, k)).astype(np.int32)
            if step == 1 and rank == 0:
                nb_idx[:] = -1                                   # a rank with an empty graph
            nb_dist = rng.random((n, k)).astype(np.float32)
            off = 100 * rank
            indptr, idx, dst = _ell_to_csr(nb_idx, nb_dist, off)
            cap = n * k
            idx_buf = np.full(cap, -7, np.int32); idx_buf[:len(idx)] = idx
            dst_buf = np.full(cap, np.nan, np.float32); dst_buf[:len(dst)] = dst
            labels = (np.arange(n) % 3).astype(np.int32)
            handles.append(ex.start(torch.from_numpy(indptr), torch.from_numpy(idx_buf), torch.from_numpy(dst_buf),
                                    torch.from_numpy(labels), 3))
            sent.append((indptr, idx, dst, labels))
        outs = []
        for h in handles:
            o = ex.finish(h)
            outs.append(dict(counts=[c.numpy().copy() for c in o["counts"]], idx=[c.numpy().copy() for c in o["idx"]],
                             dist=[c.numpy().copy() for c in o["dist"]], labels=[c.numpy().copy() for c in o["labels"]],
                             n_labels=o["n_labels"]))
        q.put((rank, sent, outs))
    finally:
        dist.destroy_process_group()


def test_sparse_graph_exchange_world2_gloo():
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 31500 + (os.getpid() % 2000)
    procs = [ctx.Process(target=_csr_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted([q.get(timeout=180) for _ in procs], key=lambda x: x[0])
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    sent = [res[0][1], res[1][1]]
    for rank in range(2):
        outs = res[rank][2]
        for step in range(2):
            o = outs[step]
            assert o["n_labels"] == 6
            for r in range(2):
                indptr, idx, dst, labels = sent[r][step]
                assert np.array_equal(o["counts"][r], np.diff(indptr))
                assert np.array_equal(o["idx"][r], idx) and np.array_equal(o["dist"][r], dst)
                assert np.array_equal(o["labels"][r], labels + 3 * r)


def test_shard_rows_and_merge_shards():
    """bucket -> rank assignment keeps buckets whole; merging the per-rank results gives globally unique labels."""
    from falcon_amd import distributed as fd
    splits = np.array([0, 5, 5, 12, 13, 40, 47], np.int64)                 # 6 buckets, one empty
    n_list = np.array([1, 1, 1, 1, 4, 1])
    cost = fd.bucket_costs(np.diff(splits), n_list, 2)
    assert cost[1] == 0 and cost[2] == 49 and cost[4] == 27 * (27 * 0.5 + 12 * 4)      # flat: n^2; IVF: probed half + k-means
    owner = fd.shard_units(cost, 3)
    seen = np.zeros(47, int)
    shards = []
    for r in range(3):
        rows, sub_splits, mine = fd.shard_rows(splits, owner, r)
        seen[rows] += 1
        assert np.array_equal(np.diff(sub_splits), np.diff(splits)[mine])
        for b, (a0, a1) in zip(mine, zip(sub_splits[:-1], sub_splits[1:])):
            assert np.array_equal(rows[a0:a1], np.arange(splits[b], splits[b + 1]))    # whole buckets, in order
        lab = (np.arange(len(rows)) // 3).astype(np.int32)                  # fake local labels
        n_lab = int(lab.max()) + 1 if len(lab) else 0
        shards.append((rows, lab, rows[:n_lab].astype(np.int32)))
    assert (seen == 1).all()                                                # every row on exactly one rank
    labels, medoids = fd.merge_shards(47, shards)
    assert labels.min() == 0 and np.array_equal(np.unique(labels), np.arange(len(medoids)))
    off = 0
    for rows, lab, med in shards:                                           # rank-major offsets
        assert np.array_equal(labels[rows], lab + off)
        off += len(med)
    with pytest.raises(ValueError):
        fd.merge_shards(48, shards)


def test_window_deal_is_balanced_and_deterministic():
    """the multi-GPU front end deals (charge, precursor window) units from their spectrum counts (`window_costs` +
    `deal_job`): every unit gets exactly one owner, the deal is a pure function of the counts, and the estimated loads balance
    within 1 % in both bucket regimes (flat 1 M windows, indexed 10 M windows, windows beyond batch_size)"""
    from falcon_amd import distributed as fd
    rng = np.random.default_rng(5)
    for counts in (rng.integers(300, 1300, 800), rng.integers(7000, 10000, 800), rng.integers(30000, 70000, 800),
                   np.array([5, 0, 12000, 3]), np.zeros(0, np.int64)):
        costs = fd.window_costs(counts, 2 ** 15, 16)
        assert costs.shape == counts.shape and (costs >= 0).all()
        for world in (1, 2, 3, 8):
            parts = [costs, costs[: len(costs) // 3] * 0.4]                      # two unlike charge partitions of one dataset
            owners = fd.deal_job(parts, world)
            assert [o.shape for o in owners] == [c.shape for c in parts]
            flat = np.concatenate(owners)
            assert ((flat >= 0) & (flat < world)).all()
            again = fd.deal_job([c.copy() for c in parts], world)
            assert all(np.array_equal(x, y) for x, y in zip(owners, again))
            if len(counts) >= 100:
                loads = np.array([sum(c[o == r].sum() for c, o in zip(parts, owners)) for r in range(world)])
                assert loads.max() <= 1.01 * loads.mean()
                if world > 1:
                    assert all(len(np.unique(o)) == world for o in owners)       # window by window: every rank, every partition
    # a window larger than batch_size is costed as its chunks (cluster.py:197-207)
    assert fd.window_costs(np.array([70000]), 2 ** 15, 16)[0] == 3 * fd.window_costs(np.array([23333]), 2 ** 15, 16)[0]


def test_weak_scaling_deal_gives_every_rank_whole_partitions():
    """bench.py --scaling weak: N statistically identical blocks as 2 N charge partitions -> every rank owns two WHOLE
    partitions (a charge-2-sized and a charge-3-sized one), i.e. it runs them exactly as a single GPU would"""
    from falcon_amd import distributed as fd
    rng = np.random.default_rng(11)
    for world in (2, 4, 8):
        costs = []
        for _ in range(world):
            for mean in (875, 375):                                              # charge 2 (70 %), charge 3 (30 %)
                c = np.zeros(1201, np.int64)
                c[400:1200] = rng.poisson(mean, 800)
                costs.append(fd.window_costs(c, 2 ** 15, 16))
        owners = fd.deal_job(costs, world)
        assert all(len(np.unique(o)) == 1 for o in owners)
        whole = np.array([o[0] for o in owners])
        for r in range(world):
            mine = np.flatnonzero(whole == r)
            assert len(mine) == 2 and sorted(mine % 2) == [0, 1]


# ---------------------------------------------------------------------------------------------
# run_sharded (one dataset, buckets dealt to ranks by LPT, ONE all-gatherv) at world size 2 over gloo.
# The HIP pipeline cannot run here, so the three phases are backed by the oracle (tests may use it);
# the bucket assignment (`ClusterPipeline._restrict`), `run_sharded` and `_gather_shards` are the product's.
# ---------------------------------------------------------------------------------------------
class _OraclePipe:
    def __init__(self, data):
        import torch
        from falcon_amd.cluster.cluster import ClusterPipeline
        self.data = data

        class Ctx:
            tdev = torch.device("cpu")

            @staticmethod
            def to_dev(a, dtype=None):
                t = a if isinstance(a, torch.Tensor) else torch.from_numpy(np.ascontiguousarray(a))
                return t.to(dtype) if dtype is not None else t
        self.ctx = Ctx()
        self._restrict = lambda c, st, p, shard: ClusterPipeline._restrict(self, c, st, p, shard)

    def _front(self, c, ds, tol, mode, rt_tol, batch_size, p):
        import torch
        from falcon_amd.cluster.cluster import n_list_rule
        from oracle import falcon_oracle as fo
        pmz = np.asarray(ds.precursor_mz, np.float32)
        order = np.argsort(pmz, kind="stable")
        splits = fo.bucket_splits(pmz[order], tol, mode, batch_size, p.mz_interval)
        return dict(order=torch.from_numpy(order), mzs=torch.from_numpy(pmz[order]), rts=None, splits=splits,
                    n_list=n_list_rule(np.diff(splits), p.n_probe))

    def _search(self, ds, st, *a):
        pass

    def _graph(self, st, tol, mode, rt_tol, p, keep):
        import torch
        from oracle import falcon_oracle as fo
        rows = st["rows"].numpy()                                  # dataset rows of my buckets, sorted order
        d = self.data
        cnt = np.diff(d["indptr"])[rows]
        ip = np.zeros(len(rows) + 1, np.int64)
        np.cumsum(cnt, out=ip[1:])
        src = np.repeat(d["indptr"][:-1][rows] - ip[:-1], cnt) + np.arange(ip[-1])
        lab, med = fo.generate_clusters(d["mz"][src], d["intensity"][src], ip, d["precursor_mz"][rows], None,
                                        precursor_tol=(tol, mode), eps=p.eps, n_probe=p.n_probe, mz_interval=p.mz_interval)
        return torch.from_numpy(lab), torch.from_numpy(med), {}


def _sharded_worker(rank, world, port, q):
    sys.path.insert(0, ROOT)
    import torch.distributed as dist
    from falcon_amd import distributed as fd, synth
    from falcon_amd.cluster.cluster import AnnParams, SpectrumDataset
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        d = synth.select_charge(synth.generate(3000, seed=3, mz_lo=500.0, mz_hi=530.0), 2)
        ds = SpectrumDataset(d["precursor_mz"], d["retention_time"], d["mz"], d["intensity"], d["indptr"])
        labels, medoids = fd.run_sharded(_OraclePipe(d), ds, 20.0, "ppm", None, 0.05, 2 ** 15, AnnParams())
        q.put((rank, labels, medoids))
    finally:
        dist.destroy_process_group()


def test_run_sharded_world2_gloo_equals_one_rank():
    """`run_sharded` through `_gather_shards` with world > 1 gives every rank the partition one rank
    computes alone (labels up to the rank-major renumbering, the same medoid rows)."""
    import torch.multiprocessing as mp
    from oracle import falcon_oracle as fo
    from falcon_amd import synth
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 33500 + (os.getpid() % 2000)
    procs = [ctx.Process(target=_sharded_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted([q.get(timeout=300) for _ in procs], key=lambda x: x[0])
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    d = synth.select_charge(synth.generate(3000, seed=3, mz_lo=500.0, mz_hi=530.0), 2)
    ref, rmed = fo.generate_clusters(d["mz"], d["intensity"], d["indptr"], d["precursor_mz"], None)
    (_, l0, m0), (_, l1, m1) = res
    assert np.array_equal(l0, l1) and np.array_equal(m0, m1)                 # every rank holds the same result
    assert l0.min() == 0 and np.array_equal(np.unique(l0), np.arange(len(m0)))
    # same partition: the label pairs are in bijection
    pairs = np.unique(np.stack([ref, l0]), axis=1)
    assert pairs.shape[1] == len(rmed) == len(m0)
    assert np.array_equal(np.sort(m0), np.sort(rmed))
    assert np.array_equal(l0[m0], np.arange(len(m0)))
    assert len(np.unique(ref)) > 50 and (np.bincount(ref) > 1).sum() > 20    # a non-trivial clustering


def _self_check_worker(rank, world, port, q, corrupt):
    sys.path.insert(0, ROOT)
    import torch
    import torch.distributed as dist
    from falcon_amd import distributed as fd
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        if corrupt and rank == 1:
            # a rank whose received bytes differ from what the sender built: flip one id of rank 0's piece on arrival
            orig = fd.SparseGraphExchange.finish

            def bad_finish(self, h):
                g = orig(self, h)
                g["idx"][0][0] += 1
                return g
            fd.SparseGraphExchange.finish = bad_finish
        q.put((rank, fd.exchange_self_check(torch.device("cpu"), n_neighbors=8, scale=300, rounds=2)))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("corrupt", [False, True])
def test_exchange_self_check_world3_gloo(corrupt):
    """`bench.py --exchange-only`'s collective self-test (ragged payloads, every rank verifies every rank's bytes) on three
    gloo ranks; a single flipped id on one rank turns the verdict red on ALL ranks"""
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 31500 + (os.getpid() % 2000) + (7 if corrupt else 0)
    procs = [ctx.Process(target=_self_check_worker, args=(r, 3, port, q, corrupt)) for r in range(3)]
    for p in procs:
        p.start()
    res = dict(q.get(timeout=180) for _ in procs)
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    for r in range(3):
        assert res[r]["world_size"] == 3 and res[r]["ranks_seen"] == [0, 1, 2]
        assert res[r]["ok"] == (not corrupt)
    assert len({res[r]["payload_bytes_this_rank"] for r in range(3)}) == 3          # ragged: every rank sends a different size
    # the padded block every rank ships is the largest rank's, and the line carries what the real 10 M job should move
    assert len({res[r]["block_bytes_per_rank"] for r in range(3)}) == 1
    assert res[0]["block_bytes_per_rank"] >= max(res[r]["payload_bytes_this_rank"] for r in range(3))
    e = res[0]["expected_strong_10M_job"]
    assert e["world_size"] == 3 and e["rows_per_rank"] == 3333334 and e["received_bytes_per_rank"] == 2 * e["block_bytes_per_rank"]
    if corrupt:
        assert res[1]["local_errors"] and not res[0]["local_errors"]


def test_deal_of_a_heavy_tailed_job_falls_back_to_longest_processing_time():
    """windows with log-normal occupancy (synth skew=True at 10 M spectra): the boustrophedon deal hands the eight most
    expensive windows one to each rank whatever they cost (worst / mean 1.29 by the cost model) -- `deal_job` then deals by
    longest-processing-time and balances within a per cent; a uniform job keeps the boustrophedon deal"""
    from falcon_amd import distributed as fd, synth
    w = synth.skew_window_weights(42, 400.0, 1200.0)
    counts = [np.round(w * 7e6).astype(np.int64), np.round(w * 3e6).astype(np.int64)]
    costs = [fd.window_costs(c, 2 ** 15, 16) for c in counts]
    allc = np.concatenate(costs)
    b = np.bincount(fd.deal_units(allc, 8), weights=allc, minlength=8)
    assert b.max() / b.mean() > 1.15
    owners = fd.deal_job(costs, 8)
    loads = sum(np.bincount(o, weights=c, minlength=8) for o, c in zip(owners, costs))
    assert loads.max() / loads.mean() < 1.01
    assert all(np.array_equal(o, o2) for o, o2 in zip(owners, fd.deal_job(costs, 8)))          # deterministic
    uni = [fd.window_costs(np.full(800, 8750), 2 ** 15, 16), fd.window_costs(np.full(800, 3750), 2 ** 15, 16)]
    assert np.array_equal(np.concatenate(fd.deal_job(uni, 8)), fd.deal_units(np.concatenate(uni), 8))


def test_expected_exchange_bytes_formula():
    """`distributed.expected_exchange_bytes`: the padded int32 block of `SparseGraphExchange.start`, priced over xGMI"""
    from falcon_amd import distributed as fd
    e = fd.expected_exchange_bytes(10_000_000, 8, 3.2)
    assert e["rows_per_rank"] == 1_250_000 and e["nnz_per_rank"] == 4_000_000
    assert e["block_bytes_per_rank"] == 4 * (3 * 1_250_000 + 2 * 4_000_000) == 47_000_000
    assert e["received_bytes_per_rank"] == 7 * 47_000_000 and e["gathered_bytes_total"] == 8 * 47_000_000
    assert abs(e["ms_fan_out_7_links"] - 7 * 47e6 / (7 * 153e9) * 1e3) < 1e-9 and abs(e["ms_ring_one_link"] - 7 * e["ms_fan_out_7_links"]) < 1e-9
    one = fd.expected_exchange_bytes(1_000_000, 1, 3.2)
    assert one["received_bytes_per_rank"] == 0 and one["ms_ring_one_link"] == 0.0


def test_window_costs_price_the_stored_neighbours_of_full_windows():
    """under skew the deal modelled 1.00 and measured 1.08 -- a window cut into several buckets keeps more
    neighbours per row (each bucket spans a fraction of the window's m/z, so a larger share of a row's best candidates passes
    the precursor tolerance) and everything behind the search (exact chains, DBSCAN, refinement, medoids) follows the stored
    neighbours.  With the tolerance given, `window_costs` adds that term: a 65 k-row window (2 buckets) costs more per row
    than two 32 k-row windows' rows, the extra is the neighbour term; without the tolerance the old figure is unchanged."""
    from falcon_amd import distributed as fd
    counts = np.zeros((1, 700), np.int64)
    counts[0, 600] = 65000
    counts[0, 601] = 32500
    plain = fd.window_costs(counts, 2 ** 15, 16)
    assert plain[0, 600] == 2 * plain[0, 601]
    priced = fd.window_costs(counts, 2 ** 15, 16, 1.0, (20.0, "ppm"), 128, 64)
    assert priced[0, 600] > 2 * priced[0, 601] > 2 * plain[0, 601]
    # per row: cluster members + k_ann x (2 tol x chunks / window width), tolerance taken at the window's m/z
    extra = (priced - plain)[0, 600:602] / counts[0, 600:602] / fd.NEIGHBOUR_UNITS
    np.testing.assert_allclose(extra, [8 + 128 * 2 * 20e-6 * 600.5 * 2, 8 + 128 * 2 * 20e-6 * 601.5], rtol=1e-12)
    # Da tolerance: the same everywhere; the count is capped at n_neighbors
    da = fd.window_costs(counts, 2 ** 15, 16, 1.0, (0.5, "Da"), 128, 64)
    assert np.allclose((da - plain)[0, 601] / 32500 / fd.NEIGHBOUR_UNITS, 64.0)
    # 1-D counts keep working (window index = position)
    assert np.array_equal(fd.window_costs(counts[0], 2 ** 15, 16, 1.0, (20.0, "ppm")), priced[0])
