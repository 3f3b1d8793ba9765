"""The multi-GPU exchange payload on the GPU: ELL -> CSR packing (fal_neighbors_to_csr) and the
asynchronous all-gatherv over RCCL (a 1-rank "nccl" group: the same calls bench.py --gpus N makes)."""
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ctx():
    from falcon_amd.device import Context
    c = Context(0)
    yield c
    c.close()


def _ell(n, k, seed, fill=0.15):
    rng = np.random.default_rng(seed)
    nb_idx = rng.integers(0, max(n, 1), (n, k)).astype(np.int32)
    nb_idx[rng.random((n, k)) > fill] = -1                   # holes anywhere in a row, not only at the end
    nb_dist = rng.random((n, k)).astype(np.float32)
    return nb_idx, nb_dist


def _csr(nb_idx, nb_dist, off):
    valid = nb_idx >= 0
    indptr = np.concatenate([[0], np.cumsum(valid.sum(1))]).astype(np.int64)
    return indptr, (nb_idx[valid] + off).astype(np.int32), nb_dist[valid]


@pytest.mark.parametrize("n,k", [(1, 1), (5, 64), (1000, 64), (4097, 200), (30000, 7)])
def test_neighbors_to_csr(ctx, n, k):
    import torch
    nb_idx, nb_dist = _ell(n, k, n + k)
    if n >= 1000:
        nb_idx[10:20] = -1                                   # empty rows
        nb_idx[30] = np.arange(k) % n                        # a full row
    indptr, idx, dist = ctx.neighbors_to_csr(torch.from_numpy(nb_idx).to(ctx.tdev), torch.from_numpy(nb_dist).to(ctx.tdev), 12345)
    e_indptr, e_idx, e_dist = _csr(nb_idx, nb_dist, 12345)
    assert np.array_equal(indptr.cpu().numpy(), e_indptr)
    nnz = int(e_indptr[-1])
    assert np.array_equal(idx.cpu().numpy()[:nnz], e_idx)
    assert np.array_equal(dist.cpu().numpy()[:nnz], e_dist)


def test_neighbors_to_csr_chained_segments(ctx):
    """two partitions appended into one CSR without a host round trip (row0 chaining)."""
    import torch
    k = 64
    (i1, d1), (i2, d2), (i3, d3) = _ell(700, k, 1), _ell(0, k, 2), _ell(301, k, 3)
    rows = 700 + 0 + 301
    out = (ctx.empty((rows + 1,), torch.int64), ctx.empty((rows * k,), torch.int32), ctx.empty((rows * k,), torch.float32))
    row0 = 0
    for nb_idx, nb_dist in ((i1, d1), (i2, d2), (i3, d3)):
        ctx.neighbors_to_csr(torch.from_numpy(nb_idx).to(ctx.tdev), torch.from_numpy(nb_dist).to(ctx.tdev), 5000 + row0,
                             out=out, row0=row0)
        row0 += nb_idx.shape[0]
    p1, x1, y1 = _csr(i1, d1, 5000)
    p3, x3, y3 = _csr(i3, d3, 5700)
    e_indptr = np.concatenate([p1, p1[-1] + p3[1:]])
    nnz = int(e_indptr[-1])
    assert np.array_equal(out[0].cpu().numpy(), e_indptr)
    assert np.array_equal(out[1].cpu().numpy()[:nnz], np.concatenate([x1, x3]))
    assert np.array_equal(out[2].cpu().numpy()[:nnz], np.concatenate([y1, y3]))


def test_neighbors_to_csr_empty(ctx):
    import torch
    indptr, idx, dist = ctx.neighbors_to_csr(ctx.empty((0, 64), torch.int32), ctx.empty((0, 64), torch.float32), 0)
    assert indptr.cpu().tolist() == [0]


def test_sparse_graph_exchange_rccl_one_rank(ctx):
    """start/finish through RCCL (`all_gather_into_tensor`, async) with two exchanges in flight."""
    import torch
    import torch.distributed as dist
    from falcon_amd import distributed as fd
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", str(29600 + os.getpid() % 300))
    dev = ctx.tdev
    created = not dist.is_initialized()
    if created:
        dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
    try:
        ex = fd.SparseGraphExchange(dev)
        cases, handles = [], []
        for step, n in enumerate((5000, 1200)):
            nb_idx, nb_dist = _ell(n, 64, 77 + step)
            labels = (np.arange(n) % 11).astype(np.int32)
            csr = ctx.neighbors_to_csr(torch.from_numpy(nb_idx).to(dev), torch.from_numpy(nb_dist).to(dev), 1000 * step)
            handles.append(ex.start(csr[0], csr[1], csr[2], torch.from_numpy(labels).to(dev), 11))
            cases.append((_csr(nb_idx, nb_dist, 1000 * step), labels))
        for h, ((e_indptr, e_idx, e_dist), labels) in zip(handles, cases):
            g = ex.finish(h)
            assert g["n_labels"] == 11 and len(g["idx"]) == 1
            assert np.array_equal(g["counts"][0].cpu().numpy(), np.diff(e_indptr))
            assert np.array_equal(g["idx"][0].cpu().numpy(), e_idx)
            assert np.array_equal(g["dist"][0].cpu().numpy(), e_dist)
            assert np.array_equal(g["labels"][0].cpu().numpy(), labels)
    finally:
        if created:
            dist.destroy_process_group()


def test_row_counts_from_fused_search_and_prefix_packing(ctx):
    """the fused a7+a8 kernel reports how many neighbours every row stores; CSR packing driven by those counts
    (prefix reads only) equals the count-by-scanning form."""
    import torch
    from falcon_amd import synth
    from falcon_amd.cluster.cluster import AnnParams, ClusterPipeline, SpectrumDataset
    d = synth.select_charge(synth.generate(5000, seed=3), 2)
    ds = SpectrumDataset(d["precursor_mz"], d["retention_time"], d["mz"], d["intensity"], d["indptr"])
    pipe = ClusterPipeline(ctx)
    pipe.run(ds, 20.0, "ppm", None, 0.05, 2 ** 15, AnnParams(eps=0.3, n_neighbors=8))      # small k: full rows occur
    nb_idx, nb_dist, cnt = pipe.last["nb_idx"], pipe.last["nb_dist"], pipe.last["nb_count"]
    ref_cnt = (nb_idx >= 0).sum(1).to(torch.int32)
    assert torch.equal(cnt, ref_cnt) and int(cnt.max()) == 8 and int(cnt.min()) == 0
    a = ctx.neighbors_to_csr(nb_idx, nb_dist, 77)
    b = ctx.neighbors_to_csr(nb_idx, nb_dist, 77, nb_count=cnt)
    nnz = int(a[0][-1])
    assert torch.equal(a[0], b[0]) and torch.equal(a[1][:nnz], b[1][:nnz]) and torch.equal(a[2][:nnz], b[2][:nnz])
