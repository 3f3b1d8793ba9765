"""The HIP pipeline under world_size 2 (VERDICT r2 missing #5): two FRESH processes (torch.distributed.run, gloo backend, both
on GPU 0) run `ClusterPipeline.run_many(shard=(rank, 2))` on their precursor buckets of ONE dataset and the one exchange step
(`start_graph_exchange` -> `SparseGraphExchange`); the assembled labels and the gathered sparse neighbour graph must equal
the single-rank result.

This file sorts first among the GPU tests on purpose: the pytest process must not have initialised the GPU when it starts the
launcher (an exec from a process that owns a GPU context is refused on the GPU boxes); the test itself creates no context."""
import os
import socket
import subprocess
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


@pytest.mark.parametrize("n_spectra,mz_lo,mz_hi,regime,mode", [
    (30000, 500.0, 560.0, "flat", "device"),  # ~350-row windows: flat buckets, thousands of units to deal
    (60000, 600.0, 606.0, "ivf", "device"),   # ~7,000-row windows: k-means index (n_list 128), prefiltered fine scan
    # the partitions HOST-resident (PartitionRunner.run(shard=)): a rank uploads the precursor columns and the peaks of its own
    # windows only -- at most 0.55 of the dataset's bytes cross PCIe per rank at world size 2
    (30000, 500.0, 560.0, "flat", "host"),
    (60000, 600.0, 612.0, "ivf", "host"),
])
def test_two_ranks_on_one_gpu_equal_one_rank(tmp_path, n_spectra, mz_lo, mz_hi, regime, mode):
    import torch
    if torch.cuda.is_initialized():
        pytest.fail("this pytest process already owns a GPU context (a GPU test file sorted in front of this one?): "
                    "tests/test_gpu_00_world2.py must run first / on its own")
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0", OMP_NUM_THREADS="4")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", str(_free_port()), os.path.join(ROOT, "tests", "world2_worker.py"), str(tmp_path), str(n_spectra),
           str(mz_lo), str(mz_hi), mode]
    proc = subprocess.Popen(cmd, env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
    try:
        log, _ = proc.communicate(timeout=900)
    except subprocess.TimeoutExpired:
        proc.kill()
        log, _ = proc.communicate()
        pytest.fail("world-size-2 job timed out:\n" + log[-3000:])
    assert proc.returncode == 0, log[-3000:]
    r0, r1 = (np.load(os.path.join(tmp_path, f"rank{r}.npz")) for r in (0, 1))
    if mode == "host":
        for r in (r0, r1):
            assert 0.2 * int(r["dataset_bytes"]) < int(r["h2d_bytes"]) <= 0.55 * int(r["dataset_bytes"]), (int(r["h2d_bytes"]), int(r["dataset_bytes"]))
    single = r0["single_labels"]
    n = len(single)
    assert (int(r0["single_n_list_max"]) > 1) == (regime == "ivf")
    # both ranks did real work and between them covered the dataset
    assert int(r0["rows_local0"]) + int(r1["rows_local0"]) == n and min(int(r0["rows_local0"]), int(r1["rows_local0"])) > n // 4
    for rep in (0, 1):
        lab = r0[f"labels{rep}"]
        assert np.array_equal(lab, r1[f"labels{rep}"])                         # every rank assembled the same global labels
        assert lab.min() == 0 and len(np.unique(lab)) == lab.max() + 1 == int(r0[f"n_local{rep}"]) + int(r1[f"n_local{rep}"])
        pairs = np.unique(np.stack([single, lab]), axis=1)
        assert pairs.shape[1] == len(np.unique(single)) == len(np.unique(lab))    # the same partition as one rank
        # the gathered sparse graph == the single rank's neighbour lists: same (row, neighbour, distance bits) triples
        e = r0[f"edges{rep}"]
        assert np.array_equal(e, r1[f"edges{rep}"])
        key = lambda t: t[:, np.lexsort((t[2], t[1], t[0]))]
        assert np.array_equal(key(e), key(r0["single_edges"]))
    assert (np.bincount(single) > 1).sum() > 20                               # a non-trivial clustering


@pytest.mark.parametrize("scaling,extra", [
    # 2 blocks -> 4 charge partitions, whole partitions per rank; then the line's `strong_scaling` leg: one shared dataset
    ("weak", ["--spectra", "150000", "--spectra-total", "200000", "--with-strong"]),
    ("strong", ["--spectra-total", "300000", "--no-configs"]),           # one dataset, window by window
])
def test_bench_runs_as_a_two_rank_job(scaling, extra):
    """bench.py's own N > 1 control flow (dataset of the scaling mode, the deal, the overlapped exchange, label assembly, max
    over ranks) as two fresh ranks on GPU 0 -- its test hooks swap RCCL for gloo and pin both ranks to one device"""
    import json
    import torch
    if torch.cuda.is_initialized():
        pytest.fail("this pytest process already owns a GPU context (a GPU test file sorted in front of this one?): "
                    "tests/test_gpu_00_world2.py must run first / on its own")
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0", OMP_NUM_THREADS="4", FALCON_BENCH_DEVICE="0", FALCON_BENCH_BACKEND="gloo")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", str(_free_port()), os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "3", "--warmup", "1",
           "--scaling", scaling, "--no-cpu-baseline"] + extra
    proc = subprocess.Popen(cmd, env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
    try:
        log, _ = proc.communicate(timeout=900)
    except subprocess.TimeoutExpired:
        proc.kill()
        log, _ = proc.communicate()
        pytest.fail("two-rank bench timed out:\n" + log[-3000:])
    assert proc.returncode == 0, log[-3000:]
    lines = [l for l in log.splitlines() if l.startswith("{")]
    assert len(lines) == 1, log[-3000:]                                         # rank 0 prints ONE JSON line
    out = json.loads(lines[0])
    assert out["n_gpus"] == 2 and out["scaling"] == scaling and out["steps"] == 3 and out["value"] > 0
    assert out["unit"] == "spectra/s" and out["roofline"]["frac"] <= 1.0
    assert abs(out["value"] - 300000 * 3 / (out["ms_per_step"] * 3e-3)) < 1e-6 * out["value"]
    assert out["rccl"]["ranks"] == 2
    # every rank's host-to-host leg: the partitions in pinned host memory, a rank uploads its own windows only
    h = out["host_to_host"]
    assert "error" not in h and h["value_host_to_host"] > 0 and 0.2 < h["uploaded_fraction_rank0"] <= (0.62 if scaling == "weak" else 0.55), h
    if scaling == "weak":
        assert out["strong_scaling"]["value"] > 0 and out["strong_scaling"]["scaling"] == "strong", out["strong_scaling"]


def test_environment_switches_do_not_change_results(tmp_path):
    """`FALCON_PROBE_LDS_LISTS=0` (the inverted probe table built with 4-byte cursors in LDS and offsets from global memory: the form
    for buckets with thousands of lists) and `FALCON_DEBUG_POISON=1` give the labels and medoids of the default run, bit for bit.
    Each setting is read once per process: fresh processes (this file runs before the pytest process owns a GPU context)."""
    import torch
    if torch.cuda.is_initialized():
        pytest.fail("this pytest process already owns a GPU context: tests/test_gpu_00_world2.py must run first / on its own")
    outs = {}
    for name, extra in (("default", {}), ("small_lds", {"FALCON_PROBE_LDS_LISTS": "0"}), ("poison", {"FALCON_DEBUG_POISON": "1"})):
        env = {k: v for k, v in os.environ.items() if k not in ("FALCON_PROBE_LDS_LISTS", "FALCON_DEBUG_POISON")}
        env.update(extra)
        out = os.path.join(tmp_path, name + ".npz")
        proc = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "env_switch_worker.py"), out], env=env,
                              stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=600)
        assert proc.returncode == 0, proc.stdout[-3000:]
        outs[name] = np.load(out)
    ref = outs["default"]
    assert int(ref["n_list_max16"]) > 1                                         # the indexed path ran
    assert (np.bincount(ref["labels16"]) > 1).sum() > 20                        # a non-trivial clustering
    for name in ("small_lds", "poison"):
        for key in ref.files:
            assert np.array_equal(outs[name][key], ref[key]), (name, key)
