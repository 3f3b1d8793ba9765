"""Host logic that runs without a GPU: config surface, n_list rule, MGF I/O, preprocessing,
synthetic generator, shard assignment, and the C1 plumbing case through the oracle."""
import io
import os

import numpy as np
import pytest

from oracle import falcon_oracle as fo


def test_config_defaults_match_reference_and_readme():
    from falcon_amd.config import Config
    c = Config()
    with pytest.raises(RuntimeError):           # config.py:203-206
        c.eps
    c.parse(["in.mgf", "out"])
    # config.py:52-183 defaults
    assert c.input_filenames == ["in.mgf"] and c.output_filename == "out"
    assert c.precursor_tol == [20.0, "ppm"] and c.rt_tol is None and c.fragment_tol == 0.05
    assert c.linkage == "complete" and c.distance_threshold == 0.1 and c.min_matched_peaks == 0
    assert c.batch_size == 2 ** 15 and c.min_peaks == 5 and c.min_mz_range == 250.0
    assert c.min_mz == 101.0 and c.max_mz == 1500.0 and c.remove_precursor_tol == 1.5
    assert c.min_intensity == 0.01 and c.max_peaks_used == 50 and c.scaling == "off"
    assert c.work_dir is None and not c.overwrite and not c.export_representatives
    # README options
    assert c.eps == 0.1 and c.n_probe == 16 and c.n_neighbors == 64 and c.n_neighbors_ann == 128 and c.low_dim == 400
    assert c["eps"] == c.eps


def test_config_cli_ini_and_aliases(tmp_path):
    from falcon_amd.config import Config
    c = Config()
    c.parse("a.mgf b.mgf out --eps 0.25 --precursor_tol 0.05 Da --rt_tol 30 --export_representatives")  # README.md:49
    assert c.input_filenames == ["a.mgf", "b.mgf"] and c.eps == 0.25 and c.distance_threshold == 0.25
    assert c.precursor_tol == [0.05, "Da"] and c.rt_tol == 30.0 and c.export_representatives
    ini = tmp_path / "falcon.ini"
    ini.write_text("eps = 0.2\nn_probe = 8\nprecursor_tol = 10 ppm\noverwrite = true\n")
    c = Config()
    c.parse(["-c", str(ini), "x.mgf", "out", "--n_probe", "4"])
    assert c.eps == 0.2 and c.n_probe == 4 and c.precursor_tol == [10.0, "ppm"] and c.overwrite
    c = Config()
    with pytest.raises(ValueError):
        c.parse(["x.mgf", "out", "--n_neighbors", "64", "--n_neighbors_ann", "32"])
    with pytest.raises(SystemExit):
        Config().parse(["x.mgf", "out", "--scaling", "bogus"])


def test_n_list_rule_matches_oracle():
    from falcon_amd.cluster.cluster import n_list_rule
    sizes = np.array([0, 1, 5, 100, 101, 624, 700, 1248, 3000, 40000, 10 ** 7])
    for n_probe in (1, 16, 32):
        got = n_list_rule(sizes, n_probe)
        for s, g in zip(sizes, got):
            ref = fo.n_list_for(int(s)) if s > 0 else 1
            assert g == (1 if ref <= n_probe else ref), (s, n_probe)


def test_mgf_roundtrip_and_preprocessing(tmp_path):
    from falcon_amd.ms_io import mgf_io, ms_io
    from falcon_amd.cluster.spectrum import process_spectrum
    specs = [{"identifier": "s1", "precursor_mz": 500.25, "precursor_charge": 2, "retention_time": 12.5,
              "mz": np.array([150.0, 300.5, 499.9, 700.25, 900.0, 1200.0]),
              "intensity": np.array([10, 200, 50, 80, 5, 1], np.float32)},
             {"identifier": "s2", "precursor_mz": 800.0, "precursor_charge": None, "retention_time": 3.0,
              "mz": np.array([120.0, 130.0]), "intensity": np.array([1, 2], np.float32)}]
    fn = str(tmp_path / "t.mgf")
    ms_io.write_spectra(fn, specs)
    back = list(ms_io.get_spectra(fn))
    assert [b["identifier"] for b in back] == ["s1", "s2"]
    assert back[0]["precursor_charge"] == 2 and back[1]["precursor_charge"] is None
    np.testing.assert_allclose(back[0]["mz"], specs[0]["mz"])
    # malformed spectrum is skipped (mgf_io.py:27-30)
    bad = "BEGIN IONS\nTITLE=x\nCHARGE=2+\n100 1\nEND IONS\n"
    assert list(mgf_io.get_spectra(io.StringIO(bad))) == []
    with pytest.raises(ValueError):
        list(ms_io.get_spectra(str(tmp_path / "nope.mgf")))
    # 6 peaks: 499.9 falls to the precursor window (charge 2 -> 500.25), 1200 to the 1 % filter
    assert process_spectrum(dict(back[0], filename="f"), 5, 250.0, 101.0, 1500.0, 1.5, 0.01, 50, None) is None
    out = process_spectrum(dict(back[0], filename="f"), 4, 250.0, 101.0, 1500.0, 1.5, 0.01, 50, None)
    assert out is not None and abs(np.linalg.norm(out["intensity"]) - 1) < 1e-6
    assert list(out["mz"]) == [150.0, 300.5, 700.25, 900.0]
    assert np.all(np.diff(out["mz"]) > 0) and out["mz"].dtype == np.float32
    assert process_spectrum(dict(back[1], filename="f"), 5, 250.0, 101.0, 1500.0, 1.5, 0.01, 50, None) is None
    root = process_spectrum(dict(back[0], filename="f"), 3, 100.0, 101.0, 1500.0, None, None, None, "root")
    assert abs(np.linalg.norm(root["intensity"]) - 1) < 1e-6


def test_synth_is_deterministic_and_well_formed():
    from falcon_amd import synth
    a, b = synth.generate(3000), synth.generate(3000)
    for k in a:
        assert np.array_equal(a[k], b[k])
    cnt = np.diff(a["indptr"])
    assert cnt.max() <= 50 and cnt.min() >= 1 and a["mz"].dtype == np.float32
    assert a["mz"].min() >= 101 and a["mz"].max() <= 1500
    for i in (0, 17, 2999):
        m = a["mz"][a["indptr"][i]:a["indptr"][i + 1]]
        assert np.all(np.diff(m) >= 0)
        it = a["intensity"][a["indptr"][i]:a["indptr"][i + 1]]
        assert abs(np.linalg.norm(it.astype(np.float64)) - 1) < 1e-5
    c2 = synth.select_charge(a, 2)
    assert set(np.unique(a["precursor_charge"])) == {2, 3} and len(c2["precursor_mz"]) == (a["precursor_charge"] == 2).sum()
    # blocks of other ranks differ
    assert not np.array_equal(synth.generate(1000, first_block=1)["precursor_mz"], synth.generate(1000)["precursor_mz"])


def test_shard_units_lpt():
    from falcon_amd.distributed import shard_units
    rng = np.random.default_rng(0)
    cost = rng.pareto(2.0, 500) + 0.1
    owner = shard_units(cost, 8)
    load = np.bincount(owner, weights=cost, minlength=8)
    assert load.max() / load.mean() < 1.05
    assert np.array_equal(owner, shard_units(cost, 8))


def test_c1_plumbing_mgf_through_the_oracle(tmp_path):
    """BASELINE config 1 (CPU plumbing): synthetic spectra -> MGF -> reader -> preprocessing ->
    the CPU restatement of the hot path; clustering equals the direct-array run."""
    from falcon_amd import synth
    from falcon_amd.ms_io import ms_io
    from falcon_amd.cluster.spectrum import process_spectrum
    from sklearn.metrics import adjusted_rand_score
    d = synth.generate(1500, seed=11)
    specs = []
    for i in range(1500):
        a, b = d["indptr"][i], d["indptr"][i + 1]
        specs.append({"identifier": f"spec_{i}", "precursor_mz": float(d["precursor_mz"][i]),
                      "precursor_charge": int(d["precursor_charge"][i]), "retention_time": float(d["retention_time"][i]),
                      "mz": d["mz"][a:b].astype(np.float64), "intensity": d["intensity"][a:b]})
    fn = str(tmp_path / "c1.mgf")
    ms_io.write_spectra(fn, specs)
    got = [process_spectrum(dict(s, filename=fn), 5, 250.0, 100.95, 1500.0, None, None, 50, None)
           for s in ms_io.get_spectra(fn)]
    got = [g for g in got if g is not None and g["precursor_charge"] == 2]
    sel = np.flatnonzero(d["precursor_charge"] == 2)
    assert len(got) == len(sel)
    indptr = np.zeros(len(got) + 1, np.int64)
    np.cumsum([len(g["mz"]) for g in got], out=indptr[1:])
    lab_mgf, _ = fo.generate_clusters(np.concatenate([g["mz"] for g in got]),
                                      np.concatenate([g["intensity"] for g in got]), indptr,
                                      np.array([g["precursor_mz"] for g in got], np.float32), None, eps=0.3)
    c2 = synth.select_charge(d, 2)
    lab_arr, med = fo.generate_clusters(c2["mz"], c2["intensity"], c2["indptr"], c2["precursor_mz"], None, eps=0.3)
    import warnings
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        assert adjusted_rand_score(lab_arr, lab_mgf) >= 0.99
    assert np.array_equal(lab_arr[med], np.arange(len(med)))


def test_oracle_ivf_exhaustive_equals_bruteforce_and_dbscan_variants_agree():
    from falcon_amd import synth
    from sklearn.metrics import adjusted_rand_score
    d = synth.select_charge(synth.generate(4000, seed=2, mz_lo=600.0, mz_hi=603.0), 2)
    nb, start, _ = fo.get_dim(101, 1500, 0.05)
    X = fo.vectorize(d["mz"], d["intensity"], d["indptr"], start, 0.05, nb, 400)[:1200]
    C, a, perm, off = fo.ivf_build(X, 16, 3)
    s1, i1 = fo.ivf_search(X, C, a, perm, off, 16, 32)
    s2, i2 = fo.exhaustive_topk(X, 32)
    assert np.array_equal(i1, i2) and np.allclose(s1, s2, atol=1e-6)
    kw = dict(eps=0.3, mz_interval=0.0, batch_size=1024, n_probe=4, kmeans_iters=3)
    l_comp, _ = fo.generate_clusters(d["mz"], d["intensity"], d["indptr"], d["precursor_mz"], None, **kw)
    l_skl, _ = fo.generate_clusters(d["mz"], d["intensity"], d["indptr"], d["precursor_mz"], None, dbscan="sklearn", **kw)
    import warnings
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        assert adjusted_rand_score(l_skl, l_comp) >= 0.99      # north_star gate, order-independent DBSCAN


def test_config_ini_values_do_not_leak_into_the_next_parse(tmp_path):
    """INI values were installed as parser defaults and survived into later parse() calls."""
    from falcon_amd.config import Config
    c = Config()
    ini = tmp_path / "a.ini"
    ini.write_text("eps = 0.3\nn_probe = 4\nscaling = root\n")
    c.parse(f"-c {ini} in.mgf out")
    assert (c.eps, c.n_probe, c.scaling) == (0.3, 4, "root")
    c.parse("in.mgf out")
    assert (c.eps, c.n_probe, c.scaling) == (0.1, 16, "off")
    c.parse("in.mgf out --clustering hierarchical --linkage average")
    assert c.rescore and c.linkage == "average"
    with pytest.raises(SystemExit):
        c.parse("in.mgf out --linkage ward")
    # the reference's --linkage values are honoured (they select the hierarchical clustering) or fail at
    # parse time, never per charge after the spectra were read
    c.parse("in.mgf out --linkage single")
    assert c.clustering == "hierarchical" and c.rescore and c.linkage == "single"
    c.parse("in.mgf out")
    assert c.clustering == "dbscan" and not c.rescore
    with pytest.raises(SystemExit):
        c.parse("in.mgf out --clustering dbscan --linkage average")
    ini2 = tmp_path / "b.ini"
    ini2.write_text("linkage = average\n")
    c.parse(f"-c {ini2} in.mgf out")
    assert c.clustering == "hierarchical" and c.linkage == "average"


def test_generate_clusters_argument_errors():
    """linkage / threshold arguments are validated before any device work (no silent ignoring)."""
    from falcon_amd.cluster.cluster import AnnParams, SpectrumDataset, generate_clusters
    ds = SpectrumDataset(np.zeros(3, np.float32), np.zeros(3, np.float32), np.zeros(3, np.float32),
                         np.zeros(3, np.float32), np.arange(4, dtype=np.int64))
    with pytest.raises(ValueError, match="only applies to the hierarchical"):
        generate_clusters(ds, "average", 0.1, 0, 20.0, "ppm", None, 0.05, 2 ** 15)
    with pytest.raises(ValueError, match="unknown linkage"):
        generate_clusters(ds, "ward", 0.1, 0, 20.0, "ppm", None, 0.05, 2 ** 15)
    ann = AnnParams(eps=0.2)
    with pytest.raises(ValueError, match="differ"):
        generate_clusters(ds, "complete", 0.1, 0, 20.0, "ppm", None, 0.05, 2 ** 15, ann=ann)
    assert ann.min_matches == 0 and ann.eps == 0.2            # the caller's parameters are never touched


def test_mgf_corrupt_peak_line_skips_the_spectrum():
    """a peak line that does not parse drops the whole spectrum (pyteomics raises and the reference's
    try/except in get_spectra yields nothing, mgf_io.py:27-30)."""
    import io
    from falcon_amd.ms_io import mgf_io
    txt = ("BEGIN IONS\nTITLE=a\nPEPMASS=500.1\nCHARGE=2+\n100.0 1.0\n2x0.0 5\n300 2\nEND IONS\n"
           "BEGIN IONS\nTITLE=b\nPEPMASS=600.1\n100.0 1.0\n200.0 5\nEND IONS\n")
    got = list(mgf_io.get_spectra(io.StringIO(txt)))
    assert [g["identifier"] for g in got] == ["b"]


def test_csv_writer_quotes_like_pandas(tmp_path):
    """minimal quoting with doubled quotes for every text column, float32 values in their own repr."""
    import csv
    from falcon_amd import falcon as fmain
    from falcon_amd.config import config
    config.parse(f"in.mgf {tmp_path / 'out'}")
    rows = [("/d/a,b.mgf", 'id "x", 1', "2", np.float32(500.23), np.float32(12.5), 3),
            ("/d/c.mgf", "plain", "None", np.float32(1199.9999), np.float32(-1), 4)]
    fmain._write_cluster_info(rows)
    lines = [l for l in open(f"{tmp_path / 'out'}.csv") if not l.startswith("#")]
    back = list(csv.reader(lines))
    assert back[0] == ["filename", "spectrum_id", "precursor_charge", "precursor_mz", "retention_time", "cluster"]
    assert back[1] == ["/d/a,b.mgf", 'id "x", 1', "2", "500.23", "12.5", "3"]
    assert back[2] == ["/d/c.mgf", "plain", "None", "1199.9999", "-1.0", "4"]
    assert '"id ""x"", 1"' in lines[1]


def test_device_generator_matches_the_numpy_recipe_statistically():
    """bench.py draws its workloads with `synth.generate_device` (torch; here on the CPU device): same recipe as the
    numpy generator of the parity tests, another random stream."""
    import torch
    from falcon_amd import synth
    n = 60000
    a = synth.generate(n)
    b = {k: v.numpy() for k, v in synth.generate_device(n, torch.device("cpu")).items()}
    assert np.array_equal(b["indptr"][:1], [0]) and b["indptr"][-1] == len(b["mz"]) and b["mz"].dtype == np.float32
    ca, cb = np.diff(a["indptr"]), np.diff(b["indptr"])
    assert cb.max() <= 50 and cb.min() >= 1 and abs(ca.mean() - cb.mean()) < 0.05
    for k, rel in (("mz", 0.005), ("intensity", 0.01), ("precursor_mz", 0.01), ("retention_time", 0.01)):
        assert abs(a[k].mean() - b[k].mean()) <= rel * abs(a[k].mean()), k
        assert abs(a[k].std() - b[k].std()) <= 2 * rel * a[k].std(), k
    assert abs((a["precursor_charge"] == 2).mean() - (b["precursor_charge"] == 2).mean()) < 0.01
    ha, hb = np.bincount(np.bincount(a["truth"]), minlength=30)[:30], np.bincount(np.bincount(b["truth"]), minlength=30)[:30]
    assert np.abs(ha - hb).max() <= 0.1 * ha.max()                        # same cluster-size distribution
    for i in (0, 77, n - 1):                                              # sorted peaks, unit norm
        m = b["mz"][b["indptr"][i]:b["indptr"][i + 1]]
        assert np.all(np.diff(m) >= 0) and m.min() >= 101 and m.max() <= 1500
        assert abs(np.linalg.norm(b["intensity"][b["indptr"][i]:b["indptr"][i + 1]].astype(np.float64)) - 1) < 1e-5
    c2 = synth.select_charge_device(synth.generate_device(5000, torch.device("cpu")), 2)
    assert int(c2["indptr"][-1]) == c2["mz"].numel() and (c2["precursor_charge"] == 2).all()


def test_skewed_generator_numpy_and_device_agree_statistically():
    """`skew=True`: log-normal occupancy of the 1 m/z precursor windows -- the SAME windows in the numpy and
    the torch generator (the weights come from the seed alone) -- and 5..50 peaks per template; the default recipe's stream is
    untouched (the golden vectors and every parity test depend on it)."""
    import torch
    from falcon_amd import synth
    a = synth.generate(150000, seed=9, skew=True)
    b = synth.generate_device(150000, torch.device("cpu"), seed=9, skew=True)
    wa = np.bincount(np.floor(a["precursor_mz"]).astype(int) - 400, minlength=800)[:800]
    wb = np.bincount(np.floor(b["precursor_mz"].numpy()).astype(int) - 400, minlength=800)[:800]
    assert wa.max() > 40 * np.median(wa) and np.corrcoef(wa, wb)[0, 1] > 0.95
    pa, pb = np.diff(a["indptr"]), np.diff(b["indptr"].numpy())
    assert pa.min() <= 5 and pa.max() == 50 and abs(pa.mean() - pb.mean()) < 1.0 and 20 < pa.mean() < 35
    w = synth.skew_window_weights(9, 400.0, 1200.0)
    assert len(w) == 800 and abs(w.sum() - 1) < 1e-12 and np.array_equal(w, synth.skew_window_weights(9, 400.0, 1200.0))
    u = synth.generate(20000, seed=9)                            # the default: uniform windows, ~50 peaks
    assert np.diff(u["indptr"]).mean() > 45


def test_spectrum_dataset_knows_where_its_columns_live():
    """`PartitionRunner.run` uploads host-resident partitions itself (cluster.py: SpectrumDataset.on_host / to_device): numpy
    columns and CPU tensors count as host-resident; the column order is the one `generate_clusters` reads (cluster.py:73-85)"""
    import torch
    from falcon_amd.cluster.cluster import SpectrumDataset
    cols = (np.zeros(3, np.float32), np.ones(3, np.float32), np.arange(5, dtype=np.float32), np.ones(5, np.float32),
            np.array([0, 2, 2, 5], np.int64))
    ds = SpectrumDataset(*cols)
    assert ds.on_host() and len(ds) == 3
    assert [c is d for c, d in zip(ds.columns(), cols)] == [True] * 5
    dt = SpectrumDataset(*[torch.from_numpy(c) for c in cols])
    assert dt.on_host() and len(dt) == 3
    moved = dt.to_device(torch.device("cpu"))                    # (same code path as the upload, no GPU needed)
    assert all(torch.equal(a, b) for a, b in zip(moved.columns(), dt.columns()))
    # a dataset without retention times (legal: the front end checks for it) and with float64 / int32 host columns: the
    # upload passes None through and hands the path the dtypes it expects
    odd = SpectrumDataset(cols[0].astype(np.float64), None, cols[2].astype(np.float64), cols[3], cols[4].astype(np.int32))
    assert odd.on_host()
    up = odd.to_device(torch.device("cpu"))
    assert up.retention_time is None
    assert [t.dtype for t in (up.precursor_mz, up.mz, up.intensity, up.indptr)] == [torch.float32] * 3 + [torch.int64]
    assert torch.equal(up.indptr, torch.from_numpy(cols[4])) and torch.equal(up.mz, torch.from_numpy(cols[2]))


def test_low_dim_is_a_free_integer_rows_are_padded_to_an_instantiated_width(tmp_path):
    """README.md:114-117: `--low_dim` is any integer; the path stores the rows `row_width(low_dim)` columns wide (64 / 128 / 256 /
    400 / 800).  Oracle side of the rule: the padded columns are zero, the first low_dim columns are the unpadded vectors bit
    for bit (same adds, same norm tree), and the k-ordered similarities of padded rows stay within 1e-6 of the unpadded ones
    (same terms, another order).  `--dtype` reaches the parser, the INI layer and rejects unknown values."""
    from falcon_amd import synth
    from falcon_amd.config import Config
    assert [fo.row_width(x) for x in (1, 64, 65, 200, 256, 257, 400, 401, 504, 800)] == [64, 64, 128, 256, 256, 400, 400, 800, 800, 800]
    with pytest.raises(ValueError):
        fo.row_width(801)
    d = synth.generate(600, seed=3)
    nb, start, _ = fo.get_dim(101.0, 1500.0, 0.05)
    for low_dim in (7, 200, 504):
        W = fo.row_width(low_dim)
        plain = fo.vectorize(d["mz"], d["intensity"], d["indptr"], start, 0.05, nb, low_dim)
        padded = fo.vectorize(d["mz"], d["intensity"], d["indptr"], start, 0.05, nb, low_dim, width=W)
        assert padded.shape == (600, W) and not padded[:, low_dim:].any() and np.array_equal(padded[:, :low_dim], plain)
        if low_dim % 2 == 0:
            np.testing.assert_allclose(fo.sims_f32(padded[:50], padded[:200]), fo.sims_f32(plain[:50], plain[:200]), atol=1e-6, rtol=0)
    c = Config()
    c.parse("in.mgf out --low_dim 504 --dtype f16")
    assert c.low_dim == 504 and c.dtype == "f16"
    c.parse("in.mgf out")
    assert c.low_dim == 400 and c.dtype == "f32"
    ini = tmp_path / "c.ini"
    ini.write_text("low_dim = 200\ndtype = f16\n")
    c.parse(f"in.mgf out -c {ini}")
    assert c.low_dim == 200 and c.dtype == "f16"
    with pytest.raises(SystemExit):
        c.parse("in.mgf out --dtype bf16")
    for bad in ("0", "801"):
        with pytest.raises(ValueError):
            c.parse(f"in.mgf out --low_dim {bad}")


def test_take_rows_is_the_host_side_csr_gather_of_a_rank():
    """a rank of a multi-GPU job uploads the peaks of ITS windows only -- `SpectrumDataset.take_rows` gathers them on
    the host in the order asked for (any order, repeats allowed, empty selection)"""
    from falcon_amd import synth
    from falcon_amd.cluster.cluster import SpectrumDataset
    d = synth.generate(500, seed=8)
    ds = SpectrumDataset(d["precursor_mz"], d["retention_time"], d["mz"], d["intensity"], d["indptr"])
    assert ds.on_host()
    rows = np.array([499, 3, 3, 120, 0], np.int64)
    sub = ds.take_rows(rows)
    assert len(sub) == 5 and np.array_equal(sub.precursor_mz.numpy(), d["precursor_mz"][rows])
    assert np.array_equal(sub.retention_time.numpy(), d["retention_time"][rows])
    ip = sub.indptr.numpy()
    for i, r in enumerate(rows):
        a, b = d["indptr"][r], d["indptr"][r + 1]
        assert np.array_equal(sub.mz.numpy()[ip[i]:ip[i + 1]], d["mz"][a:b])
        assert np.array_equal(sub.intensity.numpy()[ip[i]:ip[i + 1]], d["intensity"][a:b])
    empty = ds.take_rows(np.zeros(0, np.int64))
    assert len(empty) == 0 and empty.mz.numel() == 0 and list(empty.indptr.numpy()) == [0]
