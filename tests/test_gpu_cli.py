"""`falcon.main()` end to end on the HIP path: MGF in, CSV (+ representatives MGF) out
(reference falcon.py:33-244 contract; BASELINE configs[0] -- 10k synthetic MGF spectra, low_dim 400 -- driven through the GPU)."""
import os

import numpy as np
import pytest

from oracle import falcon_oracle as fo

pytestmark = pytest.mark.gpu


def test_main_mgf_to_csv(tmp_path):
    from falcon_amd import synth
    from falcon_amd.falcon import main
    from falcon_amd.ms_io import ms_io
    d = synth.generate(10000, seed=21)
    specs = []
    for i in range(10000):
        a, b = d["indptr"][i], d["indptr"][i + 1]
        specs.append({"identifier": f"scan={i}", "precursor_mz": float(d["precursor_mz"][i]),
                      "precursor_charge": int(d["precursor_charge"][i]), "retention_time": float(d["retention_time"][i]),
                      "mz": d["mz"][a:b].astype(np.float64), "intensity": d["intensity"][a:b]})
    mgf = str(tmp_path / "in.mgf")
    ms_io.write_spectra(mgf, specs)
    out = str(tmp_path / "res")
    args = [mgf, out, "--eps", "0.3", "--export_representatives", "--remove_precursor_tol", "0.0",
            "--min_intensity", "0.0", "--work_dir", str(tmp_path / "work")]
    assert main(args) == 0
    lines = open(out + ".csv").read().splitlines()
    head = [l for l in lines if l.startswith("#")]
    assert head[0].startswith("# falcon version") and "# eps = 0.300" in head and "# n_probe = 16" in head
    body = lines[len(head):]
    assert body[0] == "filename,spectrum_id,precursor_charge,precursor_mz,retention_time,cluster"
    rows = [l.split(",") for l in body[1:]]
    assert len(rows) == 10000
    ids = [r[1] for r in rows]
    assert ids[:3] == ["scan=0", "scan=1", "scan=2"] and ids[10] == "scan=10"      # natural sort
    lab = np.array([int(r[5]) for r in rows])
    charge = np.array([int(r[2]) for r in rows])
    assert set(np.unique(charge)) == {2, 3}
    # labels of different charges never overlap (falcon.py:189-193) and are dense
    assert lab[charge == 2].max() < lab[charge == 3].min()
    assert np.array_equal(np.unique(lab), np.arange(lab.max() + 1))
    # same clustering as the oracle, per charge
    from sklearn.metrics import adjusted_rand_score
    import warnings
    for c in (2, 3):
        sel = synth.select_charge(d, c)
        ref, _ = fo.generate_clusters(sel["mz"], sel["intensity"], sel["indptr"], sel["precursor_mz"], None, eps=0.3)
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            assert adjusted_rand_score(ref, lab[sel["rows"]]) >= 0.99
    reps = list(ms_io.get_spectra(out + ".mgf"))
    assert len(reps) == lab.max() + 1
    # outputs exist -> a second run without --overwrite aborts with 1 (falcon.py:120-122)
    assert main(args) == 1
    assert main(args + ["--overwrite"]) == 0


def test_config1_10k_mgf_at_eps_010_labels_equal_the_oracle(tmp_path):
    """BASELINE configs[0] as written -- 10k synthetic MGF spectra, low_dim 400, eps = 0.10 -- through `main()` with the
    default preprocessing: the CSV's cluster column must EQUAL the oracle's labels (not ARI-close) on the arrays main()
    clustered (work_dir's per-charge files, whose own parity is test_main_preprocesses_raw_spectra_on_the_device's subject),
    with the per-charge label offsets of falcon.py:189-193."""
    from falcon_amd import synth
    from falcon_amd.falcon import main
    from falcon_amd.ms_io import ms_io
    d = synth.generate(10000, seed=5)
    specs = []
    for i in range(10000):
        a, b = d["indptr"][i], d["indptr"][i + 1]
        specs.append({"identifier": f"scan={i}", "precursor_mz": float(d["precursor_mz"][i]),
                      "precursor_charge": int(d["precursor_charge"][i]), "retention_time": float(d["retention_time"][i]),
                      "mz": d["mz"][a:b].astype(np.float64), "intensity": d["intensity"][a:b]})
    mgf = str(tmp_path / "in.mgf")
    ms_io.write_spectra(mgf, specs)
    out, work = str(tmp_path / "res"), tmp_path / "work"
    assert main([mgf, out, "--eps", "0.1", "--low_dim", "400", "--work_dir", str(work)]) == 0
    lines = [l for l in open(out + ".csv").read().splitlines() if not l.startswith("#")]
    table = {r[1]: (int(r[2]), int(r[5])) for r in (l.split(",") for l in lines[1:])}
    offset, n_seen = 0, 0
    for charge in (2, 3):
        z = np.load(work / "spectra" / f"spectra_charge_{charge}.npz")
        ref, rmed = fo.generate_clusters(z["mz"], z["intensity"], z["indptr"], z["precursor_mz"], z["retention_time"], eps=0.1,
                                         low_dim=400)
        got = np.array([table[str(i)][1] for i in z["identifier"]])
        assert all(table[str(i)][0] == charge for i in z["identifier"][:50])
        assert np.array_equal(got - offset, ref), (charge, int((got - offset != ref).sum()))
        offset += len(rmed)
        n_seen += len(ref)
        assert (np.bincount(ref) > 1).sum() > 100                      # a real clustering at the config's eps
    assert n_seen == len(table)


def test_main_preprocesses_raw_spectra_on_the_device(tmp_path):
    """raw MGF peaks (outside the m/z window, on the precursor ions, below 1 % of the base peak, > 50 peaks) go
    through `fal_process_spectra` inside main(); what lands in work_dir equals the host `process_spectrum`."""
    from falcon_amd.cluster.spectrum import get_dim, process_spectrum
    from falcon_amd.falcon import main
    from falcon_amd.ms_io import ms_io
    from tests.prep_cases import raw_spectra
    mz, it, indptr, pmz, ch = raw_spectra(400, 9, max_peaks=300)
    ch[ch == 0] = 2                                         # (the MGF writer needs a charge)
    specs = [{"identifier": f"s{i}", "precursor_mz": float(pmz[i]), "precursor_charge": int(ch[i]),
              "retention_time": float(i), "mz": mz[indptr[i]:indptr[i + 1]], "intensity": it[indptr[i]:indptr[i + 1]]}
             for i in range(400) if indptr[i + 1] > indptr[i]]
    mgf = str(tmp_path / "raw.mgf")
    ms_io.write_spectra(mgf, specs)
    work = tmp_path / "work"
    assert main([mgf, str(tmp_path / "out"), "--work_dir", str(work), "--scaling", "root"]) == 0
    _, min_mz, max_mz = get_dim(101.0, 1500.0, 0.05)
    expected = {}
    for s in ms_io.get_spectra(mgf):                        # what the reader hands over (text round trip included)
        out = process_spectrum(dict(s), 5, 250.0, min_mz, max_mz, 1.5, 0.01, 50, "root")
        if out is not None:
            expected[(str(out["precursor_charge"]), out["identifier"])] = out
    seen = 0
    for fn in os.listdir(work / "spectra"):
        if not fn.endswith(".npz"):
            continue
        part = np.load(work / "spectra" / fn, allow_pickle=True)
        charge = fn[len("spectra_charge_"):-4]
        for r, ident in enumerate(part["identifier"]):
            e = expected[(charge, str(ident))]
            a, b = part["indptr"][r], part["indptr"][r + 1]
            assert np.array_equal(part["mz"][a:b], e["mz"])
            np.testing.assert_allclose(part["intensity"][a:b], e["intensity"], rtol=3e-6)
            seen += 1
    assert seen == len(expected) and 50 < seen < len(specs)


def test_main_with_rescore_option(tmp_path):
    """`--rescore`: the neighbours are re-scored with the matched-peak cosine (similarity.py:17-80) using
    --fragment_tol / --min_matched_peaks before DBSCAN; main() runs through and records the option."""
    from falcon_amd import synth
    from falcon_amd.falcon import main
    from falcon_amd.ms_io import ms_io
    d = synth.generate(1500, seed=4)
    specs = [{"identifier": f"scan={i}", "precursor_mz": float(d["precursor_mz"][i]),
              "precursor_charge": int(d["precursor_charge"][i]), "retention_time": float(d["retention_time"][i]),
              "mz": d["mz"][d["indptr"][i]:d["indptr"][i + 1]].astype(np.float64),
              "intensity": d["intensity"][d["indptr"][i]:d["indptr"][i + 1]]} for i in range(1500)]
    mgf = str(tmp_path / "in.mgf")
    ms_io.write_spectra(mgf, specs)
    outs = {}
    for name, extra in (("plain", []), ("rescored", ["--rescore", "--min_matched_peaks", "4"])):
        out = str(tmp_path / name)
        assert main([mgf, out, "--eps", "0.3", "--remove_precursor_tol", "0.0", "--min_intensity", "0.0",
                     "--work_dir", str(tmp_path / ("w_" + name))] + extra) == 0
        lines = open(out + ".csv").read().splitlines()
        assert f"# rescore = {name == 'rescored'}" in lines
        body = [l for l in lines if not l.startswith("#")][1:]
        outs[name] = np.array([int(l.split(",")[5]) for l in body])
    for lab in outs.values():
        assert len(lab) == 1500 and np.array_equal(np.unique(lab), np.arange(lab.max() + 1))
    assert len(np.unique(outs["rescored"])) != len(np.unique(outs["plain"])) or not np.array_equal(outs["rescored"], outs["plain"])


def test_main_hierarchical_clustering_option(tmp_path):
    """`--clustering hierarchical --linkage average` (the snapshot's own clustering, cluster.py:283-290, on the re-scored graph)
    through main(): same clustering as the oracle's run of those stages.  `--linkage average` alone selects the hierarchical
    clustering like the reference's command line does; with an explicit `--clustering dbscan` it is refused at parse time."""
    from falcon_amd import synth
    from falcon_amd.falcon import main
    from falcon_amd.ms_io import ms_io
    d = synth.generate(3000, seed=33, mz_lo=500.0, mz_hi=515.0)
    specs = []
    for i in range(3000):
        a, b = d["indptr"][i], d["indptr"][i + 1]
        specs.append({"identifier": f"scan={i}", "precursor_mz": float(d["precursor_mz"][i]),
                      "precursor_charge": int(d["precursor_charge"][i]), "retention_time": float(d["retention_time"][i]),
                      "mz": d["mz"][a:b].astype(np.float64), "intensity": d["intensity"][a:b]})
    mgf = str(tmp_path / "in.mgf")
    ms_io.write_spectra(mgf, specs)
    out = str(tmp_path / "res")
    base = [mgf, out, "--eps", "0.35", "--remove_precursor_tol", "0.0", "--min_intensity", "0.0", "--min_matched_peaks", "4",
            "--work_dir", str(tmp_path / "work")]
    with pytest.raises(SystemExit):
        main(base + ["--clustering", "dbscan", "--linkage", "average"])
    assert main(base + ["--linkage", "average"]) == 0
    implied = open(out + ".csv").read().splitlines()
    assert main(base + ["--overwrite", "--clustering", "hierarchical", "--linkage", "average"]) == 0
    lines = open(out + ".csv").read().splitlines()
    assert "# clustering = hierarchical" in lines and "# linkage = average" in lines
    assert [l for l in implied if not l.startswith("#")] == [l for l in lines if not l.startswith("#")]
    body = [l.split(",") for l in lines if not l.startswith("#")][1:]
    lab = np.array([int(r[5]) for r in body])
    charge = np.array([int(r[2]) for r in body])
    for c in (2, 3):
        sel = synth.select_charge(d, c)
        ref, _ = fo.generate_clusters(sel["mz"], sel["intensity"], sel["indptr"], sel["precursor_mz"], None, eps=0.35,
                                      clustering="hierarchical", linkage="average", min_matches=4)
        got = lab[sel["rows"]]
        pairs = np.unique(np.stack([ref, got]), axis=1)
        assert pairs.shape[1] == len(np.unique(ref)) == len(np.unique(got))          # the same partition
    assert (np.bincount(lab) > 1).sum() > 30


@pytest.mark.parametrize("extra", [["--low_dim", "200"], ["--low_dim", "504"], ["--low_dim", "800"],
                                   ["--low_dim", "800", "--dtype", "f16"], ["--low_dim", "333", "--dtype", "f16"]])
def test_main_at_any_low_dim_and_dtype_labels_equal_the_oracle(tmp_path, extra):
    """`--low_dim` is a free integer in the reference (README.md:114-117) and `main()` is the drop-in: low_dim 200 (rows of 256
    columns), 504 and 800 in float32 (rows of 800 columns: the two-K-half passes of the fp32 matrix kernels), and BASELINE
    configs[4]'s low_dim 800 float16 through `--dtype f16` -- the CSV's cluster column EQUALS the oracle's labels."""
    from falcon_amd import synth
    from falcon_amd.falcon import main
    from falcon_amd.ms_io import ms_io
    # a dense stretch (flat buckets of ~150 rows: the matrix kernels) + a sparse one (buckets of a few rows: exact chains)
    parts = [synth.generate(4000, seed=6, mz_lo=500.0, mz_hi=520.0), synth.generate(1500, seed=7, mz_lo=700.0, mz_hi=900.0)]
    specs = []
    for k, d in enumerate(parts):
        for i in range(len(d["precursor_mz"])):
            a, b = d["indptr"][i], d["indptr"][i + 1]
            specs.append({"identifier": f"scan={len(specs)}", "precursor_mz": float(d["precursor_mz"][i]),
                          "precursor_charge": int(d["precursor_charge"][i]), "retention_time": float(d["retention_time"][i]),
                          "mz": d["mz"][a:b].astype(np.float64), "intensity": d["intensity"][a:b]})
    mgf = str(tmp_path / "in.mgf")
    ms_io.write_spectra(mgf, specs)
    out, work = str(tmp_path / "res"), tmp_path / "work"
    assert main([mgf, out, "--eps", "0.1", "--work_dir", str(work)] + extra) == 0
    opts = dict(zip(extra[::2], extra[1::2]))
    low_dim, f16 = int(opts["--low_dim"]), opts.get("--dtype") == "f16"
    lines = open(out + ".csv").read().splitlines()
    assert f"# low_dim = {low_dim}" in lines and f"# dtype = {'f16' if f16 else 'f32'}" in lines
    body = [l for l in lines if not l.startswith("#")]
    table = {r[1]: (int(r[2]), int(r[5])) for r in (l.split(",") for l in body[1:])}
    offset, n_seen = 0, 0
    for charge in (2, 3):
        z = np.load(work / "spectra" / f"spectra_charge_{charge}.npz")
        ref, rmed = fo.generate_clusters(z["mz"], z["intensity"], z["indptr"], z["precursor_mz"], z["retention_time"], eps=0.1,
                                         low_dim=low_dim, dtype=np.float16 if f16 else np.float32)
        got = np.array([table[str(i)][1] for i in z["identifier"]])
        assert np.array_equal(got - offset, ref), (charge, int((got - offset != ref).sum()))
        offset += len(rmed)
        n_seen += len(ref)
        assert (np.bincount(ref) > 1).sum() > 50
    assert n_seen == len(table)
