"""`falcon` command line entry point: same contract as reference falcon/falcon.py:33-244
(`main(args) -> int`, console script `falcon = falcon.falcon:main`, setup.cfg:43-45):
read peak files, preprocess, cluster every precursor charge independently through
`cluster.generate_clusters` (the seam the HIP path sits behind), write `<out>.csv`
(+ optional `<out>.mgf` of cluster representatives).

Differences kept deliberately small and listed in DESIGN.md: spectra are held in memory /
`.npz` files in `work_dir` instead of Lance datasets (lance is not available), only MGF input is built
(mzML / mzXML are host XML parsing, out of scope), and `process_spectrum` runs as one batched device call
per peak file (`fal_process_spectra`) instead of a Python loop over spectra.
"""
from __future__ import annotations

import glob
import json
import logging
import os
import re
import shutil
import random
import sys
import tempfile
import threading
from typing import Dict, List, Union

import numpy as np

from . import __version__
from .cluster import cluster, spectrum
from .config import config
from .ms_io import ms_io

logger = logging.getLogger("falcon")

random.seed(42)            # the reference seeds `random` and NumPy with 42 on import (falcon/seed.py, falcon.py:30);
np.random.seed(42)         # the device path itself draws no random numbers


def _natural_key(s: str):
    """natural sort key (natsort is absent): digit runs compare as numbers (falcon.py:206-208)."""
    return [(0, int(t)) if t.isdigit() else (1, t) for t in re.split(r"(\d+)", str(s))]


def main(args: Union[str, List[str], None] = None) -> int:
    logging.captureWarnings(True)
    root = logging.getLogger()
    root.setLevel(logging.DEBUG)
    handler = logging.StreamHandler(sys.stderr)
    handler.setLevel(logging.DEBUG)
    handler.setFormatter(logging.Formatter(
        "{asctime} {levelname} [{name}/{processName}] {module}.{funcName} : {message}", style="{"))
    root.addHandler(handler)
    try:
        return _run(args)
    finally:
        root.removeHandler(handler)


def _option_lines() -> List[str]:
    c = config
    return [
        f"work_dir = {c.work_dir}", f"overwrite = {c.overwrite}",
        f"export_representatives = {c.export_representatives}",
        f"precursor_tol = {c.precursor_tol[0]:.2f} {c.precursor_tol[1]}", f"rt_tol = {c.rt_tol}",
        f"fragment_tol = {c.fragment_tol:.2f}", f"linkage = {c.linkage}",
        f"distance_threshold = {c.distance_threshold:.3f}", f"min_matched_peaks = {c.min_matched_peaks}",
        f"batch_size = {c.batch_size}", f"min_peaks = {c.min_peaks}", f"min_mz_range = {c.min_mz_range:.2f}",
        f"min_mz = {c.min_mz:.2f}", f"max_mz = {c.max_mz:.2f}",
        f"remove_precursor_tol = {c.remove_precursor_tol:.2f}", f"min_intensity = {c.min_intensity:.2f}",
        f"max_peaks_used = {c.max_peaks_used}", f"scaling = {c.scaling}",
        # nearest-neighbour options (README.md:101-117)
        f"eps = {c.eps:.3f}", f"n_probe = {c.n_probe}", f"n_neighbors = {c.n_neighbors}",
        f"n_neighbors_ann = {c.n_neighbors_ann}", f"low_dim = {c.low_dim}", f"mz_interval = {c.mz_interval}",
        f"rescore = {c.rescore}", f"clustering = {c.clustering}", f"dtype = {c.dtype}",
    ]


def _run(args) -> int:
    config.parse(args)
    logger.info("falcon version %s", str(__version__))
    for line in _option_lines():
        logger.debug(line)

    rm_work_dir = False
    if config.work_dir is None:
        config.work_dir = tempfile.mkdtemp()
        rm_work_dir = True
    elif os.path.isdir(config.work_dir):
        logging.warning("Working directory %s already exists, previous results might get overwritten",
                        config.work_dir)
    spectra_dir = os.path.join(config.work_dir, "spectra")
    os.makedirs(spectra_dir, exist_ok=True)

    # falcon.py:86-122: refuse to clobber existing outputs unless --overwrite
    exit_exists = False
    for ext, what in ((".csv", "cluster assignments"), (".mgf", "cluster representatives")):
        fn = f"{config.output_filename}{ext}"
        if os.path.isfile(fn):
            if config.overwrite:
                logger.warning("Output file %s (%s) already exists and will be overwritten", fn, what)
                os.remove(fn)
            else:
                logger.error("Output file %s (%s) already exists, aborting...", fn, what)
                exit_exists = True
    if exit_exists:
        return 1

    _, min_mz, max_mz = spectrum.get_dim(config.min_mz, config.max_mz, config.fragment_tol)   # falcon.py:124-126
    if config.overwrite:
        for fn in os.listdir(spectra_dir):
            os.remove(os.path.join(spectra_dir, fn))
    pipe = cluster.ClusterPipeline(device=config.device)
    pipe.ctx.plan(0)                 # the kernels' code objects, once per process: not between the kernels of the first charge's pass
    charge_path = os.path.join(spectra_dir, "charges.json")
    if os.path.isfile(charge_path) and not config.overwrite:                                   # falcon.py:143-149
        with open(charge_path) as f:
            charges = json.load(f)
    else:
        charges = _prepare_spectra(spectra_dir, min_mz, max_mz, pipe.ctx)
        with open(charge_path, "w") as f:
            json.dump(charges, f)

    ann = cluster.AnnParams(eps=config.eps, low_dim=config.low_dim, n_probe=config.n_probe,
                            n_neighbors=config.n_neighbors, n_neighbors_ann=config.n_neighbors_ann,
                            mz_interval=config.mz_interval, min_mz=config.min_mz, max_mz=config.max_mz,
                            rescore=config.rescore, clustering=config.clustering, dtype=config.dtype)
    rows_all, current_label, representatives = [], 0, []
    for charge in charges:                                                                     # falcon.py:153
        part = np.load(os.path.join(spectra_dir, f"spectra_charge_{charge}.npz"))     # plain arrays: no pickle
        n = len(part["precursor_mz"])
        if n == 0:
            continue
        ds = cluster.SpectrumDataset(part["precursor_mz"], part["retention_time"], part["mz"], part["intensity"],
                                     part["indptr"])
        labels, medoids = cluster.generate_clusters(
            ds, config.linkage, config.distance_threshold, config.min_matched_peaks, config.precursor_tol[0],
            config.precursor_tol[1], config.rt_tol, config.fragment_tol, config.batch_size, ann=ann, pipeline=pipe)
        labels = labels + current_label                                                        # falcon.py:189-193
        current_label = int(labels.max()) + 1
        for i in range(n):
            # float32 columns keep their own (shortest round-trip) text form, as pandas' to_csv prints them
            rows_all.append((str(part["filename"][i]), str(part["identifier"][i]), charge,
                             np.float32(part["precursor_mz"][i]), np.float32(part["retention_time"][i]), int(labels[i])))
        if config.export_representatives:                                                      # falcon.py:198-203
            ip = part["indptr"]
            for c, m in enumerate(medoids):
                representatives.append({
                    "identifier": str(part["identifier"][m]), "precursor_mz": float(part["precursor_mz"][m]),
                    "precursor_charge": None if charge == "None" else int(charge),
                    "retention_time": float(part["retention_time"][m]), "mz": part["mz"][ip[m]:ip[m + 1]],
                    "intensity": part["intensity"][ip[m]:ip[m + 1]], "cluster": int(labels[m])})

    rows_all.sort(key=lambda r: (_natural_key(r[0]), _natural_key(r[1])))                      # falcon.py:206-208
    n_clusters = len({r[5] for r in rows_all})
    logger.info("Export cluster assignments of %d spectra to %d unique clusters to output file %s",
                len(rows_all), n_clusters, f"{config.output_filename}.csv")
    csv_worker = threading.Thread(target=_write_cluster_info, args=(rows_all,), daemon=True)
    csv_worker.start()
    if config.export_representatives:
        logger.info("Export %d cluster representative spectra to output file %s", len(representatives),
                    f"{config.output_filename}.mgf")
        mgf_worker = threading.Thread(target=ms_io.write_spectra,
                                      args=(f"{config.output_filename}.mgf", representatives), daemon=True)
        mgf_worker.start()
        mgf_worker.join()
    csv_worker.join()
    if rm_work_dir:
        shutil.rmtree(config.work_dir)
    return 0


def _raw_csr(specs):
    """spectra read from one peak file -> raw CSR (peaks sorted by m/z inside every spectrum, which is
    what spectrum_utils does when the reference constructs an MsmsSpectrum)."""
    sizes = np.array([len(s["mz"]) for s in specs], np.int64)
    indptr = np.zeros(len(specs) + 1, np.int64)
    np.cumsum(sizes, out=indptr[1:])
    mz = np.concatenate([np.asarray(s["mz"], np.float64) for s in specs]) if len(specs) else np.zeros(0)
    it = np.concatenate([np.asarray(s["intensity"], np.float32) for s in specs]) if len(specs) else np.zeros(0, np.float32)
    order = np.lexsort((mz, np.repeat(np.arange(len(specs)), sizes)))
    return mz[order], it[order], indptr


def _take_rows(indptr: np.ndarray, rows: np.ndarray):
    """positions of the peaks of `rows` in CSR order, and the CSR offsets of the selection"""
    cnt = indptr[rows + 1] - indptr[rows]
    out = np.zeros(len(rows) + 1, np.int64)
    np.cumsum(cnt, out=out[1:])
    pos = np.repeat(indptr[rows] - out[:-1], cnt) + np.arange(out[-1])
    return pos, out


def _prepare_spectra(spectra_dir: str, min_mz: float, max_mz: float, ctx) -> List[str]:
    """falcon.py:247-328: read every peak file, preprocess (`process_spectrum`, spectrum.py:73-169 -- here one
    `fal_process_spectra` call per file on the GPU), partition by precursor charge, one CSR `.npz` per charge."""
    filenames = [fn for pattern in config.input_filenames for fn in glob.glob(pattern)]
    logger.info("Read spectra from %d peak file(s)", len(filenames))
    parts: Dict[str, Dict[str, list]] = {}
    low_quality = 0
    for fn in filenames:
        fn = os.path.abspath(fn)
        specs = list(ms_io.get_spectra(fn))
        if not specs:
            continue
        mz, it, indptr = _raw_csr(specs)
        pmz = np.array([s["precursor_mz"] for s in specs], np.float64)
        charge = np.array([int(s["precursor_charge"]) if s.get("precursor_charge") else 0 for s in specs], np.int32)
        valid, oip, omz, oit = ctx.process_spectra(
            mz, it, indptr, pmz, charge, config.min_peaks, config.min_mz_range, min_mz, max_mz,
            config.remove_precursor_tol, config.min_intensity, config.max_peaks_used,
            None if config.scaling == "off" else config.scaling)
        valid, oip, omz, oit = valid.cpu().numpy(), oip.cpu().numpy(), omz.cpu().numpy(), oit.cpu().numpy()
        low_quality += int((~valid).sum())
        ident = np.array([str(s["identifier"]) for s in specs], dtype=str)
        rt = np.array([s.get("retention_time", -1) for s in specs], np.float32)
        for z in np.unique(charge[valid]):
            rows = np.flatnonzero(valid & (charge == z))
            pos, off = _take_rows(oip, rows)
            p = parts.setdefault("None" if z == 0 else str(int(z)),
                                 dict(identifier=[], filename=[], precursor_mz=[], retention_time=[], mz=[], intensity=[],
                                      counts=[]))
            p["identifier"].append(ident[rows])
            p["filename"].append(np.array([fn] * len(rows), dtype=str))
            p["precursor_mz"].append(pmz[rows].astype(np.float32))
            p["retention_time"].append(rt[rows])
            p["mz"].append(omz[pos])
            p["intensity"].append(oit[pos])
            p["counts"].append(np.diff(off))
    n_total = 0
    for charge, p in parts.items():
        counts = np.concatenate(p["counts"])
        indptr = np.zeros(len(counts) + 1, np.int64)
        np.cumsum(counts, out=indptr[1:])
        np.savez(os.path.join(spectra_dir, f"spectra_charge_{charge}.npz"),
                 identifier=np.concatenate(p["identifier"]), filename=np.concatenate(p["filename"]),
                 precursor_mz=np.concatenate(p["precursor_mz"]), retention_time=np.concatenate(p["retention_time"]),
                 mz=np.concatenate(p["mz"]).astype(np.float32), intensity=np.concatenate(p["intensity"]).astype(np.float32),
                 indptr=indptr)
        n_total += len(counts)
    logger.info("Read %d spectra from %d peak files", n_total, len(filenames))
    logger.info("Skipped %d low-quality spectra", low_quality)
    return sorted(parts, key=_natural_key)


def _write_cluster_info(rows) -> None:
    """falcon.py:483-524: `#` header block with every option, then the CSV table (pandas `to_csv` conventions:
    minimal quoting with doubled quotes, float32 columns in their shortest round-trip form)."""
    import csv
    with open(f"{config.output_filename}.csv", "a", newline="") as f:
        f.write(f"# falcon version {__version__}\n")
        for line in _option_lines():
            f.write(f"# {line}\n")
        f.write("#\n")
        w = csv.writer(f, quoting=csv.QUOTE_MINIMAL, lineterminator="\n")
        w.writerow(["filename", "spectrum_id", "precursor_charge", "precursor_mz", "retention_time", "cluster"])
        for fn, sid, charge, pmz, rt, lab in rows:
            w.writerow([fn, sid, charge, str(np.float32(pmz)), str(np.float32(rt)), lab])


if __name__ == "__main__":
    sys.exit(main())
