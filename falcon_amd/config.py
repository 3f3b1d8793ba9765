"""Command-line / INI configuration with falcon's option surface.

Mirrors reference falcon/config.py:24-212 (names, defaults, nargs, choices, `config.ini`
default file, `-c/--config`, attribute access, RuntimeError before `parse`) and ADDS the
nearest-neighbour options the README documents but the snapshot's parser lacks
(`--eps` README.md:49,73-79; `--n_probe`, `--n_neighbors`, `--n_neighbors_ann`
README.md:107-113; `--low_dim` README.md:114-117) plus the build's `--dtype`.  configargparse is not available, so
the INI layer is stdlib configparser: values from the file become defaults, the command
line overrides them (the precedence configargparse implements).
"""
from __future__ import annotations

import argparse
import configparser
import os
import shlex
from typing import List, Optional, Union

from . import __version__

_HELP_HEADER = ("falcon: Fast spectrum clustering using nearest neighbor searching\n"
                "=================================================================\n\n"
                f"falcon (MI355X build) version {__version__}\n\n"
                "Reference code website: https://github.com/bittremieux/falcon\n\n")


class Config:
    def __init__(self) -> None:
        p = argparse.ArgumentParser(description=_HELP_HEADER, formatter_class=argparse.RawDescriptionHelpFormatter)
        p.add_argument("-c", "--config", default=None, help="Config file path (default: config.ini if present).")
        # IO  (config.py:52-74)
        p.add_argument("input_filenames", nargs="+", help="Input peak files (supported format here: .MGF).")
        p.add_argument("output_filename", help="Output file name.")
        p.add_argument("--work_dir", default=None, help="Working directory (default: temporary directory).")
        p.add_argument("--overwrite", action="store_true", help="Overwrite existing results.")
        p.add_argument("--export_representatives", action="store_true",
                       help="Export cluster representatives to an MGF file.")
        # CLUSTERING  (config.py:76-124)
        p.add_argument("--precursor_tol", nargs=2, default=[20, "ppm"],
                       help='Precursor tolerance mass and mode (default: 20 ppm). Mode is "ppm" or "Da".')
        p.add_argument("--rt_tol", type=float, default=None, help="Retention time tolerance (default: none).")
        p.add_argument("--fragment_tol", type=float, default=0.05, help="Fragment mass tolerance in m/z.")
        p.add_argument("--linkage", type=str, default="complete", choices=["complete", "single", "average"],
                       help="Linkage of the hierarchical clustering (default: complete). single / average select "
                            "--clustering hierarchical when --clustering is not given (the reference honours them, "
                            "config.py:97-103); together with an explicit --clustering dbscan they are an error.")
        p.add_argument("--clustering", type=str, default=None, choices=["dbscan", "hierarchical"],
                       help="dbscan (README: density clustering of the neighbour graph, default) or hierarchical "
                            "(the snapshot's linkage + cut at the distance threshold, on the re-scored neighbour graph; "
                            "implies --rescore).")
        p.add_argument("--distance_threshold", type=float, default=0.1,
                       help="Cosine distance threshold; alias of --eps (default: 0.1).")
        p.add_argument("--eps", type=float, default=None,
                       help="DBSCAN eps = cosine distance threshold (README); alias of --distance_threshold.")
        p.add_argument("--min_matched_peaks", type=int, default=0,
                       help="(snapshot option, accepted; unused by the nearest-neighbour path)")
        p.add_argument("--batch_size", type=int, default=2 ** 15, help="Maximum precursor-m/z block size.")
        # NEAREST NEIGHBOUR INDEXING  (README.md:101-117)
        p.add_argument("--n_probe", type=int, default=16, help="Maximum number of inverted lists to inspect.")
        p.add_argument("--n_neighbors", type=int, default=64, help="Final number of neighbours per spectrum.")
        p.add_argument("--n_neighbors_ann", type=int, default=128, help="Neighbours retrieved by the ANN search.")
        p.add_argument("--low_dim", type=int, default=400,
                       help="Low-dimensional vector length (README.md:114-117): any integer in [1, 800]. The vectors are "
                            "stored in rows of the next instantiated width (64 / 128 / 256 / 400 / 800), zero behind low_dim.")
        p.add_argument("--dtype", type=str, default="f32", choices=["f32", "f16"],
                       help="Element type of the hashed vectors: f32 (default) or f16 (half the bytes per vector; the "
                            "similarity of two float16 vectors is the float32 chain over their exact float32 images).")
        p.add_argument("--mz_interval", type=float, default=1.0,
                       help="Width in m/z of the precursor windows that bound an index (0 = off).")
        p.add_argument("--rescore", action="store_true",
                       help="Re-score the nearest neighbours with the matched-peak cosine (fragment_tol, "
                            "min_matched_peaks) before clustering.")
        p.add_argument("--device", type=int, default=0, help="HIP device ordinal.")
        # PREPROCESSING  (config.py:126-183)
        p.add_argument("--min_peaks", default=5, type=int)
        p.add_argument("--min_mz_range", default=250.0, type=float)
        p.add_argument("--min_mz", default=101.0, type=float)
        p.add_argument("--max_mz", default=1500.0, type=float)
        p.add_argument("--remove_precursor_tol", default=1.5, type=float)
        p.add_argument("--min_intensity", default=0.01, type=float)
        p.add_argument("--max_peaks_used", default=50, type=int)
        p.add_argument("--scaling", default="off", type=str, choices=["off", "root", "log", "rank"])
        self._parser = p
        self._base_defaults = {a.dest: a.default for a in p._actions if a.dest != "help"}
        self._namespace = None

    # ------------------------------------------------------------------------------------------
    def _ini_defaults(self, path: Optional[str]) -> dict:
        if path is None:
            path = "config.ini" if os.path.isfile("config.ini") else None      # config.py:46
        if path is None:
            return {}
        if not os.path.isfile(path):
            raise FileNotFoundError(f"config file {path} not found")
        with open(path) as f:
            text = f.read()
        cp = configparser.ConfigParser()
        cp.optionxform = str
        try:
            cp.read_string(text)
        except configparser.MissingSectionHeaderError:
            cp.read_string("[falcon]\n" + text)                               # configargparse: no sections
        out = {}
        known = {a.dest: a for a in self._parser._actions}
        for sec in cp.sections():
            for k, v in cp[sec].items():
                k = k.strip().lstrip("-")
                if k not in known:
                    raise ValueError(f"unknown option {k!r} in {path}")
                a = known[k]
                if isinstance(a, argparse._StoreTrueAction):
                    out[k] = v.strip().lower() in ("1", "true", "yes", "on")
                elif a.nargs in (2, "+"):
                    out[k] = shlex.split(v.strip("[] ").replace(",", " "))
                elif a.type is not None:
                    out[k] = a.type(v)
                else:
                    out[k] = v
        return out

    def parse(self, args_str: Union[str, List[str], None] = None) -> None:
        """config.py:187-201: None -> sys.argv; a string is split shell-style."""
        if isinstance(args_str, str):
            args_str = shlex.split(args_str)
        pre = argparse.ArgumentParser(add_help=False)
        pre.add_argument("-c", "--config", default=None)
        known, _ = pre.parse_known_args(args_str)
        defaults = self._ini_defaults(known.config)
        # INI values are defaults of THIS call only: start from the built-in defaults every time (a parser keeps
        # what set_defaults gave it, and main() / parse() are called repeatedly in one process)
        self._parser.set_defaults(**self._base_defaults)
        if defaults:
            self._parser.set_defaults(**defaults)
        ns = vars(self._parser.parse_args(args_str))
        ns["precursor_tol"] = [float(ns["precursor_tol"][0]), str(ns["precursor_tol"][1])]   # config.py:199-201
        if ns["precursor_tol"][1] not in ("ppm", "Da"):
            raise ValueError('precursor_tol mode must be "ppm" or "Da"')
        if ns["eps"] is None:
            ns["eps"] = ns["distance_threshold"]
        else:
            ns["distance_threshold"] = ns["eps"]
        # the reference's --linkage values keep their meaning: a non-default linkage selects the hierarchical clustering;
        # with an explicit --clustering dbscan (which has no linkage) the command line fails here, not per charge later
        if ns["clustering"] is None:
            ns["clustering"] = "hierarchical" if ns["linkage"] != "complete" else "dbscan"
        elif ns["clustering"] == "dbscan" and ns["linkage"] != "complete":
            self._parser.error(f"--linkage {ns['linkage']} needs --clustering hierarchical (DBSCAN has no linkage)")
        if ns["clustering"] == "hierarchical":
            ns["rescore"] = True
        if not 1 <= ns["low_dim"] <= 800:
            raise ValueError("low_dim must be an integer in [1, 800] (README.md:114-117; the widest rows the kernels hold)")
        if ns["n_neighbors_ann"] < ns["n_neighbors"]:
            raise ValueError("n_neighbors_ann should be equal or greater than n_neighbors (README.md:110-113)")
        self._namespace = ns

    def __getattr__(self, option):
        if option.startswith("_"):
            raise AttributeError(option)
        if self._namespace is None:
            raise RuntimeError("The configuration has not been initialized")     # config.py:203-206
        return self._namespace[option]

    def __setattr__(self, key, value):
        if key.startswith("_"):
            object.__setattr__(self, key, value)
        else:
            if self._namespace is None:
                raise RuntimeError("The configuration has not been initialized")
            self._namespace[key] = value

    def __getitem__(self, item):
        return self.__getattr__(item)


config = Config()
