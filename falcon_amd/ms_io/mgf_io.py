"""MGF reading / writing without pyteomics (absent here).

Fields as the reference reads them (falcon/ms_io/mgf_io.py:33-66): TITLE, PEPMASS (first
token), CHARGE (optional; "2+", "3+", "2" ...), RTINSECONDS (default -1), then the peak
list.  Malformed spectra are skipped silently, as mgf_io.py:27-30 does.  Writing follows
mgf_io.py:85-116 (TITLE, PEPMASS, CHARGE, RTINSECONDS, peaks).
"""
from __future__ import annotations

from typing import Dict, Iterable, Iterator

import numpy as np


def _parse_charge(txt: str):
    t = txt.strip().split()[0].split(",")[0].split("and")[0].strip()
    sign = -1 if t.endswith("-") else 1
    t = t.rstrip("+-")
    return sign * int(t)


def get_spectra(source) -> Iterator[Dict]:
    """Yield dicts: identifier, precursor_mz, precursor_charge (int or None), retention_time,
    mz f64[], intensity f32[]."""
    close = False
    if isinstance(source, str):
        f, close = open(source, "r"), True
    else:
        f = source
    try:
        params, mzs, its, inside = {}, [], [], False
        for line in f:
            line = line.strip()
            if not line or line[0] in "#;!/":
                continue
            if line == "BEGIN IONS":
                params, mzs, its, inside = {}, [], [], True
            elif line == "END IONS":
                if inside:
                    try:
                        if "__bad__" in params:          # a peak line that did not parse: the spectrum is skipped
                            raise ValueError("malformed peak line")
                        yield {
                            "identifier": params["title"],
                            "precursor_mz": float(params["pepmass"].split()[0]),
                            "precursor_charge": _parse_charge(params["charge"]) if "charge" in params else None,
                            "retention_time": float(params.get("rtinseconds", -1)),
                            "mz": np.asarray(mzs, np.float64),
                            "intensity": np.asarray(its, np.float32),
                        }
                    except (ValueError, KeyError, IndexError):
                        pass
                inside = False
            elif inside:
                if "=" in line and not (line[0].isdigit() or line[0] == "."):
                    k, v = line.split("=", 1)
                    params[k.strip().lower()] = v.strip()
                else:
                    tok = line.split()
                    try:
                        mzs.append(float(tok[0]))
                        its.append(float(tok[1]) if len(tok) > 1 else 0.0)
                    except ValueError:
                        params["__bad__"] = True
    finally:
        if close:
            f.close()


def write_spectra(filename: str, spectra: Iterable[Dict]) -> None:
    with open(filename, "w") as out:
        for s in spectra:
            out.write("BEGIN IONS\n")
            out.write(f"TITLE={s['identifier']}\n")
            out.write(f"PEPMASS={s['precursor_mz']}\n")
            ch = s.get("precursor_charge")
            if ch is not None and not (isinstance(ch, float) and np.isnan(ch)):
                ch = int(ch)
                out.write(f"CHARGE={abs(ch)}{'-' if ch < 0 else '+'}\n")
            if s.get("retention_time") is not None:
                out.write(f"RTINSECONDS={s['retention_time']}\n")
            if "cluster" in s:
                out.write(f"CLUSTER={s['cluster']}\n")
            for m, i in zip(s["mz"], s["intensity"]):
                out.write(f"{m} {i}\n")
            out.write("END IONS\n\n")
