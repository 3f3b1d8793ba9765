"""Extension dispatch (reference falcon/ms_io/ms_io.py:11-66); only MGF is built here."""
import os

from . import mgf_io


def get_spectra(filename: str):
    if not os.path.isfile(filename):
        raise ValueError(f"Non-existing peak file {filename}")
    _, ext = os.path.splitext(filename.lower())
    if ext == ".mgf":
        yield from mgf_io.get_spectra(filename)
    elif ext in (".mzml", ".mzxml"):
        raise ValueError(f"{ext} input is outside this build's scope (SURVEY section 2, row 6); convert to MGF")
    else:
        raise ValueError(f'Unknown spectrum file type with extension "{ext}"')


def write_spectra(filename: str, spectra) -> None:
    ext = os.path.splitext(filename.lower())[1]
    if ext != ".mgf":
        raise ValueError("Unsupported output file format (the reference supports only MGF too, ms_io.py:58-66)")
    mgf_io.write_spectra(filename, spectra)
