"""Peak-file front door: pick the reader / writer by file extension (the role of the reference's
falcon/ms_io/ms_io.py:11-66).  Only MGF is built in this tree; mzML / mzXML are host-side XML parsing and
outside the hot-path scope."""
import os

from . import mgf_io

_READERS = {".mgf": mgf_io.get_spectra}
_WRITERS = {".mgf": mgf_io.write_spectra}
_KNOWN_BUT_UNBUILT = {".mzml", ".mzxml"}


def _extension(path: str) -> str:
    return os.path.splitext(path.lower())[1]


def get_spectra(filename: str):
    """Iterate over the spectra of a peak file as plain dicts (see mgf_io.get_spectra)."""
    if not os.path.isfile(filename):
        raise ValueError(f"Non-existing peak file {filename}")
    ext = _extension(filename)
    reader = _READERS.get(ext)
    if reader is None:
        if ext in _KNOWN_BUT_UNBUILT:
            raise ValueError(f"{ext} input is outside this build's scope (SURVEY section 2, row 6); convert to MGF")
        raise ValueError(f'Unknown spectrum file type with extension "{ext}"')
    yield from reader(filename)


def write_spectra(filename: str, spectra) -> None:
    """Write spectra (dicts) to a peak file; like the reference (ms_io.py:58-66) only MGF can be written."""
    writer = _WRITERS.get(_extension(filename))
    if writer is None:
        raise ValueError("Unsupported output file format (only MGF can be written)")
    writer(filename, spectra)
