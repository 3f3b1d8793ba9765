"""ctypes binding of libfalcon_hip.so (the C ABI declared in include/falcon_hip.h).

There is no CPU fallback: if the shared library is missing, or no gfx950 device is
visible, every entry point raises.
"""
from __future__ import annotations

import ctypes as C
import os
from typing import Optional

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libfalcon_hip.so")

FAL_DTYPE_F32, FAL_DTYPE_F16, FAL_DTYPE_SPLIT16, FAL_OUT_F32_F16, FAL_OUT_F16_IMAGE = 0, 1, 2, 3, 4
STAGES = {"vectorize": 0, "build": 1, "coarse": 2, "scan": 3, "select": 4, "filter": 5, "dbscan": 6, "tail": 7,
          "kernel": 8}     # the cosine kernel's own launches (subset of "scan")


class FalconHipError(RuntimeError):
    pass


_lib: Optional[C.CDLL] = None

c_void_p, c_int, c_int64, c_uint32, c_float, c_double = C.c_void_p, C.c_int, C.c_int64, C.c_uint32, C.c_float, C.c_double
P = C.POINTER

_SIGNATURES = {
    "fal_version": ([], c_int),
    "fal_last_error": ([], C.c_char_p),
    "fal_device_count": ([P(c_int)], c_int),
    "fal_ctx_create": ([c_int, c_void_p, c_int, P(c_void_p)], c_int),
    "fal_ctx_destroy": ([c_void_p], c_int),
    "fal_ctx_sync": ([c_void_p], c_int),
    "fal_ctx_plan": ([c_void_p, c_int64, c_int, c_int, c_int, c_int64], c_int),
    "fal_ctx_trim": ([c_void_p], c_int),
    "fal_ctx_stage_ms": ([c_void_p, c_int, P(c_float), P(c_int64)], c_int),
    "fal_ctx_enable_timing": ([c_void_p, c_int], c_int),
    "fal_ctx_counter": ([c_void_p, c_int, P(c_int64)], c_int),
    "fal_get_dim": ([c_float, c_float, c_float, P(c_uint32), P(c_float), P(c_float)], c_int),
    "fal_hash_lookup": ([c_uint32, c_uint32, c_uint32, c_void_p], c_int),
    "fal_to_vector_indices": ([c_void_p, c_void_p, c_int64, c_double, c_double, c_void_p], c_int),
    "fal_vectorize": ([c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int64, c_double, c_double,
                       c_uint32, c_uint32, c_uint32, c_int, c_int, c_void_p], c_int),
    "fal_vectorize_pair": ([c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int64, c_double, c_double,
                            c_uint32, c_uint32, c_uint32, c_int, c_void_p, c_void_p], c_int),
    "fal_vectorize_f16_image": ([c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int64, c_double, c_double,
                                 c_uint32, c_uint32, c_uint32, c_int, c_void_p, c_void_p], c_int),
    "fal_row_width": ([c_uint32, P(c_uint32)], c_int),
    "fal_vectorize_rows": ([c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int64, c_double, c_double,
                            c_uint32, c_uint32, c_uint32, c_uint32, c_int, c_int, c_void_p, c_void_p], c_int),
    "fal_window_counts": ([c_void_p, c_void_p, c_void_p, c_int, c_double, c_int64, c_void_p, c_void_p], c_int),
    "fal_window_select": ([c_void_p, c_void_p, c_int64, c_double, c_int64, c_void_p, c_int, c_void_p, c_void_p, P(c_int64)], c_int),
    "fal_precursor_splits": ([c_void_p, c_void_p, c_int64, c_double, c_int, c_int64, c_double, c_int,
                              c_void_p, c_int64, P(c_int64)], c_int),
    "fal_ivf_build": ([c_void_p, c_void_p, c_int64, c_int, c_void_p, c_int64, c_void_p, c_int, P(c_void_p)], c_int),
    "fal_ivf_build_x16": ([c_void_p, c_void_p, c_void_p, c_int64, c_int, c_void_p, c_int64, c_void_p, c_int, P(c_void_p)], c_int),
    "fal_ivf_attach_f16": ([c_void_p, c_void_p, c_int], c_int),
    "fal_ivf_attach_prefilter": ([c_void_p, c_void_p], c_int),
    "fal_ivf_attach_prefilter_ex": ([c_void_p, c_void_p, c_int], c_int),
    "fal_ivf_destroy": ([c_void_p], c_int),
    "fal_ivf_total_lists": ([c_void_p, P(c_int64)], c_int),
    "fal_ivf_export": ([c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p], c_int),
    "fal_ivf_search_topk": ([c_void_p, c_void_p, c_int, c_int, c_void_p, c_void_p], c_int),
    "fal_filter_neighbors": ([c_void_p, c_void_p, c_void_p, c_int64, c_int, c_void_p, c_void_p, c_double, c_int,
                              c_double, c_int, c_void_p, c_void_p], c_int),
    "fal_ivf_search_neighbors": ([c_void_p, c_void_p, c_int, c_int, c_void_p, c_void_p, c_double, c_int, c_double, c_int,
                                  c_void_p, c_void_p, c_void_p], c_int),
    "fal_process_spectra": ([c_void_p, c_void_p, c_void_p, c_void_p, c_int64, c_int64, c_void_p, c_void_p, c_int, c_double,
                             c_double, c_double, c_double, c_double, c_int, c_int, c_void_p, c_void_p, c_void_p,
                             c_void_p], c_int),
    "fal_rescore_neighbors": ([c_void_p, c_void_p, c_void_p, c_int64, c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_double,
                               c_int], c_int),
    "fal_neighbors_to_csr": ([c_void_p, c_void_p, c_void_p, c_void_p, c_int64, c_int, c_int64, c_int64, c_void_p, c_void_p, c_void_p],
                             c_int),
    "fal_neighbors_to_csr_mapped": ([c_void_p, c_void_p, c_void_p, c_void_p, c_int64, c_int, c_void_p, c_int64, c_int64, c_void_p,
                                     c_void_p, c_void_p], c_int),
    "fal_dbscan": ([c_void_p, c_void_p, c_void_p, c_int64, c_int, c_float, c_void_p, P(c_int64)], c_int),
    "fal_refine_clusters": ([c_void_p, c_void_p, c_int64, c_void_p, c_void_p, c_double, c_int, c_double,
                             P(c_int64)], c_int),
    "fal_finalize": ([c_void_p, c_void_p, c_int64, c_int64, c_void_p, c_void_p, c_void_p, c_int, c_void_p,
                      c_void_p, P(c_int64)], c_int),
    "fal_cluster_graph": ([c_void_p, c_void_p, c_void_p, c_int64, c_int, c_float, c_void_p, c_void_p, c_double, c_int,
                           c_double, c_void_p, c_void_p, c_void_p, c_void_p, P(c_int64), P(c_int64)], c_int),
    "fal_cluster_graph_counted": ([c_void_p, c_void_p, c_void_p, c_void_p, c_int64, c_int, c_float, c_void_p, c_void_p, c_double,
                                   c_int, c_double, c_void_p, c_void_p, c_void_p, c_void_p, P(c_int64), P(c_int64)], c_int),
    "fal_cluster_graph_linkage": ([c_void_p, c_void_p, c_void_p, c_int64, c_int, c_float, c_int, c_void_p, c_void_p, c_double, c_int,
                                   c_double, c_void_p, c_void_p, c_void_p, c_void_p, P(c_int64), P(c_int64)], c_int),
    "fal_linkage_cluster": ([c_void_p, c_void_p, c_void_p, c_int64, c_int, c_float, c_int, c_void_p, P(c_int64)], c_int),
    "fal_sort_by_precursor": ([c_void_p, c_void_p, c_int64, c_void_p, c_void_p], c_int),
    "fal_gather_f32": ([c_void_p, c_void_p, c_void_p, c_int64, c_void_p], c_int),
}


def load() -> C.CDLL:
    """Load the HIP library; raise loudly when it has not been built."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.isfile(LIB_PATH):
        raise FalconHipError(
            f"{LIB_PATH} is missing: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
            f"(or `make -C falcon_amd/csrc`). falcon_amd has no CPU fallback.")
    # torch bundles its own libamdhip64.so.7 (same soname as /opt/rocm's).  Whichever is loaded
    # first serves the whole process, and a process that mixes the two HIP runtimes loses its
    # device; torch owns the device memory and streams we use, so its runtime goes first.
    import torch  # noqa: F401
    lib = C.CDLL(LIB_PATH)
    for name, (args, res) in _SIGNATURES.items():
        fn = getattr(lib, name)          # AttributeError if the .so is stale
        fn.argtypes = args
        fn.restype = res
    _lib = lib
    return lib


def check(rc: int, what: str = "") -> None:
    if rc != 0:
        msg = load().fal_last_error().decode("utf-8", "replace")
        raise FalconHipError(f"{what or 'libfalcon_hip'} failed with code {rc}: {msg}")


def exported_symbols():
    return list(_SIGNATURES)
