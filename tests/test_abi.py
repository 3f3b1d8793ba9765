"""The C-ABI library loads, exports every symbol include/falcon_hip.h declares, and fails
loudly (never falls back to a CPU path) when no gfx950 device is present."""
import ctypes as C
import os
import re

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared():
    txt = open(os.path.join(ROOT, "include", "falcon_hip.h")).read()
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    return sorted(set(re.findall(r"\b(fal_[a-z0-9_]+)\s*\(", txt)))


def test_every_declared_symbol_is_exported_and_bound():
    from falcon_amd import _lib
    lib = _lib.load()
    declared = _declared()
    assert len(declared) >= 25
    for name in declared:
        assert hasattr(lib, name), f"{name} declared in falcon_hip.h but not exported"
    assert sorted(_lib.exported_symbols()) == declared, "ctypes signature table out of sync with the header"
    assert lib.fal_version() >= 100


def test_no_cpu_fallback():
    """Without a HIP device the context cannot be created (FAL_ENODEV) and the Python layer raises."""
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    from falcon_amd import _lib
    from falcon_amd.device import Context
    lib = _lib.load()
    h = C.c_void_p()
    rc = lib.fal_ctx_create(0, None, 1, C.byref(h))
    assert rc == -4 and not h.value
    assert b"no HIP device" in lib.fal_last_error()
    with pytest.raises(_lib.FalconHipError):
        Context(0)
    from falcon_amd.cluster.cluster import ClusterPipeline
    with pytest.raises(_lib.FalconHipError):
        ClusterPipeline()


def test_host_entry_points(ref_golden):
    """a1 get_dim and the a3 hash table are host-side entry points: bit-exact vs the reference /
    sklearn goldens without any GPU."""
    from falcon_amd.device import get_dim, hash_lookup
    g = ref_golden
    for (lo, hi, b), dim, (s, e) in zip(g["get_dim_in"], g["get_dim_dim"], g["get_dim_start_end"]):
        d, start, end = get_dim(lo, hi, b)
        assert d == dim and np.float32(start) == s and np.float32(end) == e
    assert np.array_equal(hash_lookup(27982, 400), g["tv_hash_lookup_400"])
    assert list(hash_lookup(10, 400)) == [254, 218, 63, 305, 375, 94, 302, 321, 321, 131]


def test_bad_arguments_return_codes():
    from falcon_amd import _lib
    lib = _lib.load()
    d, s, e = C.c_uint32(), C.c_float(), C.c_float()
    assert lib.fal_get_dim(101.0, 1500.0, 0.0, C.byref(d), C.byref(s), C.byref(e)) == -1
    assert b"bin_size" in lib.fal_last_error()
    assert lib.fal_ctx_sync(None) == -1
    assert lib.fal_ctx_destroy(None) == 0
