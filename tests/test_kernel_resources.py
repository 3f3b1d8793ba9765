"""Build-time guard for the hot kernels' register allocation (no GPU needed: hipcc cross-compiles).

The cosine kernels keep a 32-row tile of vectors in registers; when a code change makes hipcc demote that
array to scratch memory the kernel still passes every parity test and silently runs several times slower
(it happened to the arg-max form once: k-means 3.5x slower).  So: no scratch in any scan kernel."""
import os
import re
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "falcon_amd", "csrc")
HIPCC = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"


def _kernel_scratch(src, tmp_path):
    out = tmp_path / (os.path.basename(src) + ".s")
    subprocess.run([HIPCC, "-O3", "-std=c++17", "--offload-arch=gfx950", "-ffp-contract=off", "-I" + os.path.join(ROOT, "include"),
                    "--cuda-device-only", "-S", "-o", str(out), src], check=True, capture_output=True, timeout=600)
    res, name = {}, None
    for line in open(out):
        m = re.match(r"\s*\.amdhsa_kernel\s+(\S+)", line)
        if m:
            name = m.group(1)
        m = re.match(r"\s*\.amdhsa_private_segment_fixed_size\s+(\d+)", line)
        if m and name:
            res[name] = int(m.group(1))
    return res


@pytest.mark.skipif(not os.path.exists(HIPCC), reason="hipcc not available")
@pytest.mark.parametrize("src,pattern,expected", [
    ("scan.hip", "dense_kernel", 10),          # DH4 in {8,16,32,50,64} x {store, arg-max}
    ("scan.hip", "dense4_kernel", 4),          # the shared-stream flat scan: DH4 in {8,16,32,50}
    ("scan.hip", "dense_tiny4_kernel", 4),
    ("scan16.hip", "scan16_kernel", 10),
    ("ivf_fine.hip", "ivf_list4_kernel", 5),
    ("assign.hip", "assign_kernel", 5),
    ("assign.hip", "assign_wave_kernel", 5),
])
def test_scan_kernels_use_no_scratch(tmp_path, src, pattern, expected):
    res = {k: v for k, v in _kernel_scratch(os.path.join(CSRC, src), tmp_path).items() if pattern in k}
    assert len(res) == expected, sorted(res)
    spilled = {k: v for k, v in res.items() if v != 0}
    assert not spilled, f"kernels with scratch memory (register tile demoted?): {spilled}"
