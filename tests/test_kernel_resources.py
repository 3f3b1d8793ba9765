"""Build-time guards on the code hipcc generates for the hot kernels (no GPU needed: hipcc cross-compiles).

1. Register allocation: the cosine kernels keep a 32-row tile of vectors in registers; when a code change makes hipcc
   demote that array to scratch memory the kernel still passes every parity test and silently runs several times
   slower (it happened to the arg-max form once: k-means 3.5x slower).  So: no scratch in any scan kernel.
2. LDS-DMA discipline (tests/isa_lint.py): in every kernel that issues `global_load_lds`, every `s_barrier` has an
   `s_waitcnt` naming `vmcnt` on every control-flow path since the last DMA instruction (round 3's assign_kernel had a
   bare barrier on its loop back-edge: rows raced their DMA on a cold box).
3. The hand-counted `s_waitcnt vmcnt(N)` of dense4_kernel / list16_kernel assume exact VM-operation counts per chunk;
   the counts are checked against the generated code."""
import glob
import os
import re

import pytest

from tests import isa_lint as L

pytestmark = pytest.mark.skipif(not os.path.exists(L.HIPCC), reason="hipcc not available")


@pytest.fixture(scope="module")
def asm_dir(tmp_path_factory):
    return tmp_path_factory.mktemp("isa")


@pytest.mark.parametrize("src,pattern,expected", [
    ("scan.hip", "dense_kernel", 10),          # DH4 in {8,16,32,50,64} x {store, arg-max}
    ("scan.hip", "dense4_kernel", 6),          # the shared-stream flat scan: DH4 in {8,16,32,50} + the two K-half passes of DH4 = 50
    ("scan.hip", "dense_tiny4_kernel", 4),
    ("dense4ab.hip", "dense4ab_kernel", 1),
    ("scan16.hip", "scan16_kernel", 10),
    ("ivf_fine.hip", "ivf_list4_kernel", 7),   # DH4 in {8,16,32,50,64} + the two K-half passes
    ("ivf16.hip", "list16_kernel", 5),
    ("list16s.hip", "list16s_kernel", 12),
    ("assign.hip", "assign_kernel", 5),
    ("assign.hip", "assign_wave_kernel", 5),
])
def test_scan_kernels_use_no_scratch(asm_dir, src, pattern, expected):
    res = {k: v for k, v in L.kernel_meta(L.compile_to_asm(src, asm_dir), "private_segment_fixed_size").items() if pattern in k}
    assert len(res) == expected, sorted(res)
    spilled = {k: v for k, v in res.items() if v != 0}
    assert not spilled, f"kernels with scratch memory (register tile demoted?): {spilled}"


def _dma_sources():
    """every source that issues LDS-DMA itself or through a header helper"""
    out = []
    for path in sorted(glob.glob(os.path.join(L.CSRC, "*.hip"))):
        text = open(path).read()
        if re.search(r"global_load_lds|lds_dma16\(", text):
            out.append(os.path.basename(path))
    return out


def test_dma_sources_are_the_known_ones():
    assert _dma_sources() == ["assign.hip", "dense4ab.hip", "ivf16.hip", "ivf_fine.hip", "list16s.hip", "scan.hip"]


@pytest.mark.parametrize("src", _dma_sources())
def test_every_barrier_behind_lds_dma_has_a_vmcnt_wait(asm_dir, src):
    bad, seen = {}, 0
    for name, body in L.kernels(L.compile_to_asm(src, asm_dir)).items():
        if any("global_load_lds" in s for s in body):
            seen += 1
            v = L.dma_barrier_violations(body)
            if v:
                bad[name] = v
    assert seen > 0
    assert not bad, f"s_barrier reachable with an un-waited LDS-DMA (block label, instruction index): {bad}"
    # the builtin is banned: hipcc's own wait insertion for it is what went wrong (common.h: lds_dma16)
    assert "__builtin_amdgcn_global_load_lds" not in open(os.path.join(L.CSRC, src)).read()


def test_the_lint_catches_the_round3_defect(asm_dir, tmp_path):
    """the shape of round 3's assign_kernel: builtin DMA, double-buffered loop, __syncthreads() as the only wait"""
    src = tmp_path / "race.hip"
    src.write_text(r'''
#include <hip/hip_runtime.h>
#define GLDS16(g, l) __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(g), \
                                                      (__attribute__((address_space(3))) void*)(l), 16, 0, 0)
__global__ void race_kernel(const float4* x, float* out, int n) {
    __shared__ float4 b0[64];
    __shared__ float4 b1[64];
    float acc = 0.f;
    GLDS16(x + threadIdx.x, b0);
    for (int c = 0; c < n; c += 2) {
        __syncthreads();
        GLDS16(x + 64 * (c + 1) + threadIdx.x, b1);
        acc += b0[(threadIdx.x + 1) & 63].x;
        __syncthreads();
        GLDS16(x + 64 * (c + 2) + threadIdx.x, b0);
        acc += b1[(threadIdx.x + 1) & 63].x;
    }
    out[threadIdx.x] = acc;
}''')
    import subprocess
    out = tmp_path / "race.s"
    subprocess.run([L.HIPCC, "-O3", "--offload-arch=gfx950", "--cuda-device-only", "-S", "-o", str(out), str(src)],
                   check=True, capture_output=True, timeout=300)
    body = next(b for k, b in L.kernels(out.read_text()).items() if "race_kernel" in k)
    # whether hipcc happens to guard this instance or not, the analysis itself must see DMA + barriers ...
    assert any("global_load_lds" in s for s in body) and any(s.startswith("s_barrier") for s in body)
    # ... and a barrier with the wait stripped is reported
    stripped = [s for s in body if not (s.startswith("s_waitcnt") and "vmcnt" in s)]
    assert L.dma_barrier_violations(stripped)


@pytest.mark.parametrize("dh4,mode,pieces", [(8, 0, 8), (16, 0, 8), (32, 0, 8), (50, 0, 16), (50, 1, 16), (50, 2, 16)])
def test_dense4_vm_operation_counts_match_the_hand_counted_waits(asm_dir, dh4, mode, pieces):
    """scan.hip: kStores = 20 stores per finished block, kPieces row DMAs per chunk and wave; chunk_barrier waits with
    vmcnt(kStores) / vmcnt(2 kStores) on exactly these counts.  (MODE 2's loads of a block's starting sums are issued in front
    of the chunk's row DMAs: VM operations retire in order, the waits cover them.)"""
    body = next(b for k, b in L.kernels(L.compile_to_asm("scan.hip", asm_dir)).items() if f"dense4_kernelILi{dh4}ELi{mode}E" in k)
    runs = [r for r in L.vm_ops_between_barriers(body) if r[2] > 0]
    assert runs, "no MFMA stretch found"
    for dma, stores, mfma in runs:
        assert mfma % (4 * dh4) == 0, (dma, stores, mfma)
        assert stores % 20 == 0 and stores >= 20, f"stores per block changed (kStores = 20): {(dma, stores, mfma)}"
        assert dma % pieces == 0 and dma >= pieces, f"row DMAs per chunk changed (kPieces = {pieces}): {(dma, stores, mfma)}"


@pytest.mark.parametrize("steps,row_ops", [(4, 8), (8, 8), (16, 8), (25, 8), (50, 16)])
def test_list16_vm_operation_counts_match_the_hand_counted_waits(asm_dir, steps, row_ops):
    """ivf16.hip: per step and wave kRowOps row DMAs + 1 metadata DMA and 16 key stores; the steps' vmcnt allowances are built
    from exactly these (and a spill would add scratch loads behind them: none allowed)."""
    asm = L.compile_to_asm("ivf16.hip", asm_dir)
    name, body = next((k, b) for k, b in L.kernels(asm).items() if f"list16_kernelILi{steps}E" in k)
    assert L.kernel_meta(asm, "private_segment_fixed_size")[name] == 0, "list16_kernel spills: scratch loads are VM operations too"
    runs = [r for r in L.vm_ops_between_barriers(body) if r[2] > 0]
    assert runs, "no MFMA stretch found"
    for dma, stores, mfma in runs:
        assert mfma == steps, (dma, stores, mfma)
        assert dma == row_ops + 1, f"DMAs per step changed (kRowOps + 1 = {row_ops + 1}): {(dma, stores, mfma)}"
        assert stores in (16, 32), f"key stores per step changed: {(dma, stores, mfma)}"


@pytest.mark.parametrize("steps", [4, 8, 16, 25, 50])
def test_list16s_vm_operation_counts_match_the_hand_counted_waits(asm_dir, steps):
    """list16s.hip: per step and wave 2 record DMAs + 1 metadata DMA and 16 key stores; the steps' vmcnt allowances are built from
    exactly these"""
    asm = L.compile_to_asm("list16s.hip", asm_dir)
    name, body = next((k, b) for k, b in L.kernels(asm).items() if f"list16s_kernelILi{steps}ELi0E" in k)
    assert L.kernel_meta(asm, "private_segment_fixed_size")[name] == 0
    runs = [r for r in L.vm_ops_between_barriers(body) if r[2] > 0]
    assert runs, "no MFMA stretch found"
    for dma, stores, mfma in runs:
        assert mfma == steps, (dma, stores, mfma)
        assert dma == 3, f"DMAs per step changed (2 records + 1 metadata): {(dma, stores, mfma)}"
        assert stores in (16, 32), f"key stores per step changed: {(dma, stores, mfma)}"
