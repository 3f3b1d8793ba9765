"""Static checks on the gfx950 assembly hipcc emits for the library's kernels (CPU box: hipcc cross-compiles).

Why: LDS-DMA (`global_load_lds_*`) is asynchronous and `s_barrier` does not wait for it.  Round 3 shipped an
`assign_kernel` whose loop-header barrier had no `s_waitcnt vmcnt` on the back-edge (hipcc's own tracking of the
builtin lost it): chunk c+2's rows could still be in flight when the MFMAs read them -- passes every test until the
DMA is slow once (a cold box), then a few k-means rows go to the wrong list.  The lint walks the control-flow graph of
every kernel that issues LDS-DMA and demands, at every `s_barrier`, a `s_waitcnt` naming `vmcnt` on EVERY path since
the last DMA instruction.
"""
import os
import re
import shutil
import subprocess

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "falcon_amd", "csrc")
HIPCC = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"

_cache = {}


def compile_to_asm(src, out_dir, extra=()):
    """hipcc -S of one source of falcon_amd/csrc with the Makefile's code-generation flags; cached per (source, flags)."""
    key = (src, tuple(extra))
    if key not in _cache:
        out = os.path.join(str(out_dir), os.path.basename(src) + ("." + "_".join(e.strip("-") for e in extra) if extra else "") + ".s")
        subprocess.run([HIPCC, "-O3", "-std=c++17", "--offload-arch=gfx950", "-ffp-contract=off",
                        "-I" + os.path.join(ROOT, "include"), "--cuda-device-only", "-S", "-o", out,
                        os.path.join(CSRC, src), *extra], check=True, capture_output=True, timeout=900)
        _cache[key] = open(out).read()
    return _cache[key]


def kernels(asm):
    """{mangled name: [instruction lines]} for every kernel (and device function) body in the listing."""
    out, name, body = {}, None, []
    for line in asm.splitlines():
        m = re.match(r"^(_Z\w+):", line)
        if m:
            name, body = m.group(1), []
            out[name] = body
            continue
        if name is None:
            continue
        if re.match(r"^\s*\.(Lfunc_end|section|rodata|amdhsa_kernel)", line) or line.startswith(".Lfunc_end"):
            name = None
            continue
        s = line.split(";")[0].strip()
        if s:
            # a hand-written barrier may declare that DMA stays in flight across it on purpose: `s_barrier ; dma-ok: why`
            body.append(s + (" ;dma-ok" if s.startswith("s_barrier") and "dma-ok" in line else ""))
    return out


def kernel_meta(asm, key):
    """{kernel: int} of one `.amdhsa_<key>` directive."""
    res, name = {}, None
    for line in asm.splitlines():
        m = re.match(r"\s*\.amdhsa_kernel\s+(\S+)", line)
        if m:
            name = m.group(1)
        m = re.match(r"\s*\.amdhsa_%s\s+(\d+)" % key, line)
        if m and name:
            res[name] = int(m.group(1))
    return res


def _blocks(body):
    """basic blocks: list of (label or None, [instructions]); successors by index."""
    blocks, cur = [], [None, []]
    for s in body:
        m = re.match(r"^(\.?\w+):$", s)
        if m:
            if cur[1] or cur[0] is not None:
                blocks.append(cur)
            cur = [m.group(1), []]
            continue
        if s.startswith("."):            # directives
            continue
        cur[1].append(s)
        if re.match(r"s_(c?branch|endpgm|setpc)", s):
            blocks.append(cur)
            cur = [None, []]
    if cur[1] or cur[0] is not None:
        blocks.append(cur)
    index = {b[0]: i for i, b in enumerate(blocks) if b[0] is not None}
    succ = []
    for i, (_, ins) in enumerate(blocks):
        last = ins[-1] if ins else ""
        nxt = []
        m = re.match(r"s_(c?branch\w*)\s+(\S+)", last)
        if m:
            if m.group(2) in index:
                nxt.append(index[m.group(2)])
            if m.group(1).startswith("cbranch") and i + 1 < len(blocks):
                nxt.append(i + 1)
        elif re.match(r"s_(endpgm|setpc)", last):
            pass
        elif i + 1 < len(blocks):
            nxt.append(i + 1)
        succ.append(nxt)
    return blocks, succ


def dma_barrier_violations(body):
    """`s_barrier`s reachable with an LDS-DMA issued and no `s_waitcnt ... vmcnt` since, on at least one path.
    Forward may-analysis over the CFG: state = 'a DMA is pending un-waited'."""
    blocks, succ = _blocks(body)
    n = len(blocks)
    state_in = [False] * n
    bad = set()
    work = list(range(n))
    while work:
        i = work.pop()
        st = state_in[i]
        for k, s in enumerate(blocks[i][1]):
            if "global_load_lds" in s or re.match(r"buffer_load\w*\s.*\blds\b", s):
                st = True
            elif s.startswith("s_waitcnt") and "vmcnt" in s:
                st = False
            elif s.startswith("s_barrier") and st and "dma-ok" not in s:
                bad.add((blocks[i][0], k))
        for j in succ[i]:
            if st and not state_in[j]:
                state_in[j] = True
                work.append(j)
    return sorted(bad, key=str)


def vm_ops_between_barriers(body):
    """[(#LDS-DMA, #global stores + atomics, #MFMA)] between consecutive `s_barrier`s in LISTING order (branches ignored:
    a conditional arm counts once).  The hand-counted `s_waitcnt vmcnt(N)` of dense4_kernel / list16_kernel assume exact
    numbers of VM operations per stretch; a compiler that merges or splits stores changes them silently."""
    runs, dma, st, mf = [], 0, 0, 0
    for s in body:
        if s.startswith("s_barrier"):
            runs.append((dma, st, mf))
            dma = st = mf = 0
        elif "global_load_lds" in s:
            dma += 1
        elif re.match(r"global_(store|atomic)", s):
            st += 1
        elif s.startswith("v_mfma"):
            mf += 1
    runs.append((dma, st, mf))
    return runs
