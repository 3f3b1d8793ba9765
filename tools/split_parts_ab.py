"""Headline step with each charge partition cut in `k` pieces at precursor-window boundaries (disjoint windows = independent
buckets), all pieces on concurrent PartitionRunner slots: does a shorter last tail / more overlap pay for the per-piece fixed
costs?   python tools/split_parts_ab.py [spectra] [pieces,slots ...]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from falcon_amd import synth
from falcon_amd.cluster.cluster import AnnParams, PartitionRunner, SpectrumDataset
from falcon_amd.device import Context

n = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
variants = [tuple(int(x) for x in v.split(",")) for v in sys.argv[2:]] or [(1, 2), (2, 2), (2, 4), (4, 4)]
ctx = Context(0)
data = synth.generate_device(n, ctx.tdev)


def pieces_of(c, k):
    pm = c["precursor_mz"]
    w = torch.floor(pm)
    qs = torch.quantile(w.float(), torch.linspace(0, 1, k + 1, device=pm.device)[1:-1]) if k > 1 else w.new_zeros(0)
    edges = [float("-inf")] + [float(torch.floor(q).item()) + 1.0 for q in qs] + [float("inf")]       # cut at window boundaries
    ip = c["indptr"]
    out = []
    for lo, hi in zip(edges[:-1], edges[1:]):
        sel = torch.nonzero((w >= lo) & (w < hi)).flatten()
        counts = (ip[1:] - ip[:-1])[sel]
        indptr = torch.zeros(sel.numel() + 1, dtype=torch.int64, device=ip.device)
        torch.cumsum(counts, 0, out=indptr[1:])
        src = torch.repeat_interleave(ip[:-1][sel] - indptr[:-1], counts) + torch.arange(int(indptr[-1].item()), device=ip.device)
        out.append(SpectrumDataset(pm[sel].contiguous(), c["retention_time"][sel].contiguous(), c["mz"][src].contiguous(),
                                   c["intensity"][src].contiguous(), indptr))
    return out


args = (20.0, "ppm", None, 0.05, 2 ** 15, AnnParams())
charges = [synth.select_charge_device(data, ch) for ch in (2, 3)]
for k, slots in variants:
    parts = [p for c in charges for p in pieces_of(c, k)]
    runner = PartitionRunner(0, slots)

    def step():
        outs = runner.run(parts, *args)
        cur, lab = 0, []
        for l, m in outs:
            lab.append(l + cur); cur += int(m.numel())
        return torch.cat(lab).cpu(), cur
    for _ in range(6):
        step()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    reps = 40
    for _ in range(reps):
        _, ncl = step()
    torch.cuda.synchronize()
    print(f"n={n} pieces per charge {k}, slots {slots}: {(time.perf_counter() - t0) / reps * 1e3:.3f} ms per step, {ncl} clusters, parts {[len(p) for p in parts]}", flush=True)
    runner.close()
