"""Timeline of ONE host-to-host step through PartitionRunner.run on host-resident partitions (bench.py's latency leg):
when each partition's bytes have arrived and when its kernels start / end, in ms from the step's start.
  python tools/h2h_timeline.py [n_spectra]"""
import sys, time
import numpy as np, torch
sys.path.insert(0, ".")
from falcon_amd import synth
from falcon_amd.cluster.cluster import AnnParams, PartitionRunner, SpectrumDataset

n = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
dev = torch.device("cuda", 0)
data = synth.generate_device(n, dev)
parts = []
for ch in (2, 3):
    c = synth.select_charge_device(data, ch)
    parts.append(SpectrumDataset(*[c[k].cpu().pin_memory() for k in ("precursor_mz", "retention_time", "mz", "intensity", "indptr")]))
del data
args = (20.0, "ppm", None, 0.05, 2 ** 15, AnnParams())


class Timed(PartitionRunner):
    def _run_one(self, ds, a, kw, shard, arrived=None):
        pipe, stream = self._pipeline()
        s0, s1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        with torch.cuda.stream(stream):
            if arrived is not None:
                stream.wait_event(arrived)
            s0.record(stream)
        out = super()._run_one(ds, a, kw, shard, None)
        s1.record(stream)
        self.marks[len(ds)] = (s0, s1, time.perf_counter())
        return out


r = Timed(0, 2)
for rep in range(6):
    r.marks = {}
    torch.cuda.synchronize()
    t0e = torch.cuda.Event(enable_timing=True)
    t0e.record(torch.cuda.current_stream())
    t0 = time.perf_counter()
    r.run(parts, *args)
    torch.cuda.synchronize()
    t1 = time.perf_counter()
    line = [f"step {1e3 * (t1 - t0):.2f} ms"]
    for ds in parts:
        s0, s1, th = r.marks[len(ds)]
        line.append(f"[{len(ds)} rows: kernels start {t0e.elapsed_time(s0):.2f} end {t0e.elapsed_time(s1):.2f}, host done {1e3 * (th - t0):.2f}]")
    print(" ".join(line), flush=True)
r.close()
