"""kernels of the last timed steps of a rocprofv3 kernel trace, per stream, with the gaps between them."""
import sqlite3, sys, collections
db = sqlite3.connect(sys.argv[1])
rows = list(db.execute("select name, start, end, stream_id, queue_id from kernels order by start"))
# steps = groups separated by vectorize kernels of the big partition; take a window near the end
vec = [i for i, r in enumerate(rows) if "vectorize_kernel" in r[0]]
i0 = vec[-9]            # a few steps before the end (2 vectorize launches per step: one per partition)
i1 = vec[-5]
t0 = rows[i0][1]
last_end = collections.defaultdict(lambda: t0)
busy = collections.defaultdict(int)
for r in rows[i0:i1]:
    name = r[0].split("(")[0].replace("void ", "").replace("fal::", "").replace("(anonymous namespace)::", "")[:44]
    key = (r[3], r[4])
    gap = (r[1] - last_end[key]) / 1e3
    print(f"{(r[1]-t0)/1e3:9.1f} us  dur {(r[2]-r[1])/1e3:8.1f}  gap {gap:8.1f}  q{r[4]}  {name}")
    last_end[key] = max(last_end[key], r[2])
    busy[key] += r[2] - r[1]
span = (rows[i1][1] - t0) / 1e3
print("window us", span, {k: round(v / 1e3, 1) for k, v in busy.items()})
