#!/bin/bash
# every kernel of ONE headline step (pipelined partitions): launches and total time per kernel name, in first-launch order
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
rm -rf /tmp/pk
rocprofv3 --kernel-trace -d /tmp/pk -o k -- python3 $R/bench.py --no-cpu-baseline --no-configs --steps 12 --warmup 3 --partitions pipelined > /tmp/o.txt 2>&1
python3 - <<'PY'
import sqlite3, collections
db = sqlite3.connect("/tmp/pk/k_results.db")
rows = list(db.execute("select name, start, end from kernels order by start"))
vec = [i for i, r in enumerate(rows) if "vectorize_kernel" in r[0]]
i0, i1 = vec[len(vec) // 2], vec[len(vec) // 2 + 2]
agg = collections.OrderedDict()
for r in rows[i0:i1]:
    n = r[0].split("(")[0].replace("void ", "").replace("fal::", "").replace("(anonymous namespace)::", "")[:52]
    a = agg.setdefault(n, [0, 0.0]); a[0] += 1; a[1] += (r[2] - r[1]) / 1e3
tot = sum(a[1] for a in agg.values()); cnt = sum(a[0] for a in agg.values())
for n, a in agg.items(): print(f"{n:54s} {a[0]:4d} {a[1]:9.1f} us")
print("kernels", cnt, "busy us", round(tot, 1), "span us", round((rows[i1][1] - rows[i0][1]) / 1e3, 1))
PY
