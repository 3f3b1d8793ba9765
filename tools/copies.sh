#!/bin/bash
# which copy kernels does a headline step hold, and how long are they?  (kernel trace of bench.py, pipelined partitions)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
rm -rf /tmp/pc
rocprofv3 --kernel-trace --memory-copy-trace -d /tmp/pc -o c -- python3 $R/bench.py --no-cpu-baseline --no-configs --steps 12 --warmup 3 --partitions ${1:-pipelined} > /tmp/o.txt 2>&1
grep '^{' /tmp/o.txt | python3 -c "import sys,json; j=json.loads(sys.stdin.read()); print(j['value'], j['ms_per_step'])"
python3 - <<'PY'
import sqlite3
db = sqlite3.connect("/tmp/pc/c_results.db")
tabs = [r[0] for r in db.execute("select name from sqlite_master where type in ('table','view')")]
print([t for t in tabs if 'copy' in t.lower() or t in ('kernels',)])
rows = list(db.execute("select name, start, end, stream_id from kernels order by start"))
vec = [i for i, r in enumerate(rows) if "vectorize_kernel" in r[0]]
i0, i1 = vec[len(vec) // 2], vec[len(vec) // 2 + 2]          # one step (two partitions) in the middle of the timed region
t0 = rows[i0][1]
print("step span us", (rows[i1][1] - t0) / 1e3)
for r in rows[i0:i1]:
    if "copyBuffer" in r[0] or "fillBuffer" in r[0]:
        d = (r[2] - r[1]) / 1e3
        if d > 7: print(f"{(r[1]-t0)/1e3:9.1f} us  dur {d:8.1f}  s{r[3]}  {r[0][:40]}")
try:
    mc = list(db.execute("select * from memory_copies order by start"))
    cols = [c[1] for c in db.execute("pragma table_info(memory_copies)")]
    print(cols)
    inwin = [m for m in mc if t0 <= m[cols.index('start')] <= rows[i1][1]]
    for m in inwin:
        d = (m[cols.index('end')] - m[cols.index('start')]) / 1e3
        if d > 12: print({c: m[i] for i, c in enumerate(cols) if c in ('name', 'size', 'src_agent_type', 'dst_agent_type')}, round(d, 1), round((m[cols.index('start')] - t0) / 1e3, 1))
except Exception as e:
    print("no memory copy view:", e)
PY
