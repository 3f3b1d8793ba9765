#!/bin/bash
# FETCH_SIZE / WRITE_SIZE of the IVF-regime kernels over three passes of 10 M spectra float32 (tools/scale_run.py)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
export PYTHONPATH=$R
rm -rf /tmp/pf /tmp/pw
rocprofv3 --kernel-trace --pmc FETCH_SIZE -d /tmp/pf -o f -- python3 $R/tools/scale_run.py ${1:-10000000} > /tmp/of.txt 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE -d /tmp/pw -o w -- python3 $R/tools/scale_run.py ${1:-10000000} > /tmp/ow.txt 2>&1
tail -2 /tmp/of.txt
python3 - <<'PY'
import sqlite3, collections
def agg(db, counter):
    d = collections.defaultdict(lambda: [0, 0.0])
    for name, v in sqlite3.connect(db).execute("select kernel_name, value from counters_collection where counter_name = ?", (counter,)):
        k = name.split("(")[0].replace("void ", "")[:48]
        d[k][0] += 1; d[k][1] += v * 1024.0
    return d
f = agg("/tmp/pf/f_results.db", "FETCH_SIZE"); w = agg("/tmp/pw/w_results.db", "WRITE_SIZE")
rows = sorted(f.items(), key=lambda kv: -(2 * kv[1][1] + w.get(kv[0], [0, 0])[1]))[:18]
print("kernel".ljust(50), "launches", "fetch GB (x2)", "write GB", " (all launches of the run: 4 passes over 10 M spectra)")
for k, (n, fb) in rows:
    print(k.ljust(50), str(n).rjust(8), f"{2 * fb / 1e9:12.2f}", f"{w.get(k, [0, 0])[1] / 1e9:9.2f}")
PY
