#!/bin/bash
# HBM bytes of SOME kernels of the 10 M pass (FETCH_SIZE, WRITE_SIZE in separate passes; --kernel-include-regex keeps the other
# kernels unprofiled) + their durations from a plain kernel trace.  -> stdout
#   bash tools/pmc_kernel_bytes.sh 'resolve_kernel|coarse16w' [scale_run.py arguments...]
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
export PYTHONPATH=$R
RX=${1:-resolve_kernel}; shift
ARGS=${@:-10000000}
rm -rf /tmp/pkb_f /tmp/pkb_w /tmp/pkb_t
timeout 600 rocprofv3 --kernel-trace --pmc FETCH_SIZE --kernel-include-regex "$RX" -d /tmp/pkb_f -o f -- python3 $R/tools/scale_run.py $ARGS > /tmp/pkb_f.txt 2>&1
timeout 600 rocprofv3 --kernel-trace --pmc WRITE_SIZE --kernel-include-regex "$RX" -d /tmp/pkb_w -o w -- python3 $R/tools/scale_run.py $ARGS > /tmp/pkb_w.txt 2>&1
timeout 600 rocprofv3 --kernel-trace --stats -d /tmp/pkb_t -o t -- python3 $R/tools/scale_run.py $ARGS > /tmp/pkb_t.txt 2>&1
tail -2 /tmp/pkb_t.txt | cut -c1-300
python3 - "$RX" <<'PY'
import sqlite3, collections, glob, re, sys
rx = re.compile(sys.argv[1])
def agg(pat, counter):
    d = collections.defaultdict(lambda: [0, 0.0])
    for db in glob.glob(pat, recursive=True):
        for name, v in sqlite3.connect(db).execute("select kernel_name, value from counters_collection where counter_name = ?", (counter,)):
            k = name.replace("void ", "").split("(")[0][:60]
            d[k][0] += 1; d[k][1] += v * 1024.0
    return d
f = agg("/tmp/pkb_f/**/*.db", "FETCH_SIZE"); w = agg("/tmp/pkb_w/**/*.db", "WRITE_SIZE")
dur = collections.defaultdict(lambda: [0, 0.0])
for db in glob.glob("/tmp/pkb_t/**/*.db", recursive=True):
    con = sqlite3.connect(db)
    tabs = [r[0] for r in con.execute("select name from sqlite_master where type in ('table','view')")]
    kd = next((t for t in tabs if t.startswith("kernels")), None) or "kernels"
    try:
        for name, s, e in con.execute(f"select name, start, end from {kd}"):
            if rx.search(name):
                k = name.replace("void ", "").split("(")[0][:60]
                dur[k][0] += 1; dur[k][1] += (e - s) * 1e-6
    except Exception as ex:
        print("durations:", ex, tabs[:12])
print("kernel | launches | fetch GB (x2, gfx950) | write GB | ms   (whole run: scale_run.py's passes together)")
for k in sorted(f):
    print(f"{k:60s} {f[k][0]:5d}  {2 * f[k][1] / 1e9:9.3f}  {w[k][1] / 1e9:9.3f}  {dur[k][1]:9.3f} ({dur[k][0]})")
PY
