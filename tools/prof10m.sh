#!/bin/bash
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
export PYTHONPATH=$R
rocprofv3 --kernel-trace --stats -d /tmp/p10 -o r -- python3 $R/tools/scale_run.py ${1:-10000000} > /tmp/o.txt 2>&1
tail -2 /tmp/o.txt
python3 $R/profiles/summarize.py stats /tmp/p10/r_results.db /tmp/k.csv
python3 - <<PY
import csv
for r in list(csv.reader(open("/tmp/k.csv")))[1:16]:
    print(r[0][:60].ljust(60), r[1].rjust(6), str(round(float(r[2])/1e6,1)).rjust(8), "ms total", str(round(float(r[3])/1e3,1)).rjust(9), "us avg")
PY
