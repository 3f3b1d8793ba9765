#!/bin/bash
# kernel trace of 10 M spectra in BASELINE configs[3]'s bucket regime (400-600 m/z: n_list 512, n_probe 32)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
export PYTHONPATH=$R
rm -rf /tmp/p10
rocprofv3 --kernel-trace --stats -d /tmp/p10 -o r -- python3 $R/tools/scale_run.py ${1:-10000000} f32 32 f32 400 400 600 > /tmp/o.txt 2>&1
tail -2 /tmp/o.txt
python3 $R/profiles/summarize.py stats /tmp/p10/r_results.db /tmp/k.csv
python3 - <<PY
import csv
for r in list(csv.reader(open("/tmp/k.csv")))[1:22]:
    print(r[0].split("(")[0].replace("void ","")[:56].ljust(56), r[1].rjust(6), str(round(float(r[2])/1e6/4,1)).rjust(8), "ms/pass", str(round(float(r[3])/1e3,1)).rjust(9), "us avg")
PY
