#!/bin/bash
# per-kernel average times of the graph/tail kernels in the bench (rocprofv3 kernel trace)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --stats -d /tmp/pk -o r -- python3 $R/bench.py --steps 6 --warmup 1 --no-cpu-baseline > /dev/null 2>&1
python3 $R/profiles/summarize.py stats /tmp/pk/r_results.db /tmp/k.csv
python3 - <<PY
import csv
for r in list(csv.reader(open("/tmp/k.csv")))[1:]:
    if any(s in r[0] for s in ("dbscan", "medoid", "refine", "relabel", "label_", "member_", "finalize", "cluster_size", "select", "dense", "vectorize")):
        print(r[0].split("(")[0][-45:].ljust(46), r[1].rjust(5), str(round(float(r[3])/1e3,1)).rjust(8), "us avg")
PY
