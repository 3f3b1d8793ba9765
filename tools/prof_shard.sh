#!/bin/bash
# kernel trace of ONE rank's share of the 8-GPU job (tools/shard_share.py), on one GPU
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
export PYTHONPATH=$R
rm -rf /tmp/prof_shard
rocprofv3 --kernel-trace --stats -d /tmp/prof_shard -o shard -- python3 $R/tools/shard_share.py ${1:-8} ${2:-1250000} 0 > /tmp/o.txt 2>&1
grep "^rank" /tmp/o.txt
python3 $R/profiles/summarize.py stats /tmp/prof_shard/shard_results.db /tmp/k.csv
cp /tmp/k.csv $R/gpurun_out/shard${1:-8}_rank0_kernel_stats.csv
python3 - <<PY
import csv
rows = list(csv.reader(open("/tmp/k.csv")))[1:]
tot = sum(float(r[2]) for r in rows)
print("total kernel ms over 5 passes", round(tot/1e6,1))
for r in rows[:28]:
    print(r[0][:60].ljust(60), r[1].rjust(6), str(round(float(r[2])/1e6,1)).rjust(8), "ms total", str(round(float(r[3])/1e3,1)).rjust(9), "us avg")
PY
python3 $R/tools/timeline2.py /tmp/prof_shard/shard_results.db > $R/gpurun_out/shard${1:-8}_timeline.txt
tail -1 $R/gpurun_out/shard${1:-8}_timeline.txt
