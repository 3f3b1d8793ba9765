#!/bin/bash
# PMC counters for the select kernel (separate passes), summarised per kernel
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for set in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES" "SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_INSTS_SMEM"; do
  tag=$(echo $set | tr ' ' '_')
  rocprofv3 --kernel-trace --pmc $set -d /tmp/pmc_sel_$tag -o r -- python3 $R/bench.py --steps 1 --warmup 0 --no-cpu-baseline > /dev/null 2>&1
done
python3 - <<PY
import sqlite3, glob, collections
for db in sorted(glob.glob("/tmp/pmc_sel_*/r_results.db")):
    c = sqlite3.connect(db)
    agg = collections.defaultdict(lambda: [0, 0.0])
    for name, cn, v in c.execute("select kernel_name, counter_name, value from counters_collection"):
        key = None
        for sub in ("select_kernel", "dense_kernel", "filter_kernel", "refine_kernel"):
            if sub in name: key = sub
        if key:
            agg[(key, cn)][0] += 1; agg[(key, cn)][1] += v
    for (k, cn), (n, v) in sorted(agg.items()):
        print(f"{k:16s} {cn:24s} launches {n:3d} total {v:.4g} per-launch {v/n:.4g}")
PY
