#!/bin/bash
# SQ wait/active breakdown per kernel (one PMC pass) of the headline steps: where do the waves of the scan spend their cycles?
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
export PYTHONPATH=$R
rm -rf /tmp/psq
rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES -d /tmp/psq -o s -- python3 $R/bench.py --steps 5 --warmup 1 --no-configs --no-cpu-baseline --partitions pipelined > /tmp/osq.txt 2>&1
python3 - <<'PY'
import sqlite3, collections
db = sqlite3.connect("/tmp/psq/s_results.db")
d = collections.defaultdict(lambda: collections.defaultdict(float))
for name, cn, v in db.execute("select kernel_name, counter_name, value from counters_collection"):
    d[name.split("(")[0].replace("void ", "")[:40]][cn] += v
rows = sorted(d.items(), key=lambda kv: -kv[1].get("SQ_WAVE_CYCLES", 0))[:8]
for k, c in rows:
    wc = c.get("SQ_WAVE_CYCLES", 1)
    print(k.ljust(42), "wave_cycles %.3g" % wc, " wait_any %.2f" % (c.get("SQ_WAIT_ANY", 0) / wc), " wait_inst %.2f" % (c.get("SQ_WAIT_INST_ANY", 0) / wc),
          " active %.2f" % (c.get("SQ_ACTIVE_INST_ANY", 0) / wc), " mfma_busy/busy %.2f" % (c.get("SQ_VALU_MFMA_BUSY_CYCLES", 0) / max(c.get("SQ_BUSY_CYCLES", 1), 1)))
PY
