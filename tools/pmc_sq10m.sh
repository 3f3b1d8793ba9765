#!/bin/bash
# SQ wait / active breakdown per kernel over passes of 10 M spectra float32 (one PMC pass): what the list scan's waves wait for
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
export PYTHONPATH=$R
rm -rf /tmp/psq
rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_LDS SQ_INSTS_VALU -d /tmp/psq -o s -- python3 $R/tools/scale_run.py ${1:-10000000} ${2:-f32} ${3:-16} ${4:-f32} ${5:-400} > /tmp/osq.txt 2>&1
tail -2 /tmp/osq.txt
python3 - <<'PY'
import sqlite3, collections
db = sqlite3.connect("/tmp/psq/s_results.db")
d = collections.defaultdict(lambda: collections.defaultdict(float))
n = collections.Counter()
for name, cn, v in db.execute("select kernel_name, counter_name, value from counters_collection"):
    k = name.split("(")[0].replace("void ", "")[:44]
    d[k][cn] += v
    if cn == "SQ_WAVE_CYCLES": n[k] += 1
rows = sorted(d.items(), key=lambda kv: -kv[1].get("SQ_BUSY_CYCLES", 0))[:16]
for k, c in rows:
    wc = max(c.get("SQ_WAVE_CYCLES", 1), 1)
    print(k.ljust(46), "launches %4d" % n[k], "busy %.3g" % c.get("SQ_BUSY_CYCLES", 0), "wave_cycles %.3g" % wc, " wait_any %.2f" % (c.get("SQ_WAIT_ANY", 0) / wc),
          " wait_inst %.2f" % (c.get("SQ_WAIT_INST_ANY", 0) / wc), "(lds %.2f)" % (c.get("SQ_WAIT_INST_LDS", 0) / wc), " active %.2f" % (c.get("SQ_ACTIVE_INST_ANY", 0) / wc),
          " mfma_busy/busy %.2f" % (c.get("SQ_VALU_MFMA_BUSY_CYCLES", 0) / max(c.get("SQ_BUSY_CYCLES", 1), 1)), " valu insts %.3g" % c.get("SQ_INSTS_VALU", 0))
PY
