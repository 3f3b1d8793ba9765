#!/bin/bash
# counters of dense4_kernel / dense4ab_kernel on the steady-state workload of tools/dense4_ab.py (64 flat buckets of 8,192 rows)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
export PYTHONPATH=$R
i=0
for grp in \
  "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_INSTS_MFMA SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_BUSY_CYCLES" \
  "SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR" \
  "GRBM_GUI_ACTIVE GRBM_COUNT SQ_WAVES SQ_INST_LEVEL_LDS SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT" ; do
  rm -rf /tmp/pd_$i
  timeout 600 rocprofv3 --kernel-trace --pmc $grp --kernel-include-regex "dense4" -d /tmp/pd_$i -o c -- python3 $R/tools/dense4_ab.py > /tmp/pd_$i.txt 2>&1
  tail -2 /tmp/pd_$i.txt | cut -c1-120
  i=$((i+1))
done
python3 - <<'PY'
import sqlite3, collections, glob
out = collections.defaultdict(lambda: collections.defaultdict(float))
for i in range(3):
    for dbp in glob.glob(f"/tmp/pd_{i}/**/*.db", recursive=True):
        rows = sqlite3.connect(dbp).execute("select kernel_name, counter_name, value from counters_collection").fetchall()
        n = collections.Counter()
        for k, c, v in rows:
            k = k.replace("void ", "").split("(")[0][:40]
            out[k][c] += v; n[(k, c)] += 1
        for (k, c), cnt in n.items(): out[k]["launches"] = cnt
    for dbp in glob.glob(f"/tmp/pd_{i}/**/*.db", recursive=True):
        if i == 0:
            for name, dur, cnt in sqlite3.connect(dbp).execute("select name, sum(duration), count(*) from kernels group by name"):
                if "dense4" in name: print("trace:", name[:50], "launches", cnt, "total ms", round(dur / 1e6, 3))
for k, d in out.items():
    print("==", k, "launches", int(d.get("launches", 0)))
    for c in sorted(d):
        if c != "launches": print(f"   {c:32s} {d[c]:.4g}")
    if d.get("SQ_VALU_MFMA_BUSY_CYCLES"):
        print("   matrix pipe busy %.1f %% of 4 x SQ_BUSY_CU_CYCLES;  MFMA busy cycles per MFMA instruction %.1f" % (
            100 * d["SQ_VALU_MFMA_BUSY_CYCLES"] / max(4 * d.get("SQ_BUSY_CU_CYCLES", 1), 1), d["SQ_VALU_MFMA_BUSY_CYCLES"] / max(d.get("SQ_INSTS_MFMA", 1), 1)))
PY
