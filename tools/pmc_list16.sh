#!/bin/bash
# Counter passes over ONE kernel family of the 10 M pass (rocprofv3 --kernel-include-regex keeps every other kernel at full
# speed; separate --pmc passes, --kernel-trace only).  -> gpurun_out/pmc_<regex>.txt
#   bash tools/pmc_list16.sh [regex] [scale_run.py arguments...]
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
export PYTHONPATH=$R
RX=${1:-list16_kernel}; shift
ARGS=${@:-10000000}
i=0
NG=${NGRP:-6}
for grp in \
  "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU" \
  "SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_SALU SQ_WAVES SQ_INST_CYCLES_VMEM_WR" \
  "SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_INSTS_VALU_MFMA_MOPS_F16 SQ_BUSY_CU_CYCLES SQ_WAIT_ANY SQ_WAVE_CYCLES" \
  "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_LDS_UNALIGNED_STALL SQ_LDS_DATA_FIFO_FULL SQ_LDS_CMD_FIFO_FULL SQ_INST_CYCLES_VMEM_RD" \
  "TA_BUSY_avr TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum TA_ADDR_STALLED_BY_TD_CYCLES_sum TCP_PENDING_STALL_CYCLES_sum" \
  "TCC_HIT_sum TCC_MISS_sum TCP_TCC_READ_REQ_sum TCP_TCC_READ_REQ_LATENCY_sum TA_FLAT_READ_LDS_WAVEFRONTS_sum TA_FLAT_WRITE_WAVEFRONTS_sum" ; do
  [ $i -ge $NG ] && break
  rm -rf /tmp/pl_$i
  timeout 600 rocprofv3 --kernel-trace --pmc $grp --kernel-include-regex "$RX" -d /tmp/pl_$i -o c -- python3 $R/tools/scale_run.py $ARGS > /tmp/pl_$i.txt 2>&1
  tail -1 /tmp/pl_$i.txt | cut -c1-120
  i=$((i+1))
done
python3 - "$RX" <<'PY'
import sqlite3, collections, glob, sys, os
out = collections.defaultdict(lambda: collections.defaultdict(float))
for i in range(6):
    for dbp in glob.glob(f"/tmp/pl_{i}/**/*.db", recursive=True):
        try:
            rows = sqlite3.connect(dbp).execute("select kernel_name, counter_name, value from counters_collection").fetchall()
        except Exception as e:
            print("pass", i, e); continue
        n = collections.Counter()
        for k, c, v in rows:
            k = k.replace("void ", "").split("(")[0][:40]
            out[k][c] += v
            n[(k, c)] += 1
        for (k, c), cnt in n.items():
            out[k]["launches"] = cnt
for k, d in out.items():
    print("==", k, "launches", int(d.get("launches", 0)))
    for c in sorted(d):
        if c != "launches":
            print(f"   {c:40s} {d[c]:.4g}")
    wc = max(d.get("SQ_WAVE_CYCLES", 1), 1)
    if d.get("SQ_VALU_MFMA_BUSY_CYCLES"):
        # (SQ_BUSY_CU_CYCLES counts CU-cycles with a wave resident; SQ_VALU_MFMA_BUSY_CYCLES sums the four SIMDs' matrix pipes)
        print("   derived: matrix pipe busy %.1f %% of 4 x SQ_BUSY_CU_CYCLES" % (100 * d["SQ_VALU_MFMA_BUSY_CYCLES"] / max(4 * d.get("SQ_BUSY_CU_CYCLES", 1), 1)))
    print("   derived: wait%% %.1f  waitLDS%% %.1f  vmem-active%% %.1f  lds-active%% %.1f  valu-active%% %.1f  bankconf%% %.1f  L2 hit %.2f  rd latency %.0f" % (
        100 * d.get("SQ_WAIT_INST_ANY", 0) / wc, 100 * d.get("SQ_WAIT_INST_LDS", 0) / wc, 100 * d.get("SQ_ACTIVE_INST_VMEM", 0) / wc,
        100 * d.get("SQ_ACTIVE_INST_LDS", 0) / wc, 100 * d.get("SQ_ACTIVE_INST_VALU", 0) / wc,
        100 * d.get("SQ_LDS_BANK_CONFLICT", 0) / max(d.get("SQ_LDS_IDX_ACTIVE", 1), 1),
        d.get("TCC_HIT_sum", 0) / max(d.get("TCC_HIT_sum", 0) + d.get("TCC_MISS_sum", 0), 1),
        d.get("TCP_TCC_READ_REQ_LATENCY_sum", 0) / max(d.get("TCP_TCC_READ_REQ_sum", 1), 1)))
PY
