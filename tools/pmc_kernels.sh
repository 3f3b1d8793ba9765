#!/bin/bash
# Per-kernel bottleneck counters of one workload (default: 10 M spectra float32), several rocprofv3 --pmc passes with
# --kernel-trace only (separate runs per counter group).  -> gpurun_out/pmc_kernels_<tag>.json
#   bash tools/pmc_kernels.sh [tag] [scale_run.py arguments...]
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
export PYTHONPATH=$R
TAG=${1:-10M_f32}; shift
ARGS=${@:-10000000}
i=0
for grp in \
  "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU" \
  "TA_BUSY_avr TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum TA_ADDR_STALLED_BY_TD_CYCLES_sum TCP_PENDING_STALL_CYCLES_sum" \
  "SQ_INST_CYCLES_VMEM_RD SQ_INST_CYCLES_VMEM_WR SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS SQ_INSTS_VALU" \
  "TCC_HIT_sum TCC_MISS_sum TCP_TCC_READ_REQ_sum TCP_TCC_READ_REQ_LATENCY_sum SQ_INSTS_MFMA SQ_WAVES" ; do
  rm -rf /tmp/pk_$i
  rocprofv3 --kernel-trace --pmc $grp -d /tmp/pk_$i -o c -- python3 $R/tools/scale_run.py $ARGS > /tmp/pk_$i.txt 2>&1
  tail -1 /tmp/pk_$i.txt | cut -c1-160
  i=$((i+1))
done
python3 - $TAG <<'PY'
import sqlite3, collections, json, os, sys, glob
R = os.environ["GRAFT_REPO_ROOT"]
out = collections.defaultdict(dict)
for i in range(4):
    dbs = glob.glob(f"/tmp/pk_{i}/**/*.db", recursive=True)
    if not dbs:
        continue
    db = sqlite3.connect(dbs[0])
    try:
        rows = db.execute("select kernel_name, counter_name, value from counters_collection").fetchall()
    except Exception as e:
        print("pass", i, e); continue
    agg = collections.defaultdict(lambda: [0, 0.0])
    for k, c, v in rows:
        k = k.replace("void ", "").split("(")[0][:48]
        agg[(k, c)][0] += 1; agg[(k, c)][1] += v
    for (k, c), (n, v) in agg.items():
        out[k][c] = v
        out[k]["launches"] = n
json.dump(out, open(os.path.join(R, "gpurun_out", f"pmc_kernels_{sys.argv[1]}.json"), "w"), indent=1)
def g(d, k): return d.get(k, 0.0)
top = sorted(out.items(), key=lambda kv: -g(kv[1], "SQ_BUSY_CYCLES"))[:16]
print(f"{'kernel':40s} {'busyMcyc':>9s} {'wait%':>6s} {'waitLDS%':>8s} {'vmem%':>6s} {'lds%':>6s} {'valu%':>6s} {'TAbusy':>7s} {'TAstallTC':>10s} {'TCPpend':>9s} {'L2hit':>6s} {'rdLat':>7s} {'bankConf%':>9s}")
for k, d in top:
    wc = max(g(d, "SQ_WAVE_CYCLES"), 1.0)
    print(f"{k[:40]:40s} {g(d,'SQ_BUSY_CYCLES')/1e6:9.1f} {100*g(d,'SQ_WAIT_INST_ANY')/wc:6.1f} {100*g(d,'SQ_WAIT_INST_LDS')/wc:8.1f} "
          f"{100*g(d,'SQ_ACTIVE_INST_VMEM')/wc:6.1f} {100*g(d,'SQ_ACTIVE_INST_LDS')/wc:6.1f} {100*g(d,'SQ_ACTIVE_INST_VALU')/wc:6.1f} "
          f"{g(d,'TA_BUSY_avr')/max(d.get('launches',1),1):7.1f} {g(d,'TA_ADDR_STALLED_BY_TC_CYCLES_sum')/1e6:10.1f} {g(d,'TCP_PENDING_STALL_CYCLES_sum')/1e6:9.1f} "
          f"{g(d,'TCC_HIT_sum')/max(g(d,'TCC_HIT_sum')+g(d,'TCC_MISS_sum'),1):6.2f} {g(d,'TCP_TCC_READ_REQ_LATENCY_sum')/max(g(d,'TCP_TCC_READ_REQ_sum'),1):7.0f} "
          f"{100*g(d,'SQ_LDS_BANK_CONFLICT')/max(g(d,'SQ_LDS_IDX_ACTIVE'),1):9.1f}")
PY
