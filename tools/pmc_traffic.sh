#!/bin/bash
# HBM traffic (rocprofv3 PMC: FETCH_SIZE and WRITE_SIZE in SEPARATE passes, with --kernel-trace only) of the dominant cosine
# kernel of every bench workload -> gpurun_out/r5_pmc_traffic.json (copied to profiles/).  FETCH_SIZE is doubled for gfx950
# (MI355X_MICROARCH.md, HBM section); counters are KiB.  Run on the GPU box:  bash tools/pmc_traffic.sh
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
export PYTHONPATH=$R
O=$R/gpurun_out
run() {   # name, program args...
    name=$1; shift
    rm -rf /tmp/pf_$name /tmp/pw_$name
    rocprofv3 --kernel-trace --pmc FETCH_SIZE -d /tmp/pf_$name -o f -- python3 "$@" > /tmp/of_$name.txt 2>&1
    rocprofv3 --kernel-trace --pmc WRITE_SIZE -d /tmp/pw_$name -o w -- python3 "$@" > /tmp/ow_$name.txt 2>&1
    tail -1 /tmp/of_$name.txt | cut -c1-200
}
run headline $R/bench.py --steps 5 --warmup 1 --no-configs --no-cpu-baseline --no-cold --partitions pipelined
run 10M_f32 $R/tools/scale_run.py 10000000
run 10M_f32_dense $R/tools/scale_run.py 10000000 f32 32 f32 400 400 600
run 10M_f16 $R/tools/scale_run.py 10000000 f32 16 f16 800
run 10M_f32_b64k $R/tools/scale_run.py 10000000 f32 32 f32 400 400 560 65536
python3 - <<'PY'
import sqlite3, collections, json, os
R = os.environ["GRAFT_REPO_ROOT"]
KERNEL = {"headline": "dense4_kernel<50", "10M_f32": "list16", "10M_f32_dense": "list16", "10M_f16": "list16",
          "10M_f32_b64k": "list16"}          # ("list16": list16_kernel (dense query rows) and list16s_kernel (sparse records, round 6))
def agg(db, counter):
    d = collections.defaultdict(lambda: [0, 0.0])
    for name, v in sqlite3.connect(db).execute("select kernel_name, value from counters_collection where counter_name = ?", (counter,)):
        k = name.replace("void ", "").split("(")[0][:64]
        d[k][0] += 1; d[k][1] += v * 1024.0
    return d
out = {"_note": "rocprofv3 --kernel-trace --pmc FETCH_SIZE / --pmc WRITE_SIZE in separate passes (tools/pmc_traffic.sh); bytes per "
                "LAUNCH of each workload's dominant cosine kernel, averaged over all its launches in the run; FETCH_SIZE doubled for "
                "gfx950 (MI355X_MICROARCH.md); `kernels` = every kernel of the run: [launches, fetch GB (x2), write GB] over the whole run"}
for w, sub in KERNEL.items():
    try:
        f = agg(f"/tmp/pf_{w}/f_results.db", "FETCH_SIZE"); wr = agg(f"/tmp/pw_{w}/w_results.db", "WRITE_SIZE")
    except Exception as e:
        out[w] = {"error": repr(e)}; continue
    ks = [k for k in f if sub in k]
    n = sum(f[k][0] for k in ks); fb = 2.0 * sum(f[k][1] for k in ks); wb = sum(wr[k][1] for k in ks if k in wr)
    table = {k: [v[0], round(2 * v[1] / 1e9, 3), round(wr.get(k, [0, 0.0])[1] / 1e9, 3)]
             for k, v in sorted(f.items(), key=lambda kv: -(2 * kv[1][1] + wr.get(kv[0], [0, 0.0])[1]))[:24]}
    out[w] = {"kernel": ", ".join(ks), "launches": n, "fetch_bytes_per_launch": fb / max(n, 1), "write_bytes_per_launch": wb / max(n, 1),
              "hbm_bytes_per_launch": (fb + wb) / max(n, 1), "kernels": table}
json.dump(out, open(os.path.join(R, "gpurun_out", os.environ.get("FALCON_PMC_OUT", "r5_pmc_traffic.json")), "w"), indent=1)
for w in KERNEL:
    e = out[w]
    print(w, e.get("kernel"), e.get("launches"), "GB/launch", round(e.get("hbm_bytes_per_launch", 0) / 1e9, 3))
PY
