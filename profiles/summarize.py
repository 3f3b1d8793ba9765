"""Turn rocprofv3's rocpd databases (gpurun_out/prof_*/r1_results.db) into the summaries kept here.

  python profiles/summarize.py stats  <kernel-trace db>  <out.csv>
  python profiles/summarize.py pmc    <FETCH_SIZE db> <WRITE_SIZE db> <passes> <out.json>

`passes` = how many passes over the hot path the profiled command made (bench.py runs
warmup + steps + 1 per-stage timing pass), so the JSON holds PER-STEP and PER-LAUNCH figures.
FETCH_SIZE / WRITE_SIZE are KiB; FETCH_SIZE is doubled for gfx950 as MI355X_MICROARCH.md prescribes.
"""
import csv
import json
import sqlite3
import statistics
import sys

GROUPS = {  # kernel-name substring -> stage key
    "vectorize_kernel": "vectorize", "dense_kernel": "scan", "dense4_kernel": "scan", "dense_tiny4_kernel": "scan", "scan16_kernel": "scan", "ivf_list_kernel": "scan",
    "ivf_list4_kernel": "scan", "approx_kernel": "prefilter_approx", "band_kernel": "prefilter_band",
    "resolve_kernel": "prefilter_resolve", "fused_fallback_kernel": "prefilter_fallback",
    "assign16_kernel": "kmeans_assign16", "assign_exact_rows": "kmeans_exact_rows", "assign_kernel": "kmeans_assign",
    "list_walk_kernel": "kmeans_update",
    "select_kernel": "select", "filter_kernel": "filter", "dbscan_core": "dbscan_core", "dbscan_edges": "dbscan_edges",
    "refine_kernel": "refine", "medoid_score": "medoid_score",
}


def stats(db_path, out_csv):
    db = sqlite3.connect(db_path)
    per = {}
    for name, dur in db.execute("select name, duration from kernels"):
        per.setdefault(name, []).append(dur)
    total = sum(sum(v) for v in per.values())
    rows = sorted(per.items(), key=lambda kv: -sum(kv[1]))
    with open(out_csv, "w", newline="") as f:
        w = csv.writer(f, quoting=csv.QUOTE_NONNUMERIC)
        w.writerow(["Name", "Calls", "TotalDurationNs", "AverageNs", "Percentage", "MinNs", "MaxNs", "StdDev"])
        for name, v in rows:
            w.writerow([name, len(v), sum(v), round(sum(v) / len(v), 1), round(100.0 * sum(v) / total, 2), min(v), max(v),
                        round(statistics.pstdev(v), 1)])


def _counter(db_path, counter):
    db = sqlite3.connect(db_path)
    out = {}
    for name, value in db.execute("select kernel_name, value from counters_collection where counter_name = ?", (counter,)):
        for sub, key in GROUPS.items():
            if sub in name:
                g = out.setdefault(key, [0, 0.0])
                g[0] += 1
                g[1] += value * 1024.0
                break
    return out


def pmc(fetch_db, write_db, passes, out_json, cmd="python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline"):
    fetch = _counter(fetch_db, "FETCH_SIZE")
    write = _counter(write_db, "WRITE_SIZE")
    res = {"_note": f"rocprofv3 --kernel-trace --pmc FETCH_SIZE / --pmc WRITE_SIZE in separate passes over `{cmd}` "
                    f"({passes} passes of the hot path: priming, warmup, timed steps, the per-stage timing pass and the "
                    "host-to-host steps); figures are PER STEP (one pass over 1,000,000 spectra) and per launch. Counters are KiB; "
                    "FETCH_SIZE is doubled as MI355X_MICROARCH.md prescribes for gfx950."}
    for key in fetch:
        launches, fb = fetch[key]
        wb = write.get(key, [0, 0.0])[1]
        fb = 2.0 * fb / passes
        wb = wb / passes
        lp = launches / passes
        res[key] = {"launches_per_step": lp, "fetch_bytes": fb, "write_bytes": wb, "hbm_bytes": fb + wb,
                    "hbm_bytes_per_launch": (fb + wb) / lp}
    with open(out_json, "w") as f:
        json.dump(res, f, indent=1)


if __name__ == "__main__":
    if sys.argv[1] == "stats":
        stats(sys.argv[2], sys.argv[3])
    else:
        pmc(sys.argv[2], sys.argv[3], int(sys.argv[4]), sys.argv[5], *sys.argv[6:7])
