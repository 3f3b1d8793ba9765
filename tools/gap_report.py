"""Idle time of the GPU inside the busiest stretch of a rocprofv3 kernel trace: union of the kernels' [start, end) intervals, the gaps
between them, and which kernels stand on either side of the largest gaps (what the next kernel waited for: host work, a sync, a copy).

    python tools/gap_report.py <kernel-trace db> [window_ms] [n_gaps]
The window is `window_ms` around the median kernel start of the trace (run bench.py with enough steps that the timed steps hold most kernels)."""
import sqlite3, sys, collections

db = sqlite3.connect(sys.argv[1])
win_ms = float(sys.argv[2]) if len(sys.argv) > 2 else 60.0
n_gaps = int(sys.argv[3]) if len(sys.argv) > 3 else 25
rows = db.execute("select name, start, end from kernels order by start").fetchall()
try:
    rows += db.execute("select name, start, end from memory_copies order by start").fetchall()
    rows.sort(key=lambda r: r[1])
except Exception:
    pass
# the window: `win_ms` around the MEDIAN kernel start (bench.py with many steps: the timed steps hold most of the trace's kernels)
t_mid = rows[len(rows) // 2][1]
rows = [r for r in rows if t_mid - win_ms * 0.5e6 <= r[1] <= t_mid + win_ms * 0.5e6]
busy = 0
cur_s, cur_e = rows[0][1], rows[0][2]
last_name = rows[0][0]
gaps = []
for name, s, e in rows[1:]:
    if s > cur_e:
        busy += cur_e - cur_s
        gaps.append((s - cur_e, last_name, name))
        cur_s, cur_e = s, e
        last_name = name
    elif e > cur_e:
        cur_e = e
        last_name = name
busy += cur_e - cur_s
span = rows[-1][2] - rows[0][1]
short = lambda n: n.replace("void ", "").split("(")[0][-48:]
print(f"window {span / 1e6:.2f} ms, busy {busy / 1e6:.2f} ms ({100 * busy / span:.1f} %), {len(rows)} kernels, {len(gaps)} gaps totalling {(span - busy) / 1e6:.2f} ms")
by_pair = collections.defaultdict(lambda: [0, 0])
for g, a, b in gaps:
    by_pair[(short(a), short(b))][0] += 1
    by_pair[(short(a), short(b))][1] += g
for (a, b), (n, tot) in sorted(by_pair.items(), key=lambda kv: -kv[1][1])[:n_gaps]:
    print(f"{tot / 1e3:9.1f} us in {n:4d} gaps (avg {tot / n / 1e3:6.1f})  {a}  ->  {b}")
