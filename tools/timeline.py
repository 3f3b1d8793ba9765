import sqlite3, sys
db = sqlite3.connect(sys.argv[1])
rows = list(db.execute("select name, start, end, stream_id, queue_id from kernels order by start"))
# last ~N kernels covering the final timed step: find the last 'iota_i64' occurrences (sort start of each partition)
idx = [i for i, r in enumerate(rows) if "iota_i64" in r[0]]
# take the step starting at the 4th-from-last partition start (2 partitions per step; last step is the timing pass)
start_i = idx[-4]
end_i = idx[-2]
t0 = rows[start_i][1]
prev_end = t0
busy = 0
for r in rows[start_i:end_i]:
    name = r[0].split("(")[0].replace("void ", "").replace("fal::", "")[:40]
    gap = (r[1] - prev_end) / 1e3
    print(f"{(r[1]-t0)/1e3:9.1f} us  dur {(r[2]-r[1])/1e3:8.1f}  gap {gap:8.1f}  s{r[3]} q{r[4]}  {name}")
    prev_end = max(prev_end, r[2])
print("step span us", (rows[end_i][1] - t0) / 1e3)
