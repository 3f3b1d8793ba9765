cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
rm -rf gpurun_out/prof_fused
rocprofv3 --kernel-trace --stats -d gpurun_out/prof_fused -o p -- python3 tools/fused_phases.py 1000000 0 > gpurun_out/phases.txt 2>&1
grep -E "dbg|staged|rows" gpurun_out/phases.txt
find gpurun_out/prof_fused -name "*kernel_stats*" | head
python3 - <<'PY'
import csv,glob
f=glob.glob('gpurun_out/prof_fused/**/*kernel_stats.csv',recursive=True)[0]
rows=list(csv.DictReader(open(f)))
for r in rows[:14]:
    print(r['Name'][:70], r['Calls'], r['TotalDurationNs'], r['AverageNs'])
PY
