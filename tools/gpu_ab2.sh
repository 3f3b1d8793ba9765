cd $GRAFT_REPO_ROOT
for extra in "" "--prefilter" "--scan f16x3"; do
  timeout 600 python bench.py --no-configs --no-cpu-baseline $extra 2>&1 | grep '^{' | python3 -c "import sys,json; j=json.loads(sys.stdin.read()); print('[$extra]', round(j['value']/1e6,1), 'M/s', round(j['ms_per_step'],3), 'ms', j['stage_ms'])"
done
