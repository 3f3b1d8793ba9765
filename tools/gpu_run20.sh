cd $GRAFT_REPO_ROOT
for m in auto pipelined auto pipelined; do
  timeout 600 python bench.py --no-configs --no-cpu-baseline --partitions $m 2>&1 | grep '^{' | python3 -c "import sys,json; j=json.loads(sys.stdin.read()); print('$m', round(j['value']/1e6,1), 'M/s', round(j['ms_per_step'],3), 'ms  h2h', round(j['value_host_to_host']/1e6,1), j['config']['partitions'][:30])"
done
