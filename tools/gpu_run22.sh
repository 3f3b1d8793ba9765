cd $GRAFT_REPO_ROOT
timeout 1800 python -m pytest tests -m gpu -x -q --timeout 900 > gpurun_out/pytest22.txt 2>&1
tail -4 gpurun_out/pytest22.txt
for m in auto auto; do
  timeout 600 python bench.py --no-configs --no-cpu-baseline --partitions $m 2>&1 | grep '^{' | python3 -c "import sys,json; j=json.loads(sys.stdin.read()); print('$m', round(j['value']/1e6,1), 'M/s', round(j['ms_per_step'],3), 'ms', j['stage_ms'])"
done
timeout 600 python tools/scale_run.py 10000000 2>&1 | tail -2
