cd $GRAFT_REPO_ROOT
timeout 1200 python -m pytest tests/test_gpu_search.py tests/test_gpu_linkage.py -q --timeout 600 -k "prefilter or linkage" > gpurun_out/pytest10.txt 2>&1
tail -12 gpurun_out/pytest10.txt
timeout 900 python bench.py --no-cpu-baseline --steps 20 > gpurun_out/bench10.txt 2>&1
python3 - <<'PY'
import json
l=[x for x in open('gpurun_out/bench10.txt') if x.startswith('{')]
if l:
    j=json.loads(l[-1]); print('value',j['value'],'ms',j['ms_per_step'], j['stage_ms'])
    for c in j.get('configs',[]): print(c.get('dtype'), c.get('ms_per_step'), c.get('stage_ms'), c.get('error'))
else:
    print(open('gpurun_out/bench10.txt').read()[-2000:])
PY
