cd $GRAFT_REPO_ROOT
timeout 3000 python -m pytest tests -m gpu -q --timeout 1500 > gpurun_out/pytest12.txt 2>&1
tail -15 gpurun_out/pytest12.txt
timeout 900 python bench.py > gpurun_out/bench12.txt 2>&1
python3 - <<'PY'
import json
l=[x for x in open('gpurun_out/bench12.txt') if x.startswith('{')]
if l:
    j=json.loads(l[-1]); print('value',j['value'],'ms',j['ms_per_step'], j['stage_ms'], 'h2h', j.get('value_host_to_host'))
    for c in j.get('configs',[]): print(c.get('dtype'), c.get('ms_per_step'), c.get('stage_ms'), c.get('error'))
    print(j.get('cpu_baseline'))
else:
    print(open('gpurun_out/bench12.txt').read()[-2000:])
PY
