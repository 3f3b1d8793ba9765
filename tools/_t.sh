#!/bin/bash
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
timeout 1200 python -m pytest tests/test_gpu_search.py tests/test_gpu_pipeline.py tests/test_gpu_regimes.py tests/test_gpu_edge_cases.py tests/test_gpu_fused.py -x -q 2>&1 | tail -3
rm -rf gpurun_out/pa
rocprofv3 --kernel-trace --stats -d gpurun_out/pa -o t -- python3 bench.py --partitions pipelined --no-configs --no-cpu-baseline --steps 10 > gpurun_out/pa.txt 2>&1
python3 profiles/summarize.py stats gpurun_out/pa/t_results.db gpurun_out/pa.csv > /dev/null
grep -E "dense" gpurun_out/pa.csv | sed 's/"\(void \)*fal::\([a-z0-9_]*\)[^"]*"/\2/' | cut -c1-120
timeout 300 python bench.py --no-configs --no-cpu-baseline --steps 50 2>/dev/null | python3 -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        j = json.loads(l); r = j['roofline']
        print('ms_per_step', round(j['ms_per_step'], 3), 'avg', r['avg_launch_ms'], 'frac', round(r['frac'],3), j['stage_ms'])
"
