#!/bin/bash
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_pipeline.py tests/test_gpu_edge_cases.py -x -q 2>&1 | tail -2
rm -rf gpurun_out/pa
rocprofv3 --kernel-trace --stats -d gpurun_out/pa -o t -- python3 bench.py --partitions pipelined --no-configs --no-cpu-baseline --steps 10 > gpurun_out/pa.txt 2>&1
python3 profiles/summarize.py stats gpurun_out/pa/t_results.db gpurun_out/pa.csv > /dev/null
grep -E "dbscan|medoid|refine" gpurun_out/pa.csv | sed 's/"\(void \)*fal::\([a-z0-9_]*\)[^"]*"/\2/' | cut -c1-120
